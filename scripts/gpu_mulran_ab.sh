#!/bin/bash
set -u
T="${TAG:-r05_k}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu -k "whole_grid or stage_in or sharing or two_pipelines or deskew or mulran or cpp_program or batch" --durations=6 2>&1 | tail -14 ) > $O/${T}_pytest_subset.log
for rep in 1 2; do
  for v in 1 0; do
    timeout 300 python bench.py --workload mulran --steps 100 --warmup 10 --no-cpu-baseline --no-extras --opt stage_in=$v > $O/${T}_bench_mulran_stage_in${v}_r${rep}.json 2> $O/${T}_bench_mulran_stage_in${v}_r${rep}.err
  done
done
cat $O/${T}_pytest_subset.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "gap", round(d["host_side"]["device_gap_ms"] / d["steps"], 4))
    except Exception as e:
        print(f, "FAILED", e)
PY
