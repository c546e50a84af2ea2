#!/bin/bash
set -u
T="${TAG:-r05_ab}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OPT="${OPT:-icp_group_prune}"
( timeout 600 python -m pytest tests -x -q -m gpu -k "${KEXPR:-pair or norm or ties or align or pipeline or vegetated or closest or thread_per_query or stability or flat}" 2>&1 | tail -8 ) > $O/${T}_pytest_subset.log
for rep in 1 2; do
  for v in 1 0; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench_${OPT}${v}_r${rep}.json 2> $O/${T}_bench_${OPT}${v}_r${rep}.err
  done
done
for v in 1 0; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench20_${OPT}${v}.json 2> $O/${T}_bench20_${OPT}${v}.err
done
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
cat $O/${T}_pytest_subset.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "us/iter", round(1e3 * d["ms_per_icp_iter"], 2), "frac", round(d["roofline"]["frac"], 4), d["icp_last_launch"])
    except Exception as e:
        print(f, "FAILED", e)
PY
grep -A8 "^iteration 23" $O/${T}_icp_probe_steady.txt | head -12
