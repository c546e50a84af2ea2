#!/bin/bash
# Round 6, session j: sort keys left by the scatter (rank sort reads them); the C++ RegisterFrame leg of the bench line.
set -u
T="${TAG:-r06_j}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu -k "not cold and not streams_sharing and not deadlines" 2>&1 | tail -8 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
for rep in 1 2; do
  timeout 300 python bench.py --workload mulran --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/${T}_bench_mulran_r${rep}.json 2> $O/${T}_bench_mulran_r${rep}.err
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/${T}_bench_r${rep}.json 2> $O/${T}_bench_r${rep}.err
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_20_5.json 2> $O/${T}_bench_20_5.err
( STEPS=40 BENCH_ARGS="--workload mulran" timeout 200 bash scripts/timeline.sh > $O/${T}_timeline_mulran.txt 2>&1 )
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "gap us", round(1e3 * d["host_side"]["device_gap_ms"] / d["steps"], 2), "frac", round(d["roofline"]["frac"], 4),
              {k: (round(d[k]["scans_per_s"]) if "scans_per_s" in d[k] else d[k]) for k in ("sync_per_frame", "sync_with_outputs", "sync_with_outputs_cpp", "device_resident") if k in d})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $O/${T}_pytest_gpu.log; tail -3 $O/${T}_bench_20_5.err
tail -16 $O/${T}_timeline_mulran.txt
