#!/bin/bash
# Round 6, session i: the whole GPU suite on the working tree, then one option on / off interleaved on the MulRan-like and the
# KITTI-like bench commands, kernel timelines of both.
# Usage (through gpurun): TAG=r06_i OPT=sort_by_rank bash scripts/gpu_r6_i.sh
set -u
T="${TAG:-r06_i}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OPT="${OPT:-sort_by_rank}"
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=5 ${PYTEST_ARGS:-} 2>&1 | tail -25 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; [ "${STOP_ON_FAIL:-1}" = 1 ] && exit 1; fi
for rep in 1 2; do
  for v in 1 0; do
    timeout 300 python bench.py --workload mulran --steps 100 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench_mulran_${OPT}${v}_r${rep}.json 2> $O/${T}_bench_mulran_${OPT}${v}_r${rep}.err
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench_${OPT}${v}_r${rep}.json 2> $O/${T}_bench_${OPT}${v}_r${rep}.err
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench20_${OPT}${v}_r${rep}.json 2> $O/${T}_bench20_${OPT}${v}_r${rep}.err
  done
done
( STEPS=40 timeout 200 bash scripts/timeline.sh > $O/${T}_timeline.txt 2>&1 )
( STEPS=40 BENCH_ARGS="--workload mulran" timeout 200 bash scripts/timeline.sh > $O/${T}_timeline_mulran.txt 2>&1 )
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "gap us", round(1e3 * d["host_side"]["device_gap_ms"] / d["steps"], 2), "frac", round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -8 $O/${T}_pytest_gpu.log
tail -26 $O/${T}_timeline.txt
tail -28 $O/${T}_timeline_mulran.txt
