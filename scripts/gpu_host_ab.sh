#!/bin/bash
# The driver's suite command as the first process of the box, then the HOST side of the bench with and without an option
# (OPT, default relaxed_backpressure): throughput and the process's CPU seconds per frame, one stream and eight stacked on GPU 0.
# Usage (through gpurun): TAG=r05_s bash scripts/gpu_host_ab.sh
set -u
T="${TAG:-r05_host}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OPT="${OPT:-relaxed_backpressure}"
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=6 2>&1 | tail -14 ) > $O/${T}_pytest_gpu.log
cp $O/test_cpp_api_last.log $O/${T}_test_cpp_api_laps.log 2>/dev/null
for rep in 1 2; do
  for v in 1 0; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench_${OPT}${v}_r${rep}.json 2> $O/${T}_bench_${OPT}${v}_r${rep}.err
  done
done
for v in 1 0; do
  timeout 300 python bench.py --gpus 8 --device 0 --steps 20 --warmup 5 --opt $OPT=$v > $O/${T}_bench8_${OPT}${v}.json 2> $O/${T}_bench8_${OPT}${v}.err
done
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_20_5.json 2> $O/${T}_bench_20_5.err
cat $O/${T}_pytest_gpu.log
grep "^\[" $O/${T}_test_cpp_api_laps.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        hs = d.get("host_side") or {}
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", d.get("host_cpu"), {k: hs.get(k) for k in ("backpressure_waits", "backpressure_ms", "stage_ms", "device_numa_node", "staging_numa_node", "staging_helpers", "helpers_bound", "max_device_gap_ms")})
    except Exception as e:
        print(f, "FAILED", e)
PY
