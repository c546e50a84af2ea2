#!/bin/bash
# Round 6, closing state: the GPU suite (no -x) and two bench lines on the bounds-asserting build (make DEBUG_BOUNDS=1, built on
# the box) -- the round's later kernels (transposed reduction rows, register copies of LDS records, packed scan keys) under the
# same check the round's first commit passed.
# Usage (through gpurun): TAG=r06_dbg2 bash scripts/gpu_r6_dbg.sh
set -u
T="${TAG:-r06_dbg2}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
make -C kiss-icp_amd/csrc clean > /dev/null
( make -C kiss-icp_amd/csrc -j8 DEBUG_BOUNDS=1 2>&1 | grep -E "error|Error" ; make -C kiss-icp_amd/cpp 2>&1 | grep -E "error|Error"; make -C tests/cpp 2>&1 | grep -E "error|Error" ) > $O/${T}_build_debug.log
( timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -30 ) > $O/${T}_pytest_debug.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_debug_20_5.json 2> $O/${T}_bench_debug_20_5.err
timeout 400 python bench.py --workload livox --steps 30 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_debug_livox.json 2> $O/${T}_bench_debug_livox.err
for f in $O/${T}_build_debug.log $O/${T}_pytest_debug.log; do echo "== $f"; tail -12 $f; done
for f in $O/${T}_bench_debug_20_5 $O/${T}_bench_debug_livox; do echo "== $f"; python - <<PY
import json
try:
    d = json.loads(open("$f.json").read().strip().splitlines()[-1]); print(round(d["value"], 1), d.get("pose_error_vs_cpu"), d["host_side"].get("map_grows"))
except Exception as e:
    print("failed", e)
PY
tail -3 $f.err; done
