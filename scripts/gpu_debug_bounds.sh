#!/bin/bash
# One box session: the GPU suite on the release build, then the SAME suite and two bench lines on the bounds-asserting build
# (make DEBUG_BOUNDS=1: every index taken from device memory is checked, kicp_internal.hpp KICP_IDX), built on the box.
# Usage (through gpurun): TAG=r05_e bash scripts/gpu_debug_bounds.sh
set -u
T="${TAG:-r05_dbg}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -40 ) > $O/${T}_pytest_release.log
make -C kiss-icp_amd/csrc clean > /dev/null
( make -C kiss-icp_amd/csrc -j8 DEBUG_BOUNDS=1 2>&1 | grep -E "error|Error" ; make -C kiss-icp_amd/cpp 2>&1 | grep -E "error|Error"; make -C tests/cpp 2>&1 | grep -E "error|Error" ) > $O/${T}_build_debug.log
( timeout 1200 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -30 ) > $O/${T}_pytest_debug.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_debug_20_5.json 2> $O/${T}_bench_debug_20_5.err
timeout 400 python bench.py --workload livox --steps 30 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_debug_livox.json 2> $O/${T}_bench_debug_livox.err
for f in $O/${T}_pytest_release.log $O/${T}_build_debug.log $O/${T}_pytest_debug.log; do echo "== $f"; tail -25 $f; done
for f in $O/${T}_bench_debug_20_5 $O/${T}_bench_debug_livox; do echo "== $f"; tail -c 600 $f.json; tail -5 $f.err; done
