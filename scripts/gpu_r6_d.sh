#!/bin/bash
# Round 6, session d: registration tests on the working tree, then library variants interleaved on one box (scripts/ab_bench.sh), probe.
set -u
T="${TAG:-r06_d}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu -k "not cold" 2>&1 | tail -8 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
TAG=${T} REPS=2 ARGS="--steps 200 --warmup 10 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh ${VARIANTS:-v1 cur}
cp $O/${T}_ab.txt $O/${T}_ab_200_10.txt
TAG=${T} REPS=2 ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh ${VARIANTS:-v1 cur}
cp $O/${T}_ab.txt $O/${T}_ab_20_5.txt
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
tail -3 $O/${T}_pytest_gpu.log
head -60 $O/${T}_icp_probe_steady.txt
