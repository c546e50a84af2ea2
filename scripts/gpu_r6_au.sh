#!/bin/bash
# Round 6, session au: group form: all eight waves solve (kIcpSolveAll: no hand-off of the update through LDS, one barrier less per
# iteration) -- sv8 against the last commit (cur), three interleaved repetitions of both bench commands; the solve / registration
# tests on sv8's tree first.
# Usage (through gpurun): TAG=r06_au bash scripts/gpu_r6_au.sh
set -u
T="${TAG:-r06_au}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "align or registration or solve or golden or smoke or linear" 2>&1 | tail -6 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=3 ARGS="--steps 200 --warmup 10 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh cur sv8 > /dev/null; cp $O/${T}_ab.txt $O/${T}_ab_200_10.txt
TAG=$T REPS=3 ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh cur sv8 > /dev/null; cp $O/${T}_ab.txt $O/${T}_ab_20_5.txt
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
