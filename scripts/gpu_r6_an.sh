#!/bin/bash
# Round 6, session an: roots only where they can decide (phase C's sqrt(d2) < max_dist, the loop's sqrt(|dx|^2) < convergence_criterion:
# a squared value 2^-40 off the squared bound is on its side whatever the roundings do), the tile's origin read once -- group form
# only -- mc2 against the last commit (cur).  The registration tests on mc2's tree first.
# Usage (through gpurun): TAG=r06_an bash scripts/gpu_r6_ah.sh
set -u
T="${TAG:-r06_an}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or norms or timeout or give_up or linear or solve"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-cur mc2} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
