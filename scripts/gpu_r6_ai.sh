#!/bin/bash
# Round 6, session ai: the exchange's two staged reductions (a leader's members, the leaders' group sums) by DPP rows too -- a scalar's
# sixteen addends on the lanes of one row: one LDS read each, four row operations; a leader's lane c stores copy c -- rd against
# the last commit (cur).  The registration tests on rd's tree first.
# Usage (through gpurun): TAG=r06_ai bash scripts/gpu_r6_ah.sh
set -u
T="${TAG:-r06_ai}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or norms or timeout or give_up or linear or solve"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-cur rd} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
