#!/bin/bash
# Round 6, session am: two small things on the iteration's critical path -- a leader's 14 members added without the selects of the
# general row sum, a group's first search index asked for beside the search count -- mc against the last commit (cur).
# Usage (through gpurun): TAG=r06_am bash scripts/gpu_r6_ah.sh
set -u
T="${TAG:-r06_am}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or norms or timeout or give_up or linear or solve"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-cur mc} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
