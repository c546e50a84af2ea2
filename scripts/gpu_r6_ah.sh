#!/bin/bash
# Round 6, session ah: session z's whole wave per list scan again (kIcpWaveScan), the exchanges between the two halves of the wave
# by v_permlane32_swap instead of three trips through the LDS crossbar, the list's base and length by readfirstlane -- ws2 against
# the last commit (cur).  The registration tests on ws2's tree first.
# Usage (through gpurun): TAG=r06_ah bash scripts/gpu_r6_ah.sh
set -u
T="${TAG:-r06_ah}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or norms"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-cur ws2} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
