#!/bin/bash
# Round 6, session ab: the exchange's polls with three loads in flight (kIcpPollPipelined: granule_poll_pair3, written out in
# assembly, the loads landing in v244 .. v255) -- a load every 4 / 8 / 16 x 64 cycles (pp4 / pp8 / pp16) against the last commit
# (head).  The registration and timeout tests on pp8's tree (the working tree) first.
# Usage (through gpurun): TAG=r06_ab bash scripts/gpu_r6_ab2.sh
set -u
T="${TAG:-r06_ab}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or timeout or give_up or deadline"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-head pp4 pp8 pp16} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
