#!/bin/bash
# Round 6, session aa: the exchange's first hop through the XCD's L2 (kIcpLocalHop: members of a group that sits on one XCD store
# their partial sums plainly from the second iteration on; placement checked inside the launch by HW_REG_XCC_ID) -- lh against
# lh0 (the same tree with the switch off) and head (the last commit).  The registration tests on lh's tree first.
# Usage (through gpurun): TAG=r06_aa bash scripts/gpu_r6_aa.sh
set -u
T="${TAG:-r06_aa}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or timeout or give_up"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh lh0 lh head > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
