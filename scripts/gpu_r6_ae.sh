#!/bin/bash
# Round 6, session ae: the leaders' group sums in eight copies (a workgroup polls copy b mod 8) and the release build's exchange
# without the profiling slot (gc8); on top of it a workgroup's sums over its 16 groups by DPP row operations, stored by the lane
# that holds them -- no LDS round trip, barrier or chain of sixteen additions in front of the first hop (dpp = the working tree)
# -- against the last commit (gs24).  The registration and timeout tests on dpp's tree first.
# Usage (through gpurun): TAG=r06_ae bash scripts/gpu_r6_ae.sh
set -u
T="${TAG:-r06_ae}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or timeout or give_up or deadline or weights or linear or solve"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-gs24 gc8 dpp} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
