#!/bin/bash
# Round 6, session ad: every workgroup's block of granule pairs on lines of its own (kIcpGranStride 24: 384 bytes instead of the
# packed 304) -- gs24 against the last commit (head); the registration and timeout tests on gs24's tree first.  What the exchange
# alone does with it: scripts/probes/xchg_bench.hip (profiles/r06_ac_*).
# Usage (through gpurun): TAG=r06_ad bash scripts/gpu_r6_ad.sh
set -u
T="${TAG:-r06_ad}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config or timeout or give_up or deadline or weights"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-head gs24} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
