#!/bin/bash
# Round 6, session u: t2 (session t's tree) / u1 (+ the scan-list build's prefix sum by DPP, the store position riding in the
# scan's key, no queue atomic when every search had its group) / u2 (+ the launch's epilogue without round trips to memory:
# last_pose from LDS, the sums and counts as the prologue read them).  The GPU suite on u2's tree, then the same-box A/B.
# Usage (through gpurun): TAG=r06_u bash scripts/gpu_r6_u.sh
set -u
T="${TAG:-r06_u}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh t2 u1 u2 > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
grep "searches by scan list" $O/${T}_icp_probe_u2.txt
