#!/bin/bash
# Round 6, session z: a whole wave per list scan when a workgroup has at most eight searches (kIcpWaveScan: the upper group of a
# wave lends its lanes to the lower one's scan -- tile_scan_list<64>) -- ws against head (the last commit).  The registration
# tests on ws's tree first, then the same-box A/B.
# Usage (through gpurun): TAG=r06_z bash scripts/gpu_r6_z.sh
set -u
T="${TAG:-r06_z}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:--k "align or registration or stability or closest or golden or smoke or config"} 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh head ws > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
