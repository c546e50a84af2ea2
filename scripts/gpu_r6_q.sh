#!/bin/bash
# Round 6, session q: prologues that ask for everything they need from memory in one round trip (k_icp, k_icp_weights),
# the run boundaries' prefix loops four words per trip, k_map_apply's block allocation without the head atomic when the free
# queue is empty -- variant q1 against cur (session p's tree), same box; the GPU suite on q1's tree; then option sweeps of the
# thread-per-query form on the 1M-point configuration (scripts/opt_sweep.py), whose defaults date from before its spills went.
# Usage (through gpurun): TAG=r06_q bash scripts/gpu_r6_q.sh
set -u
T="${TAG:-r06_q}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-2} bash scripts/gpu_ab_variants.sh cur q1 > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
( STEPS=40 timeout 200 bash scripts/timeline.sh > $O/${T}_timeline.txt 2>&1 ); tail -14 $O/${T}_timeline.txt
sw() { ( timeout 400 python scripts/opt_sweep.py "$@" livox=1 kitti=0 livox_frames=30 2>&1 | grep -v Warning | tail -12 ) > $O/${T}_sweep_$1.txt; echo "== $*"; cat $O/${T}_sweep_$1.txt; }
sw icp_wide_per_round 2,4,6,8,12
sw icp_wide_promote_from 0,1,2
sw icp_wide_prefill 0,4,8
sw icp_wide_group_max 64,128,256
sw icp_wide_load_eighths 4,5,6
