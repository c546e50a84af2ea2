#!/bin/bash
# Round 6, session af: the WHOLE GPU suite on the tree with the exchange's new layout and the DPP publish (cur), then cur against
# the same tree with s_sleep 0 in the exchange's polls (sl0; the probe: -0.04 us per round).
# Usage (through gpurun): TAG=r06_af bash scripts/gpu_r6_af.sh
set -u
T="${TAG:-r06_af}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-cur sl0} > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
