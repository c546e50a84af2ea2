#!/bin/bash
# Round 6, session y: phase C's rows formed during phase B (kIcpTermsInB: the last wave serves the stable points while the others
# search, a searching group forms its point's row itself) and the exchange's failure flag read beside the rows instead of in
# front of them -- y1 against y0 (the same tree with kIcpTermsInB off).  The GPU suite on y1's tree, then the same-box A/B.
# Usage (through gpurun): TAG=r06_y bash scripts/gpu_r6_y.sh
set -u
T="${TAG:-r06_y}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh y0 y1 > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
