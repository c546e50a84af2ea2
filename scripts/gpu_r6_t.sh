#!/bin/bash
# Round 6, session t: phase B of the group form -- t0 as session q left it, t1 the first eight searches of a workgroup on eight
# different waves (kIcpSpreadSearches), t2 = t1 + the searched query's point slot and record as register copies, the list
# build handing base / length back in registers.  The GPU suite on t2's tree, then the same-box A/B and the probes.
# Usage (through gpurun): TAG=r06_t bash scripts/gpu_r6_t.sh
set -u
T="${TAG:-r06_t}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-2} bash scripts/gpu_ab_variants.sh t0 t1 t2 > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
grep "searches by scan list" $O/${T}_icp_probe_t0.txt $O/${T}_icp_probe_t2.txt
