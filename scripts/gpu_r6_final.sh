#!/bin/bash
# Round 6's closing call: scripts/gpu_final.sh (suite first, smoke, PMC traffic, bench lines, kernel traces, other configurations,
# plumbing runs, probes, timeline) and then what round 6 adds: SQ_* counter passes of k_icp on the closing code, the MulRan-like
# timeline (deskew: the front stages on the serial chain).
# Usage (through gpurun): TAG=r06_final bash scripts/gpu_r6_final.sh
set -u
T="${TAG:-r06_final}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$T STOP_ON_FAIL=1 bash scripts/gpu_final.sh || exit 1
pmc_pass() {  # name, counters...
  local name=$1; shift
  local d=$O/${T}_sq_$name
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$d -o r -- python $R/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10 --gen-procs 1 > /dev/null 2> $R/$d.err )
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
  rm -rf $d
}
pmc_pass a SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pmc_pass b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pmc_pass c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT
pmc_pass d GRBM_GUI_ACTIVE GRBM_COUNT
python scripts/sq_to_json.py "k_icp<false, false>" $O/${T}_sq_k_icp.json $O/${T}_sq_a.txt $O/${T}_sq_b.txt $O/${T}_sq_c.txt $O/${T}_sq_d.txt > /dev/null 2>&1
( STEPS=40 BENCH_ARGS="--workload mulran" timeout 200 bash scripts/timeline.sh > $O/${T}_timeline_mulran.txt 2>&1 )
head -40 $O/${T}_sq_k_icp.json | tail -20
tail -12 $O/${T}_timeline.txt
