#!/bin/bash
# Round 6, first box session on round 5's HEAD: what the review asked for before any kernel work --
#   (1) SQ_* counter passes over k_icp on the steady bench command (where do a search's cycles go),
#   (2) the r04_au sweep (forced thread-per-query form on the KITTI-like scene) three times on the fixed release build,
#   (3) baseline bench lines + in-kernel probe + timeline of this box,
#   (4) LAST (it replaces the library of the box's copy): the whole GPU suite, without -x, and two bench lines on the
#       bounds-asserting build (make DEBUG_BOUNDS=1).
# Usage (through gpurun): TAG=r06_a bash scripts/gpu_r6_baseline.sh
set -u
T="${TAG:-r06_a}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' ) > $O/${T}_counters_available.txt
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
# (2) the sweep that faulted once in round 4
for rep in 1 2 3; do
  ( timeout 400 python scripts/opt_sweep.py icp_wide 0,1,0,1 outer=icp_wide_prefill:0,8 frames=60 2>&1 | tail -12 ) > $O/${T}_kitti_wide_sweep_$rep.txt
done
# (3) this box's baseline
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_20_5.json 2> $O/${T}_bench_20_5.err
timeout 400 python3 bench.py --gpus 1 --steps 200 --warmup 10 > $O/${T}_bench_200_10.json 2> $O/${T}_bench_200_10.err
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
( STEPS=40 timeout 200 bash scripts/timeline.sh > $O/${T}_timeline.txt 2>&1 )
# (4) the bounds-asserting build
make -C kiss-icp_amd/csrc clean > /dev/null
( make -C kiss-icp_amd/csrc -j8 DEBUG_BOUNDS=1 2>&1 | grep -E "error|Error" ; make -C kiss-icp_amd/cpp 2>&1 | grep -E "error|Error"; make -C tests/cpp 2>&1 | grep -E "error|Error" ) > $O/${T}_build_debug.log
( timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -40 ) > $O/${T}_pytest_debug.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_debug_20_5.json 2> $O/${T}_bench_debug_20_5.err
timeout 400 python bench.py --workload livox --steps 30 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_debug_livox.json 2> $O/${T}_bench_debug_livox.err
for f in $O/${T}_sq_*.txt $O/${T}_kitti_wide_sweep_*.txt $O/${T}_pytest_debug.log; do echo "== $f"; tail -12 $f; done
for f in $O/${T}_bench_20_5 $O/${T}_bench_200_10 $O/${T}_bench_debug_20_5 $O/${T}_bench_debug_livox; do echo "== $f"; tail -c 400 $f.json; tail -3 $f.err; done
true
