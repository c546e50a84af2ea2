#!/bin/bash
# The round's closing GPU call: everything the documentation quotes, in one box session, every piece under its own timeout.
# Usage (through gpurun): TAG=r05_final bash scripts/gpu_final.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T="${TAG:-r05_final}"
O=gpurun_out
# the driver's exact suite command as the FIRST process of the box (one more fresh-lease run), then the smoke
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=8 2>&1 | tail -${PYTEST_TAIL:-30} ) > $O/${T}_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > $O/${T}_smoke.log
# STOP_ON_FAIL=1: a red suite ends the call here (GPU minutes are for the fix, not for profiles of a wrong kernel)
if [ "${STOP_ON_FAIL:-0}" = 1 ] && ! grep -q " passed" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
if [ "${STOP_ON_FAIL:-0}" = 1 ] && grep -q " failed" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
# HBM traffic (PMC), separate passes, corrected by a known-size copy on this box
for wl in kitti $([ "${LEAN:-0}" = 1 ] || echo livox); do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/${T}_pmc_${wl}_${ctr}
    st=60; wu=10; [ $wl = livox ] && { st=30; wu=4; }
    ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --no-extras --steps $st --warmup $wu --gen-procs 1 > /dev/null 2> $GRAFT_REPO_ROOT/$d.err )
    f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
    rm -rf $d
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$O/${T}_pmc_cal_${ctr}
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python $GRAFT_REPO_ROOT/scripts/pmc_calibrate.py > /dev/null 2> $GRAFT_REPO_ROOT/$d.err )
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
  rm -rf $d
done
# ... folded into profiles/pmc_traffic.json HERE, so that the bench lines below quote this session's traffic (a copy comes back under gpurun_out/)
for wl in kitti $([ "${LEAN:-0}" = 1 ] || echo livox); do
  [ -s $O/${T}_pmc_${wl}_FETCH_SIZE.txt ] && [ -s $O/${T}_pmc_cal_FETCH_SIZE.txt ] && python scripts/pmc_to_json.py $O/${T}_pmc_${wl}_FETCH_SIZE.txt $O/${T}_pmc_${wl}_WRITE_SIZE.txt $O/${T}_pmc_cal_FETCH_SIZE.txt $O/${T}_pmc_cal_WRITE_SIZE.txt $wl profiles/pmc_traffic.json > $O/${T}_pmc_to_json_${wl}.log 2>&1
done
cp profiles/pmc_traffic.json $O/${T}_pmc_traffic.json
# the driver's own command, first thing a fresh process does on the device
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_20_5.json 2> $O/${T}_bench_20_5.err
# the steady-state map
timeout 400 python3 bench.py --gpus 1 --steps 200 --warmup 10 > $O/${T}_bench_200_10.json 2> $O/${T}_bench_200_10.err
# kernel trace + stats of the same command (no extras: one pipeline)
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/${T}_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/${T}_prof.err )
f=$(find $O/${T}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv
rm -rf $O/${T}_prof
# ... of the driver's own command (20 / 5) and of the 1M-point configuration's 100-frame run
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/${T}_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/${T}_bench_20_5_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/${T}_prof_20_5.err )
f=$(find $O/${T}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats_20_5.csv
rm -rf $O/${T}_prof
( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/${T}_prof -o r -- python $GRAFT_REPO_ROOT/bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/${T}_bench_livox100_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/${T}_prof_livox.err )
f=$(find $O/${T}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats_livox100.csv
rm -rf $O/${T}_prof
# the other configurations (parity-test cases, not the bench line)
timeout 300 python3 bench.py --workload mulran --steps 60 --warmup 10 --no-extras > $O/${T}_bench_mulran.json 2> $O/${T}_bench_mulran.err
timeout 300 python3 bench.py --workload kitti-street --steps 100 --warmup 10 --no-extras > $O/${T}_bench_street.json 2> $O/${T}_bench_street.err
timeout 500 python3 bench.py --workload livox --steps 100 --warmup 4 --no-extras > $O/${T}_bench_livox100.json 2> $O/${T}_bench_livox100.err
timeout 300 python3 bench.py --gpus 2 --device 0 --steps 20 --warmup 5 > $O/${T}_bench_2streams_1gpu.json 2> $O/${T}_bench_2streams_1gpu.err
# eight streams stacked on the one GPU through the host communicator: plumbing of the N = 8 path, and what it costs the HOST (host_cpu in the line)
timeout 400 python3 bench.py --gpus 8 --device 0 --steps 20 --warmup 5 > $O/${T}_bench_8streams_1gpu.json 2> $O/${T}_bench_8streams_1gpu.err
timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --device 0 > $O/${T}_bench_2rank_gloo.json 2> $O/${T}_bench_2rank_gloo.err
# in-kernel phase timers and the kernel timeline of two frames
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
timeout 400 python scripts/icp_probe.py livox=1 frames=100 > $O/${T}_icp_probe_livox100.txt 2>&1
( cd /tmp; cd $GRAFT_REPO_ROOT; STEPS=40 timeout 200 bash scripts/timeline.sh > $O/${T}_timeline.txt 2>&1 )
for f in $O/${T}_smoke.log $O/${T}_pytest_gpu.log; do echo "== $f"; tail -6 $f; done
for f in 20_5 200_10 mulran street livox100 2streams_1gpu 8streams_1gpu 2rank_gloo; do python3 - <<PY
import json
try:
    d = json.loads(open("$O/${T}_bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"], 1), d.get("speedup_vs_cpu"), d.get("roofline", {}).get("frac"), d.get("pose_error_vs_cpu"), d.get("n_gpus"))
except Exception as e:
    print("$f FAILED", e)
PY
done
true
