#!/bin/bash
# One GPU-box call, assembled from pieces: WHAT="tests bench prof probe livox mulran pmc 2rank smoke" (any subset).
# Everything lands under gpurun_out/ (merged back by gpurun); the tail of each piece is echoed at the end.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${WHAT:-tests bench}"
TAG="${TAG:-x}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has smoke; then ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > gpurun_out/${TAG}_smoke.log; fi
if has tests; then ( timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --durations=10 ${TEST_ARGS:-} 2>&1 | tail -40 ) > gpurun_out/${TAG}_pytest_gpu.log; fi
if has bench; then ( timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ); fi
if has street; then ( timeout 600 python bench.py --workload kitti-street --no-extras ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_street.json 2> gpurun_out/${TAG}_bench_street.err ); fi
if has prof; then
  ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o r -- python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err )
  f=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv
  rm -rf gpurun_out/${TAG}_prof
fi
if has probe; then ( timeout 400 python scripts/icp_probe.py ${PROBE_ARGS:-} > gpurun_out/${TAG}_icp_probe.txt 2>&1 ); fi
if has livox; then ( timeout 600 python bench.py --workload livox --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_livox.json 2> gpurun_out/${TAG}_bench_livox.err ); fi
if has mulran; then ( timeout 400 python bench.py --workload mulran --steps 60 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench_mulran.json 2> gpurun_out/${TAG}_bench_mulran.err ); fi
if has pmc; then
  for wl in ${PMC_WORKLOADS:-kitti}; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=gpurun_out/${TAG}_pmc_${wl}_${ctr}
      # (--gen-procs 1: a worker pool torn down under rocprofv3 --pmc has hung the run before)
      ( timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o r -- python bench.py --workload $wl --no-cpu-baseline --no-extras --steps 12 --warmup 4 --gen-procs 1 > /dev/null 2> $d.err )
      f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
      rm -rf $d
    done
  done
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/${TAG}_pmc_cal_${ctr}
    ( timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o r -- python scripts/pmc_calibrate.py > /dev/null 2> $d.err )
    f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
    rm -rf $d
  done
fi
if has 2rank; then ( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --device 0 --workload kitti-street > gpurun_out/${TAG}_bench_2rank_gloo.json 2> gpurun_out/${TAG}_bench_2rank.err ); fi
if has rccl; then ( timeout 300 python scripts/rccl_probe.py > gpurun_out/${TAG}_rccl_probe.json 2> gpurun_out/${TAG}_rccl_probe.err ); fi
if has extra; then ( timeout ${EXTRA_TIMEOUT:-600} bash -c "${EXTRA_CMD}" > gpurun_out/${TAG}_extra.log 2>&1 ); fi
for f in gpurun_out/${TAG}_*.log gpurun_out/${TAG}_*.json gpurun_out/${TAG}_*.txt; do [ -f "$f" ] && { echo "== $f"; tail -c 2500 "$f"; echo; }; done
for f in gpurun_out/${TAG}_*.err; do [ -s "$f" ] && { echo "== $f"; tail -5 "$f"; }; done
true
