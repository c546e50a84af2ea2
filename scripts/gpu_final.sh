#!/bin/bash
# Round-end GPU call: full parity suite, bench line, kernel stats, PMC traffic (+ calibration),
# multi-rank plumbing of bench.py on one GPU (gloo), the 1M-point configuration.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then
( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err )
fi
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err )
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o r -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2> gpurun_out/pmc_fetch.err )
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o r -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2> gpurun_out/pmc_write.err )
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal_fetch -o r -- python scripts/pmc_calibrate.py > /dev/null 2> gpurun_out/pmc_cal.err )
( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_cal_write -o r -- python scripts/pmc_calibrate.py > /dev/null 2>> gpurun_out/pmc_cal.err )
for d in pmc_fetch pmc_write pmc_cal_fetch pmc_cal_write; do
  f=$(find gpurun_out/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python scripts/pmc_summary.py "$f" > gpurun_out/$d.txt 2>&1
  rm -rf gpurun_out/$d
done
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --device 0 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank.err )
( timeout 400 python bench.py --workload livox --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_livox.json 2> gpurun_out/bench_livox.err )
( timeout 300 python bench.py --workload mulran --no-cpu-baseline > gpurun_out/bench_mulran.json 2> gpurun_out/bench_mulran.err )
( timeout 300 python scripts/icp_probe.py > gpurun_out/icp_probe.txt 2>&1 )
tail -14 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.json; echo; cut -c1-400 gpurun_out/bench_2rank_gloo.json; tail -3 gpurun_out/bench_2rank.err; cut -c1-900 gpurun_out/bench_livox.json; tail -3 gpurun_out/bench_livox.err; cut -c1-300 gpurun_out/bench_mulran.json; head -5 gpurun_out/pmc_fetch.txt gpurun_out/pmc_cal_fetch.txt gpurun_out/pmc_cal_write.txt
