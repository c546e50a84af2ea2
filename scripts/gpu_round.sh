#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel stats, ICP phase probe.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err )
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/bench_prof.json 2> gpurun_out/prof.err )
( timeout 300 python scripts/icp_probe.py > gpurun_out/icp_probe.txt 2>&1 )
tail -25 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
find gpurun_out/prof -name '*stats*' | head
