#!/bin/bash
# GPU round for the 16-lane ICP variant: parity subset under both group widths, bench + probes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
F="align or golden or kitti_like or async or degenerate or max_iterations"
( KICP_ICP_GROUP_LANES=16 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$F" 2>&1 | tail -15 ) > gpurun_out/pytest_gl16.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$F" 2>&1 | tail -15 ) > gpurun_out/pytest_gl32.log
( timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_gl32.json 2> gpurun_out/bench_a.err )
( timeout 300 python bench.py --no-cpu-baseline --opt icp_group_lanes=16 > gpurun_out/bench_gl16.json 2>> gpurun_out/bench_a.err )
( timeout 300 python scripts/icp_probe.py > gpurun_out/icp_probe_gl32.txt 2>&1 )
( timeout 300 python scripts/icp_probe.py icp_group_lanes=16 > gpurun_out/icp_probe_gl16.txt 2>&1 )
tail -4 gpurun_out/pytest_gl16.log; tail -3 gpurun_out/pytest_gl32.log; cut -c1-330 gpurun_out/bench_gl32.json; echo; cut -c1-330 gpurun_out/bench_gl16.json; echo; head -8 gpurun_out/icp_probe_gl16.txt
