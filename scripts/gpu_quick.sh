#!/bin/bash
# Short GPU round: subset of parity tests, ICP probe, bench variants.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "${KICP_TEST_FILTER:-align or golden or kitti_like or map_ or closest}" 2>&1 | tail -15 ) > gpurun_out/pytest_gpu_subset.log
( timeout 300 python scripts/icp_probe.py > gpurun_out/icp_probe.txt 2>&1 )
( timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err )
for v in "--opt map_apply_threads=256" "--opt map_apply_threads=1024" "--icp-ppg 2" ${KICP_EXTRA_VARIANTS:-}; do
  echo "== $v" >> gpurun_out/bench_variants.txt
  ( timeout 300 python bench.py --no-cpu-baseline $v >> gpurun_out/bench_variants.txt 2>/dev/null )
done
tail -5 gpurun_out/pytest_gpu_subset.log; head -20 gpurun_out/icp_probe.txt; cat gpurun_out/bench_a.json
