#!/bin/bash
# Short GPU round: subset of parity tests, bench (+variants), kernel trace, ICP probes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "${KICP_TEST_FILTER:-async or kitti_like or align_points or golden_seq or map_update}" 2>&1 | tail -15 ) > gpurun_out/pytest_gpu_subset.log
( timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err )
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r -- python bench.py --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err )
rm -f gpurun_out/bench_variants.txt
for v in ${KICP_VARIANTS:-"--opt icp_groups=8"}; do
  echo "== $v" >> gpurun_out/bench_variants.txt
  ( timeout 300 python bench.py --no-cpu-baseline $v >> gpurun_out/bench_variants.txt 2>/dev/null )
done
( timeout 300 python scripts/icp_probe.py > gpurun_out/icp_probe.txt 2>&1 )
( timeout 300 python scripts/icp_probe.py icp_groups=8 > gpurun_out/icp_probe_g8.txt 2>&1 )
tail -5 gpurun_out/pytest_gpu_subset.log; cat gpurun_out/bench_a.json; cat gpurun_out/bench_variants.txt | cut -c1-400
