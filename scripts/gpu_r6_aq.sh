#!/bin/bash
# Round 6, session aq: two options whose defaults date from slower kernels, against today's: a point's base weight in the run
# partition ("icp_weight_base" 128: 64 / 256) and k_map_apply's workgroup size ("map_apply_threads" 512: 256 / 1024); the steady
# bench line, two interleaved repetitions (options, no rebuild).
# Usage (through gpurun): TAG=r06_aq bash scripts/gpu_r6_aq.sh
set -u
T="${TAG:-r06_aq}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
out=$O/${T}_options.txt; : > $out
for r in 1 2; do
  for v in "icp_weight_base=128" "icp_weight_base=64" "icp_weight_base=256" "map_apply_threads=256" "map_apply_threads=1024" "icp_weight_dense_min=100" "icp_weight_dense_min=400"; do
    timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); l = d.get('icp_last_launch', {})
print('%-26s rep $r  %7.1f scans/s  k_icp/iter %.2f us  first %.1f later %.2f  ms/launch %.4f  roofline %.4f' % ('$v', d['value'], d['ms_per_icp_iter'] * 1000, l.get('first_iteration_us', 0), l.get('later_iterations_us', 0), d['roofline']['ms_per_launch'], d['roofline']['frac']))" >> $out
  done
done
cat $out
