#!/bin/bash
# same-box A/B of library variants (scripts/build_variant.sh): REPS interleaved rounds of the steady bench line per variant.
#   TAG=r03_x REPS=2 scripts/ab_bench.sh head cur sel        -> gpurun_out/<TAG>_ab.txt
TAG=${TAG:-ab}; REPS=${REPS:-2}; ARGS=${ARGS:---steps 200 --warmup 10 --no-cpu}
out=gpurun_out/${TAG}_ab.txt; mkdir -p gpurun_out; : > $out
for r in $(seq $REPS); do
  for v in "$@"; do
    lib=$PWD/kiss-icp_amd/csrc/variants/libkicp_$v.so
    KICP_LIB=$lib timeout ${TMO:-120} python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); l = d.get('icp_last_launch', {})
print('%-8s rep $r  %7.1f scans/s  k_icp/iter %.2f us  last launch: first %.1f later %.2f total %.1f  roofline %.4f  local-hop groups %s' % ('$v', d['value'], d['ms_per_icp_iter'] * 1000, l.get('first_iteration_us', 0), l.get('later_iterations_us', 0), l.get('total_us', 0), d['roofline']['frac'], d['config'].get('icp_local_hop_groups', '-')))" >> $out
  done
done
cat $out
