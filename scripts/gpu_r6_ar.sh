#!/bin/bash
# Round 6, session ar: k_map_apply's workgroup size ("map_apply_threads": 512 against 256) once more, three interleaved repetitions on
# both bench commands and the MulRan-like scene (session aq's two repetitions: 256 ahead by 0.3 - 1 %, inside the noise).
# Usage (through gpurun): TAG=r06_ar bash scripts/gpu_r6_ar.sh
set -u
T="${TAG:-r06_ar}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
out=$O/${T}_apply_threads.txt; : > $out
for args in "--steps 200 --warmup 10" "--steps 20 --warmup 5" "--workload mulran --steps 60 --warmup 10"; do
  for r in 1 2 3; do
    for v in 512 256; do
      timeout 120 python bench.py $args --no-cpu-baseline --no-extras --opt map_apply_threads=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); hs = d['host_side']
print('%-44s apply threads %4d rep $r  %7.1f scans/s  between registrations %.1f us' % ('$args', $v, d['value'], 1e3 * hs['device_gap_ms'] / max(1, hs['frames'] - 1)))" >> $out
    done
  done
done
cat $out
