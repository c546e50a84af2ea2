#!/bin/bash
# Round 6, session ap: the CUs k_icp leaves to the front stages of the next frame ("icp_reserve_cus", 32 since round 3) against the
# round's shorter registration: 24 / 32 / 40 / 48, the steady bench line, three interleaved repetitions (no rebuild: an option).
# Usage (through gpurun): TAG=r06_ap bash scripts/gpu_r6_ap.sh
set -u
T="${TAG:-r06_ap}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
out=$O/${T}_reserve_cus.txt; : > $out
for r in 1 2 3; do
  for v in 24 32 40 48; do
    timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt icp_reserve_cus=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); l = d.get('icp_last_launch', {})
print('reserve %2d rep $r  %7.1f scans/s  workgroups %d  k_icp/iter %.2f us  later %.2f  ms/launch %.4f  roofline %.4f' % ($v, d['value'], d['config']['icp_workgroups'], d['ms_per_icp_iter'] * 1000, l.get('later_iterations_us', 0), d['roofline']['ms_per_launch'], d['roofline']['frac']))" >> $out
  done
done
cat $out
