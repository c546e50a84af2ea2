#!/bin/bash
# Round 6, session r: the thread-per-query form's options on the bench's own 1M-point command (100 frames), after session q's
# sweeps on a young map (prefill 8: -8 %, promote from the first iteration: -9 % per launch): one at a time and together.
# Usage (through gpurun): TAG=r06_r bash scripts/gpu_r6_r.sh
set -u
T="${TAG:-r06_r}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
run() {  # name, options...
  local name=$1; shift
  local opts=""; for o in "$@"; do opts="$opts --opt $o"; done
  ( timeout 300 python bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras $opts > $O/${T}_bench_livox100_$name.json 2>/dev/null )
  python - <<PY
import json
try:
    d = json.loads(open("$O/${T}_bench_livox100_$name.json").read().strip().splitlines()[-1])
    print("%-28s %7.1f scans/s  roofline %.4f  %s" % ("$name", d["value"], d["roofline"]["frac"], d.get("icp_last_launch")))
except Exception as e:
    print("$name failed", e)
PY
}
run base
run prefill8 icp_wide_prefill=8
run promote0 icp_wide_promote_from=0
run prefill8_promote0 icp_wide_prefill=8 icp_wide_promote_from=0
run prefill8_promote0_round8 icp_wide_prefill=8 icp_wide_promote_from=0 icp_wide_per_round=8
run all4 icp_wide_prefill=8 icp_wide_promote_from=0 icp_wide_per_round=8 icp_wide_group_max=256
run base_again
( timeout 300 python bench.py --workload mulran --steps 60 --no-cpu-baseline --no-extras > $O/${T}_bench_mulran.json 2>/dev/null )
python -c "
import json
d = json.loads(open('$O/${T}_bench_mulran.json').read().strip().splitlines()[-1]); print('mulran', round(d['value'], 1))"
