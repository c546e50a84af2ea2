#!/bin/bash
# KITTI + 1M-point A/B of the default library (parity subset first).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.txt
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "align_points or async or ties" 2>&1 | tail -3 ) > gpurun_out/pytest_ab.log
run() { echo "== $1" >> gpurun_out/ab.txt; shift; ( "$@" >> gpurun_out/ab.txt 2>/dev/null ); }
run "kitti" timeout 200 python bench.py --no-cpu-baseline
run "livox" timeout 300 python bench.py --workload livox --steps 8 --warmup 3 --no-cpu-baseline
for so in kiss-icp_amd/csrc/variants/*.so; do
  [ -e "$so" ] || continue
  run "kitti $so" env KICP_LIB=$PWD/$so timeout 200 python bench.py --no-cpu-baseline
  run "livox $so" env KICP_LIB=$PWD/$so timeout 300 python bench.py --workload livox --steps 8 --warmup 3 --no-cpu-baseline
done
cat gpurun_out/pytest_ab.log
python - <<'PY'
import json
for l in open('gpurun_out/ab.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print('   scans/s %.1f ms/step %.4f ms/iter %.5f icp ms/launch %.4f frac %.4f'%(d['value'],d['ms_per_step'],d['ms_per_icp_iter'],d.get('roofline',{}).get('ms_per_launch',0),d.get('roofline',{}).get('frac',0)))
PY
