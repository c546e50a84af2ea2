#!/bin/bash
# A/B bench runs: default library and variant builds under kiss-icp_amd/csrc/variants/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.txt
( echo skipped ) > gpurun_out/pytest_ab.log
run() { echo "== $1" >> gpurun_out/ab.txt; shift; ( "$@" >> gpurun_out/ab.txt 2>/dev/null ); }
run "default" timeout 200 python bench.py --no-cpu-baseline
run "default again" timeout 200 python bench.py --no-cpu-baseline
for so in kiss-icp_amd/csrc/variants/*.so; do
  run "$so" env KICP_LIB=$PWD/$so timeout 200 python bench.py --no-cpu-baseline
done

cat gpurun_out/pytest_ab.log
python - <<'PY'
import json
for l in open('gpurun_out/ab.txt'):
    l=l.strip()
    if l.startswith('=='): print(l); continue
    if l.startswith('{'):
        d=json.loads(l); print('   scans/s %.1f ms/step %.4f ms/iter %.5f icp ms/launch %.4f'%(d['value'],d['ms_per_step'],d['ms_per_icp_iter'],d.get('roofline',{}).get('ms_per_launch',0)))
PY
