#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/kiss-icp_amd/csrc/variants/libkicp_contig.so
( KICP_LIB=$V timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "align_points or ties or 16_lane or golden_align" 2>&1 | tail -3 ) > gpurun_out/pytest_last.log
( timeout 100 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2>/dev/null )
( KICP_LIB=$V timeout 100 python bench.py --no-cpu-baseline > gpurun_out/bench_contig.json 2>/dev/null )
cat gpurun_out/pytest_last.log
python - <<'PY'
import json
for f in ('bench_default','bench_contig'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, '%.1f scans/s  icp %.1f us/launch  first it %.1f us later %.2f us'%(d['value'], 1e3*d['roofline']['ms_per_launch'], d['icp_last_launch']['first_iteration_us'], d['icp_last_launch']['later_iterations_us']))
PY
