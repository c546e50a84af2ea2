#!/bin/bash
# A/B bench runs: the default library (twice, for the noise floor) and every variant build found under
# kiss-icp_amd/csrc/variants/ (selected through KICP_LIB).  KICP_AB_TESTS="<pytest -k filter>" runs a
# parity subset first; KICP_AB_BENCH_ARGS adds bench.py arguments (e.g. "--workload livox --steps 8 --warmup 3").
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.txt gpurun_out/pytest_ab.log
if [ -n "${KICP_AB_TESTS:-}" ]; then
  ( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$KICP_AB_TESTS" 2>&1 | tail -3 ) > gpurun_out/pytest_ab.log
  cat gpurun_out/pytest_ab.log
fi
run() { echo "== $1" >> gpurun_out/ab.txt; shift; ( "$@" >> gpurun_out/ab.txt 2>/dev/null ); }
run "default" timeout 300 python bench.py --no-cpu-baseline ${KICP_AB_BENCH_ARGS:-}
run "default again" timeout 300 python bench.py --no-cpu-baseline ${KICP_AB_BENCH_ARGS:-}
for so in kiss-icp_amd/csrc/variants/*.so; do
  [ -e "$so" ] || continue
  run "$so" env KICP_LIB=$PWD/$so timeout 300 python bench.py --no-cpu-baseline ${KICP_AB_BENCH_ARGS:-}
done
python - <<'PY'
import json
for l in open('gpurun_out/ab.txt'):
    l = l.strip()
    if l.startswith('=='):
        print(l)
    elif l.startswith('{'):
        d = json.loads(l)
        last = d.get('icp_last_launch', {})
        print('   scans/s %.1f  ms/step %.4f  icp us/launch %.1f  first iteration %.1f us  later %.2f us' % (
            d['value'], d['ms_per_step'], 1e3 * d.get('roofline', {}).get('ms_per_launch', 0),
            last.get('first_iteration_us', 0), last.get('later_iterations_us', 0)))
PY
