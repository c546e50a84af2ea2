#!/bin/bash
# Confirmation of a tree: the driver's suite command as the first process of the lease, the smoke, the driver's bench command
# and the steady-state one.  Usage (through gpurun): TAG=r05_v bash scripts/gpu_confirm.sh
set -u
T="${TAG:-r05_confirm}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=8 2>&1 | tail -14 ) > $O/${T}_pytest_gpu.log
cp $O/test_cpp_api_last.log $O/${T}_test_cpp_api_laps.log 2>/dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 ) > $O/${T}_smoke.log
if [ "${SKIP_BENCH:-0}" != 1 ]; then
timeout 300 python3 bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
timeout 400 python3 bench.py --gpus 1 --steps 200 --warmup 10 > $O/${T}_bench_200_10.json 2> $O/${T}_bench_200_10.err
fi
cat $O/${T}_pytest_gpu.log; grep "^\[" $O/test_cpp_api_front.log $O/${T}_test_cpp_api_laps.log; cat $O/${T}_smoke.log
python3 - <<PY
import json
for f in ("default", "200_10"):
    try:
        d = json.loads(open("$O/${T}_bench_%s.json" % f).read().strip().splitlines()[-1])
        hs = d.get("host_side") or {}
        print(f, round(d["value"], 1), "scans/s steps", d["steps"], "warmup", d["warmup"], "speedup", d.get("speedup_vs_cpu"), "roofline", d["roofline"]["frac"], "host_cpu", d.get("host_cpu"), "max_call_ms", hs.get("max_call_ms"), "max_gap", hs.get("max_device_gap_ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
