#!/bin/bash
# Round 6, session c: the registration tests, then the stability shortcut (now also for queries without a scan list) on / off,
# what the timing events around k_icp cost (icp_timing 0), and the probe with its search counts.
set -u
T="${TAG:-r06_c}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -x -q -m gpu -k "not cold" 2>&1 | tail -8 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
for rep in 1 2; do
  for v in "icp_group_stable=1" "icp_group_stable=0" "icp_timing=0"; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $v > $O/${T}_bench_${v}_r${rep}.json 2> $O/${T}_bench_${v}_r${rep}.err
  done
done
for v in "icp_group_stable=1" "icp_group_stable=0" "icp_timing=0"; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt $v > $O/${T}_bench20_${v}.json 2> $O/${T}_bench20_${v}.err
done
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "us/iter", round(1e3 * d["ms_per_icp_iter"], 2), "frac", round(d["roofline"]["frac"], 4), d.get("icp_last_launch"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $O/${T}_pytest_gpu.log
head -62 $O/${T}_icp_probe_steady.txt
