#!/bin/bash
# Round 6, session k: the whole GPU suite on the working tree (the driver's command), then the driver's bench line with its
# secondary keys (sync_with_outputs_cpp: the C++ RegisterFrame through kicp_pipeline_collect_outputs).
set -u
T="${TAG:-r06_k}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=5 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; exit 1; fi
for rep in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_20_5_r${rep}.json 2> $O/${T}_bench_20_5_r${rep}.err
done
timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline > $O/${T}_bench_200_10.json 2> $O/${T}_bench_200_10.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "frac", round(d["roofline"]["frac"], 4),
              {k: (round(d[k]["scans_per_s"]) if "scans_per_s" in d[k] else d[k]) for k in ("sync_per_frame", "sync_with_outputs", "sync_with_outputs_cpp", "device_resident", "host_float32_input") if k in d})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -4 $O/${T}_pytest_gpu.log
