#!/bin/bash
# One option on / off, alternating, REPS times each, on the driver's command (20 / 5) and the steady-state one (200 / 10):
# throughput, the longest host call and the longest device gap of every run.  Usage: TAG=r05_t OPT=staging_numa bash scripts/gpu_opt_reps.sh
set -u
T="${TAG:-r05_reps}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OPT="${OPT:-staging_numa}"; A="${A:-1}"; B="${B:-0}"
for rep in $(seq 1 ${REPS:-4}); do
  for v in $A $B; do
    timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_b20_${OPT}${v}_r${rep}.json 2> $O/${T}_b20_${OPT}${v}_r${rep}.err
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_b200_${OPT}${v}_r${rep}.json 2> $O/${T}_b200_${OPT}${v}_r${rep}.err
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        hs = d.get("host_side") or {}
        print(f.split("/")[-1], round(d["value"], 1), "scans/s  max_call_ms", round(hs.get("max_call_ms", 0), 3), "max_gap_ms", round(hs.get("max_device_gap_ms", 0), 3), "stage_ms", round(hs.get("stage_ms", 0), 2), "busy", round((d.get("host_cpu") or {}).get("cpu_busy", 0), 2), "helpers", hs.get("staging_helpers"), "bound", hs.get("helpers_bound"))
    except Exception as e:
        print(f, "FAILED", e)
PY
