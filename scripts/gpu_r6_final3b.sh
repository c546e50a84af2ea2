#!/bin/bash
# After the closing session r06_final3: its MulRan-like line carried ONE host call of 4.5 ms (a 3.7 ms device gap in a 20 ms run --
# the busy-neighbour hiccup of profiles/r05_s_*), so the line is repeated three times on another lease, the driver's command twice.
# Usage (through gpurun): TAG=r06_final3b bash scripts/gpu_r6_final3b.sh
set -u
T="${TAG:-r06_final3b}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3; do timeout 300 python3 bench.py --workload mulran --steps 60 --warmup 10 --no-extras > $O/${T}_bench_mulran_r$r.json 2> /dev/null; done
for r in 1 2; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_20_5_r$r.json 2> /dev/null; done
python3 - <<PY
import json
for n in ("mulran_r1", "mulran_r2", "mulran_r3", "20_5_r1", "20_5_r2"):
    try:
        d = json.loads(open("$O/${T}_bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), round(d.get("speedup_vs_cpu", 0), 1), d["roofline"]["frac"], "max host call ms", round(d["host_side"]["max_call_ms"], 2), "max device gap ms", round(d["host_side"]["max_device_gap_ms"], 2))
    except Exception as e:
        print(n, "FAILED", e)
PY
