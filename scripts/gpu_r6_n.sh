#!/bin/bash
# Round 6, session n: the exchange's three reduction stages with all addends in flight (icp_row_sum), phases A and C of the group
# form with their LDS reads up front, tile_find four slots per round trip -- against the last commit, same box:
# the GPU suite on the new library first (bitwise tests), then scripts/gpu_ab_variants.sh base n1 (both bench commands,
# interleaved, + the in-kernel probe of both).
# Usage (through gpurun): TAG=r06_n bash scripts/gpu_r6_n.sh
set -u
T="${TAG:-r06_n}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
tail -3 $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-3} bash scripts/gpu_ab_variants.sh ${VARIANTS:-base n1}
( KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_n1.so timeout 300 python bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_livox100_n1.json 2>/dev/null )
( KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_base.so timeout 300 python bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_livox100_base.json 2>/dev/null )
python - <<PY
import json
for v in ("base", "n1"):
    try:
        d = json.loads(open("$O/${T}_bench_livox100_%s.json" % v).read().strip().splitlines()[-1])
        print("livox100", v, round(d["value"], 1), d["roofline"]["frac"], d.get("icp_last_launch"))
    except Exception as e:
        print("livox100", v, "failed", e)
PY
