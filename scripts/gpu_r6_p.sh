#!/bin/bash
# Round 6, session p: the GPU suite on the tree with the watcher-less second hop (+ scan prefetch, thread-per-query phase C in
# batches of four), then same-box A/B: n1 (session n's library) / cur / p1 (cur with s_sleep 1 instead of 2 in the exchange's
# polls), the 1M-point line of n1 and cur, and the probe of cur (later iterations: the list build's share of a search).
# Usage (through gpurun): TAG=r06_p bash scripts/gpu_r6_p.sh
set -u
T="${TAG:-r06_p}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
TAG=$T REPS=${REPS:-2} bash scripts/gpu_ab_variants.sh ${VARIANTS:-n1 cur p1}
for v in n1 cur; do
  ( KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_$v.so timeout 300 python bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_livox100_$v.json 2>/dev/null )
done
python - <<PY
import json
for v in ("n1", "cur"):
    try:
        d = json.loads(open("$O/${T}_bench_livox100_%s.json" % v).read().strip().splitlines()[-1])
        print("livox100", v, round(d["value"], 1), d["roofline"]["frac"], d.get("icp_last_launch"))
    except Exception as e:
        print("livox100", v, "failed", e)
PY
grep "searches by scan list" $O/${T}_icp_probe_cur.txt
