#!/bin/bash
# Round 6, session x: the window phases' dedup loops with the query's record read once (x1) against session w's library (w1):
# GPU suite on x1's tree, the 1M-point line of both interleaved, both KITTI-like bench commands of both.
# Usage (through gpurun): TAG=r06_x bash scripts/gpu_r6_x.sh
set -u
T="${TAG:-r06_x}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
for r in 1 2; do for v in w1 x1; do
  ( KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_$v.so timeout 300 python bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_livox100_${v}_r$r.json 2>/dev/null )
  python - <<PY
import json
try:
    d = json.loads(open("$O/${T}_bench_livox100_${v}_r$r.json").read().strip().splitlines()[-1])
    print("$v rep $r  %7.1f scans/s  roofline %.4f  k_icp %.3f ms/launch  %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d.get("icp_last_launch")))
except Exception as e:
    print("$v failed", e)
PY
done; done
TAG=$T REPS=2 bash scripts/gpu_ab_variants.sh w1 x1 > $O/${T}_ab_all.txt 2>&1
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
grep -E "bulk fill windows" $O/${T}_icp_probe_w1.txt $O/${T}_icp_probe_x1.txt | head
