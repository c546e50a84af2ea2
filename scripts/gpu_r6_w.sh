#!/bin/bash
# Round 6, session w: the thread-per-query form -- w0 the closing state, w1 the window test from one read of the record and the
# table's chains four slots per round trip (wide_entry, the search's long-chain loop).  The GPU suite on w1's tree, then the
# 100-frame 1M-point bench line of both, interleaved, and w1's probe.
# Usage (through gpurun): TAG=r06_w bash scripts/gpu_r6_w.sh
set -u
T="${TAG:-r06_w}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/${T}_pytest_gpu.log
grep -E "passed|failed" $O/${T}_pytest_gpu.log
for r in 1 2; do for v in w0 w1; do
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
( KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_w1.so timeout 400 python scripts/icp_probe.py livox=1 frames=100 > $O/${T}_icp_probe_livox100_w1.txt 2>&1 )
grep -E "wave chains|wave lookups|^  0 " $O/${T}_icp_probe_livox100_w1.txt | head -4
