#!/bin/bash
# Round 6 A/B session: GPU suite first (STOP_ON_FAIL), then one kernel option on / off interleaved on the steady and the driver's
# command, the in-kernel probe, the 1M-point configuration (bench + kernel trace + PMC write traffic: the spills).
# Usage (through gpurun): TAG=r06_b OPT=icp_group_stable bash scripts/gpu_r6_ab.sh
set -u
T="${TAG:-r06_b}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OPT="${OPT:-icp_group_stable}"
( timeout 1150 python -m pytest tests/ -x -q -m gpu --durations=8 ${PYTEST_ARGS:-} 2>&1 | tail -30 ) > $O/${T}_pytest_gpu.log
if ! grep -q " passed" $O/${T}_pytest_gpu.log || grep -q " failed\| error" $O/${T}_pytest_gpu.log; then cat $O/${T}_pytest_gpu.log; [ "${STOP_ON_FAIL:-1}" = 1 ] && exit 1; fi
for rep in 1 2; do
  for v in 1 0; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench_${OPT}${v}_r${rep}.json 2> $O/${T}_bench_${OPT}${v}_r${rep}.err
  done
done
for v in 1 0; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt $OPT=$v > $O/${T}_bench20_${OPT}${v}.json 2> $O/${T}_bench20_${OPT}${v}.err
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_20_5.json 2> $O/${T}_bench_20_5.err
timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_steady.txt 2>&1
timeout 300 python bench.py --workload mulran --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $O/${T}_bench_mulran.json 2> $O/${T}_bench_mulran.err
if [ "${LIVOX:-1}" = 1 ]; then
  timeout 500 python3 bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $O/${T}_bench_livox100.json 2> $O/${T}_bench_livox100.err
  ( cd /tmp; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${T}_prof -o r -- python $R/bench.py --workload livox --steps 100 --warmup 4 --no-cpu-baseline --no-extras > $R/$O/${T}_bench_livox100_under_rocprof.json 2> $R/$O/${T}_prof_livox.err )
  f=$(find $O/${T}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats_livox100.csv
  rm -rf $O/${T}_prof
  for ctr in WRITE_SIZE FETCH_SIZE; do
    d=$O/${T}_pmc_livox_${ctr}
    ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$d -o r -- python $R/bench.py --workload livox --no-cpu-baseline --no-extras --steps 30 --warmup 4 --gen-procs 1 > /dev/null 2> $R/$d.err )
    f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python scripts/pmc_summary.py "$f" > $d.txt 2>&1
    rm -rf $d
  done
  timeout 400 python scripts/icp_probe.py livox=1 frames=100 > $O/${T}_icp_probe_livox100.txt 2>&1
fi
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${T}_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "scans/s", round(d["ms_per_step"], 4), "ms/step", "icp ms/launch", round(d["roofline"]["ms_per_launch"], 4), "us/iter", round(1e3 * d["ms_per_icp_iter"], 2), "frac", round(d["roofline"]["frac"], 4), d.get("icp_last_launch"), {k: round(d[k]["scans_per_s"]) for k in ("sync_per_frame", "sync_with_outputs") if k in d})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 $O/${T}_pytest_gpu.log
head -32 $O/${T}_icp_probe_steady.txt
grep k_icp $O/${T}_pmc_livox_*.txt $O/${T}_kernel_stats_livox100.csv 2>/dev/null | cut -c1-220
true
