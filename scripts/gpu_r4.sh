#!/bin/bash
# Round-4 GPU call: STEPS="t_wide t_all livox livox_group probe_livox kitti kitti20 ..." (any subset); TAG names the outputs.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS="${STEPS:-t_wide livox}"
TAG="${TAG:-r04_x}"
has() { [[ " $STEPS " == *" $1 "* ]]; }
o=gpurun_out/${TAG}
if has t_wide; then ( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 -k "thread_per_query or config5 or align or ties" 2>&1 | tail -30 ) > ${o}_pytest_wide.log; fi
if has t_all; then ( timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -40 ) > ${o}_pytest_gpu.log; fi
if has smoke; then ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 ) > ${o}_smoke.log; fi
if has livox; then ( timeout 500 python bench.py --workload livox --steps ${LIVOX_STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras ${LIVOX_ARGS:-} > ${o}_bench_livox.json 2> ${o}_bench_livox.err ); fi
if has livox_group; then ( timeout 500 python bench.py --workload livox --steps ${LIVOX_STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --opt icp_wide=0 > ${o}_bench_livox_group.json 2> ${o}_bench_livox_group.err ); fi
if has livox_p0; then ( timeout 500 python bench.py --workload livox --steps ${LIVOX_STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --opt icp_wide_prune=0 > ${o}_bench_livox_prune0.json 2> ${o}_bench_livox_prune0.err ); fi
if has livox_p1; then ( timeout 500 python bench.py --workload livox --steps ${LIVOX_STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --opt icp_wide_prune=1 > ${o}_bench_livox_prune1.json 2> ${o}_bench_livox_prune1.err ); fi
if has livox100; then ( timeout 900 python bench.py --workload livox --steps 100 --warmup 3 --no-extras ${LIVOX_ARGS:-} > ${o}_bench_livox100.json 2> ${o}_bench_livox100.err ); fi
if has probe_livox; then ( timeout 400 python scripts/icp_probe.py livox=1 ${PROBE_ARGS:-} > ${o}_icp_probe_livox.txt 2>&1 ); fi
if has probe; then ( timeout 400 python scripts/icp_probe.py ${PROBE_ARGS:-} > ${o}_icp_probe_steady.txt 2>&1 ); fi
if has kitti; then ( timeout 900 python bench.py ${BENCH_ARGS:-} > ${o}_bench_200_10.json 2> ${o}_bench_200_10.err ); fi
if has kitti20; then ( timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > ${o}_bench_20_5.json 2> ${o}_bench_20_5.err ); fi
if has kitti_wide; then ( timeout 600 python bench.py --no-cpu-baseline --no-extras --opt icp_wide=1 > ${o}_bench_kitti_wide.json 2> ${o}_bench_kitti_wide.err ); fi
if has mulran; then ( timeout 400 python bench.py --workload mulran --steps 60 --no-cpu-baseline --no-extras ${MULRAN_ARGS:-} > ${o}_bench_mulran.json 2> ${o}_bench_mulran.err ); fi
if has mulran_wide; then ( timeout 400 python bench.py --workload mulran --steps 60 --no-cpu-baseline --no-extras --opt icp_wide=1 > ${o}_bench_mulran_wide.json 2> ${o}_bench_mulran_wide.err ); fi
if has extra; then ( timeout ${EXTRA_TIMEOUT:-600} bash -c "${EXTRA_CMD}" > ${o}_extra.log 2>&1 ); fi
for f in ${o}_*.log ${o}_*.json ${o}_*.txt; do [ -f "$f" ] && { echo "== $f"; tail -c ${TAILC:-1800} "$f"; echo; }; done
for f in ${o}_*.err; do [ -s "$f" ] && { echo "== $f"; tail -5 "$f"; }; done
true
