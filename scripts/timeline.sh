#!/bin/bash
# kernel timeline of the last frames of a short bench run (rocprofv3 --kernel-trace): where a frame's time goes
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o r -- python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-40} ${BENCH_ARGS:-} > /dev/null 2>&1
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
icp = [i for i, r in enumerate(rows) if 'k_icp' in r['Kernel_Name']]
a, b = icp[-4], icp[-2]
for r in rows[a - 1:b + 1]:
    n = r['Kernel_Name'].split('(')[0].replace('kicp::', '').replace('void ', '')
    print("%10.2f %10.2f %8.2f  q%s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                           (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), n[:60]))
PY
rm -rf gpurun_out/tl
