#!/bin/bash
# Where do multi-millisecond stalls in the host-side staging copy come from?  (r03: the driver's 20/5 bench line)
mkdir -p gpurun_out
O=gpurun_out/${TAG:-r03_b}_stall_diag.txt
{
echo "== cgroup / numa"; cat /sys/fs/cgroup/cpu.max 2>&1; cat /sys/fs/cgroup/cpu.stat 2>&1 | head -8
echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>&1)"; echo "thp: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1)"
grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status
which numactl taskset; numactl -H 2>&1 | head -12
for st in 0 1 3 7; do
  KICP_HOST_TRACE=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --opt staging_threads=$st > gpurun_out/${TAG:-r03_b}_st$st.json 2> /tmp/st$st.err
  echo "== staging_threads=$st: $(python -c "import json;d=json.load(open('gpurun_out/${TAG:-r03_b}_st$st.json'));print(d['value'], d['host_side'])")"
  grep "kicp host" /tmp/st$st.err | awk '{for(i=1;i<=NF;i++) if($i=="call"){c=$(i+1)}; if (c>1.0) print}' | head -8
done
echo "== staging_threads=3 pinned to cpus 0-15"
KICP_HOST_TRACE=1 taskset -c 0-15 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --gen-procs 16 > gpurun_out/${TAG:-r03_b}_pin.json 2> /tmp/pin.err
python -c "import json;d=json.load(open('gpurun_out/${TAG:-r03_b}_pin.json'));print(d['value'], d['host_side'])"
grep "kicp host" /tmp/pin.err | awk '{for(i=1;i<=NF;i++) if($i=="call"){c=$(i+1)}; if (c>1.0) print}' | head -8
cat /sys/fs/cgroup/cpu.stat 2>&1 | head -8
} > $O 2>&1
cat $O
