#!/bin/bash
# Where does the FIRST GPU process of a fresh lease spend its minutes?  tests/cpp/test_cpp_api (everything before its batch
# entry takes 0.5 s; the batch entry's first ncclCommInitRank has taken 6 s on most leases and 54 - 295 s on a quarter of
# them) as the first process, RCCL's INFO log into a file, and a sampler (1 Hz) of the log's length, the process's I/O
# counters, and state / wait channel / kernel stack of every thread that is not running.  Then the same again.
# Usage (through gpurun): TAG=r05_x bash scripts/first_rccl_probe.sh
set -u
T="${TAG:-r05_rccl}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export KISS_ORACLE_THREADS=8   # (what pytest's conftest sets: the CPU checker must not oversubscribe the container)
for pass in first second; do
  L=$O/${T}_${pass}
  rm -f /tmp/rccl_$pass.log
  t0=$(date +%s.%N)
  ( NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV,GRAPH,NET NCCL_DEBUG_FILE=/tmp/rccl_$pass.log timeout 900 tests/cpp/test_cpp_api > ${L}_program.log 2>&1 ) &
  sleep 0.2
  pid=$(pgrep -n -x test_cpp_api)
  echo "pass $pass pid $pid" > ${L}_samples.txt
  while [ -n "$pid" ] && kill -0 $pid 2>/dev/null; do
    {
      echo "== t=$(python3 -c "import time;print('%.1f' % (time.time() - $t0))") rccl_log_lines=$(wc -l < /tmp/rccl_$pass.log 2>/dev/null || echo 0) $(grep -E "read_bytes|rchar" /proc/$pid/io 2>/dev/null | tr '\n' ' ') $(grep -E "RssFile" /proc/$pid/status 2>/dev/null | tr -s ' \t' ' ')"
      for t in /proc/$pid/task/*; do
        st=$(cut -d' ' -f3 $t/stat 2>/dev/null)
        [ "$st" = "R" ] && { echo "  $(cat $t/comm 2>/dev/null) R"; continue; }
        echo "  $(cat $t/comm 2>/dev/null) $st $(cat $t/wchan 2>/dev/null) | $(head -5 $t/stack 2>/dev/null | awk '{print $2}' | tr '\n' ' ')"
      done | sort | uniq -c
    } >> ${L}_samples.txt
    sleep 1
  done
  echo "pass $pass took $(python3 -c "import time;print('%.1f' % (time.time() - $t0))") s" | tee -a ${L}_samples.txt
  grep "^\[" ${L}_program.log
  tail -3 /tmp/rccl_$pass.log | cut -c1-200
  cp /tmp/rccl_$pass.log ${L}_rccl.log
done
