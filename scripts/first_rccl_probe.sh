#!/bin/bash
# Where does the FIRST GPU process of a fresh box spend its minutes?  tests/cpp/test_cpp_api (everything it does before the
# batch entry takes 0.5 s; the batch entry's first ncclCommInitRank has taken 6 s warm and 70 / 100 / 295 s cold) as the first
# process, RCCL's own log with wall-clock stamps, and a sampler of the process's I/O counters and of what its threads wait in.
# Then the same again (warm).  Usage (through gpurun): TAG=r05_u bash scripts/first_rccl_probe.sh
set -u
T="${TAG:-r05_rccl}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
stamp() { while IFS= read -r l; do printf '%s %s\n' "$(date +%s.%N | cut -c1-14)" "$l"; done; }
for pass in cold warm; do
  L=$O/${T}_${pass}
  t0=$(date +%s.%N)
  ( NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV,GRAPH,NET timeout 600 tests/cpp/test_cpp_api 2>&1 | stamp > ${L}_program.log ) &
  sleep 0.3
  pid=$(pgrep -n -x test_cpp_api)
  echo "pass $pass pid $pid t0 $t0" > ${L}_samples.txt
  while [ -n "$pid" ] && kill -0 $pid 2>/dev/null; do
    {
      echo "== $(date +%s.%N | cut -c1-14)"
      grep -E "rchar|read_bytes|syscr" /proc/$pid/io 2>/dev/null | tr '\n' ' '; echo
      grep -E "VmRSS|RssFile" /proc/$pid/status 2>/dev/null | tr '\n' ' '; echo
      for t in /proc/$pid/task/*; do
        printf '%s %s %s | ' "$(cat $t/comm 2>/dev/null)" "$(cut -d' ' -f3 $t/stat 2>/dev/null)" "$(cat $t/wchan 2>/dev/null)"
        head -4 $t/stack 2>/dev/null | tr '\n' ' '
        echo
      done
    } >> ${L}_samples.txt
    sleep 2
  done
  echo "pass $pass took $(echo "$(date +%s.%N) - $t0" | bc) s" | tee -a ${L}_samples.txt
  grep "^\S* \[" ${L}_program.log | head -8
done
grep -c . $O/${T}_cold_program.log $O/${T}_warm_program.log
