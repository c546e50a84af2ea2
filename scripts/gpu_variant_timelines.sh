#!/bin/bash
# kernel timelines (scripts/timeline.sh) of library variants (scripts/build_variant.sh), one after the other on one box:
#   TAG=r06_g bash scripts/gpu_variant_timelines.sh cur nofence ...     -> gpurun_out/<TAG>_timeline_<variant>.txt
T="${TAG:-tl}"; O=gpurun_out; mkdir -p $O
for v in "$@"; do
  KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_$v.so STEPS=${STEPS:-40} timeout 200 bash scripts/timeline.sh > $O/${T}_timeline_$v.txt 2>&1
  echo "== $v"; grep -v "q2" $O/${T}_timeline_$v.txt | tail -12
done
