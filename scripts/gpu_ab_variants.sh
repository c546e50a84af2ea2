#!/bin/bash
# same-box A/B of library variants (scripts/build_variant.sh) on both bench commands + the in-kernel probe of the first variant given last
#   TAG=r06_l bash scripts/gpu_ab_variants.sh cur nowts
T="${TAG:-ab}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=${T} REPS=${REPS:-2} ARGS="--steps 200 --warmup 10 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh "$@" > /dev/null
cp $O/${T}_ab.txt $O/${T}_ab_200_10.txt
TAG=${T} REPS=${REPS:-2} ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras" bash scripts/ab_bench.sh "$@" > /dev/null
cp $O/${T}_ab.txt $O/${T}_ab_20_5.txt
cat $O/${T}_ab_200_10.txt $O/${T}_ab_20_5.txt
for v in "$@"; do
  KICP_LIB=$PWD/kiss-icp_amd/csrc/variants/libkicp_$v.so timeout 300 python scripts/icp_probe.py frames=160 > $O/${T}_icp_probe_$v.txt 2>&1
  echo "== probe $v"; head -8 $O/${T}_icp_probe_$v.txt | tail -6; grep -A5 "^iteration 0" $O/${T}_icp_probe_$v.txt | head -6
done
