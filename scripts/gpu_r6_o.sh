#!/bin/bash
# Round 6, session o: same-box A/B of build-time switches on top of session n's library (variant n1):
#   n2  kIcpPollAll    the second hop of the exchange without watcher lanes (every lane of the sweep polls its own pair)
#   n3  kScanPrefetch  tile_scan_list asks for a trip's list entries during the trip before
#   n4  both
# Usage (through gpurun): TAG=r06_o bash scripts/gpu_r6_o.sh
set -u
T="${TAG:-r06_o}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=$T REPS=${REPS:-2} bash scripts/gpu_ab_variants.sh ${VARIANTS:-n1 n2 n3 n4}
