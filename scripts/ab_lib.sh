#!/bin/bash
# A/B of two builds of libvolpath_hip.so on the three bench workloads (C2 philox, C2 sampler.h, C3 philox).
# usage: scripts/ab_lib.sh <alt-lib-path> [frames]
ALT=$1; FR=${2:-64}
for lib in cuda-volpath_amd/libvolpath_hip.so $ALT; do
  for cfg in "0 1 1" "0 1 0" "1 8 1"; do
    echo "== $lib est/brick/rng=$cfg"
    VOLPATH_LIB=$lib timeout -k 10 120 python scripts/prof_case.py $cfg $FR || exit 1
    VOLPATH_LIB=$lib timeout -k 10 120 python scripts/prof_case.py $cfg $FR || exit 1
  done
done
