#!/bin/bash
# PMC + kernel-trace digests of several bench workloads in one GPU call:
#   scripts/profile_workloads.sh TAG wl1 [wl2 ...]     -> gpurun_out/prof_TAG_<wl>/digest.json
set -e
TAG=$1; shift
for WL in "$@"; do
  echo "== $WL"
  bash scripts/profile_bench.sh ${TAG}_$WL --workload $WL > gpurun_out/prof_${TAG}_$WL.log 2>&1
  tail -3 gpurun_out/prof_${TAG}_$WL.log
done
