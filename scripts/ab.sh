#!/bin/bash
# A/B of kernel builds / knobs on the GPU box: scripts/ab.sh "WORKLOADS" FRAMES "ENV1" "ENV2" ...   (each ENV like "VOLPATH_LIB=... VP_X=1")
WL=$1; FR=$2; shift 2
for E in "$@"; do
  echo "== $E"
  env $E timeout -k 10 300 python3 scripts/perf_workloads.py $WL $FR 2 || exit 1
done
