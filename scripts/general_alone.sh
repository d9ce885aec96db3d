#!/bin/bash
# the general kernel alone (VP_DEBUG_ONLY_CLASS=0: INCOMPLETE images) at 3..8 resident workgroups per CU: what the general class costs
# when nothing runs beside it.  scripts/general_alone.sh "c2,c3,c3ref" FRAMES
WL=${1:-c2,c3}; FR=${2:-256}
for B in 3 4 5 6 8; do
  echo "== general kernel alone, $B workgroups per CU"
  VP_PERF_RNG=2 VP_DEBUG_ONLY_CLASS=0 VP_GENERAL_BLOCKS_PER_CU=$B timeout -k 10 300 python3 scripts/perf_workloads.py $WL $FR 2 || exit 1
done
echo "== light kernel alone"
VP_PERF_RNG=2 VP_DEBUG_ONLY_CLASS=1 VP_LIGHT_BLOCKS_PER_CU=8 timeout -k 10 300 python3 scripts/perf_workloads.py $WL $FR 2 || exit 1
echo "== both (default)"
VP_PERF_RNG=2 timeout -k 10 300 python3 scripts/perf_workloads.py $WL $FR 2
