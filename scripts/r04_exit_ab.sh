#!/bin/bash
# Round 4: exit flights A/B (bit-identical images expected in every row).  scripts/r04_exit_ab.sh [FRAMES] [LIB] ["K list"]
FR=${1:-256}; LIB=${2:-libvolpath_hip.so}; KS=${3:-"2 4 8 16"}
cd $GRAFT_REPO_ROOT
export VOLPATH_LIB=$GRAFT_REPO_ROOT/cuda-volpath_amd/$LIB VP_PERF_RNG=2
echo "== VP_NO_EXIT=1"
VP_NO_EXIT=1 timeout -k 10 600 python3 scripts/perf_workloads.py c2,c3,c3ref,c4s $FR 2 || exit 1
for K in $KS; do
  echo "== VP_EXIT_K=$K"
  VP_EXIT_K=$K timeout -k 10 600 python3 scripts/perf_workloads.py c2,c3,c3ref,c4s $FR 2 || exit 1
done
for WL in c2 c3ref; do
  echo "=== tallies $WL K=4 (counting build ends paths where the timed build does)"
  VP_EXIT_K=4 VP_DEBUG_COUNT_CLIPS=1 VP_DEBUG_ONLY_CLASS=0 timeout -k 10 300 python3 scripts/block_profile.py $WL 32 2>&1 | grep -v "wave-iterations 0\|lanes per execution\|0 tests"
done
