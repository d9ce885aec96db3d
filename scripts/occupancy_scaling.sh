#!/bin/bash
export VP_PERF_RNG=2 VP_NO_HANDOFF=1 VP_DEBUG_ONLY_CLASS=0
for wl in c2 c3ref c4s; do
for b in 2 3 4 5 6; do
  echo "== $wl general alone blocks/CU=$b"
  VP_GENERAL_BLOCKS_PER_CU=$b VP_BLOCKS_PER_CU=$b timeout -k 10 120 python3 scripts/perf_workloads.py $wl 128 2 || exit 1
done
done
