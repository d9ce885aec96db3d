#!/bin/bash
# development sweep: when a wave leaves its tracking loop for the event pass (general kernel)
export VP_PERF_RNG=${VP_PERF_RNG:-2}
for wl in ${WLS:-c2 c3ref}; do
for lanes in ${LANES:-16 24 32}; do
 for iters in ${ITERS:-8 16 32 64}; do
  echo "== $wl wait_lanes=$lanes wait_iters=$iters"
  VP_WAIT_LANES=$lanes VP_WAIT_ITERS=$iters timeout -k 10 120 python3 scripts/perf_workloads.py $wl ${FRAMES:-256} 2 || exit 1
 done
done
done
