#!/bin/bash
# wait-policy sweep: scripts/sweep4.sh EST BRICK RNG FRAMES
for wl in 8 12 16 24; do for wi in 8 16 24; do
  echo -n "wait_lanes=$wl wait_iters=$wi: "
  VP_WAIT_LANES=$wl VP_WAIT_ITERS=$wi timeout -k 10 100 python3 scripts/prof_case.py "$@" | tail -1 || exit 1
done; done
