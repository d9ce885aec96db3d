#!/bin/bash
for wl in 8 12 16 24 32; do for wi in 4 8 16 32 64; do
  echo -n "wait_lanes=$wl wait_iters=$wi: "
  VP_WAIT_LANES=$wl VP_WAIT_ITERS=$wi python3 scripts/prof_case.py "$@" | tail -1
done; done
