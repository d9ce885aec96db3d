#!/bin/bash
for wl in 4 8 16; do for wk in 2 4 8 16 24; do
  echo -n "wait_lanes=$wl wait_lookups=$wk: "
  VP_WAIT_LANES=$wl VP_WAIT_LOOKUPS=$wk VP_WAIT_ITERS=16 python3 scripts/prof_case.py "$@" | tail -1
done; done
