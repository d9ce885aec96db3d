#!/bin/bash
# usage: scripts/ab_multi.sh lib1 lib2 ...   (C2 philox and C3 philox, two wait_iters settings, 64 frames)
for lib in "$@"; do
  for wi in 16 32; do
    for cfg in "0 1 1" "1 8 1"; do
      echo -n "$lib wait_iters=$wi cfg=$cfg: "
      VOLPATH_LIB=$lib VP_WAIT_ITERS=$wi timeout -k 10 120 python scripts/prof_case.py $cfg 64 | tail -1 || exit 1
    done
  done
done
