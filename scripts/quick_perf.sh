#!/bin/bash
# the three bench workloads (C2 philox, C2 sampler.h, C3 philox), two runs each; optional VOLPATH_LIB
FR=${1:-64}
for cfg in "0 1 1" "0 1 0" "1 8 1"; do
  echo "== est/brick/rng=$cfg"
  timeout -k 10 120 python scripts/prof_case.py $cfg $FR || exit 1
  timeout -k 10 120 python scripts/prof_case.py $cfg $FR || exit 1
done
