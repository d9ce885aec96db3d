#!/bin/bash
# Round 4: A/B of development builds of the kernels (bit-identical images expected in every row: compare the hashes).
#   scripts/r04_lib_ab.sh "deva devb devc" "c3ref,c4s,c4f" [FRAMES] [RNG]      (through gpurun)
LIBS=${1:-"dev"}; WLS=${2:-"c3ref,c4s,c4f"}; FR=${3:-256}; RNG=${4:-2}
cd $GRAFT_REPO_ROOT
export VP_PERF_RNG=$RNG
for L in $LIBS; do
  echo "== libvolpath_hip_$L.so (rng $RNG)"
  VOLPATH_LIB=$GRAFT_REPO_ROOT/cuda-volpath_amd/libvolpath_hip_$L.so timeout -k 10 900 python3 scripts/perf_workloads.py $WLS $FR 2 || exit 1
done
