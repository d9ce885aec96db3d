#!/bin/bash
# Round 5: perturbation profile of render_k (vp_pad in vp_kernels.hip).  Needs the development builds
#   make dev DEVNAME=pad0;  for B in STEP FETCH EOF SETUP COLL END: make dev DEVNAME=pad_$B DEVFLAGS=-DVP_PAD_$B=64
#   make dev DEVNAME=prof DEVFLAGS=-DVP_PROFILE_BLOCKS=1      (block executions of the same launches)
# scripts/r05_pad_profile.sh "WORKLOADS" FRAMES OUT      (through gpurun)
WL=${1:-c4f}; FR=${2:-128}; OUT=${3:-gpurun_out/r05/pad_profile.txt}
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/cuda-volpath_amd
ARGS=("VOLPATH_LIB=$L/libvolpath_hip_pad0.so")
for B in STEP FETCH EOF SETUP COLL END; do ARGS+=("VOLPATH_LIB=$L/libvolpath_hip_pad_$B.so"); done
scripts/r05_sweep.sh $WL $FR $OUT "${ARGS[@]}" > /dev/null || exit 1
for W in ${WL//,/ }; do
  echo "=== block tallies $W, $FR frames (profiling form of the counting build, Philox2x32-10 streams; all classes)" >> $OUT
  VP_DEBUG_COUNT_CLIPS=1 timeout -k 10 600 python3 scripts/block_profile.py $W $FR 2>&1 | grep -v "^Read" >> $OUT
done
cat $OUT
