#!/bin/bash
# development sweep: block split between the general and the light kernel (global-majorant estimator), light wait policy
L=/root/repo/cuda-volpath_amd
export VP_PERF_RNG=2
for lib in ${LIBS:-ls16}; do
 for blk in ${SPLITS:-4,3 5,1 6,1 5,2 4,2} ; do
  g=${blk%,*}; l=${blk#*,}
  for wi in ${WAITS:-128 256}; do
    echo "== lib=$lib general=$g light=$l light_wait_iters=$wi"
    VOLPATH_LIB=$L/libvolpath_hip_$lib.so VP_GENERAL_BLOCKS_PER_CU=$g VP_LIGHT_BLOCKS_PER_CU=$l VP_LIGHT_WAIT_ITERS=$wi timeout -k 10 120 python3 scripts/perf_workloads.py ${WL:-c2} 256 2 || exit 1
  done
 done
done
