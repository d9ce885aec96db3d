#!/bin/bash
# development sweep: block split between the general and the light kernel, light wait policy
#   LIBS="dev ls16" SPLITS="4,3 5,1" WAITS="128" WL=c2 EXTRA="VP_NO_HANDOFF=1" scripts/sweep_light.sh
L=/root/repo/cuda-volpath_amd
export VP_PERF_RNG=${VP_PERF_RNG:-2}
for lib in ${LIBS:-.}; do
 for blk in ${SPLITS:-4,3 5,1 6,1 5,2 4,2} ; do
  g=${blk%,*}; l=${blk#*,}
  for wi in ${WAITS:-128}; do
    echo "== lib=$lib general=$g light=$l light_wait_iters=$wi $EXTRA"
    lp=$L/libvolpath_hip_$lib.so; [ "$lib" = "." ] && lp=$L/libvolpath_hip.so
    env $EXTRA VOLPATH_LIB=$lp VP_GENERAL_BLOCKS_PER_CU=$g VP_LIGHT_BLOCKS_PER_CU=$l VP_LIGHT_WAIT_ITERS=$wi timeout -k 10 120 python3 scripts/perf_workloads.py ${WL:-c2} ${FRAMES:-256} 2 || exit 1
  done
 done
done
