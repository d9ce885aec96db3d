#!/bin/bash
# Round 4: what bounds render_k?  The general kernel ALONE (VP_DEBUG_ONLY_CLASS=0: incomplete images) on the given workloads:
#   (1) block tallies + cycle stamps of the counting build (where the lane slots and the wave cycles go; shadow rays and exit
#       flights end where the timed build ends them), with and without exit flights,
#       (needs cuda-volpath_amd/libvolpath_hip_prof.so: make dev DEVNAME=prof DEVFLAGS=-DVP_PROFILE_BLOCKS=1)
#   (2) rocprofv3 pc sampling if the box allows it (stochastic first, host trap second).
# The SQ / TCC counter passes of the timed launches are part of scripts/profile_bench.sh (profiles/r04_<wl>_digest.json).
# scripts/r04_stalls.sh "c2 c3ref c4f" [FRAMES]     (through gpurun; writes gpurun_out/r04_stalls/)
WLS=${1:-"c2 c3ref c4f"}; FR=${2:-32}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VP_PERF_RNG=2 VP_DEBUG_ONLY_CLASS=0 VP_DEBUG_COUNT_CLIPS=1
for WL in $WLS; do
  F=$FR; [ $WL = c4f ] && F=$((FR / 2)); [ $WL = c4s ] && F=$((FR / 2))
  for E in "VP_NO_EXIT=1" "VP_EXIT_K=8"; do
    echo "=== $WL general kernel alone, $F frames, $E: block tallies of the counting build"
    env $E timeout -k 10 300 python3 $ROOT/scripts/block_profile.py $WL $F 2>&1 | grep -v "wave-iterations 0, \|lanes per execution\|^Read\|: 0 tests, 0 paths ended; 0 null" | tee $OUT/blocks_${WL}_${E%%=*}.txt
  done
done
unset VP_DEBUG_ONLY_CLASS VP_DEBUG_COUNT_CLIPS
WL=$(echo $WLS | awk '{print $1}')
for M in "stochastic cycles 1048576" "host_trap time 100"; do
  set -- $M
  D=$OUT/pcs_$1; rm -rf $D; mkdir -p $D
  echo "=== pc sampling ($1, $2, interval $3) on $WL"
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d $D -o t -- python3 $ROOT/scripts/perf_workloads.py $WL 32 1 > $D/run.log 2>&1
  echo "rc $?"; grep -v "^W2\|^$" $D/run.log | tail -3
done
exit 0
