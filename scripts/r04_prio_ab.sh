#!/bin/bash
# Round 4: s_setprio A/B and the wait policy re-swept with exit flights on.  Dev builds, Philox2x32-7.
cd $GRAFT_REPO_ROOT
export VP_PERF_RNG=2
for L in dev prio3 prioT; do
  echo "== lib $L"
  VOLPATH_LIB=$GRAFT_REPO_ROOT/cuda-volpath_amd/libvolpath_hip_$L.so timeout -k 10 600 python3 scripts/perf_workloads.py c2,c3ref,c3 256 3 || exit 1
  VOLPATH_LIB=$GRAFT_REPO_ROOT/cuda-volpath_amd/libvolpath_hip_$L.so timeout -k 10 600 python3 scripts/perf_workloads.py c4f 32 2 2>&1 | grep -v "^Read" || exit 1
done
export VOLPATH_LIB=$GRAFT_REPO_ROOT/cuda-volpath_amd/libvolpath_hip_dev.so
WLS="c2" LANES="16 24 32 40" ITERS="8 16 32" FRAMES=256 bash scripts/sweep_wait.sh
for E in "VP_END_LANES=2" "VP_END_LANES=8" "VP_END_LANES=16"; do echo "== $E"; env $E timeout -k 10 120 python3 scripts/perf_workloads.py c2 256 2; done
