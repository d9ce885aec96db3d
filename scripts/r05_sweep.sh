#!/bin/bash
# Round 5: knob / build sweeps with a stated noise floor.  scripts/r05_sweep.sh "WORKLOADS" FRAMES OUTFILE "ENV1" "ENV2" ...   (through gpurun)
# Every ENV ("VP_X=1 VOLPATH_LIB=...", or "-" for the defaults) is run as its own process (perf_workloads.py, best of 3 launches, Philox2x32-7,
# image hash printed: every row of a workload must show the same hash); the FIRST setting is run three times in all -- at the
# start, in the middle and at the end -- and the spread of those three is the noise floor the other rows are read against.
WL=$1; FR=$2; OUT=$3; shift 3
cd $GRAFT_REPO_ROOT; mkdir -p $(dirname $OUT)
export VP_PERF_RNG=${VP_PERF_RNG:-2}
run() { echo "== $1" | tee -a $OUT; if [ "$1" = "-" ]; then timeout -k 10 600 python3 scripts/perf_workloads.py $WL $FR 3 2>/dev/null | tee -a $OUT; else env $1 timeout -k 10 600 python3 scripts/perf_workloads.py $WL $FR 3 2>/dev/null | tee -a $OUT; fi; }
N=$#; H=$(( (N + 1) / 2 )); I=0
BASE=$1
for E in "$@"; do
  run "$E" || exit 1
  I=$((I + 1))
  if [ $I -eq $H ] && [ $N -gt 1 ]; then run "$BASE" || exit 1; fi
done
[ $N -gt 1 ] && run "$BASE"
exit 0
