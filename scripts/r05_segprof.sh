#!/bin/bash
# Round 5: the approach walk's kernel with and without the per-pixel segment table (kernel trace of perf_workloads.py).  Through gpurun.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05/segprof; mkdir -p $OUT
export VP_PERF_RNG=2
WL=${1:-c3,c4s}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tab -- python3 scripts/perf_workloads.py $WL 1024 2 > $OUT/tab.log 2>&1 || { tail -5 $OUT/tab.log; exit 1; }
export VP_NO_APPROACH_TABLE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/notab -- python3 scripts/perf_workloads.py $WL 1024 2 > $OUT/notab.log 2>&1 || { tail -5 $OUT/notab.log; exit 1; }
for d in tab notab; do echo "== $d"; find $OUT/$d -name "*kernel_stats.csv" | xargs grep -h "approach\|render_k" | sed -e 's/(vp::SceneDev, vp::LaunchDev)//' | cut -c1-150; done
find $OUT -name "*kernel_trace.csv" -delete
