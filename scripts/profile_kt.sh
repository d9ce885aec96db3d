#!/bin/bash
# kernel-trace stats of one bench workload: scripts/profile_kt.sh TAG [bench args]   -> gpurun_out/kt_TAG/
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kt_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json; echo; cat $OUT/*/*_kernel_stats.csv | head -4 | cut -c1-200
