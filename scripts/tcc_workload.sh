#!/bin/bash
# L2 hit rate and fabric reads of one bench workload (timed kernels, one launch): scripts/tcc_workload.sh TAG WORKLOAD FRAMES  (env: VP_CELL_BRICKS ...)
set -e
TAG=$1; WL=$2; FR=${3:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tcc_$TAG; rm -rf $OUT; mkdir -p $OUT
VP_PERF_RNG=2 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT -- python3 scripts/perf_workloads.py $WL $FR 1 > $OUT/run.log 2>&1
grep "Msamples" $OUT/run.log | tail -1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.Counter(); last = {}
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "render_k" in r["Kernel_Name"]]
did = max(int(r["Dispatch_Id"]) for r in rows)          # the timed launch (the last render_k dispatch)
for r in rows:
    if int(r["Dispatch_Id"]) == did: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in tot.items()}, "L2 hit rate %.3f" % (tot["TCC_HIT_sum"] / max(tot["TCC_REQ_sum"], 1)), "fabric read GB %.1f" % (tot["TCC_EA0_RDREQ_sum"] * 128 / 1e9))
PY
