#!/bin/bash
# L2 hit rate and fabric traffic of one prof_case.py configuration: scripts/tcc_case.sh TAG EST BRICK RNG FRAMES
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tcc_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT -- python3 scripts/prof_case.py "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_k" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in tot.items()}, "hit rate %.3f" % (tot["TCC_HIT_sum"] / max(tot["TCC_REQ_sum"], 1)), "fabric GB %.1f" % (tot["TCC_EA0_RDREQ_sum"] * 128 / 1e9))
PY
