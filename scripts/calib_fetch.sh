#!/bin/bash
# FETCH_SIZE calibration for 8-byte gathers (guide: "other access widths are uncalibrated: calibrate on a known byte count")
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- scripts/ubench/gather_calib > $OUT/run_fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- scripts/ubench/gather_calib > $OUT/run_tcc.log 2>&1 || true
cat $OUT/run_fetch.log | grep mode
python3 - <<'PY'
import csv, glob
for d in ("fetch", "tcc"):
    for f in glob.glob(f"gpurun_out/calib/{d}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            if "gather_k" in r.get("Kernel_Name", ""):
                print(d, r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"])
PY
