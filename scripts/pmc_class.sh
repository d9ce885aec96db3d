#!/bin/bash
# SQ counters of ONE kernel class running alone: scripts/pmc_class.sh WORKLOAD CLASS BLOCKS_PER_CU [FRAMES]   (through gpurun)
WL=$1; CLS=$2; B=$3; FR=${4:-128}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${WL}_c${CLS}_b${B}
rm -rf $OUT; mkdir -p $OUT
export VP_PERF_RNG=2 VP_DEBUG_ONLY_CLASS=$CLS VP_GENERAL_BLOCKS_PER_CU=$B VP_LIGHT_BLOCKS_PER_CU=$B VP_BLOCKS_PER_CU=$B
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/scripts/perf_workloads.py $WL $FR 1 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); last = {}
rows = list(csv.DictReader(open(f)))
# the long render_k dispatch: the last one of the class
disp = [r for r in rows if "render_k" in r["Kernel_Name"]]
did = max(int(r["Dispatch_Id"]) for r in disp)
for r in disp:
    if int(r["Dispatch_Id"]) == did:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(open(out + "/run.log").read().strip().splitlines()[-1])
print({k: f"{v:.4g}" for k, v in acc.items()})
i, a, t = acc["SQ_INSTS_VALU"], acc["SQ_ACTIVE_INST_VALU"], acc["SQ_THREAD_CYCLES_VALU"]
print(f"lane_util {t / 64 / i:.3f}  wait_any/wave_cycles {acc['SQ_WAIT_ANY'] / acc['SQ_WAVE_CYCLES']:.3f}  wait_inst/wave_cycles {acc['SQ_WAIT_INST_ANY'] / acc['SQ_WAVE_CYCLES']:.3f}  active_valu/insts {a / i:.3f}")
PY
