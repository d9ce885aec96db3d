#!/bin/bash
# Round 4: what bounds render_k?  The general kernel ALONE (VP_DEBUG_ONLY_CLASS=0: incomplete images) on the given workloads:
#   (1) block tallies + cycle stamps of the counting build (where the lane slots and the wave cycles go),
#   (2) SQ counter passes beyond lane utilisation: instruction mix, VMEM / LDS / scalar activity, issue stalls,
#   (3) rocprofv3 pc sampling if the box allows it (stochastic first, host trap second).
# scripts/r04_stalls.sh "c2 c3ref c4f" [FRAMES]     (through gpurun; writes gpurun_out/r04_stalls/)
WLS=${1:-"c3ref c4f"}; FR=${2:-64}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VP_PERF_RNG=2 VP_DEBUG_ONLY_CLASS=0
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
pick() {  # keep the counter names this box knows
  local keep=""
  for c in "$@"; do grep -q -w "$c" $OUT/avail.txt && keep="$keep $c"; done
  echo $keep
}
P1=$(pick SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY)
P2=$(pick SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH)
P3=$(pick SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS)
P4=$(pick SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU)
echo "P1: $P1"; echo "P2: $P2"; echo "P3: $P3"; echo "P4: $P4"
for WL in $WLS; do
  F=$FR; [ $WL = c4f ] && F=$((FR / 4)); [ $WL = c4s ] && F=$((FR / 2))
  echo "=== $WL general kernel alone, $F frames: block tallies (counting build, shadow rays end where the timed build ends them)"
  VP_DEBUG_COUNT_CLIPS=1 timeout -k 10 300 python3 $ROOT/scripts/block_profile.py $WL $F > $OUT/blocks_$WL.txt 2>&1 || { tail -5 $OUT/blocks_$WL.txt; exit 1; }
  cat $OUT/blocks_$WL.txt
  n=1
  for P in "$P1" "$P2" "$P3" "$P4"; do
    [ -z "$P" ] && { n=$((n + 1)); continue; }
    D=$OUT/pmc_${WL}_p$n; rm -rf $D; mkdir -p $D
    timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d $D -o t -- python3 $ROOT/scripts/perf_workloads.py $WL $F 1 > $D/run.log 2>&1 || { tail -5 $D/run.log; exit 1; }
    python3 - $D <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for kern in ("render_k", "approach"):
    disp = [r for r in rows if kern in r["Kernel_Name"]]
    if not disp: continue
    did = max(int(r["Dispatch_Id"]) for r in disp)   # the long dispatch: the last one
    acc = collections.defaultdict(float)
    for r in disp:
        if int(r["Dispatch_Id"]) == did: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(kern, {k: f"{v:.5g}" for k, v in acc.items()})
print(open(out + "/run.log").read().strip().splitlines()[-1])
PY
    n=$((n + 1))
  done
done
# pc sampling: one workload, short
WL=$(echo $WLS | awk '{print $1}')
for M in "stochastic cycles 1048576" "host_trap time 100"; do
  set -- $M
  D=$OUT/pcs_$1; rm -rf $D; mkdir -p $D
  echo "=== pc sampling ($1, $2, interval $3) on $WL"
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d $D -o t -- python3 $ROOT/scripts/perf_workloads.py $WL 32 1 > $D/run.log 2>&1
  echo "rc $?"; tail -3 $D/run.log; find $D -type f | head -20
  python3 - $D <<'PY'
# aggregate the samples per (code object, offset[, instruction, stall columns]); the raw file can be hundreds of MB: keep the histogram
import csv, glob, sys, os, collections
for f in glob.glob(sys.argv[1] + "/**/*pc_sampling*.csv", recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        print(f, cols)
        keys = [c for c in cols if not any(w in c.lower() for w in ("timestamp", "dispatch", "correlation", "exec_mask", "wave", "chiplet", "workgroup", "hw_id"))]
        hist = collections.Counter(); n = 0; head = []
        for r in rd:
            if n < 5: head.append(r)
            hist[tuple(r[k] for k in keys)] += 1; n += 1
    with open(f + ".hist", "w") as o:
        o.write("# %d samples; columns: count,%s\n" % (n, ",".join(keys)))
        for k, v in hist.most_common(): o.write("%d,%s\n" % (v, ",".join(k)))
    print(n, "samples;", len(hist), "distinct; head:", head[:2])
    if os.path.getsize(f) > 8 << 20: os.remove(f)
PY
done
exit 0
