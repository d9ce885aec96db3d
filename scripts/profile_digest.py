"""Digest of scripts/profile_bench.sh output: per-launch averages of the render kernel."""
import collections, csv, glob, json, re, sys
out = sys.argv[1]
res = {}
ks = glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True)
if ks:
    rows = [r for r in csv.DictReader(open(ks[0])) if "render_k" in r["Name"]]
    r = max(rows, key=lambda r: int(r["Calls"]))  # the timed (non-counting) variant
    res["kernel_trace"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
for d in ("fetch", "write", "sq", "tcc"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); disp = set()
        for r in csv.DictReader(open(f)):
            # the timed variant only: template args <EST, RNG, QUANT, COUNT=false>
            m = re.search(r"render_k<([^>]*)>", r["Kernel_Name"])
            if m and m.group(1).split(",")[3].strip() == "false":
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        n = max(len(disp), 1)
        res[d] = {"launches": len(disp), "per_launch": {k: v / n for k, v in agg.items()}}
for d in ("kt", "fetch"):
    try:
        line = [l for l in open(f"{out}/bench_{d}.log") if l.startswith("{")][-1]
        res[f"bench_line_{d}"] = json.loads(line)
    except Exception as e:  # noqa
        pass
if "fetch" in res and "write" in res:
    f = res["fetch"]["per_launch"].get("FETCH_SIZE", 0.0); w = res["write"]["per_launch"].get("WRITE_SIZE", 0.0)
    res["hbm_bytes_per_launch_raw"] = (f + w) * 1024.0
    # gfx950: one fabric read request per distinct 128-byte line, tallied at 64 bytes -- also for this kernel's 8-byte
    # gathers (scripts/ubench/gather_calib.hip, profiles/r01_fetch_calibration.txt): reads count double
    res["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
print(json.dumps(res, indent=1))
