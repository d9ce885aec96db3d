"""Digest of scripts/profile_bench.sh output: per-launch averages of the render launch.

A render launch is the general kernel render_k<..., LIGHT=false> -- behind approach_k / approach_local_k / approach_local_tab_k, the
camera rays' walk ahead of it -- and, where the workload has light pixels that are not per-pixel constants, the light kernel
render_k<..., LIGHT=true> beside it on a second stream; bench.py times the pair with HIP events from the start of the first
to the end of the last.  Counters are summed over these kernels of the timed (non-counting) variant and divided by the number
of launches (= dispatches of the general kernel)."""
import collections, csv, glob, json, re, sys
out = sys.argv[1]
res = {}


def variant(name):
    name = re.sub(r"RngPhiloxR<\d+>", "RngPhiloxR", name)
    if "approach_k<" in name or "approach_local_k<" in name or "approach_local_tab_k<" in name:
        # the camera rays' free flights ahead of the global-majorant general kernel: part of the launch, summed like the light kernel
        return {"count": False, "light": False, "approach": True}
    m = re.search(r"render_k<([^>]*)>", name)
    if not m:
        return None
    a = [x.strip() for x in m.group(1).split(",")]
    return {"count": a[3] == "true", "light": len(a) > 8 and a[8] == "true", "approach": False}


import os
# (gpurun merges a call's files into what earlier calls left: the newest pass counts)
ks = sorted(glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)
if ks:
    rows = [(r, variant(r["Name"])) for r in csv.DictReader(open(ks[0])) if "render_k" in r["Name"] or "approach_k" in r["Name"] or "approach_local_k" in r["Name"] or "approach_local_tab_k" in r["Name"]]
    rows = [(r, v) for r, v in rows if v and not v["count"]]
    for key, light, appr in (("kernel_trace", False, False), ("kernel_trace_light", True, False), ("kernel_trace_approach", False, True)):
        sel = [r for r, v in rows if v["light"] == light and v["approach"] == appr]
        if sel:
            r = max(sel, key=lambda r: float(r["TotalDurationNs"]))
            res[key] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                        "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
for d in ("fetch", "write", "sq", "sq2", "tcc"):
    for f in sorted(glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(float); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            v = variant(r["Kernel_Name"])
            if v and not v["count"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if not v["light"] and not v["approach"]:
                    disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
        # a launch may hold two general kernels (the LDS-table kernel and its helper workgroups without the LDS stage): each is
        # dispatched once per launch, so the launches are the dispatches of any one of them
        n = max([len(x) for x in disp.values()] + [1])
        res[d] = {"launches": n, "per_launch": {k: v / n for k, v in agg.items()}}
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
