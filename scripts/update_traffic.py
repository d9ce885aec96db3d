"""profiles/traffic.json from the digests of scripts/profile_workloads.sh:
   python scripts/update_traffic.py TAG wl1 [wl2 ...]     (reads gpurun_out/prof_TAG_<wl>/digest.json, copies each digest and
   the kernel-trace summary to profiles/TAG_<wl>_digest.json / _kernel_stats.csv)"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wls = sys.argv[1], sys.argv[2:]
tp = os.path.join(ROOT, "profiles", "traffic.json")
T = json.load(open(tp)) if os.path.exists(tp) else {}
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "unknown"
for wl in wls:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
    d = json.load(open(os.path.join(src, "digest.json")))
    dst = f"profiles/{tag}_{wl}_digest.json"
    shutil.copy(os.path.join(src, "digest.json"), os.path.join(ROOT, dst))
    # the kernel-trace summary the digest was computed from: gpurun merges every call's files into gpurun_out/, so an earlier
    # profile of the same tag may lie beside it -- take the one whose render_k row carries the digest's average
    import csv
    picked = None
    for ks in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
        rows = {r["Name"]: r for r in csv.DictReader(open(ks))}
        r = rows.get(d["kernel_trace"]["name"])
        if r and abs(float(r["AverageNs"]) / 1e6 - d["kernel_trace"]["avg_ms"]) <= 1e-6 * d["kernel_trace"]["avg_ms"]:
            picked = ks
    if not picked:
        raise SystemExit(f"{wl}: no kernel_stats.csv under {src}/kt matches the digest")
    shutil.copy(picked, os.path.join(ROOT, f"profiles/{tag}_{wl}_kernel_stats.csv"))
    sq, tcc, b = d["sq"]["per_launch"], d["tcc"]["per_launch"], d["bench_line_kt"]
    sq2 = d.get("sq2", {}).get("per_launch", {})
    # where the wave-cycles go (MI355X_MICROARCH.md, PMC section: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint) and
    # what is issued besides vector arithmetic; all counters in units of four cycles, summed over the launch's kernels
    stalls = {"wait_any_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], "wait_inst_frac": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]}
    if sq2:
        stalls.update({"active_inst_frac": sq2["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                       "salu_per_valu": sq2["SQ_INSTS_SALU"] / sq["SQ_INSTS_VALU"],
                       "valu_per_vmem_read": sq["SQ_INSTS_VALU"] / max(sq2["SQ_INSTS_VMEM_RD"], 1.0),
                       "branch_per_valu": sq2["SQ_INSTS_BRANCH"] / sq["SQ_INSTS_VALU"],
                       "vmem_level_per_read": sq2["SQ_INST_LEVEL_VMEM"] / max(sq2["SQ_INSTS_VMEM_RD"], 1.0)})
    T[wl] = {
        "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
        "valu_insts_per_launch": sq["SQ_INSTS_VALU"],
        "lane_util": sq["SQ_THREAD_CYCLES_VALU"] / 64.0 / sq["SQ_INSTS_VALU"],
        "l2_hit_rate": tcc["TCC_HIT_sum"] / tcc["TCC_REQ_sum"],
        "launch": f"{b['config']['spp_per_step']} frames x {b['config']['image']} ({b['config']['samples_per_step']} samples)",
        "launch_ms_kernel_trace": d["kernel_trace"]["avg_ms"],
        "kernel": d["kernel_trace"]["name"],
        "method": "rocprofv3 --pmc, one pass each for FETCH_SIZE, WRITE_SIZE, the SQ counters and the TCC counters (scripts/profile_bench.sh), "
                  "averaged over the render_k dispatches of the pass.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: the factor 2 is "
                  "the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section), calibrated for this kernel's 8-byte gathers "
                  "(profiles/r01_fetch_calibration.txt); these counters sit on the L2's fabric side, so Infinity-Cache hits are INCLUDED: it is "
                  "an upper bound of the HBM bytes.  lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU).",
        "stalls": stalls,
        "source": dst, "commit": commit,
    }
json.dump(T, open(tp, "w"), indent=1)
print(json.dumps({k: {q: v.get(q) for q in ("hbm_bytes_per_launch", "valu_insts_per_launch", "lane_util", "l2_hit_rate")} for k, v in T.items()}, indent=1))
