"""The counter table of profiles/r04_stalls.md (section 2) from the committed digests: python scripts/stall_table.py [TAG]"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
W = ["c2", "c3", "c3ref", "c3ref_samplerh", "c4s", "c4f"]
D = {}
for w in W:
    d = json.load(open(f"profiles/{tag}_{w}_digest.json"))
    sq, s2, t = d["sq"]["per_launch"], d["sq2"]["per_launch"], d["tcc"]["per_launch"]
    wc = sq["SQ_WAVE_CYCLES"]
    kt = d["kernel_trace"]["avg_ms"]; ka = d.get("kernel_trace_approach", {}).get("avg_ms", 0)
    D[w] = dict(kt=kt, ka=ka, valu=sq["SQ_INSTS_VALU"] / 1e9, issue=sq["SQ_INSTS_VALU"] / ((kt + ka) * 1e-3) / 1.2288e12,
                lane=sq["SQ_THREAD_CYCLES_VALU"] / 64 / sq["SQ_INSTS_VALU"], wait=sq["SQ_WAIT_ANY"] / wc, winst=sq["SQ_WAIT_INST_ANY"] / wc,
                act=s2["SQ_ACTIVE_INST_ANY"] / wc, actv=sq["SQ_ACTIVE_INST_VALU"] / wc, salu=s2["SQ_INSTS_SALU"] / sq["SQ_INSTS_VALU"],
                br=s2["SQ_INSTS_BRANCH"] / sq["SQ_INSTS_VALU"], vrd=sq["SQ_INSTS_VALU"] / s2["SQ_INSTS_VMEM_RD"],
                lvl=s2["SQ_INST_LEVEL_VMEM"] / s2["SQ_INSTS_VMEM_RD"], l2=t["TCC_HIT_sum"] / t["TCC_REQ_sum"],
                hbm=d["hbm_bytes_per_launch"] / 1e12, hbmr=d["hbm_bytes_per_launch"] / ((kt + ka) * 1e-3) / 1e12,
                waves=wc * 4 / (sq["SQ_BUSY_CYCLES"] / 32.0 * 1024.0 * 4) if sq.get("SQ_BUSY_CYCLES") else 0, bench=d["bench_line_kt"]["value"])


def row(label, f):
    return "| " + label + " | " + " | ".join(f(D[w]) for w in W) + " |"


print("\n".join([
    "| | C2 | C3 | c3ref | c3ref sampler.h | c4s | c4f |", "|---|---|---|---|---|---|---|",
    row("integrator / approach kernel, ms (kernel trace)", lambda x: f"{x['kt']:.1f} / {x['ka']:.1f}"),
    row("`SQ_INSTS_VALU` per launch", lambda x: f"{x['valu']:.1f} G"),
    row("vector issue slots taken = INSTS_VALU · 2 cycles / (1024 SIMDs · 2.4 GHz · launch time)", lambda x: f"{x['issue']:.2f}"),
    row("lane utilisation `THREAD_CYCLES_VALU / (64 · INSTS_VALU)`", lambda x: f"{x['lane']:.2f}"),
    row("`WAIT_ANY / WAVE_CYCLES` (in `s_waitcnt`)", lambda x: f"{x['wait']:.2f}"),
    row("`WAIT_INST_ANY / WAVE_CYCLES` (ready, not issued)", lambda x: f"{x['winst']:.2f}"),
    row("`ACTIVE_INST_ANY / WAVE_CYCLES` (vector: `ACTIVE_INST_VALU`)", lambda x: f"{x['act']:.2f} ({x['actv']:.2f})"),
    row("scalar instructions per vector instruction", lambda x: f"{x['salu']:.2f}"),
    row("branches per vector instruction", lambda x: f"{x['br']:.3f}"),
    row("vector instructions per vector-memory read", lambda x: f"{x['vrd']:.1f}"),
    row("`INST_LEVEL_VMEM` per read (reads in flight, sampled)", lambda x: f"{x['lvl']:.1f}"),
    row("L2 hit rate", lambda x: f"{x['l2']:.2f}"),
    row("fabric bytes per launch `(2·FETCH + WRITE)·1024`", lambda x: f"{x['hbm']:.2f} TB ({x['hbmr']:.2f} TB/s)"),
    row("Msamples/s of the kernel-trace run", lambda x: f"{x['bench']:.0f}"),
]))
