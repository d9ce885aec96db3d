#!/usr/bin/env python3
"""bench.py -- Msamples/s of the volumetric radiance integrator on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

runs as typed for any N: with N > 1 and no WORLD_SIZE in the environment it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...` as a CHILD process (before anything
touches the GPU) and relays its JSON line and exit code; under an external torch.distributed.run it is one rank.

A step = one complete render job of the Julia-256^3 scene at 800x600: `spp` samples per pixel, pixel tiles dealt over
the ranks (vp_tile_owner), followed (N > 1) by one RCCL reduce of the HDR accumulators to rank 0.  --scaling weak
(default): spp = 1024 * N, every GPU integrates 800*600*1024 samples per step; at N = 1 this is BASELINE.json's
configs[1] exactly.  --scaling strong: spp = 1024 whatever N (the job is fixed, the ranks share it); --scaling both
prints the weak line with the strong measurement inside it ("strong").  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0 -- the LAST line of stdout, at most LINE_LIMIT bytes (compact_line(): the driver
keeps an 8 KB tail; round 4's 20 KB line was cut in the middle and never parsed): the contract fields, a trimmed roofline,
the CPU baseline, the headline's own general class (the 12 % of its pixels that scatter) as
"general_class_msamples_per_s", and -- at N = 1 -- under "secondary" one compact dict per workload that does physics in every
pixel or on the reference's own streams, all inside the one run the driver times: BASELINE configs[2] (c3), the reference's
live configuration on its own sampler.h streams (c3ref_samplerh), and the two flagged stand-ins of configs[3] (c4s; c4f, the
frame-filling cloud, at that config's 4096 spp).  Everything else (per-class tables, stall buckets, lookups per sample, the
sentences) goes to the file named by --full-out (default gpurun_out/bench_last_full.json), never to stdout.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # same guide, chip table: what streaming kernels reach
# vector-instruction issue peak of the chip: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz
# (= 157.3 TFLOP/s fp32 / 2 flops / 64 lanes, same guide)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
# bytes per sample of the REFERENCE's live estimator on Julia-256^3 800x600 (97.6 density + 51.3
# bound + 1.0 environment lookups): 8*97.6 + 2*51.3 + 16*1.0 + 32.  Quoted next to this build's own figure so that a
# workload that does more (C2's global majorant: ~525 lookups) or less work per sample is not mis-read as bandwidth.
REFERENCE_ESTIMATOR_BYTES_PER_SAMPLE = 8 * 97.6 + 2 * 51.3 + 16 * 1.0 + 32
LINE_LIMIT = 3072  # bytes of the final stdout line (tests/test_host_cpu.py pins it)
WORKLOAD_CHOICES = ["c2", "c3", "c3ref", "c1", "c4s", "c4f"]
RNG_NAMES = {"philox": "philox2x32-10", "philox7": "philox2x32-7", "samplerh": "sampler.h"}
# what the default N = 1 line carries besides the headline: (key, workload, stream, spp per step, timed steps, CPU seconds)
SECONDARY = [
    ("c3", "c3", "philox7", 1024, 3, 5.0),                  # BASELINE configs[2]
    ("c3ref_samplerh", "c3ref", "samplerh", 1024, 3, 5.0),  # the reference's live configuration on its own streams (host.cpp:631)
    ("c4s", "c4s", "philox7", 1024, 1, 5.0),                # configs[3] shape, Julia stand-in
    ("c4f", "c4f", "philox7", 4096, 1, 6.0),                # configs[3] shape and spp, frame-filling cloud stand-in
]


def bytes_per_sample(c, loads):
    """8*L_d + 2*L_b + 32*L_o + 16*L_e + 32 (accumulator read+write).
    loads=True: L_d = the density lookups that ISSUED A LOAD in the timed kernels (the build's own counter: free-flight steps of
    a camera ray through certified-empty cells, and of a sun shadow ray that has only empty cells left, use the +0 such a fetch
    returns without fetching -- DESIGN.md section 5).  loads=False: every lookup the estimator asks for (the oracle counts the
    same number): what a build without those certificates would load."""
    n = max(c["samples"], 1)
    ld = c["density_loads"] if loads else c["density_lookups"]
    return (8.0 * ld + 2.0 * c["bound_lookups"] + 32.0 * c["opacity_lookups"] + 16.0 * c["env_lookups"]) / n + 32.0


def effective_cores():
    """CPUs this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(workload, seconds_hint=15.0, rng="philox7", grid=None, opacity=None):
    """The oracle (CPU restatement, 'port') on this host's cores, on a bounded sample of the same workload.
    grid: the workload's uchar volume if the caller holds it already (the GPU leg's: the two voxelisers are tested bit for bit
    against each other); else the oracle voxelises it itself.
    opacity: the optical-depth table the GPU leg built for this scene (precompute_opacity: an INPUT of the frames from 11 on, quirk
    Q5; its CPU precompute is an N^4 march, minutes at 256^3 and above, and not part of the metric -- host.cpp:634-638 times the render
    loop).  The GPU table is the oracle's own, bit for bit (tests/test_c4_gpu.py: 4152 voxels x 3 directions at 256^3 and 512^3),
    so with it the CPU sample runs across the frame-11 switch on every workload, as the GPU job does."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from volpath import scene as vscene
    cfg = vscene.WORKLOADS[workload]
    O.build()
    if grid is None:
        grid = vscene.host_volume(workload, oracle=O)
    env, sun_dir, sun_power = vscene.default_sunsky()
    orng = {"philox": O.RNG_PHILOX, "philox7": O.RNG_PHILOX7, "samplerh": O.RNG_SAMPLERH}[rng]
    osc = O.OracleScene(grid, env, sun_dir, sun_power,
                        brick=cfg["brick"], estimator=cfg["est"], rng_mode=orng, seed=(0x9E3779B9, 0x85EBCA6B),
                        inv_view=vscene.camera_of(cfg))
    P = O.default_param(cfg["width"], cfg["height"])
    if cfg["chromatic"]:
        O.mat(P, *vscene.PRESET1)
    cores = effective_cores()
    # the live kernel reads the optical-depth volume from frame 11 on (quirk Q5).  Its CPU precompute is an N^4 march
    # (minutes at 256^3 and above), so the timed sample stays within frames 0..10 there and says so; at 128^3 the table is
    # built (untimed) and the sample runs across the switch.
    across_q5 = cfg["est"] == O.EST_DECOMP and (cfg["n"] <= 128 or opacity is not None)
    if across_q5 and opacity is not None:
        import numpy as np
        osc.opacity = np.ascontiguousarray(opacity, np.float32)
        osc.S.opacity = osc.opacity.ctypes.data
    elif across_q5:
        osc.precompute_opacity()
    max_frames = 16 if (cfg["est"] != O.EST_DECOMP or across_q5) else 11
    # decomposition workloads with the table: start at frame 10, so that a bounded sample of a few frames (the cloud: two seconds each on
    # 16 cores) has both estimators in it: frame 10 is the last one that tracks every shadow ray, 11 the first that reads the table
    first_frame = 10 if (cfg["est"] == O.EST_DECOMP and across_q5 and cfg["n"] > 128) else 0
    nframes, acc, tot, t0 = 0, None, 0, time.time()
    while True:
        acc, c = osc.render_frame(P, first_frame + nframes, acc, threads=cores)
        tot += c.samples
        nframes += 1
        if time.time() - t0 > seconds_hint or first_frame + nframes >= max_frames:
            break
    dt = time.time() - t0
    f0, f1 = first_frame, first_frame + nframes - 1
    return {"value": tot / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample_short": f"frames {f0}..{f1} of {cfg['width']}x{cfg['height']}, {tot} samples, {dt:.1f} s, {RNG_NAMES[rng]}",
            "sample": f"frames {f0}..{f1} of {cfg['width']}x{cfg['height']} ({tot} samples, {dt:.1f} s, "
                      f"OpenMP over rows, {RNG_NAMES[rng]} streams"
                      + ("; frames 11+ would read the optical-depth table, whose CPU precompute is not affordable here)"
                         if cfg["est"] == O.EST_DECOMP and not across_q5 else
                         ("; the optical-depth table read from frame 11 on is the GPU leg's, the oracle's own bit for bit)" if opacity is not None and cfg["est"] == O.EST_DECOMP else ")"))}


def _sig(x, digits=5):
    """x to `digits` significant digits for the compact line (None, strings, ints and bools pass)."""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float(f"{float(x):.{digits}g}")


def compact_line(full):
    """The line the driver parses: <= LINE_LIMIT bytes whatever the run carried.  `full` is the complete record
    (what round 4 printed); this keeps the contract fields, a trimmed roofline, the CPU baseline, the general class and one small
    dict per secondary workload / per split / per rank list.  Pure function of `full`: tests/test_host_cpu.py runs it on the
    committed round-4 record and on an 8-rank stub without a GPU."""
    cfg = full.get("config", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _sig(line["value"], 6), _sig(line["ms_per_step"], 6)
    line["config"] = {k: cfg[k] for k in ("workload", "image", "spp_per_step", "samples_per_step", "estimator", "bound_brick", "rng",
                                          "parallelism") if k in cfg}

    def roof(r):
        if not r:
            return None
        b = r.get("bounded_by") or ""
        short = None if not b else ("memory_system + valu_issue (committed pmc)" if b.startswith("memory") else "valu_issue x lane_util (committed pmc)")
        return {"bound": r["bound"], "achieved": _sig(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _sig(r["frac"], 4),
                "traffic": _sig(r.get("traffic")), "traffic_over_loaded_bytes": _sig(r.get("traffic_over_loaded_bytes"), 4),
                "kernel": "vp::render_k", "launch_ms": _sig(r.get("launch_ms")), "launches": r.get("launches"),
                "lane_util": _sig(r.get("lane_util"), 3), "valu_issue_frac": _sig(r.get("valu_issue_frac"), 3), "bounded_by": short,
                "bytes_per_sample": _sig(r.get("loaded_bytes_per_sample"), 5)}

    def cpu(c):
        if not c:
            return None
        return {"value": _sig(c["value"], 4), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                "sample": (c.get("sample_short") or c.get("sample", ""))[:96]}

    if "roofline" in full:
        line["roofline"] = roof(full["roofline"])
    if "cpu_baseline" in full:
        line["cpu_baseline"] = cpu(full["cpu_baseline"])
    line["general_class_msamples_per_s"] = _sig(full.get("general_class_msamples_per_s"))
    if full.get("secondary"):
        sec = {}
        for key, w in full["secondary"].items():
            r = w.get("roofline") or {}
            c = w.get("cpu_baseline") or {}
            sec[key] = {"value": _sig(w["value"]), "general": _sig(w.get("general_class_msamples_per_s")), "ms_per_step": _sig(w["ms_per_step"]),
                        "steps": w["steps"], "spp": w["config"]["spp_per_step"], "frac": _sig(r.get("frac"), 3),
                        "traffic_ratio": _sig(r.get("traffic_over_loaded_bytes"), 3), "lane_util": _sig(r.get("lane_util"), 3),
                        "cpu": _sig(c.get("value"), 3)}
        line["secondary"] = sec
    if full.get("ranks"):
        rk = full["ranks"]
        line["ranks"] = {"kernel_ms": [_sig(v, 4) for v in rk["kernel_ms"]], "wall_s": [_sig(v, 4) for v in rk["wall_s"]],
                         "balance_max_over_mean": _sig(rk.get("balance_max_over_mean"), 4), "collective_ranks": rk.get("collective_ranks"),
                         "backend": rk.get("backend")}
    if full.get("strong"):
        st = full["strong"]
        line["strong"] = {"value": _sig(st["value"]), "unit": st.get("unit"), "ms_per_step": _sig(st["ms_per_step"]), "scaling": "strong",
                          "spp_per_step": st.get("spp_per_step"), "split": st.get("split"),
                          "balance_max_over_mean": _sig((st.get("ranks") or {}).get("balance_max_over_mean"), 4),
                          "by_split": {k: {"value": _sig(v["value"]), "ms_per_step": _sig(v["ms_per_step"]),
                                           "balance_max_over_mean": _sig(v.get("balance_max_over_mean"), 4)}
                                       for k, v in (st.get("by_split") or {}).items()}}
    if full.get("reference_call_pattern") and "msamples_per_s" in full["reference_call_pattern"]:
        cp = full["reference_call_pattern"]
        line["call_pattern"] = {"msamples_per_s": _sig(cp["msamples_per_s"]), "orbit_msamples_per_s": _sig(cp["orbit"]["msamples_per_s"])}
    if full.get("full"):
        line["full"] = full["full"]
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:   # cannot happen with the fields above (an 8-rank line with four secondaries is ~2.3 KB); never print more
        for k in ("call_pattern", "strong", "ranks", "secondary"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def self_launch(args, argv):
    """`python bench.py --gpus N` typed by hand (no WORLD_SIZE): start the ranks as a child torchrun.  Nothing in this process
    has touched the GPU yet, and it never will: it relays the child's output and exit code."""
    # a port nobody holds right now: bound to port 0 here, released, handed to the launcher (no pid arithmetic)
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env)
    raise SystemExit(p.returncode)


def run_workload(workload, args, ctx, spp_per_gpu, steps, warmup, scaling, full=True, rng=None, warmup_spp=None, count_frames=None,
                 split="tiles"):
    """Time `steps` render jobs of one workload; returns the fields of the JSON line (rank 0) or None.
    rng: the stream (default args.rng); warmup_spp: samples per pixel of a warm-up step (default: as a timed step).
    split: how N ranks share a step.  "tiles" (north_star): every rank renders ITS pixel tiles (vp_tile_owner) in all frames; the
    accumulators are disjoint, so the reduce is exact and the image bit-identical to one rank's.  "frames" (SURVEY section 8e's
    alternative): every rank renders ALL pixels in its contiguous share of the step's frames -- perfectly balanced whatever the
    image -- and rank 0 adds the partial images IN RANK ORDER, ((p0 + p1) + p2) + ...: a defined result (tested against the same
    sum made on one GPU), not the one-rank bits: binary32 addition does not associate."""
    import torch
    import torch.distributed as dist
    import volpath as vp
    from volpath import scene as vscene
    from volpath import dist as vdist
    rank, world, dev, stream, rehearsal = ctx["rank"], ctx["world"], ctx["dev"], ctx["stream"], ctx["rehearsal"]
    rng = rng or args.rng
    spp_step = spp_per_gpu * world if scaling == "weak" else spp_per_gpu
    warmup_spp = spp_step if warmup_spp is None else warmup_spp
    by_frames = split == "frames" and world > 1
    if by_frames and (spp_step % world or warmup_spp % world):
        raise SystemExit(f"--split frames: {spp_step} samples per pixel do not divide over {world} ranks")
    rng_mode = {"philox": vp.RNG_PHILOX, "philox7": vp.RNG_PHILOX7, "samplerh": vp.RNG_SAMPLERH}[rng]
    cfg = vscene.WORKLOADS[workload]
    if count_frames is None:   # frames of the (untimed) work-counter pass
        count_frames = 4 if (cfg["est"] == vp.EST_GLOBAL or cfg["n"] > 256) else 8
    # the live kernel reads the optical-depth table from frame 11 on (quirk Q5): a job of hundreds of frames is counted there
    count_first = 16 if (cfg["est"] == vp.EST_DECOMP and spp_step * steps >= 64) else 0
    P, info = vscene.setup(workload, rng_mode=rng_mode, rank=0 if by_frames else rank, world=1 if by_frames else world,
                           last_frame=max(warmup_spp * warmup + spp_step * steps, count_first + count_frames), sunsky=ctx.setdefault("sunsky", None))
    ctx["sunsky"] = info["sunsky"]
    W, H = P.width, P.height

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        acc = torch.zeros(H, W, 4, device=dev, dtype=torch.float32)
        image = torch.zeros(H, W, 4, device=dev, dtype=torch.float32) if rank == 0 else None

        # per-camera set-up at full size (untimed region, reported): the per-pixel tables of the camera, the class split and
        # the pixel lists (GPU kernels + one 12-byte read-back), the sun table.  Forced by moving the camera away and back.
        cam = info["camera"]
        moved = list(cam)
        moved[3] += 0.01
        vp.set_camera(moved)
        vp.prepare(P)
        vp.set_camera(cam)
        vp.synchronize()
        t0 = time.perf_counter()
        vp.prepare(P)
        per_camera_setup_ms = (time.perf_counter() - t0) * 1e3

        # work counters (untimed, counting kernel variant), per pixel class: per-sample lookups of THIS build
        vp.enable_counters(True)
        vp.read_counters(reset=True)
        vp.render_frames(acc.data_ptr(), count_first, count_frames, P)
        counters = vp.read_counters(reset=True)
        vp.enable_counters(False)

        def step(first, spp):
            acc.zero_()
            if by_frames:
                # this rank's contiguous share of the step's frames, all pixels; partial images gathered and added in rank order
                share = spp // world
                vp.render_frames(acc.data_ptr(), first + rank * share, share, P)
                total = vdist.gather_sum_in_rank_order(acc.cpu() if rehearsal else acc, dst=0)
                if rank == 0:
                    image.add_(total.to(dev))
                return
            vp.render_frames(acc.data_ptr(), first, spp, P)
            if world > 1 and rehearsal:
                host = acc.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                acc.copy_(host)
            elif world > 1:
                dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI
            if rank == 0:
                image.add_(acc)

        for i in range(warmup):
            step(i * warmup_spp, warmup_spp)
        # buffers are sized before the clock starts, as the reference's host sizes its own at start-up: a short warm-up step must
        # not leave a 15 GB hipMalloc to the first timed launch
        vp.reserve_frames(P, (spp_step // world) if by_frames else spp_step)
        barrier()
        vp.render_time_ms(reset=True)
        vp.render_class_time_ms(reset=True)
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup * warmup_spp + i * spp_step, spp_step)
        barrier()
        dt = time.perf_counter() - t0
        kern_ms, launches = vp.render_time_ms(reset=True)
        class_ms, class_px = vp.render_class_time_ms(reset=True)
        light_const_flag = vp.last_light_const()

    kern_ms, launches = max(kern_ms, 1e-9), max(launches, 1)
    # per-rank diagnostics: wall time of the timed region and kernel time of each rank (rank order), so that an N > 1 line
    # shows whether the tile deal balanced the work
    t = torch.tensor([dt, kern_ms], device=torch.device("cpu") if rehearsal else dev, dtype=torch.float64)
    per_rank = [t.clone() for _ in range(world)] if world > 1 else [t]
    if world > 1:
        dist.all_gather(per_rank, t)
    walls = [float(x[0]) for x in per_rank]
    kerns = [float(x[1]) for x in per_rank]
    dt = max(walls)
    if rank != 0:
        return None

    samples_total = float(W) * H * spp_step * steps                 # all ranks together
    frames_rank = (spp_step // world if by_frames else spp_step) * steps   # frames this rank rendered (of its own pixels)
    pixels_rank = sum(class_px.values())
    samples_rank = float(pixels_rank) * frames_rank
    value = samples_total / dt / 1e6
    launch_ms = kern_ms / launches
    n_cnt = max(counters["samples"], 1)
    # Bytes the TIMED kernels move (SURVEY section 8d formula on the build's own counters, class by class -- ADVICE r3): the general
    # class by the counters (8 B per density lookup that issued a load, 2 B per bound lookup, 32 B per optical-depth lookup, 16 B
    # per environment lookup, 32 B of accumulator traffic per sample: a 16-byte staging write, read once by the reduce); a class
    # that is a per-pixel constant (the box-missing pixels always; the light class where a null collision in empty space leaves a
    # throughput of 1 as it is) costs ONE environment lookup per pixel and launch and its staging slot, staged once per launch; a
    # light class that is integrated (its kernel beside the general one) fetches no cells: bound lookups + one environment lookup.
    # The counting pass walks every light path, so its env / bound counts are split by the classes' sample shares.
    light_const = class_px["light"] > 0 and light_const_flag
    cper = {k: counters[k] / n_cnt for k in ("density_lookups", "density_loads", "bound_lookups", "opacity_lookups", "env_lookups", "scatters")}
    smp_g = float(class_px["general"]) * frames_rank
    smp_l = float(class_px["light"]) * frames_rank
    smp_m = float(class_px["misses_box"]) * frames_rank
    # per general sample: every load, optical-depth lookup and scatter is the general class's; bound / environment lookups of the
    # counting pass are spread over all samples that make them
    bound_other = 1.0 if cfg["est"] != vp.EST_GLOBAL else 0.0          # a box-missing sample of a local estimator: one bound fetch
    bytes_gen = (8.0 * cper["density_loads"] + 32.0 * cper["opacity_lookups"]) * samples_rank \
        + (2.0 * max(cper["bound_lookups"] * samples_rank - bound_other * smp_m, 0.0) * (smp_g / max(smp_g + smp_l, 1.0))) \
        + (16.0 + 32.0) * smp_g
    # (a per-pixel constant is staged ONCE per launch -- LaunchDev::const_from -- and added frame by frame by the add-kernel, which
    # reads it once and the accumulator once: 16 B written + 16 B read + 32 B of accumulator per PIXEL and launch; VP_NO_CONST_ROWS=1
    # stages it for every frame as before: 32 B per sample)
    const_once = os.environ.get("VP_NO_CONST_ROWS", "0") != "1"
    stage_const = (lambda px, smp: 64.0 * px * launches) if const_once else (lambda px, smp: 32.0 * smp)
    if light_const:
        bytes_light = 16.0 * class_px["light"] * launches + stage_const(class_px["light"], smp_l)
    else:
        bytes_light = 2.0 * max(cper["bound_lookups"] * samples_rank - bound_other * smp_m, 0.0) * (smp_l / max(smp_g + smp_l, 1.0)) + (16.0 + 32.0) * smp_l
    bytes_miss = 16.0 * class_px["misses_box"] * launches + stage_const(class_px["misses_box"], smp_m)
    loaded_total = bytes_gen + bytes_light + bytes_miss
    loaded_bps = loaded_total / max(samples_rank, 1.0)
    estimator_bps = bytes_per_sample(counters, loads=False)
    achieved = loaded_total / launches / (launch_ms * 1e-3) / 1e9
    # per pixel class: which kernel integrates how many samples at what rate (the general and the light kernel run side by
    # side, so their times overlap; each is measured with its own pair of HIP events)
    per_class = {}
    for name in ("general", "light", "misses_box"):
        px, ms = class_px[name], class_ms[name]
        smp = float(px) * frames_rank
        per_class[name] = {"pixels": px, "pixel_fraction": px / max(pixels_rank, 1), "samples": smp, "kernel_ms": ms,
                           "kernel_ms_per_launch": ms / launches, "msamples_per_s": (smp / (ms * 1e-3) / 1e6) if ms > 0 else None}
    per_class["light"]["per_pixel_constant"] = bool(light_const)
    # the density lookups of the light and box-missing classes issue no load; every load belongs to the general kernel
    gen = per_class["general"]
    if gen["kernel_ms"] > 0:
        lookups_general = counters["density_lookups"] / n_cnt * samples_rank
        gen["density_loads_per_s"] = counters["density_loads"] / n_cnt * samples_rank / (gen["kernel_ms"] * 1e-3)
        gen["density_loads_per_sample"] = counters["density_loads"] / n_cnt * samples_rank / max(gen["samples"], 1.0)
        gen["loaded_bytes_per_sample"] = bytes_gen / max(smp_g, 1.0)
        gen["loaded_GBps"] = bytes_gen / (gen["kernel_ms"] * 1e-3) / 1e9
        per_class["all"] = {"density_lookups_per_s": lookups_general / (kern_ms * 1e-3)}
    out = {
        "metric": "Msamples/sec (WxHxspp) + achieved HBM GB/s, Julia 256^3 @ 800x600",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": info["name"], "volume": info.get("volume", f"{info['n']}^3 uchar Julia set"), "image": f"{W}x{H}",
                   "spp_per_step": spp_step, "samples_per_step": int(W * H * spp_step),
                   "estimator": "global_majorant" if info["est"] == vp.EST_GLOBAL else "decomposition",
                   "bound_brick": info["brick"],
                   "rng": RNG_NAMES[rng],
                   "parallelism": (f"frame-ranges x{world} + RCCL gather, partial images added in rank order" if by_frames else
                                   f"pixel-tiles x{world}" + (" + RCCL reduce" if world > 1 else "")),
                   "sky": "Hosek sun/sky bake, setup_sunsky(0.5, 0.2), 1024x512"},
        # the class that does the physics: the pixels whose camera ray can meet the medium (the others are per-pixel constants in
        # this scene: quirk Q3, no pixel jitter) -- the number to track; `value` is BASELINE's metric, all pixels
        "general_class_msamples_per_s": gen["msamples_per_s"],
        "per_camera_setup_ms": per_camera_setup_ms,
        "per_class": per_class,
    }
    if warmup_spp != spp_step:
        out["config"]["warmup_spp"] = warmup_spp
    if info.get("note"):
        out["config"]["note"] = info["note"]
    if full:
        # counters of the render kernels from the committed rocprofv3 PMC passes of THIS workload and stream (not measured in this
        # run: PMC collection serialises kernels); scaled to this run's launch size
        pmc, traffic, valu_frac, lane_util, pmc_src, stalls = {}, None, None, None, None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        tkey = workload if rng == "philox7" else f"{workload}_{rng}"
        if os.path.exists(tp):
            pmc = json.load(open(tp)).get(tkey, {})
        if pmc:
            ref_samples = float(pmc["launch"].split("(")[1].split()[0])
            scale = (float(W) * H * frames_rank / launches) / ref_samples
            traffic = pmc.get("hbm_bytes_per_launch") and pmc["hbm_bytes_per_launch"] * scale
            if pmc.get("valu_insts_per_launch"):
                valu_frac = pmc["valu_insts_per_launch"] * scale / (launch_ms * 1e-3) / VALU_ISSUE_PEAK
            lane_util = pmc.get("lane_util")
            stalls = pmc.get("stalls")
            pmc_src = f"{pmc.get('source')} @ {pmc.get('commit')} (rocprofv3 --pmc passes of `bench.py --workload {workload} --rng {rng}`; " \
                      f"fabric-side counters, Infinity-Cache hits included)"
        loaded_per_launch = loaded_total / launches
        traffic_gbps = (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None
        # What bounds the launch, from the committed counters (profiles/r04_stalls.md has the derivation): the kernel issues
        # valu_issue_frac of the chip's wave-instruction peak (at 2 cycles each; its mix of conversions, compares, 64-bit multiplies
        # and divide sequences costs ~1.4x that) with lane_util of the lanes active, its waves spend wait_frac of their cycles in
        # s_waitcnt -- a vector-issue-bound state machine with latency left over, not a memory-bound gather.  The frame-filling
        # 512^3 cloud alone also sits near what the fabric delivers in 128-byte lines.
        bounded_by = None
        if valu_frac is not None:
            bounded_by = "vector_issue x lane_utilisation (from the committed profile)"
            if traffic_gbps and traffic_gbps / HBM_ACHIEVABLE_GBS > valu_frac:
                bounded_by = "memory_system (line-granular gathers) and vector_issue (from the committed profile)"
        out["roofline"] = {
            # bound/achieved/peak/frac: the contract's HBM roofline on the bytes the timed kernels move (the build's own counters,
            # SURVEY section 8d, class by class).  The kernel is not bound by bytes: its working set is cache-resident; see bounded_by
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": pmc_src,
            "traffic_over_loaded_bytes": (traffic / loaded_per_launch) if traffic else None,
            # what the memory system really moves per second (128-byte lines for 8-byte cells) against the 6.3 TB/s the guide gives
            # as achievable: the frame-filling 512^3 cloud (c4f) sits near it, the Julia workloads far below
            "traffic_GBps": traffic_gbps, "traffic_frac_of_achievable_hbm": (traffic_gbps / HBM_ACHIEVABLE_GBS) if traffic_gbps else None,
            "bounded_by": bounded_by, "valu_issue_frac": valu_frac, "valu_issue_peak_per_s": VALU_ISSUE_PEAK, "lane_util": lane_util,
            "stalls": stalls,
            "kernel": "vp::render_k (a launch = the general kernel -- behind vp::approach_k / approach_local_k, which walk the camera rays "
                      "through their certified-empty stretch -- and, where the light pixels are not per-pixel constants, their kernel beside "
                      "it on a second stream; HIP events from the start of the first to the end of the last; the counters are summed over "
                      "these kernels)",
            "launch_ms": launch_ms, "launches": launches,
            "loaded_bytes_per_sample": loaded_bps,
            "loaded_bytes_per_general_sample": bytes_gen / max(smp_g, 1.0),
            "estimator_bytes_per_sample": estimator_bps,
            "estimator_equivalent_GBps": estimator_bps * samples_rank / launches / (launch_ms * 1e-3) / 1e9,
            "reference_estimator_bytes_per_sample": REFERENCE_ESTIMATOR_BYTES_PER_SAMPLE,
            "lookups_per_sample": cper,
            "lookups_counted_on": f"frames {count_first}..{count_first + count_frames - 1} (counting kernel variant, untimed)"}
    else:
        out["loaded_GBps"] = achieved
        out["launch_ms"] = launch_ms
    if world > 1:
        mean_k = sum(kerns) / world
        out["ranks"] = {"wall_s": walls, "kernel_ms": kerns, "balance_max_over_mean": max(kerns) / mean_k if mean_k > 0 else None,
                        "collective_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                        "tile_deal": "vp_tile_owner: every world-th 8x8 tile of a row, rows shifted by a hash of the row index"}
    if rehearsal:
        out["config"]["parallelism"] += " (REHEARSAL: all ranks on one GPU, gloo)"
    out["_image"] = image
    out["_grid"] = info.get("grid")
    # (for the CPU baseline of a decomposition workload: the optical-depth table of this scene, read back while the scene is current)
    out["_opacity"] = vp.opacity_table((info["n"],) * 3) if (info["est"] == vp.EST_DECOMP and ctx.get("want_opacity")) else None
    return out


def reference_call_pattern(workload, args, ctx, frames=1200, moves=12, frames_per_move=32):
    """The reference host's OWN call pattern under the driver's clock: one render_kernel per frame with a synchronisation after each
    (src/volumeRender.cpp:631-632), `frames` of them on the resting camera; then the interactive case -- the camera orbits, every move
    resets the accumulation (:617-625) -- `moves` x `frames_per_move`.  Same entry point, same bits as the batched job (render_kernel
    stages frames ahead: DESIGN.md section 7); reported beside the headline, never as `value`."""
    import numpy as np
    import torch
    import volpath as vp
    from volpath import scene as vscene, host as vhost
    rng_mode = {"philox": vp.RNG_PHILOX, "philox7": vp.RNG_PHILOX7, "samplerh": vp.RNG_SAMPLERH}[args.rng]
    P, info = vscene.setup(workload, rng_mode=rng_mode, last_frame=frames + 8, sunsky=ctx.get("sunsky"))
    W, H = P.width, P.height
    with torch.cuda.stream(ctx["stream"]):
        acc = torch.zeros(H, W, 4, device=ctx["dev"], dtype=torch.float32)
        for f in range(4):                      # clocks, tables, staging slots
            vp.render_kernel(acc.data_ptr(), f, P); vp.synchronize()
        acc.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            vp.render_kernel(acc.data_ptr(), f, P); vp.synchronize()
        wall = time.perf_counter() - t0
        first, t1 = [], time.perf_counter()
        for m in range(moves):
            a = 2.0 * np.pi * m / moves
            cam = vhost.camera_matrix((3.9 * np.cos(a), -0.78, 3.9 * np.sin(a)), (-np.cos(a), 0.2, -np.sin(a)), (0.0, 1.0, 0.0))
            vp.set_camera(tuple(float(v) for v in cam))
            acc.zero_()
            for f in range(frames_per_move):
                t = time.perf_counter(); vp.render_kernel(acc.data_ptr(), f, P); vp.synchronize()
                if f == 0:
                    first.append((time.perf_counter() - t) * 1e3)
        orbit = time.perf_counter() - t1
        vp.set_camera(info["camera"])
        vp.synchronize()
    return {"workload": workload, "calls": "one render_kernel per frame, a synchronisation after each (src/volumeRender.cpp:631-632)",
            "frames": frames, "msamples_per_s": W * H * frames / wall / 1e6, "wall_ms": wall * 1e3,
            "orbit": {"moves": moves, "frames_per_move": frames_per_move, "msamples_per_s": W * H * moves * frames_per_move / orbit / 1e6,
                      "first_frame_after_a_move_ms_median": float(np.median(first)), "note": "set_camera + first frame: the per-camera tables, stopping the batches in flight, one one-frame launch"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=WORKLOAD_CHOICES)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU per step (weak) / per step (strong)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong", "both"],
                    help="weak: spp * N per step (per-GPU work fixed); strong: spp per step whatever N; both: the weak line with "
                         "the strong measurement inside it; auto (default): weak at N = 1, both at N > 1 (the fixed-job run costs "
                         "1/N of a weak step and shows the per-shard tail the weak line hides)")
    ap.add_argument("--split", default="auto", choices=["auto", "tiles", "frames"],
                    help="how the ranks share a FIXED job (--scaling strong / both): tiles = pixel tiles (north_star; the weak line always), "
                         "frames = contiguous frame ranges of all pixels, partial images added in rank order; auto = measure both, report "
                         "the better")
    ap.add_argument("--rng", default="philox7", choices=["philox", "philox7", "samplerh"],
                    help="philox7 = Philox2x32-7 (default: the fewest rounds Random123 documents as Crush-resistant; oracle parity "
                         "like the others), philox = Philox2x32-10 (the round-1 default, 4-6 %% slower), samplerh = the reference's "
                         "sampler.h streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads of the N = 1 line")
    ap.add_argument("--secondary", default=None, help="comma-separated subset of the secondary workloads (c3,c3ref_samplerh,c4s,c4f)")
    ap.add_argument("--call-pattern", action="store_true",
                    help="also time the reference host's own call pattern (one render_kernel per frame; an orbiting camera): the "
                         "interactive shell is out of scope (SURVEY section 2 rows 15-16), so this is off by default")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_last_full.json"),
                    help="where rank 0 writes the COMPLETE record (per-class tables, stalls, lookups per sample, notes); stdout gets "
                         "the compact line only")
    ap.add_argument("--dump-image", default=None, help="rank 0 saves the summed HDR image (.npy) -- used by tests")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # VP_BENCH_REHEARSAL=1: every rank on GPU 0 and the reduce over gloo through host memory -- lets the N>1
    # code path (sharding, reduce, timing, JSON) be rehearsed on a one-GPU box.  Never used for reported numbers.
    rehearsal = os.environ.get("VP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import volpath as vp
    vp.set_device(local_rank)
    stream = torch.cuda.Stream(device=dev)
    vp.set_stream(stream.cuda_stream)
    ctx = {"rank": rank, "world": world, "dev": dev, "stream": stream, "rehearsal": rehearsal,
           "want_opacity": world == 1 and not args.no_cpu_baseline}

    if args.scaling == "auto":
        args.scaling = "both" if world > 1 else "weak"
    first = "strong" if args.scaling == "strong" else "weak"
    out = run_workload(args.workload, args, ctx, args.spp, args.steps, args.warmup, first,
                       split="frames" if (first == "strong" and args.split == "frames") else "tiles")
    strong, strong_alt = None, None
    if args.scaling == "both" and world > 1:
        # the fixed job (spp whatever N) both ways where the frames divide: by pixel tiles (every shard pays the tail of its own
        # deepest paths and a few per cent of imbalance) and by frame ranges (balanced by construction, every rank pays the fixed
        # cost of a launch); the better one is reported, the other kept beside it
        strong = run_workload(args.workload, args, ctx, args.spp, args.steps, args.warmup, "strong", full=False, split="tiles")
        if args.split in ("auto", "frames") and args.spp % world == 0:
            strong_alt = run_workload(args.workload, args, ctx, args.spp, args.steps, args.warmup, "strong", full=False, split="frames")
    call_pattern = None
    if world == 1 and args.call_pattern:
        try:
            call_pattern = reference_call_pattern(args.workload, args, ctx)
        except Exception as e:   # a side measurement must not take the line down with it
            call_pattern = {"error": repr(e)}
    secondary = {}
    if world == 1 and args.workload == "c2" and not args.no_secondary:
        # The workloads that do physics in every pixel, or on the reference's own streams, under the same clock as the headline
        # (VERDICT r3 item 1): BASELINE configs[2] (its 8^3 brick table staged through LDS), the reference's LIVE configuration on its own sampler.h
        # streams (what src/volumeRender.cpp:631 computes, sample for sample), and the two flagged stand-ins of configs[3] -- the
        # frame-filling cloud at that config's 4096 spp.  One warm-up step of 64 spp each: tables, lists and clocks, not 13 s of cloud.
        only = set(args.secondary.split(",")) if args.secondary else None
        for key, wl, rng, spp, steps, cpu_s in SECONDARY:
            if only is not None and key not in only:
                continue
            sec = run_workload(wl, args, ctx, spp, min(steps, max(args.steps, 1)), 1, "weak", full=True, rng=rng, warmup_spp=64)
            sec.pop("_image")
            grid, opa = sec.pop("_grid"), sec.pop("_opacity")
            if not args.no_cpu_baseline:
                sec["cpu_baseline"] = cpu_baseline(wl, seconds_hint=cpu_s, rng=rng, grid=grid, opacity=opa)
            del grid, opa
            secondary[key] = sec
        if "c3" in secondary:
            secondary["c3"]["note"] = ("BASELINE configs[2]: 8^3 bricks staged through LDS as the config asks (+2-3 % over the same brick table "
                                       "read from global memory, VP_NO_LDS_BOUNDS=1); the same estimator on the reference's finer per-voxel "
                                       "table (workload c3ref) is faster still: tighter bounds")
        if "c3ref_samplerh" in secondary:
            secondary["c3ref_samplerh"]["note"] = ("the reference's live configuration (decomposition tracking, its per-voxel bound table, its "
                                                   "default scene) on its own sampler.h streams: the parity mode; a sequential stream has no "
                                                   "shadow-ray sub-streams, so its shadow rays walk to their end")

    if rank == 0:
        image = out.pop("_image")
        grid, opa = out.pop("_grid"), out.pop("_opacity")
        if strong:
            both = {"tiles": strong}
            if strong_alt:
                both["frames"] = strong_alt
            for q in both.values():
                q.pop("_image")
                q.pop("_grid")
                q.pop("_opacity", None)
            best = max(both, key=lambda k: both[k]["value"]) if args.split == "auto" else ("frames" if (args.split == "frames" and strong_alt) else "tiles")
            out["strong"] = {k: both[best][k] for k in ("value", "unit", "ms_per_step", "scaling", "per_class", "ranks") if k in both[best]}
            out["strong"]["spp_per_step"] = both[best]["config"]["spp_per_step"]
            out["strong"]["split"] = best
            out["strong"]["by_split"] = {k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "parallelism": v["config"]["parallelism"],
                                             "balance_max_over_mean": v.get("ranks", {}).get("balance_max_over_mean")} for k, v in both.items()}
        if call_pattern:
            out["reference_call_pattern"] = call_pattern
        if secondary:
            out["secondary"] = secondary
        if args.dump_image:
            import numpy as np
            np.save(args.dump_image, image.cpu().numpy())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, rng=args.rng, grid=grid, opacity=opa)
        # the complete record to a file (never to stdout), the compact line -- <= LINE_LIMIT bytes -- as the LAST line of stdout
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.full_out)), exist_ok=True)
            with open(args.full_out, "w") as f:
                json.dump(out, f, indent=1)
            out["full"] = os.path.relpath(os.path.abspath(args.full_out), ROOT)
        except OSError as e:
            print(f"bench.py: could not write {args.full_out}: {e}", file=sys.stderr)
        sys.stdout.flush()
        print(compact_line(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
