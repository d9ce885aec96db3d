"""GPU parity, randomised: seeded random scenes -- volume shape and content, box, camera (outside, inside, looking away),
medium, estimator, random stream, filter, brick size, image size, first frame -- rendered by the HIP path through the C ABI and
by the CPU oracle.  Bar: bit-exact accumulators and equal work counters (tolerance 0).

The round-2 optimisations are geometric claims about a camera ray (the restart crawl in front of the volume, the distance it
runs through certified-empty cells, the pixel classes, the light kernel): the hand-picked scenes of test_parity_gpu.py look at
the volume from the reference's default camera; these do not.
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _random_volume(rng):
    kind = rng.integers(0, 5)
    if kind == 0:
        nz, ny, nx = (int(rng.integers(8, 33)) for _ in range(3))       # ragged, mostly empty with a few dense boxes
        g = np.zeros((nz, ny, nx), np.float32)
        for _ in range(int(rng.integers(1, 5))):
            z0, y0, x0 = (int(rng.integers(0, n)) for n in (nz, ny, nx))
            z1, y1, x1 = (int(min(n, a + rng.integers(1, 9))) for n, a in ((nz, z0), (ny, y0), (nx, x0)))
            g[z0:z1, y0:y1, x0:x1] = rng.random() * rng.random((z1 - z0, y1 - y0, x1 - x0), dtype=np.float32)
    elif kind == 1:
        n = int(rng.integers(16, 33))
        g = scenes.blob_volume_f32(n, seed=int(rng.integers(0, 1000)))
    elif kind == 2:
        nz, ny, nx = (int(rng.integers(6, 25)) for _ in range(3))       # nowhere empty: no pixel is "light"
        g = 0.05 + 0.95 * rng.random((nz, ny, nx), dtype=np.float32)
    elif kind == 3:
        nz, ny, nx = (int(rng.integers(12, 29)) for _ in range(3))      # a shell: empty inside and outside
        z, y, x = np.mgrid[0:nz, 0:ny, 0:nx].astype(np.float32)
        r = np.sqrt(((x - nx / 2) / nx) ** 2 + ((y - ny / 2) / ny) ** 2 + ((z - nz / 2) / nz) ** 2)
        g = ((r > 0.25) & (r < 0.4)).astype(np.float32) * rng.random((nz, ny, nx), dtype=np.float32)
    else:
        nz, ny, nx = (int(rng.integers(4, 21)) for _ in range(3))       # sparse single voxels
        g = (rng.random((nz, ny, nx)) < 0.02).astype(np.float32) * rng.random((nz, ny, nx), dtype=np.float32)
    if rng.random() < 0.6:
        return np.ascontiguousarray((g * 255.0).astype(np.uint8))
    return np.ascontiguousarray(g.astype(np.float32))


def _random_camera(rng, host, centre, extent):
    """position on a shell around the box centre (sometimes inside the box), looking at a point near it (sometimes past it)"""
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    radius = float(rng.choice([0.15, 0.6, 1.5, 3.0, 6.0])) * float(np.linalg.norm(extent))
    pos = centre + d * radius
    target = centre + rng.normal(size=3) * extent * float(rng.choice([0.1, 0.3, 0.5, 2.5], p=[0.4, 0.3, 0.2, 0.1]))
    fwd = target - pos
    fwd /= np.linalg.norm(fwd)
    up = np.cross(fwd, rng.normal(size=3))
    up /= np.linalg.norm(up)
    return host.camera_matrix(pos.astype(np.float32), fwd.astype(np.float32), up.astype(np.float32))


def _case(seed, host):
    """everything a random scene consists of, from its seed"""
    rng = np.random.default_rng(1000 + seed)
    grid = _random_volume(rng)
    nz, ny, nx = grid.shape
    if rng.random() < 0.5:
        box = None
        bmin, bmax = np.array([-1.0, -ny / nx, -nz / nx]), np.array([1.0, ny / nx, nz / nx])
    else:
        bmin = rng.uniform(-2.0, 0.5, 3)
        bmax = bmin + rng.uniform(0.3, 3.0, 3)
        box = (tuple(float(np.float32(v)) for v in bmin), tuple(float(np.float32(v)) for v in bmax))
    est = int(rng.integers(0, 3))
    rng_mode = int(rng.integers(0, 3))
    linear = bool(rng.random() < 0.75)
    brick = int(rng.choice([1, 2, 4, 8])) if est else 1
    W, H = int(rng.integers(3, 57)), int(rng.integers(3, 41))
    # optical thickness of the box diagonal at the volume's densest: 2 to ~600 mean free paths
    thickness = 10.0 ** rng.uniform(0.3, 2.8)
    scale = float(np.linalg.norm(bmax - bmin)) * max(float(grid.max()) / (255.0 if grid.dtype == np.uint8 else 1.0), 1e-3)
    kw = dict(density=float(np.float32(thickness / scale)), g=float(np.float32(rng.uniform(-0.9, 0.95))))
    if rng.random() < 0.6:
        kw["sigma_t"] = tuple(float(np.float32(v)) for v in rng.uniform(0.2, 1.0, 3))
        kw["albedo"] = tuple(float(np.float32(v)) for v in rng.uniform(0.3, 1.0, 3))
    sun = rng.normal(size=3)
    sun /= np.linalg.norm(sun)
    sun_dir = tuple(float(np.float32(v)) for v in sun)
    sun_power = tuple(float(np.float32(v)) for v in rng.uniform(0.0, 5.0e4, 3))
    env = scenes.synthetic_env(w=int(rng.integers(1, 40)), h=int(rng.integers(1, 20)), seed=seed)
    cam = _random_camera(rng, host, (bmin + bmax) / 2, (bmax - bmin) / 2)
    key = (int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31)))
    late = est == 1 and nx * ny * nz <= 16 ** 3 and rng.random() < 0.5     # across the frame-11 estimator switch (quirk Q5)
    first = 9 if late else int(rng.choice([0, 0, 3, 977, 123456]))
    nframes = 4 if late else int(rng.integers(1, 5))
    if est == 1 and not late and first + nframes - 1 > 10:
        first = 0
    # the reference's compiled-out builds: active environment sampling (one-sample MIS), scalar / multi-channel tracking
    build = rng.random()
    env_mis = bool(build < 0.15)
    track = int(rng.integers(1, 3)) if 0.15 <= build < 0.3 else 0
    if (env_mis or track) and rng_mode == 2:
        rng_mode = 1            # Philox2x32-7 is built for the shipped configuration only
    world = int(rng.choice([1, 1, 2, 3, 8]))
    return dict(grid=grid, box=box, est=est, rng_mode=rng_mode, linear=linear, brick=brick, W=W, H=H, kw=kw, sun_dir=sun_dir,
                sun_power=sun_power, env=env, cam=cam, key=key, late=late, first=first, nframes=nframes, env_mis=env_mis,
                track=track, world=world)


def _oracle_render(oracle, c):
    osc = oracle.OracleScene(c["grid"], c["env"], c["sun_dir"], c["sun_power"], box=c["box"], brick=c["brick"], linear=c["linear"],
                             estimator=c["est"], rng_mode=c["rng_mode"], seed=c["key"], inv_view=c["cam"], env_mis=c["env_mis"],
                             track_mode=c["track"])
    oP = oracle.default_param(c["W"], c["H"], **c["kw"])
    if c["late"]:
        osc.precompute_opacity()
    ref, cnt = None, None
    for f in range(c["first"], c["first"] + c["nframes"]):
        ref, k = osc.render_frame(oP, f, ref)
        d = k.as_dict()
        cnt = d if cnt is None else {q: cnt[q] + d[q] for q in d}
    return ref, cnt


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VP_FUZZ_SEEDS", "64"))))
def test_random_scene_bit_exact(vp, oracle, seed):
    from volpath import host
    c = _case(seed, host)
    grid, box, est, rng_mode, linear, brick, W, H, kw = (c[k] for k in ("grid", "box", "est", "rng_mode", "linear", "brick", "W", "H", "kw"))
    sun_dir, sun_power, env, cam, key, late, first, nframes = (c[k] for k in ("sun_dir", "sun_power", "env", "cam", "key", "late", "first", "nframes"))
    ref, cnt = _oracle_render(oracle, c)
    vP = vp.make_param(W, H, **kw)
    what = dict(seed=seed, grid=grid.shape, dtype=str(grid.dtype), box=box, est=est, rng=rng_mode, linear=linear, brick=brick,
                size=(W, H), first=first, nframes=nframes, env_mis=c["env_mis"], track=c["track"], world=c["world"], **kw)
    buf = vp.DeviceBuffer(W, H)
    try:
        vp.init_volume(grid, box=box, brick=brick, linear=linear)
        vp.init_envmap(env)
        vp.set_sun(sun_dir, sun_power)
        vp.set_camera(cam)
        vp.set_estimator(est)
        vp.set_rng(rng_mode, key)
        vp.set_tracking(c["track"])
        vp.set_envmap_sampling(vp.ENV_MIS if c["env_mis"] else vp.ENV_PASSIVE)
        vp.set_shard(0, 1)
        vp.set_exit_flights(seed % 3)             # exit flights: off / global-majorant estimator (default) / local majorants too
        if late:
            vp.precompute_opacity(sun_dir)
        counted = not c["track"]                  # the work counters are not built for the scalar tracking kernels
        vp.enable_counters(counted)
        vp.read_counters(reset=True)
        vp.render_frames(buf.ptr, first, nframes, vP)
        got = buf.download()
        k = vp.read_counters()
        vp.enable_counters(False)
        assert np.array_equal(got, ref, equal_nan=True), (what, float(np.nanmax(np.abs(got - ref))))
        # no shadow ray of the oracle's render outran the 2^20 pairs of its sub-stream (include/volpath.h, VP_RNG_PHILOX)
        assert oracle.lib().vpo_debug_shadow_overflow() == 0, what
        if counted:
            for q in ("samples", "density_lookups", "bound_lookups", "opacity_lookups", "env_lookups", "scatters"):
                assert k[q] == cnt[q], (what, q, k[q], cnt[q])
        # frame by frame through the reference's entry point (direct accumulation, no staging), counters off; the image as the
        # sum of the shards of `world` ranks (disjoint pixel tiles: the sum is exact)
        buf.reset()
        for r in range(c["world"]):
            vp.set_shard(r, c["world"])
            for f in range(first, first + nframes):
                vp.render_kernel(buf.ptr, f, vP)
        assert np.array_equal(buf.download(), ref, equal_nan=True), what
    finally:
        vp.enable_counters(False)
        vp.set_shard(0, 1)
        vp.set_exit_flights(1)
        vp.set_tracking(0)
        vp.set_envmap_sampling(0)
        vp.set_camera()
        buf.free()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VP_FUZZ_LONG", "12"))))
def test_random_scene_long_launch_bit_exact(vp, oracle, seed, monkeypatch):
    """Launches of 64 frames and more take paths of their own: the approach walk's waves are one pixel in 64 frames, and the
    decomposition estimator's walk then reads the per-view segment table (round 5, approach_segments_k -> approach_local_tab_k).
    Random scenes as above -- cropped to 20^3 voxels so that the oracle's optical-depth precompute stays cheap -- with the
    decomposition or the global-majorant estimator, spectral tracking, 64...100 frames across the frame-11 switch, a few pixels.
    Odd seeds render in a context that never takes a volume for dense (VP_DENSE_PERCENT=101: these small random grids mostly
    are, and a dense volume gets no walk), so that the walk -- and its table, on uchar volumes -- runs in most of them."""
    from volpath import host
    c = _case(5000 + seed, host)
    rng = np.random.default_rng(77 + seed)
    c["grid"] = np.ascontiguousarray(c["grid"][:20, :20, :20])
    if c["grid"].dtype != np.uint8 and rng.random() < 0.7:
        c["grid"] = np.ascontiguousarray((np.clip(c["grid"], 0.0, 1.0) * 255.0).astype(np.uint8))
    c["est"] = int(rng.choice([1, 1, 1, 0]))
    c["brick"] = int(rng.choice([1, 2, 4, 8])) if c["est"] else 1
    c["env_mis"], c["track"] = False, 0
    c["W"], c["H"] = int(rng.integers(3, 20)), int(rng.integers(3, 14))
    c["first"], c["nframes"] = int(rng.choice([0, 5, 11, 40])), int(rng.integers(64, 101))
    c["late"] = c["est"] == 1
    ref, cnt = _oracle_render(oracle, c)
    vP = vp.make_param(c["W"], c["H"], **c["kw"])
    what = dict(seed=seed, grid=c["grid"].shape, dtype=str(c["grid"].dtype), box=c["box"], est=c["est"], rng=c["rng_mode"], linear=c["linear"],
                brick=c["brick"], size=(c["W"], c["H"]), first=c["first"], nframes=c["nframes"], **c["kw"])
    ctx = None
    if seed & 1:
        monkeypatch.setenv("VP_DENSE_PERCENT", "101")
        ctx = vp.Context(0)
        monkeypatch.delenv("VP_DENSE_PERCENT")
        ctx.__enter__()
    buf = vp.DeviceBuffer(c["W"], c["H"])
    try:
        vp.init_volume(c["grid"], box=c["box"], brick=c["brick"], linear=c["linear"])
        vp.init_envmap(c["env"])
        vp.set_sun(c["sun_dir"], c["sun_power"])
        vp.set_camera(c["cam"])
        vp.set_estimator(c["est"])
        vp.set_rng(c["rng_mode"], c["key"])
        vp.set_tracking(0)
        vp.set_envmap_sampling(vp.ENV_PASSIVE)
        vp.set_shard(0, 1)
        if c["late"]:
            vp.precompute_opacity(c["sun_dir"])
        vp.render_frames(buf.ptr, c["first"], c["nframes"], vP)
        got = buf.download()
        assert np.array_equal(got, ref, equal_nan=True), (what, float(np.nanmax(np.abs(got - ref))))
        # the table is read exactly where it can be: the decomposition estimator's walk over a uchar volume
        want_table = c["est"] == 1 and vp.last_approach_mode() == 1 and c["grid"].dtype == np.uint8
        assert vp.last_approach_table() == (1 if want_table else 0), what
        if __import__("os").environ.get("VP_FUZZ_VERBOSE"):
            print(f"seed {seed}: est {c['est']} {c['grid'].dtype} approach mode {vp.last_approach_mode()} table {vp.last_approach_table()}")
        vp.enable_counters(True)
        vp.read_counters(reset=True)
        buf.reset()
        vp.render_frames(buf.ptr, c["first"], c["nframes"], vP)
        k = vp.read_counters()
        vp.enable_counters(False)
        assert np.array_equal(buf.download(), ref, equal_nan=True), what
        for q in ("samples", "density_lookups", "bound_lookups", "opacity_lookups", "env_lookups", "scatters"):
            assert k[q] == cnt[q], (what, q, k[q], cnt[q])
    finally:
        vp.enable_counters(False)
        vp.set_camera()
        buf.free()
        if ctx is not None:
            ctx.__exit__(None, None, None)
            ctx.destroy()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VP_FUZZ_BINARY", "16"))))
def test_random_binary_volume_takes_the_compact_lds_table(vp, oracle, seed):
    """Round 5: a brick table with at most four distinct (max,min) pairs goes through LDS as 2-bit codes beside the cold per-path state
    (render_k<..., LDSB = 2>) -- timed launches of the counter-based streams; the sampler.h stream's read it from global memory, counting
    launches and frame-by-frame render_kernel calls keep the 16-bit table or global memory.  The randomised scenes above are
    soft volumes (hundreds of pairs) and never take that form: these are BINARY volumes {0, v} of random ragged shapes, random boxes,
    cameras, media, brick sizes and first frames (half of them across the frame-11 switch), decomposition estimator.  The timed
    launch, the counting launch and the frame-by-frame path must all equal the oracle bit for bit; vp_last_lds_form says which form
    the timed launch took."""
    from volpath import host
    rng = np.random.default_rng(5000 + seed)
    nz, ny, nx = (int(rng.integers(9, 41)) for _ in range(3))
    z, y, x = np.mgrid[0:nz, 0:ny, 0:nx].astype(np.float32)
    g = np.zeros((nz, ny, nx), bool)
    for _ in range(int(rng.integers(1, 6))):                         # a few ellipsoids
        c = rng.uniform(0.2, 0.8, 3) * (nx, ny, nz)
        r = rng.uniform(0.08, 0.35, 3) * (nx, ny, nz)
        g |= ((x - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((z - c[2]) / r[2]) ** 2 < 1.0
    level = int(rng.choice([255, 255, 200, 37]))
    grid = np.ascontiguousarray(g.astype(np.uint8) * np.uint8(level))
    if rng.random() < 0.5:
        box, bmin, bmax = None, np.array([-1.0, -ny / nx, -nz / nx]), np.array([1.0, ny / nx, nz / nx])
    else:
        bmin = rng.uniform(-2.0, 0.5, 3)
        bmax = bmin + rng.uniform(0.5, 3.0, 3)
        box = (tuple(float(np.float32(v)) for v in bmin), tuple(float(np.float32(v)) for v in bmax))
    brick = int(rng.choice([2, 4, 8]))
    rng_mode = int(rng.choice([1, 2, 2, 0]))
    chromatic = bool(rng.random() < 0.3)
    W, H = int(rng.integers(8, 57)), int(rng.integers(8, 41))
    thickness = 10.0 ** rng.uniform(0.5, 2.5)
    kw = dict(density=float(np.float32(thickness / (float(np.linalg.norm(bmax - bmin)) * level / 255.0))), g=float(np.float32(rng.uniform(-0.5, 0.95))))
    if chromatic:
        kw["sigma_t"] = tuple(float(np.float32(v)) for v in rng.uniform(0.2, 1.0, 3))
        kw["albedo"] = tuple(float(np.float32(v)) for v in rng.uniform(0.3, 1.0, 3))
    sun = rng.normal(size=3)
    sun /= np.linalg.norm(sun)
    sun_dir = tuple(float(np.float32(v)) for v in sun)
    sun_power = tuple(float(np.float32(v)) for v in rng.uniform(0.0, 5.0e4, 3))
    env = scenes.synthetic_env(w=int(rng.integers(1, 40)), h=int(rng.integers(1, 20)), seed=seed)
    cam = _random_camera(rng, host, (bmin + bmax) / 2, (bmax - bmin) / 2)
    key = (int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31)))
    late = bool(rng.random() < 0.5) and nx * ny * nz <= 24 ** 3
    first, nframes = (9, 4) if late else (int(rng.choice([0, 3])), int(rng.integers(2, 5)))
    c = dict(grid=grid, box=box, est=1, rng_mode=rng_mode, linear=True, brick=brick, W=W, H=H, kw=kw, sun_dir=sun_dir, sun_power=sun_power,
             env=env, cam=cam, key=key, late=late, first=first, nframes=nframes, env_mis=False, track=0, world=1)
    ref, cnt = _oracle_render(oracle, c)
    what = dict(seed=seed, grid=grid.shape, level=level, box=box, rng=rng_mode, brick=brick, size=(W, H), first=first, nframes=nframes, chromatic=chromatic, **kw)
    vP = vp.make_param(W, H, **kw)
    buf = vp.DeviceBuffer(W, H)
    try:
        vp.init_volume(grid, box=box, brick=brick, linear=True)
        vp.init_envmap(env)
        vp.set_sun(sun_dir, sun_power)
        vp.set_camera(cam)
        vp.set_estimator(1)
        vp.set_rng(rng_mode, key)
        vp.set_tracking(0)
        vp.set_envmap_sampling(vp.ENV_PASSIVE)
        vp.set_shard(0, 1)
        if late:
            vp.precompute_opacity(sun_dir)
        vp.render_frames(buf.ptr, first, nframes, vP)               # the timed launch
        got = buf.download()
        form = vp.last_lds_form()
        assert np.array_equal(got, ref, equal_nan=True), (what, form, float(np.nanmax(np.abs(got - ref))))
        # (chromatic media too, since their kernel fits six waves; the sampler.h stream has no codes kernel and reads the table from global memory: round 5)
        assert form == (2 if rng_mode != 0 else 0), (what, form)
        vp.enable_counters(True)                                     # the counting launch: the 16-bit table, the estimator's counters
        vp.read_counters(reset=True)
        buf.reset()
        vp.render_frames(buf.ptr, first, nframes, vP)
        k = vp.read_counters()
        vp.enable_counters(False)
        assert vp.last_lds_form() == 1 and np.array_equal(buf.download(), ref, equal_nan=True), what
        for q in ("samples", "density_lookups", "bound_lookups", "opacity_lookups", "env_lookups", "scatters"):
            assert k[q] == cnt[q], (what, q, k[q], cnt[q])
        buf.reset()
        for f in range(first, first + nframes):                     # frame by frame through the reference's entry point
            vp.render_kernel(buf.ptr, f, vP)
        assert np.array_equal(buf.download(), ref, equal_nan=True), what
    finally:
        vp.enable_counters(False)
        vp.set_camera()
        buf.free()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VP_FUZZ_SEQUENCES", "8"))))
def test_random_call_sequences(vp, oracle, seed):
    _run_call_sequence(vp, oracle, seed, extra=False)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VP_FUZZ_SEQUENCES_EXTRA", "4"))))
def test_random_call_sequences_with_invisible_calls(vp, oracle, seed):
    """The same with the calls of round 3 mixed in that must not change anything: vp_prepare (builds the per-camera tables, pixel
    lists and sun table ahead of a frame) and the timing read-outs (which synchronise every stream of the context)."""
    _run_call_sequence(vp, oracle, 500 + seed, extra=True)


def _run_call_sequence(vp, oracle, seed, extra):
    """The reference's entry points in random order: render_kernel for consecutive, repeated and far-away frames (the frame
    look-ahead stages consecutive ones in batches), batched vp_render_frames, and between them every setter a host may call --
    Param, camera, estimator, stream, sun, environment, look-ahead depth, accumulator.  Whatever is staged must be dropped when
    it no longer applies: after every call sequence the accumulators equal the oracle's, rendered call by call."""
    from volpath import host
    rng = np.random.default_rng(7000 + seed)
    W, H = 40, 24
    grid = oracle.julia(16) if seed % 2 == 0 else scenes.blob_volume_u8(14, seed=seed)
    st = dict(est=int(rng.integers(0, 3)), rng_mode=int(rng.integers(0, 3)), key=(int(rng.integers(0, 1 << 30)), 5), density=200.0, g=0.6,
              cam=None, sun=scenes.DEFAULT_SUN_DIR, env_seed=3)

    def make_oracle():
        env = scenes.synthetic_env(seed=st["env_seed"])
        o = oracle.OracleScene(grid, env, st["sun"], scenes.DEFAULT_SUN_POWER, brick=1, estimator=st["est"], rng_mode=st["rng_mode"], seed=st["key"],
                               inv_view=st["cam"])
        if st["est"] == 1:
            o.precompute_opacity()
        return o

    def apply_all():
        vp.init_volume(grid, brick=1)
        vp.init_envmap(scenes.synthetic_env(seed=st["env_seed"]))
        vp.set_sun(st["sun"], scenes.DEFAULT_SUN_POWER)
        vp.set_camera() if st["cam"] is None else vp.set_camera(st["cam"])
        vp.set_estimator(st["est"]); vp.set_rng(st["rng_mode"], st["key"]); vp.set_shard(0, 1)
        vp.set_tracking(0); vp.set_envmap_sampling(0)
        if st["est"] == 1:
            vp.precompute_opacity(st["sun"])

    apply_all()
    osc = make_oracle()
    bufs = [vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)]
    refs = [np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)]
    cur, frame, log = 0, 0, []
    try:
        for step in range(110):
            u = rng.random()
            P_o = oracle.default_param(W, H, density=st["density"], g=st["g"])
            P_v = vp.make_param(W, H, density=st["density"], g=st["g"])
            if u < 0.60:
                frame += 1
                vp.render_kernel(bufs[cur].ptr, frame, P_v); refs[cur], _ = osc.render_frame(P_o, frame, refs[cur]); log.append(("k", frame))
            elif u < 0.64:
                vp.render_kernel(bufs[cur].ptr, frame, P_v); refs[cur], _ = osc.render_frame(P_o, frame, refs[cur]); log.append(("again", frame))
            elif u < 0.68:
                frame = int(rng.integers(0, 5000))
                vp.render_kernel(bufs[cur].ptr, frame, P_v); refs[cur], _ = osc.render_frame(P_o, frame, refs[cur]); log.append(("jump", frame))
            elif u < 0.74:
                n = int(rng.integers(1, 6))
                vp.render_frames(bufs[cur].ptr, frame + 1, n, P_v)
                for f in range(frame + 1, frame + 1 + n):
                    refs[cur], _ = osc.render_frame(P_o, f, refs[cur])
                frame += n; log.append(("batch", n))
            elif u < 0.79:
                st["density"] = float(np.float32(rng.uniform(20, 900))); st["g"] = float(np.float32(rng.uniform(-0.5, 0.9))); log.append("param")
            elif u < 0.83:
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                pos = (d * rng.uniform(0.4, 4.0)).astype(np.float32)
                fwd = (-d + 0.2 * rng.normal(size=3)); fwd /= np.linalg.norm(fwd)
                up = np.cross(fwd, rng.normal(size=3)); up /= np.linalg.norm(up)
                st["cam"] = host.camera_matrix(pos, fwd.astype(np.float32), up.astype(np.float32))
                vp.set_camera(st["cam"]); osc = make_oracle(); log.append("camera")
            elif u < 0.86:
                st["est"] = int(rng.integers(0, 3)); vp.set_estimator(st["est"])
                if st["est"] == 1:
                    vp.precompute_opacity(st["sun"])
                osc = make_oracle(); log.append(("est", st["est"]))
            elif u < 0.89:
                st["rng_mode"] = int(rng.integers(0, 3)); st["key"] = (int(rng.integers(0, 1 << 30)), int(rng.integers(0, 99)))
                vp.set_rng(st["rng_mode"], st["key"]); osc = make_oracle(); log.append(("rng", st["rng_mode"]))
            elif u < 0.91:
                s3 = rng.normal(size=3); s3 /= np.linalg.norm(s3)
                st["sun"] = tuple(float(np.float32(v)) for v in s3)
                vp.set_sun(st["sun"], scenes.DEFAULT_SUN_POWER)
                if st["est"] == 1:
                    vp.precompute_opacity(st["sun"])
                osc = make_oracle(); log.append("sun")
            elif u < 0.93:
                st["env_seed"] = int(rng.integers(0, 100)); vp.init_envmap(scenes.synthetic_env(seed=st["env_seed"])); osc = make_oracle(); log.append("env")
            elif u < 0.96:
                vp.set_lookahead(int(rng.choice([0, 2, 8, 64, 256]))); log.append("lookahead")
            elif extra and u < 0.975:
                vp.prepare(P_v); log.append("prepare")
            elif extra and u < 0.985:
                vp.render_time_ms(reset=bool(rng.integers(0, 2))); vp.render_class_time_ms(reset=bool(rng.integers(0, 2))); log.append("timers")
            else:
                cur = 1 - cur; log.append("buffer")
            if step % 23 == 22:
                for b, r in zip(bufs, refs):
                    assert np.array_equal(b.download(), r, equal_nan=True), (seed, step, log[-12:])
        for b, r in zip(bufs, refs):
            assert np.array_equal(b.download(), r, equal_nan=True), (seed, "end", log[-12:])
    finally:
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
        vp.set_camera()
        for b in bufs:
            b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("workload,seed", [("c2" if k % 4 == 2 else "c1", k) for k in range(int(__import__("os").environ.get("VP_FUZZ_INFLIGHT", "8")))])
def test_call_sequences_with_batches_in_flight(vp, workload, seed):
    """The call-sequence fuzz above runs on scenes a launch finishes in microseconds: nothing is ever in flight when a setter comes.
    Here the same kind of sequence at sizes where render_kernel's staged batches ARE running when the camera moves, a Param changes,
    the look-ahead depth changes or the host jumps to another frame or accumulator -- with and without a synchronisation after each
    call.  Every accumulator must equal its replay through the explicit batch call with the look-ahead off (which the parity tests tie
    to the oracle), bit for bit; vp_lookahead_stats must show that batches were stopped while running."""
    from volpath import scene as vscene, host
    rng = np.random.default_rng(4200 + seed)
    rng_mode = vp.RNG_SAMPLERH if seed % 4 == 1 else vp.RNG_PHILOX7
    key = (0x9E3779B9, 0x85EBCA6B)
    P0, info = vscene.setup(workload, rng_mode=rng_mode, key=key, last_frame=400)
    env, sun_dir, sun_power = info["sunsky"]
    suns = [tuple(sun_dir), tuple(float(np.float32(v)) for v in np.array([0.3, 0.8, -0.52]) / np.linalg.norm([0.3, 0.8, -0.52]))]
    sun, sun_now = 0, [0]

    def apply_sun(k):
        if sun_now[0] != k:
            vp.set_sun(suns[k], sun_power)
            if info["est"] == vp.EST_DECOMP:
                vp.precompute_opacity(suns[k])
            sun_now[0] = k
    W, H = P0.width, P0.height
    bufs = [vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)]
    ref = vp.DeviceBuffer(W, H)
    cams = [info["camera"]]
    for a in rng.uniform(0, 2 * np.pi, 5):
        cams.append(tuple(float(v) for v in host.camera_matrix((3.9 * np.cos(a), -0.78, 3.9 * np.sin(a)), (-np.cos(a), 0.2, -np.sin(a)), (0.0, 1.0, 0.0))))
    segs, cur, cam, density, frame = [], 0, 0, 100.0, 0
    vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
    vp.synchronize()
    l0, c0 = vp.lookahead_stats()
    try:
        for seg in range(28):
            u = rng.random()
            if u < 0.45:
                cam = int(rng.integers(0, len(cams))); vp.set_camera(cams[cam]); frame = 0
            elif u < 0.60:
                density = float(np.float32(rng.uniform(60, 300)))
            elif u < 0.70:
                vp.set_lookahead(int(rng.choice([0, 8, 64, 256])))
            elif u < 0.80:
                cur = 1 - cur
            elif u < 0.88:
                frame = int(rng.integers(0, 300))          # a frame jump inside a run
            elif u < 0.94:
                key = (int(rng.integers(0, 1 << 30)), int(rng.integers(0, 99))); vp.set_rng(rng_mode, key)
            elif u < 0.97:
                sun = 1 - sun; apply_sun(sun)               # the sun table (and the optical-depth table) are rebuilt
            P = vp.make_param(W, H, density=density)
            n = int(rng.choice([1, 2, 5, 12, 30, 45, 70]))
            sync = rng.random() < 0.7                       # the reference's loop synchronises after every call
            for f in range(frame, frame + n):
                vp.render_kernel(bufs[cur].ptr, f, P)
                if sync:
                    vp.synchronize()
            segs.append((cur, cam, density, frame, n, key, sun))
            frame += n
        got = [b.download() for b in bufs]
        l1, c1 = vp.lookahead_stats()
        if seed < 8:   # (the default seeds are known to run many batches and to stop several in flight; a random sequence may not)
            assert l1 - l0 >= 10 and c1 - c0 >= 1, (l1 - l0, c1 - c0)
        vp.set_lookahead(0)
        for i in range(2):
            ref.reset()
            for (b, cm, dn, f0, n, ky, sn) in segs:
                if b == i:
                    vp.set_camera(cams[cm]); vp.set_rng(rng_mode, ky); apply_sun(sn)
                    vp.render_frames(ref.ptr, f0, n, vp.make_param(W, H, density=dn))
            assert np.array_equal(got[i], ref.download(), equal_nan=True), (workload, seed, i, segs)
    finally:
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
        vp.set_camera()
        apply_sun(0)
        for b in bufs:
            b.free()
        ref.free()
