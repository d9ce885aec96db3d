"""BASELINE configs 4 / 5 on the GPU: ~512^3 dense grid, chromatic medium (preset #1), 1280x720, decomposition tracking.

The WDAS cloud itself (and OpenVDB) is not available offline, so the workload is the FLAGGED SYNTHETIC STAND-IN `c4s`
(volpath/scene.py): same grid size, medium, image and estimator.  At full size the checks are size-independent properties
(batched == frame by frame, shard union == whole for 2 and 8 ranks -- config 5's partition --, finite / non-negative,
decomposition vs global-majorant convergence); the data path real cloud data takes -- dense float dump -> loadBinaryFile ->
init_cuda (src/volumeRender.cpp:915-965, vdbloader/load_vdb.cpp:52-69) -- is compared bit for bit with the oracle at
reduced size on a non-cubic, float, chromatic volume, for the library and for the C++ driver.
"""
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c4s(vp):
    from volpath import scene as vscene
    P, info = vscene.setup("c4s", rng_mode=vp.RNG_PHILOX, last_frame=64)
    assert (P.width, P.height) == (1280, 720) and info["n"] == 512 and info["chromatic"]
    return P, info


def test_c4s_batched_equals_frame_by_frame_and_shards_add_up(vp, c4s):
    P, info = c4s
    W, H = P.width, P.height
    vp.set_estimator(vp.EST_DECOMP)
    vp.set_shard(0, 1)
    a, b = vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)
    frames = range(9, 13)                                   # across the frame-11 estimator switch (quirk Q5)
    vp.render_frames(a.ptr, frames[0], len(frames), P)      # one launch, staged, added in frame order
    for f in frames:
        vp.render_kernel(b.ptr, f, P)                       # the reference's call pattern
    whole = a.download()
    assert np.array_equal(whole, b.download())
    assert np.isfinite(whole).all() and (whole >= 0).all()
    assert (whole[..., 3] > 0).mean() > 0.05
    b.free()
    from volpath import dist as vd
    for world in (2, 8):                                    # 8 ranks at 160 tiles per row: BASELINE config 5's partition
        tot = np.zeros_like(whole)
        for r in range(world):
            a.reset()
            vp.set_shard(r, world)
            vp.render_frames(a.ptr, frames[0], len(frames), P)
            part = a.download()
            assert not part[~vd.owned_mask(r, world, W, H)].any()      # a rank writes its own tiles only
            tot += part
        assert np.array_equal(tot, whole), world
    vp.set_shard(0, 1)
    a.free()


def test_c4s_decomposition_and_global_majorant_converge(vp, c4s):
    """the live kernel (16^3 bricks in LDS) and the global-majorant kernel estimate the same image on the 512^3 chromatic
    workload: 64 spp each; 40x40 block means within 3 % of the image mean + 4 standard errors, image means within 1.5 %
    (the bound is the optical-depth table's bias, quirk Q5, as at 256^3)"""
    P, info = c4s
    W, H, frames = P.width, P.height, 64
    imgs = {}
    for est in (vp.EST_DECOMP, vp.EST_GLOBAL):
        vp.set_estimator(est)
        vp.set_rng(vp.RNG_PHILOX, (77, est))
        buf = vp.DeviceBuffer(W, H)
        vp.render_frames(buf.ptr, 0, frames, P)
        imgs[est] = buf.download()[..., :3].astype(np.float64) / frames
        buf.free()
    vp.set_estimator(vp.EST_DECOMP)
    a, b = imgs[vp.EST_DECOMP], imgs[vp.EST_GLOBAL]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.allclose(a.mean((0, 1)), b.mean((0, 1)), rtol=1.5e-2), (a.mean((0, 1)), b.mean((0, 1)))
    blocks = lambda im: im.reshape(18, 40, 32, 40, 3).mean(axis=(1, 3))
    spread = lambda im: im.reshape(18, 40, 32, 40, 3).std(axis=(1, 3)) / np.sqrt(1600.0)
    tol = 0.03 * b.mean() + 4.0 * np.sqrt(spread(a) ** 2 + spread(b) ** 2)
    assert (np.abs(blocks(a) - blocks(b)) > tol).mean() < 0.01
    # chromatic medium: the three channels really differ
    assert abs(a[..., 0].mean() / a[..., 2].mean() - 1) > 0.02


def _cloudlet(shape=(20, 28, 36), seed=12):
    """a small non-cubic float density field with empty space, values partly outside [0,1] (the dump is clamped on load)"""
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    z, y, x = np.mgrid[0:nz, 0:ny, 0:nx].astype(np.float32)
    r = np.sqrt(((x - nx / 2) / nx) ** 2 + ((y - ny / 2) / ny) ** 2 + ((z - nz / 2) / nz) ** 2)
    v = np.clip(1.5 - 4.0 * r, -0.2, 1.3) * (0.5 + 0.5 * rng.random(shape, dtype=np.float32))
    v[r > 0.33] = 0
    return v.astype(np.float32)


@pytest.mark.parametrize("quantized", [True, False])
def test_dense_dump_ingest_path_is_bit_exact(vp, oracle, tmp_path, quantized):
    """dump_dense_volume -> loadBinaryFile -> init_cuda, chromatic preset #1, decomposition estimator: == oracle.
    quantized = the path of BASELINE config 4 (uchar(clamp(v,0,1)*255), host.cpp:955); float = the reference's
    `quantized = false` branch of the same loader."""
    from volpath import host
    W, H = 64, 48
    vol = _cloudlet()
    path = str(tmp_path / "cloudlet.bin")
    assert host.dump_dense(path, vol)
    grid = host.load_binary(path, quantized=quantized)
    assert grid.shape == vol.shape and grid.dtype == (np.uint8 if quantized else np.float32)
    if quantized:
        assert np.array_equal(grid, (np.clip(vol, 0, 1) * np.float32(255)).astype(np.uint8))
    else:
        assert np.array_equal(grid, vol)
        grid = np.clip(grid, 0, 1)            # the float texture path wants densities in [0,1] like the uchar one
    nz, ny, nx = grid.shape
    box = ((-1.0, -ny / nx, -nz / nx), (1.0, ny / nx, nz / nx))            # host.cpp:1336-1339
    env = scenes.synthetic_env()
    osc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, brick=1, estimator=oracle.EST_DECOMP,
                             rng_mode=oracle.RNG_PHILOX, seed=(3, 4), box=box)
    oP, vP = oracle.default_param(W, H, density=120.0), vp.make_param(W, H, density=120.0)
    oracle.mat(oP, *scenes.PRESET1)
    vp.mat(vP, *scenes.PRESET1)
    vp.init_volume(grid, box=box, brick=1, linear=True)
    vp.init_envmap(env)
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(vp.EST_DECOMP)
    vp.set_rng(vp.RNG_PHILOX, (3, 4))
    vp.set_shard(0, 1)
    ref = None
    for f in range(4):
        ref, _ = osc.render_frame(oP, f, ref)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, 4, vP)
    got = buf.download()
    buf.free()
    assert (ref[..., 3] > 0).mean() > 0.05
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_cli_bin_ingest_matches_oracle_and_two_contexts_match_one(vp, oracle, tmp_path):
    """volpath_render --bin (C++ host: loadBinaryFile, box from the dims, Hosek bake, preset, reference entry points, gamma,
    PPM) end to end == oracle; and --gpus 2 on one device (two contexts, disjoint tile shards, on-device sum -- the
    single-process multi-GPU host with the RCCL reduce replaced by vp_accumulate because both contexts share the GPU)
    writes the same file byte for byte."""
    import ctypes as C
    from volpath import host
    exe = os.path.join(ROOT, "cuda-volpath_amd", "volpath_render")
    vol = _cloudlet()
    path = str(tmp_path / "cloudlet.bin")
    assert host.dump_dense(path, vol)
    W, H, spp = 72, 40, 5
    common = ["--bin", path, "--size", str(W), str(H), "--spp", str(spp), "--preset", "0", "--density", "150", "--rng", "philox"]
    out1, out2, out3 = (str(tmp_path / n) for n in ("one.ppm", "two.ppm", "three.hdr"))
    r = subprocess.run([exe] + common + ["--out", out1], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r2 = subprocess.run([exe] + common + ["--gpus", "2", "--devices", "0,0", "--out", out2], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "2 ranks (shared device" in r2.stdout and "balance max/mean" in r2.stdout
    assert open(out1, "rb").read() == open(out2, "rb").read()
    r3 = subprocess.run([exe] + common + ["--gpus", "3", "--devices", "0,0,0", "--batch", "2", "--out", out3], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    # EIGHT ranks (BASELINE configs[4]'s shard count) as eight contexts of this one process on the one GPU: the tile deal for
    # world = 8, eight launches side by side, the sum on the root context -- the same file byte for byte
    out8 = str(tmp_path / "eight.ppm")
    r8 = subprocess.run([exe] + common + ["--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--out", out8], capture_output=True, text=True)
    assert r8.returncode == 0, r8.stdout + r8.stderr
    assert "8 ranks (shared device" in r8.stdout and "multi-GPU reducer: no collective" in r8.stdout, r8.stdout
    assert open(out1, "rb").read() == open(out8, "rb").read()
    # the oracle on the same inputs
    grid = (np.clip(vol, 0, 1) * np.float32(255)).astype(np.uint8)
    nz, ny, nx = grid.shape
    box = ((-1.0, -ny / nx, -nz / nx), (1.0, ny / nx, nz / nx))
    env, sun_dir, sun_power = host.bake_sunsky(0.5, 0.2)
    osc = oracle.OracleScene(grid, env, sun_dir, sun_power, inv_view=host.camera_matrix(), rng_mode=oracle.RNG_PHILOX,
                             seed=(0x9E3779B9, 0x85EBCA6B), box=box)
    oP = oracle.default_param(W, H, density=150.0)
    oracle.mat(oP, *scenes.PRESET1)
    acc = None
    for f in range(spp):
        acc, _ = osc.render_frame(oP, f, acc)
    disp = np.empty_like(acc)
    oracle.lib().vpo_gamma_correct(disp.ctypes.data_as(C.c_void_p), acc.ctypes.data_as(C.c_void_p), W * H, 1.0 / spp, 2.2)
    expect = (np.minimum(disp[::-1, :, :3], 1.0) * np.float32(255)).astype(np.uint8)
    raw = open(out1, "rb").read()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3), expect)
    assert (acc[..., 3] > 0).mean() > 0.03


def test_two_contexts_on_one_device_sum_to_the_one_context_image(vp, oracle):
    """explicit contexts.  Two contexts on device 0 hold their own scene copies and render disjoint tile
    shards concurrently; their on-device sum (vp_accumulate) is bit-identical to the default context's image, and the
    default context -- the one the reference's 14 entry points act on -- is untouched by them."""
    W, H = 120, 56           # 15 tiles per row: not a multiple of the world size
    grid = oracle.julia(32)
    env = scenes.synthetic_env()

    def scene(rank, world):
        vp.init_volume(grid, brick=1, linear=True)
        vp.init_envmap(env)
        vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
        vp.set_camera()
        vp.set_estimator(vp.EST_DECOMP)
        vp.set_rng(vp.RNG_PHILOX, (8, 1))
        vp.set_shard(rank, world)

    scene(0, 1)
    P = vp.make_param(W, H)
    one = vp.DeviceBuffer(W, H)
    vp.render_frames(one.ptr, 0, 6, P)
    want = one.download()
    ctxs = [vp.Context(0), vp.Context(0)]
    bufs = []
    for r, c in enumerate(ctxs):
        with c:
            assert vp.lib().vp_ctx_device() == 0
            scene(r, 2)
            bufs.append(vp.DeviceBuffer(W, H))
            vp.render_frames(bufs[r].ptr, 0, 6, P)          # asynchronous: both contexts' launches are in flight together
    with ctxs[1]:
        vp.synchronize()
        part1 = bufs[1].download()
    with ctxs[0]:
        vp.accumulate(bufs[0].ptr, bufs[1].ptr, W * H)
        got = bufs[0].download()
    assert part1.any() and not np.array_equal(part1, want)
    assert np.array_equal(got, want)
    # the default context still renders its own (unsharded) scene
    one.reset()
    vp.render_frames(one.ptr, 0, 6, P)
    assert np.array_equal(one.download(), want)
    for r, c in enumerate(ctxs):
        with c:
            bufs[r].free()
    for c in ctxs:
        c.destroy()
    one.free()


def test_event_ring_stays_bounded_and_oom_is_reported(vp, oracle):
    """a host that renders forever and never asks for timings must not accumulate events; a failed device
    allocation is VP_E_NOMEM (-5), not "no device"."""
    W, H = 32, 24
    vp.init_volume(oracle.julia(16), brick=1)
    vp.init_envmap(scenes.synthetic_env())
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(vp.EST_GLOBAL)
    vp.set_shard(0, 1)
    vp.set_lookahead(0)
    try:
        P = vp.make_param(W, H)
        buf = vp.DeviceBuffer(W, H)
        vp.render_time_ms(reset=True)
        for f in range(300):
            vp.render_kernel(buf.ptr, f, P)
        ms, n = vp.render_time_ms(reset=True)
        assert n == 300 and ms > 0                       # all 300 launches counted, although only 64 pairs are kept pending
        buf.free()
    finally:
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
    L = vp.lib()
    assert L.vp_malloc(1 << 50) is None                  # 1 PiB
    assert b"VP_E_NOMEM" in L.vp_last_error() or b"memory" in L.vp_last_error().lower()
    # the library is still usable afterwards (an out-of-memory error is not sticky)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, 2, vp.make_param(W, H))
    assert np.isfinite(buf.download()).all()
    buf.free()


def test_tuning_knobs_are_validated_and_never_change_results(tmp_path):
    """VP_WAIT_LANES=0 used to hang the persistent kernel, VP_BLOCKS_PER_CU=0 was an empty launch.  Out-of-range or
    malformed values are ignored with a message; valid ones change speed, never bits (here: the per-pixel tables, the light
    kernel and its overlap switched off one by one, a two-entry null-collision table on a medium whose weights are not 1)."""
    import hashlib
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, volpath as vp, scenes\n"
        "vp.set_device(0); W, H = 72, 40\n"
        "out = []\n"
        "for est in (0, 1):\n"
        "    vp.init_volume(vp.julia_volume(32), brick=4 if est else 1); vp.init_envmap(scenes.synthetic_env())\n"
        "    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER); vp.set_camera(); vp.set_estimator(est)\n"
        "    if est: vp.precompute_opacity(scenes.DEFAULT_SUN_DIR)\n"
        "    vp.set_rng(vp.RNG_PHILOX, (3, 1)); b = vp.DeviceBuffer(W, H); vp.render_frames(b.ptr, 0, int(__import__('os').environ.get('VP_TEST_FRAMES', '6')), vp.make_param(W, H, density=209.0, sigma_t=(0.3, 0.7, 1.0), albedo=(0.9, 0.8, 0.95)))\n"
        "    out.append(hashlib.sha1(b.download().tobytes()).hexdigest())\n"
        "print('HASH', *out)\n"
    ) % (os.path.join(ROOT, "cuda-volpath_amd"), os.path.join(ROOT, "tests"))
    hashes = {}
    # (VP_CHUNK_FRAMES_LOG2=8: a chunk of 128 samples cannot span 256 frames -- it would hold no pixel and the kernel would divide by
    # zero; refused since the chunk went from 256 to 128 samples in round 5.  7, the largest shape that exists, is rendered: one pixel.)
    for name, env in (("default", {}), ("bad", {"VP_WAIT_LANES": "0", "VP_BLOCKS_PER_CU": "0", "VP_STAGE_MB": "-5", "VP_WAIT_ITERS": "x", "VP_CHUNK_FRAMES_LOG2": "8"}),
                      ("default128", {"VP_TEST_FRAMES": "128"}), ("one_pixel_chunks128", {"VP_TEST_FRAMES": "128", "VP_CHUNK_FRAMES_LOG2": "7"}),
                      ("no_tables", {"VP_NO_CRAWL_TABLE": "1", "VP_NO_EMPTY_TABLE": "1"}), ("no_light", {"VP_NO_LIGHT": "1"}),
                      ("no_overlap", {"VP_NO_LIGHT_OVERLAP": "1", "VP_SETUP_LANES": "1", "VP_WAIT_LANES": "32"}),
                      ("short_table", {"VP_THR_TABLE": "2", "VP_LIGHT_WAIT_ITERS": "16", "VP_LIGHT_BLOCKS_PER_CU": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        hashes[name] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
        if name == "bad":
            assert r.stderr.count("ignoring VP_") == 5, r.stderr
    assert hashes.pop("default128") == hashes.pop("one_pixel_chunks128")
    assert len(set(hashes.values())) == 1, hashes


def test_full_size_tables_do_not_change_a_bit_for_any_camera():
    """The per-pixel tables, the pixel classes and the light kernel at BASELINE size (256^3, 800x600) for cameras the
    hand-picked tests never use: inside the volume, close to a face, far away and off-axis.  Same process state, same
    frames, the optimisations switched off by their knobs in a second process: identical accumulators (hash) for the
    global-majorant and the decomposition estimator, the latter across the frame-11 switch."""
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, volpath as vp\n"
        "from volpath import scene, host\n"
        "vp.set_device(0); out = []\n"
        "cams = [((0.05, 0.1, -0.02), (0.3, -0.2, 0.93), (0, 1, 0)), ((1.02, 0.4, 0.3), (-1, -0.3, -0.2), (0, 0, 1)),\n"
        "        ((-7.0, 5.0, 3.0), (0.75, -0.55, -0.36), (0, 1, 0)), ((0.3, 0.2, 2.5), (0.0, 0.0, -1.0), (1, 0, 0))]\n"
        "for wl, first, n in (('c2', 0, 3), ('c3ref', 9, 4)):\n"
        "    P, info = scene.setup(wl, rng_mode=vp.RNG_PHILOX7, last_frame=first + n)\n"
        "    for pos, fwd, up in cams:\n"
        "        f = np.array(fwd, np.float32); f /= np.linalg.norm(f)\n"
        "        u = np.cross(np.cross(f, np.array(up, np.float32)), f); u /= np.linalg.norm(u)\n"
        "        vp.set_camera(host.camera_matrix(np.array(pos, np.float32), f, u.astype(np.float32)))\n"
        "        b = vp.DeviceBuffer(P.width, P.height); vp.render_frames(b.ptr, first, n, P)\n"
        "        a = b.download(); out.append(hashlib.sha1(a.tobytes()).hexdigest()[:16] + ':%%.4g' %% float(a[..., :3].mean())); b.free()\n"
        "print('HASH', *out)\n"
    ) % (os.path.join(ROOT, "cuda-volpath_amd"), os.path.join(ROOT, "tests"))
    res = {}
    for name, env in (("default", {}), ("plain", {"VP_NO_CRAWL_TABLE": "1", "VP_NO_EMPTY_TABLE": "1", "VP_NO_LIGHT": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res[name] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()[1:]
    assert res["default"] == res["plain"], res
    assert len(set(res["default"])) == 8          # eight different images ...
    assert all(float(h.split(":")[1]) > 0 for h in res["default"])     # ... none of them black


def test_rccl_calls_of_the_multi_gpu_host_run_on_this_gpu():
    """The one-GPU box cannot run a collective between ranks, but it can run the calls: `volpath_render --rccl-selftest` loads
    librccl.so.1 (dlopen, as the N > 1 path does), creates a one-rank communicator with ncclCommInitAll and runs one
    ncclReduce(sum, float, 1280x720x4) in place on a stream; the sum over one rank is the rank's own data."""
    exe = os.path.join(ROOT, "cuda-volpath_amd", "volpath_render")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe, "--rccl-selftest", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ncclReduce(sum, float, 3686400) on device 0: ok" in r.stdout, r.stdout
    #  the run describes itself: RCCL's version and the communicator's rank count, so that the first run on
    # more than one GPU says what it ran on
    import re
    m = re.search(r"RCCL (\d+), ncclCommCount (\d+):", r.stdout)
    assert m and int(m.group(1)) > 20000 and int(m.group(2)) == 1, r.stdout


def test_torch_rccl_backend_runs_a_one_rank_reduce():
    """bench.py --gpus N sums the ranks' accumulators with torch.distributed over backend "nccl" (= RCCL on ROCm).  With one
    GPU the collective has one rank, but process-group creation, the communicator and a reduce on a side stream -- the calls of
    bench.py's N > 1 branch -- do run here."""
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
        "dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev)\n"
        "s = torch.cuda.Stream(device=dev)\n"
        "with torch.cuda.stream(s):\n"
        "    a = torch.arange(1280 * 720 * 4, device=dev, dtype=torch.float32).reshape(720, 1280, 4) * 0.5\n"
        "    b = a.clone(); dist.reduce(a, dst=0, op=dist.ReduceOp.SUM)\n"
        "s.synchronize(); dist.barrier(); ok = bool(torch.equal(a, b)); dist.destroy_process_group(); print('REDUCE', ok)\n"
    )
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "REDUCE True" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,world", [(160, 120, 1), (67, 45, 3), (1280, 720, 8), (9, 7, 4), (800, 600, 2)])
def test_pixel_lists_are_the_stable_partition_of_the_tile_order(vp, W, H, world):
    """The pixel lists are built on the GPU (pixlist_*_k) since round 3.  Restated here in numpy from their definition: the
    rank's pixels tile by tile (row-major 8x8 tiles owned per vp_tile_owner, row-major pixels within a tile, partial edge tiles
    clipped), stably partitioned by the class in the pixel table.  Every rank, ragged sizes, more ranks than tiles in a row."""
    import scenes
    grid = vp.julia_volume(32)
    vp.init_volume(grid, brick=1, linear=True)
    vp.init_envmap(scenes.synthetic_env())
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(vp.EST_GLOBAL)
    vp.set_tracking(0)
    vp.set_rng(vp.RNG_PHILOX7, (1, 2))
    P = vp.make_param(W, H)
    seen = np.zeros((H, W), int)
    try:
        for rank in range(world):
            vp.set_shard(rank, world)
            cls = vp.pixel_table(P)[..., 5].astype(int)
            want = [[], [], []]
            for ty in range((H + 7) // 8):
                for tx in range((W + 7) // 8):
                    if vp.tile_owner(tx, ty, world) != rank:
                        continue
                    for y in range(ty * 8, min(ty * 8 + 8, H)):
                        for x in range(tx * 8, min(tx * 8 + 8, W)):
                            want[cls[y, x]].append(y << 16 | x)
            got = vp.pixel_lists(P)
            for c in range(3):
                assert np.array_equal(got[c], np.asarray(want[c], np.uint32)), (rank, c)
                seen[got[c] >> 16, got[c] & 0xffff] += 1
        assert np.all(seen == 1)
    finally:
        vp.set_shard(0, 1)


@pytest.mark.gpu
def test_cloud_generator_matches_the_oracle_and_survives_the_ingest_path(vp, oracle, tmp_path):
    """The flagged synthetic cloud of workload c4f: GPU generator == the oracle's restatement bit for bit; float densities that
    are not binary; through dump_dense_volume -> loadBinaryFile + quantiser it arrives as the quantised grid."""
    from volpath import host
    for n, seed in ((40, 1), (33, 7)):
        g = vp.cloud_volume(n, seed)
        assert np.array_equal(g, oracle.cloud(n, seed))
        assert g.min() == 0.0 and 0.9 < g.max() <= 1.0
        frac = ((g > 0.02) & (g < 0.98)).mean()
        assert frac > 0.2, "the cloud is meant to have soft values, not a binary mask"
    path = str(tmp_path / "cloud.bin")
    assert host.dump_dense(path, g)
    q = host.load_binary(path, quantized=True)
    assert q.shape == g.shape and np.array_equal(q, host.quantize(g))


@pytest.mark.gpu
def test_class_times_and_prepare(vp):
    """vp_prepare builds the per-camera tables without rendering; vp_render_class_time_ms reports one event pair per kernel of a
    launch and the pixels of each class."""
    import scenes
    grid = vp.julia_volume(64)
    vp.init_volume(grid, brick=1, linear=True)
    vp.init_envmap(scenes.synthetic_env())
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(vp.EST_GLOBAL)
    vp.set_tracking(0)
    vp.set_shard(0, 1)
    vp.set_rng(vp.RNG_PHILOX7, (1, 2))
    W, H = 320, 240
    P = vp.make_param(W, H)
    vp.prepare(P)
    vp.render_class_time_ms(reset=True)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, 8, P)
    vp.synchronize()
    ms, px = vp.render_class_time_ms(reset=True)
    buf.free()
    assert sum(px.values()) == W * H and all(v > 0 for v in px.values())
    assert all(ms[k] > 0 for k in ms)
    total, launches = vp.render_time_ms(reset=True)
    assert launches >= 1 and total >= max(ms.values()) * 0.9
    ms2, _ = vp.render_class_time_ms(reset=True)
    assert all(v == 0 for v in ms2.values())


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c1", "c2"])
def test_camera_move_stops_lookahead_batches_in_flight(vp, workload):
    """The reference's interactive loop: render_kernel frame by frame, then the camera moves (host.cpp:617-632).  The batches
    render_kernel has staged ahead are told to stop while they run (LaunchDev::cancel, render_k<..., CANCEL>); frames already handed
    out must be whole, the frames after the move must be those of the new camera: both equal the explicit batch call without
    look-ahead, bit for bit.  At a size where batches ARE still running when the move comes (vp_lookahead_stats says so)."""
    from volpath import scene as vscene, host
    P, info = vscene.setup(workload, rng_mode=vp.RNG_PHILOX7, last_frame=80)
    W, H = P.width, P.height
    cam1 = info["camera"]
    cam2 = tuple(float(v) for v in host.camera_matrix((3.9 * np.cos(0.7), -0.78, 3.9 * np.sin(0.7)), (-np.cos(0.7), 0.2, -np.sin(0.7)), (0.0, 1.0, 0.0)))
    vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
    a, b, ref = vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)
    try:
        stopped = 0
        # (whether a batch is still running at the move is a matter of a few milliseconds: the scene is played with the move at several
        # points of the ramp; the images must be right every time, and a batch must have been stopped in flight at least once)
        for n1 in (34, 40, 36, 66, 35, 50, 98, 38):
            if stopped >= 1 and n1 not in (34, 40):   # (the first two always; the others only while no batch was caught running)
                break
            n2 = 12
            vp.set_camera(cam1)
            vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
            a.reset(); b.reset()
            vp.synchronize()
            l0, c0 = vp.lookahead_stats()
            for f in range(n1):
                vp.render_kernel(a.ptr, f, P)          # the reference's loop: one call per frame, a synchronisation after each
                vp.synchronize()                       # (host.cpp:631-632); the batch behind the one being served runs meanwhile
            vp.set_camera(cam2)                        # the move: what is staged ahead is dropped, what runs is stopped
            for f in range(n2):
                vp.render_kernel(b.ptr, f, P)
            l1, c1 = vp.lookahead_stats()              # (the batches are stopped where the new camera is first used)
            got_a, got_b = a.download(), b.download()
            assert l1 - l0 >= 2, "the look-ahead did not ramp"
            stopped += c1 - c0
            vp.set_lookahead(0)
            ref.reset()
            vp.render_frames(ref.ptr, 0, n2, P)
            assert np.array_equal(got_b, ref.download()), ("frames after the move", n1)
            vp.set_camera(cam1)
            ref.reset()
            vp.render_frames(ref.ptr, 0, n1, P)
            assert np.array_equal(got_a, ref.download()), ("frames handed out before the move", n1)
        assert stopped >= 1, "no batch was in flight at any of the moves: the test does not exercise the cancellation"
    finally:
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
        a.free(); b.free(); ref.free()


# ------------------------------------------------------------------ c4f: the same shape with a volume that fills the frame
@pytest.fixture(scope="module")
def c4f(vp):
    from volpath import scene as vscene
    P, info = vscene.setup("c4f", rng_mode=vp.RNG_PHILOX7, last_frame=64)
    assert (P.width, P.height) == (1280, 720) and info["n"] == 512 and info["chromatic"] and "SYNTHETIC" in info["volume"]
    return P, info


def test_c4f_fills_the_frame_and_has_soft_densities(vp, c4f):
    """What the frame-filling stand-in is for: every camera ray enters the box, at least 70 % of the pixels are `general` (a cloud
    leaves few rays that never meet the medium -- the Julia stand-in c4s leaves 88 %), and the bound table sees minima and
    maxima that differ from 0 / 255 (non-binary densities: quirk Q4's local majorant and the control component are active)."""
    P, info = c4f
    vp.set_estimator(vp.EST_DECOMP)
    vp.set_shard(0, 1)
    cls = np.bincount(vp.pixel_table(P)[..., 5].astype(int).ravel(), minlength=3) / float(P.width * P.height)
    assert cls[2] == 0.0 and cls[0] >= 0.70, cls
    bounds, brick, radius = vp.bound_table()
    assert brick == 16
    mx, mn = bounds[..., 0], bounds[..., 1]
    assert ((mx > 0) & (mx < 255)).mean() > 0.05 and (mn > 0).mean() > 0.01
    assert 0.03 < info["occupancy"] < 0.5


def test_c4f_batched_equals_frame_by_frame_and_shards_add_up(vp, c4f):
    P, info = c4f
    W, H = P.width, P.height
    vp.set_estimator(vp.EST_DECOMP)
    vp.set_rng(vp.RNG_PHILOX7, (0x9E3779B9, 0x85EBCA6B))
    vp.set_shard(0, 1)
    a, b = vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)
    frames = range(10, 13)                                  # across the frame-11 estimator switch (quirk Q5)
    vp.render_frames(a.ptr, frames[0], len(frames), P)
    for f in frames:
        vp.render_kernel(b.ptr, f, P)
    whole = a.download()
    assert np.array_equal(whole, b.download())
    assert np.isfinite(whole).all() and (whole >= 0).all()
    assert (whole[..., 3] > 0).mean() > 0.6                 # most pixels scatter
    b.free()
    from volpath import dist as vd
    for world in (2, 8):
        tot = np.zeros_like(whole)
        for r in range(world):
            a.reset()
            vp.set_shard(r, world)
            vp.render_frames(a.ptr, frames[0], len(frames), P)
            part = a.download()
            assert not part[~vd.owned_mask(r, world, W, H)].any()
            tot += part
        assert np.array_equal(tot, whole), world
    vp.set_shard(0, 1)
    a.free()


def test_cloud_workload_is_bit_exact_at_reduced_size(vp, oracle, tmp_path):
    """The c4f recipe -- synthetic cloud as float, dump_dense_volume, loadBinaryFile + quantiser, chromatic preset #1, the
    workload's own camera, decomposition tracking over a brick table -- at 48^3 / 96x54 against the oracle, tolerance 0,
    images and work counters, across the frame-11 switch; Philox2x32-7 (shadow sub-streams, sun table, light class)."""
    from volpath import host, scene as vscene
    cfg = vscene.WORKLOADS["c4f"]
    n, W, H = 48, 96, 54
    vol = vp.cloud_volume(n, cfg["seed"])
    path = str(tmp_path / "cloud.bin")
    assert host.dump_dense(path, vol)
    grid = host.load_binary(path, quantized=True)
    cam = vscene.camera_of(cfg)
    env = scenes.synthetic_env()
    for est, brick in ((oracle.EST_DECOMP, 4), (oracle.EST_GLOBAL, 1)):
        osc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, brick=brick, estimator=est,
                                 rng_mode=oracle.RNG_PHILOX7, seed=(3, 4), inv_view=cam)
        osc.precompute_opacity()
        oP, vP = oracle.default_param(W, H, density=60.0), vp.make_param(W, H, density=60.0)
        oracle.mat(oP, *scenes.PRESET1)
        vp.mat(vP, *scenes.PRESET1)
        vp.init_volume(grid, brick=brick, linear=True)
        vp.init_envmap(env)
        vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
        vp.set_camera(cam)
        vp.set_estimator(est)
        vp.set_tracking(0)
        vp.set_rng(vp.RNG_PHILOX7, (3, 4))
        vp.set_shard(0, 1)
        vp.precompute_opacity(scenes.DEFAULT_SUN_DIR)
        frames = range(9, 13)
        ref, tot = None, None
        for f in frames:
            ref, c = osc.render_frame(oP, f, ref)
            d = c.as_dict()
            tot = d if tot is None else {k: tot[k] + d[k] for k in d}
        buf = vp.DeviceBuffer(W, H)
        vp.enable_counters(True)
        vp.read_counters(reset=True)
        vp.render_frames(buf.ptr, frames[0], len(frames), vP)
        cnt = vp.read_counters(reset=True)
        vp.enable_counters(False)
        assert np.array_equal(buf.download(), ref), est
        for k in ("density_lookups", "bound_lookups", "env_lookups", "scatters", "opacity_lookups"):
            assert cnt[k] == tot[k], (est, k, cnt[k], tot[k])
        buf.reset()
        vp.render_frames(buf.ptr, frames[0], len(frames), vP)          # the timed kernels (sun table, one event visit per collision)
        assert np.array_equal(buf.download(), ref), est
        buf.free()
        assert (ref[..., 3] > 0).mean() > 0.5
    vp.set_camera()


@pytest.mark.parametrize("est,brick", [(0, 1), (1, 8), (1, 1)])
def test_results_do_not_depend_on_the_tuning_knobs(vp, oracle, est, brick):
    """"results never depend on the knobs".  A context reads the VP_* environment when it is created:
    contexts created under different settings -- tables, sun table, constant light class, light kernel, helper workgroups, LDS
    stage, cell order, chunk shape, wait / set-up / end policies, one staged frame per launch, the approach kernels -- render the same bits as the
    default context, which in turn equals the oracle."""
    W, H = 96, 64
    grid = oracle.julia(64)
    env = scenes.synthetic_env()
    frames = range(8, 14)                       # across the frame-11 switch of the live kernel

    def scene():
        vp.init_volume(grid, brick=brick, linear=True)
        vp.init_envmap(env)
        vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
        vp.set_camera()
        vp.set_estimator(est)
        vp.set_tracking(0)
        vp.set_rng(vp.RNG_PHILOX7, (4, 2))
        vp.set_shard(0, 1)
        if est == 1:
            vp.precompute_opacity(scenes.DEFAULT_SUN_DIR)

    def render():
        P = vp.make_param(W, H)
        buf = vp.DeviceBuffer(W, H)
        vp.render_frames(buf.ptr, frames[0], len(frames), P)
        out = buf.download()
        buf.free()
        return out

    scene()
    want = render()
    osc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, brick=brick, estimator=est,
                             rng_mode=oracle.RNG_PHILOX7, seed=(4, 2))
    if est == 1:
        osc.precompute_opacity()
    ref, oP = None, oracle.default_param(W, H)
    for f in frames:
        ref, _ = osc.render_frame(oP, f, ref)
    assert np.array_equal(want, ref)
    settings = [
        dict(VP_NO_SUN_CLIP="1"), dict(VP_NO_LIGHT_CONST="1"), dict(VP_NO_LIGHT="1"), dict(VP_NO_CRAWL_TABLE="1", VP_NO_EMPTY_TABLE="1"),
        dict(VP_NO_LDS_HELPER="1"), dict(VP_NO_LDS_BOUNDS="1"), dict(VP_CELL_BRICKS="1"), dict(VP_CHUNK_FRAMES_LOG2="1"),
        dict(VP_WAIT_LANES="5", VP_WAIT_ITERS="4", VP_SETUP_LANES="1", VP_END_LANES="1"), dict(VP_END_LANES="40", VP_SETUP_LANES="33"),
        dict(VP_STAGE_MB="1", VP_BLOCKS_PER_CU="2"), dict(VP_NO_LIGHT_OVERLAP="1", VP_NO_LIGHT_CONST="1"),
        # the camera rays' walk ahead of the integrator (approach_k / approach_local_k): off, cut short after a few steps / segments
        dict(VP_NO_APPROACH="1"), dict(VP_NO_APPROACH_LOCAL="1"), dict(VP_NO_APPROACH_TABLE="1"), dict(VP_NO_APPROACH_TABLE="1", VP_APPROACH_STEPS="3"), dict(VP_APPROACH_STEPS="3"), dict(VP_APPROACH_STEPS="0"),
        dict(VP_APPROACH_STEPS="40", VP_NO_LDS_HELPER="1"), dict(VP_APPROACH_FRAMES_LOG2="0"), dict(VP_APPROACH_FRAMES_LOG2="1"),
        # exit flights (paths that can only leave the box are ended at once): off, tested at once, tested late
        dict(VP_NO_EXIT="1"), dict(VP_EXIT_K="1"), dict(VP_EXIT_K="40"), dict(VP_EXIT_LOCAL="1"), dict(VP_EXIT_LOCAL="1", VP_EXIT_K="2"), dict(VP_EXIT_LOCAL="0"),
        # per-pixel constants staged for every frame instead of once per launch
        dict(VP_NO_CONST_ROWS="1"), dict(VP_NO_CONST_ROWS="1", VP_NO_LIGHT_CONST="1"),
        # round 5: the optical-depth table built by the gather kernel instead of through LDS tiles; read by the integrator from the plain
        # table instead of the packed cells (what happens by itself where the 8x copy cannot be allocated: ADVICE r4); the chromatic
        # kernels' wait policy pinned to the general default
        # the brick table through LDS as 16-bit pairs instead of 2-bit codes (and without the codes for a chromatic medium)
        dict(VP_NO_LDS_COMPACT="1"), dict(VP_LDS_COMPACT_CHROMATIC="0"), dict(VP_NO_LDS_COMPACT="1", VP_NO_LDS_HELPER="1"),
        # ... as 16-bit pairs where it cannot go as codes (the default reads it from global memory then), with and without the helper
        dict(VP_LDS_PAIRS="1"), dict(VP_LDS_PAIRS="1", VP_NO_LDS_COMPACT="1"), dict(VP_LDS_PAIRS="1", VP_NO_LDS_COMPACT="1", VP_NO_LDS_HELPER="1"),
        # every volume treated as dense (no sun table, no approach walk for the decomposition estimator) / none
        dict(VP_DENSE_PERCENT="0"), dict(VP_DENSE_PERCENT="101"),
        dict(VP_NO_OPACITY_LDS="1"), dict(VP_NO_OPACITY_CELLS="1"), dict(VP_NO_OPACITY_CELLS="1", VP_NO_OPACITY_LDS="1"), dict(VP_WAIT_LANES="24"),
    ]
    for env_set in settings:
        saved = {k: os.environ.get(k) for k in env_set}
        os.environ.update(env_set)
        try:
            ctx = vp.Context(0)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        try:
            with ctx:
                scene()
                got = render()
        finally:
            ctx.destroy()
        assert np.array_equal(got, want), env_set
    scene()                                     # the default context is untouched by all of it
    assert np.array_equal(render(), want)


@pytest.mark.parametrize("workload", ["c3", "c4s", "c4f"])
def test_full_size_opacity_tables_match_the_oracle_on_sampled_voxels(vp, oracle, workload):
    """precompute_opacity (A10, kernel.cu:483-553) AT FULL SIZE: the optical-depth table the GPU builds for the 256^3 and 512^3 bench
    volumes against the oracle's march of the same voxel (vpo_opacity_voxel), tolerance 0, on 4096 + 56 voxels per light direction
    -- random ones, the eight corners, and voxels on every face -- for the default sun and for two more directions, so that the
    marches leave the box through each of its six faces.  (The whole table is compared at 32^3 and the fuzz sizes; the oracle's
    whole-table precompute is an N^4 march.)  This is what closes the loop of
    test_full_size_workloads_match_the_oracle_on_sampled_pixels, whose oracle reads the GPU-built table from frame 11 on."""
    from volpath import scene as vscene
    cfg = vscene.WORKLOADS[workload]
    n = cfg["n"]
    P, info = vscene.setup(workload, rng_mode=2, last_frame=0)
    env, sun_dir, sun_power = info["sunsky"]
    grid = vscene.host_volume(workload)
    osc = oracle.OracleScene(grid, env, sun_dir, sun_power, brick=cfg["brick"], estimator=cfg["est"], rng_mode=2,
                             seed=(0x9E3779B9, 0x85EBCA6B), inv_view=vscene.camera_of(cfg))
    rng = np.random.default_rng(11)
    ijk = [rng.integers(0, n, size=(4096, 3))]
    ijk.append(np.array([[a, b, c] for a in (0, n - 1) for b in (0, n - 1) for c in (0, n - 1)]))
    for axis in range(3):                      # eight voxels on each of the six faces
        for side in (0, n - 1):
            f = rng.integers(0, n, size=(8, 3))
            f[:, axis] = side
            ijk.append(f)
    ijk = np.concatenate(ijk).astype(np.int32)
    exits = set()
    for d in (sun_dir, (0.6, -0.5, 0.62), (-0.7, 0.1, 0.7)):
        d = np.asarray(d, np.float32)
        d = tuple(float(v) for v in d / np.float32(np.sqrt(np.float32((d * d).sum()))))
        vp.precompute_opacity(d)
        table = vp.opacity_table((n, n, n))                      # [k][j][i]
        got = table[ijk[:, 2], ijk[:, 1], ijk[:, 0]]
        want = osc.opacity_voxels(ijk, light_dir=d)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, f"{len(bad)} of {len(ijk)} voxels differ toward {d}: first {ijk[bad[0]]} {got[bad[0]]} vs {want[bad[0]]}"
        assert (want > 0).mean() > 0.02          # the marches meet the medium
        # through which face the marches leave (float64 slab test from the voxel centres)
        c = ((ijk + 0.5) / n) * 2.0 - 1.0
        with np.errstate(divide="ignore"):
            t = np.where(np.array(d) > 0, (1.0 - c) / np.array(d, np.float64), (-1.0 - c) / np.array(d, np.float64))
        t = np.where(np.array(d) == 0, np.inf, t)
        ax = t.argmin(1)
        exits |= {(int(a), bool(d[a] > 0)) for a in ax}
    assert len(exits) == 6, exits
    vp.precompute_opacity(sun_dir)


@pytest.mark.parametrize("workload,first,rng_mode", [("c2", 0, 2), ("c3", 0, 2), ("c4s", 0, 2), ("c4f", 0, 2), ("c3", 9, 2), ("c4f", 9, 2),
                                                     ("c3ref", 0, 0), ("c3ref", 9, 0), ("c2", 0, 0)])
def test_full_size_workloads_match_the_oracle_on_sampled_pixels(vp, oracle, workload, first, rng_mode):
    """The bench workloads AT FULL SIZE -- BASELINE configs[1] and [2] (Julia-256^3, 800x600) and the two 512^3 / 1280x720 chromatic
    stand-ins of configs[3] (Julia and the frame-filling cloud through the dense-dump ingest path), the baked Hosek sky, the bench's
    Philox2x32-7 streams -- against the oracle pixel by pixel on a sample of the image: up to 1200 pixels that scatter, 300 whose
    ray meets empty cells only and 300 that miss the box, four frames, tolerance 0.  (The whole-image comparisons stop at 120x56
    and 64^3; the oracle takes a quarter of an hour per full-size frame.)  first = 9: frames 9..12, across the live kernel's
    switch to the optical-depth table at frame 11 (quirk Q5); the oracle's N^4 precompute of that table is not affordable at these
    sizes, so it reads the table the GPU built -- itself compared with the oracle's march voxel by voxel on 4152 sampled voxels per
    light direction at these very sizes (test_full_size_opacity_tables_match_the_oracle_on_sampled_voxels) and whole at 32^3.
    rng_mode 0: the reference's own sampler.h streams, on c3ref = the reference's live configuration (its estimator, its per-voxel
    bound table, its default scene and window-independent camera) -- what a run of the reference itself computes, sample by sample."""
    import ctypes as C
    from volpath import scene as vscene
    cfg = vscene.WORKLOADS[workload]
    P, info = vscene.setup(workload, rng_mode=rng_mode, key=(0x9E3779B9, 0x85EBCA6B), last_frame=first + 4)
    W, H, frames = P.width, P.height, 4
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, first, frames, P)
    got = buf.download()
    buf.free()
    # the camera rays were walked ahead of the integrator (approach_k / approach_local_k) -- except on the frame-filling cloud: more than
    # 40 % of its cells have a non-empty neighbour, so the tables that pay in empty space (that walk, the sun table) are off (round 5)
    assert vp.last_approach_mode() == (0 if workload == "c4f" else 1)
    cls = vp.pixel_table(P)[..., 5].astype(int)
    rng = np.random.default_rng(5)
    pick = []
    for c, n in ((0, 1200), (1, 300), (2, 300)):
        ys, xs = np.nonzero(cls == c)
        if len(ys):
            sel = rng.choice(len(ys), min(n, len(ys)), replace=False)
            pick += list(zip(ys[sel], xs[sel]))
    env, sun_dir, sun_power = info["sunsky"]
    grid = vscene.host_volume(workload)
    osc = oracle.OracleScene(grid, env, sun_dir, sun_power, brick=cfg["brick"], estimator=cfg["est"], rng_mode=rng_mode,
                             seed=(0x9E3779B9, 0x85EBCA6B), inv_view=vscene.camera_of(cfg))
    oP = oracle.default_param(W, H)
    if cfg["chromatic"]:
        oracle.mat(oP, *scenes.PRESET1)
    if first + frames > 11 and cfg["est"] == oracle.EST_DECOMP:
        n = cfg["n"]
        osc.opacity = vp.opacity_table((n, n, n))
        osc.S.opacity = osc.opacity.ctypes.data
    L = oracle.lib()
    out = (C.c_float * 4)()
    cnt = oracle.Counters()
    bad = 0
    for y, x in pick:
        acc = np.zeros(4, np.float32)
        for f in range(first, first + frames):
            L.vpo_render_sample(C.byref(osc.S), C.byref(oP), int(x), int(y), f, out, C.byref(cnt))
            acc = acc + np.array(out[:], np.float32)
        if not np.array_equal(acc, got[y, x]):
            bad += 1
    assert bad == 0, f"{bad} of {len(pick)} sampled pixels differ"
    assert (got[..., 3] > 0).mean() > 0.05
    vp.set_camera()


def test_ragged_grid_beyond_512_matches_the_oracle_on_sampled_pixels_and_voxels(vp, oracle):
    """"nothing tests a non-cubic or > 512^3 grid against [the packed optical-depth cells]".  A soft 640 x 352 x 416 volume
    (x beyond 512, every edge different, none a multiple of the 16-voxel brick on y) in the box the reference derives from the
    dimensions (kernel.cu:373-378), chromatic preset, the live estimator with 16^3 bricks, frames 9..12 -- across the switch to the
    optical-depth table -- on the bench's streams: (1) the table opacity_lds_k builds (8x8x8 blocks of a grid whose y and z edges are
    not multiples of 8 or of the tile) against the oracle's march on 2048 + 56 voxels, tolerance 0; (2) the image against the oracle
    on up to 700 sampled pixels, tolerance 0, the integrator reading that table from its packed cells (3 GB here)."""
    import ctypes as C
    from scipy import ndimage
    nx, ny, nz = 640, 352, 416
    rs = np.random.default_rng(77)
    low = rs.random((nz // 16, ny // 16, nx // 16)).astype(np.float32)
    g = ndimage.zoom(low, 16, order=1)                      # soft: trilinear upsampling of a coarse random field
    g = np.clip((g - 0.45) * 3.0, 0.0, 1.0)
    grid = np.ascontiguousarray((g * 255).astype(np.uint8))
    assert grid.shape == (nz, ny, nx) and 0.2 < (grid > 0).mean() < 0.8
    env = scenes.synthetic_env()
    W, H, first, frames = 160, 96, 9, 4
    vp.init_volume(grid, brick=16, linear=True)
    vp.init_envmap(env)
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(vp.EST_DECOMP)
    vp.set_tracking(0)
    vp.set_rng(vp.RNG_PHILOX7, (5, 6))
    vp.set_shard(0, 1)
    vp.precompute_opacity(scenes.DEFAULT_SUN_DIR)
    P = vp.make_param(W, H, density=120.0)
    vp.mat(P, *scenes.PRESET1)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, first, frames, P)
    got = buf.download()
    buf.free()
    osc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, brick=16, estimator=oracle.EST_DECOMP,
                             rng_mode=oracle.RNG_PHILOX7, seed=(5, 6))
    # (1) the table
    table = vp.opacity_table((nz, ny, nx))
    ijk = [np.stack([rs.integers(0, nx, 2048), rs.integers(0, ny, 2048), rs.integers(0, nz, 2048)], 1)]
    ijk.append(np.array([[a, b, c] for a in (0, nx - 1) for b in (0, ny - 1) for c in (0, nz - 1)]))
    for axis, n in enumerate((nx, ny, nz)):
        for side in (0, n - 1):
            f = np.stack([rs.integers(0, nx, 8), rs.integers(0, ny, 8), rs.integers(0, nz, 8)], 1)
            f[:, axis] = side
            ijk.append(f)
    ijk = np.concatenate(ijk).astype(np.int32)
    want = osc.opacity_voxels(ijk)
    tgot = table[ijk[:, 2], ijk[:, 1], ijk[:, 0]]
    bad = np.nonzero(tgot != want)[0]
    assert len(bad) == 0, f"{len(bad)} of {len(ijk)} voxels differ: first {ijk[bad[0]]} {tgot[bad[0]]} vs {want[bad[0]]}"
    assert (want > 0).mean() > 0.5
    # (2) the image, the oracle reading the table just checked
    osc.opacity = table
    osc.S.opacity = osc.opacity.ctypes.data
    oP = oracle.default_param(W, H, density=120.0)
    oracle.mat(oP, *scenes.PRESET1)
    ys, xs = np.nonzero(got[..., 3] > 0)
    sel = rs.choice(len(ys), min(600, len(ys)), replace=False)
    pick = list(zip(ys[sel], xs[sel]))
    ys0, xs0 = np.nonzero(got[..., 3] == 0)
    if len(ys0):
        sel0 = rs.choice(len(ys0), min(100, len(ys0)), replace=False)
        pick += list(zip(ys0[sel0], xs0[sel0]))
    L = oracle.lib()
    out = (C.c_float * 4)()
    cnt = oracle.Counters()
    wrong = 0
    for y, x in pick:
        acc = np.zeros(4, np.float32)
        for f in range(first, first + frames):
            L.vpo_render_sample(C.byref(osc.S), C.byref(oP), int(x), int(y), f, out, C.byref(cnt))
            acc = acc + np.array(out[:], np.float32)
        if not np.array_equal(acc, got[y, x]):
            wrong += 1
    assert wrong == 0, f"{wrong} of {len(pick)} sampled pixels differ"
    assert cnt.opacity_lookups > 1000 and (got[..., 3] > 0).mean() > 0.2        # deep paths read the table; the volume is in view
    vp.set_camera()


def test_lookahead_habit_decays_when_nobody_reads_the_speculative_batches(vp):
    """once a caller had been served a staged frame, EVERY first frame of a run had a speculative 32-frame batch queued
    beside it -- an interactive drag (each call is frame 0 of a new camera) paid for a batch per move that the next move threw away.
    Now a speculative batch that goes unserved clears the habit until a staged frame is served again: a drag of twelve moves after a
    converged run launches ONE more batch, not twelve; the images stay the explicit launches' (the look-ahead is invisible)."""
    from volpath import scene as vscene, host
    P, info = vscene.setup("c1", rng_mode=vp.RNG_PHILOX7, last_frame=80)
    W, H = P.width, P.height
    vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
    a, ref = vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)
    try:
        for f in range(40):                              # a converging run: the ramp starts, staged frames are served -> the habit is on
            vp.render_kernel(a.ptr, f, P)
            vp.synchronize()
        l0, _ = vp.lookahead_stats()
        cams = [tuple(float(v) for v in host.camera_matrix((3.9 * np.cos(t), -0.78, 3.9 * np.sin(t)), (-np.cos(t), 0.2, -np.sin(t)), (0.0, 1.0, 0.0)))
                for t in np.linspace(0.1, 1.2, 12)]
        for cam in cams:                                 # the drag: one frame per camera
            vp.set_camera(cam)
            a.reset()
            vp.render_kernel(a.ptr, 0, P)
            vp.synchronize()
        l1, _ = vp.lookahead_stats()
        assert l1 - l0 <= 2, f"{l1 - l0} batches launched during a drag of {len(cams)} moves"
        got = a.download()
        vp.set_lookahead(0)
        ref.reset()
        vp.render_frames(ref.ptr, 0, 1, P)
        assert np.array_equal(got, ref.download())
        # ... and the habit comes back with the next converging run
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
        a.reset()
        for f in range(40):
            vp.render_kernel(a.ptr, f, P)
            vp.synchronize()
        l2, _ = vp.lookahead_stats()
        assert l2 - l1 >= 1
        ref.reset()
        vp.set_lookahead(0)
        vp.render_frames(ref.ptr, 0, 40, P)
        assert np.array_equal(a.download(), ref.download())
    finally:
        vp.set_lookahead(vp.LOOKAHEAD_DEFAULT)
        vp.set_camera()
        a.free(); ref.free()


def test_exit_flights_switched_on_after_the_volume_build_their_table(vp, oracle):
    """vp_set_exit_flights(1 / 2) after a volume was initialised with exit flights off did nothing, and vp_get_exit_table
    then failed.  A context created under VP_NO_EXIT=1 initialises its volume without the table; switching the mode on builds it, the
    table equals the one a default context builds, and the image is the same bits either way."""
    grid = oracle.julia(48)
    env = scenes.synthetic_env()
    W, H = 64, 48

    def scene():
        vp.init_volume(grid, brick=1, linear=True)
        vp.init_envmap(env)
        vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
        vp.set_camera()
        vp.set_estimator(vp.EST_GLOBAL)
        vp.set_tracking(0)
        vp.set_rng(vp.RNG_PHILOX7, (9, 9))
        vp.set_shard(0, 1)

    def render():
        P = vp.make_param(W, H)
        buf = vp.DeviceBuffer(W, H)
        vp.render_frames(buf.ptr, 0, 6, P)
        out = buf.download()
        buf.free()
        return out

    scene()
    want_img, want_tab = render(), vp.exit_table((48, 48, 48))
    saved = os.environ.get("VP_NO_EXIT")
    os.environ["VP_NO_EXIT"] = "1"
    ctx = vp.Context(0)
    try:
        with ctx:
            scene()
            with pytest.raises(vp.VolpathError):
                vp.exit_table((48, 48, 48))              # off: no table
            off_img = render()
            vp.set_exit_flights(1)                       # on, after the volume: the table is built now
            assert np.array_equal(vp.exit_table((48, 48, 48)), want_tab)
            on_img = render()
    finally:
        ctx.destroy()
        if saved is None:
            del os.environ["VP_NO_EXIT"]
        else:
            os.environ["VP_NO_EXIT"] = saved
    assert np.array_equal(off_img, want_img) and np.array_equal(on_img, want_img)
