"""CPU suite, part 1: the oracle against reference-derived anchors and committed golden vectors."""
import json
import os

import numpy as np
import pytest

import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ANCH = json.load(open(os.path.join(HERE, "golden", "ref_anchors.json")))
GOLD = np.load(os.path.join(HERE, "golden", "oracle_renders.npz"))


def test_hash_and_cudarng_match_reference_anchors(oracle):
    for k, v in ANCH["hash"].items():
        assert oracle.lib().vpo_hash(int(k)) == v
    for k, vals in ANCH["cudarng"].items():
        x, y, f = (int(t) for t in k.split(","))
        got = oracle.rng_stream(oracle.RNG_SAMPLERH, x, y, f, len(vals))
        assert np.array_equal(got, np.array(vals, np.float32))


def test_philox_random123_kat(oracle):
    for kat in ANCH["philox2x32_10_random123_kat"]:
        assert oracle.philox(kat["ctr"], kat["key"]) == kat["out"]
    # draw n of sample (x,y,frame) = word n%2 of philox2x32_10((n//2, x<<16|y), (frame ^ k0) + k1)
    s = oracle.rng_stream(oracle.RNG_PHILOX, 3, 5, 7, 8, key=(11, 22))
    for n in range(8):
        w = oracle.philox([n // 2, (3 << 16) | 5], ((7 ^ 11) + 22) & 0xffffffff)[n % 2]
        assert s[n] == np.float32(np.uint32(0x3f800000 | (w >> 9)).view(np.float32) - np.float32(1.0))


def test_philox7_random123_kat(oracle):
    """VP_RNG_PHILOX7: philox2x32 with 7 rounds, Random123's published known answers (kat_vectors) and the stream layout"""
    for kat in ANCH["philox2x32_7_random123_kat"]:
        assert oracle.philox(kat["ctr"], kat["key"], rounds=7) == kat["out"]
    s = oracle.rng_stream(oracle.RNG_PHILOX7, 3, 5, 7, 8, key=(11, 22))
    for n in range(8):
        w = oracle.philox([n // 2, (3 << 16) | 5], ((7 ^ 11) + 22) & 0xffffffff, rounds=7)[n % 2]
        assert s[n] == np.float32(np.uint32(0x3f800000 | (w >> 9)).view(np.float32) - np.float32(1.0))
    # statistical sanity of the streams the integrator sees: per-pixel streams are uncorrelated and uniform
    a = np.stack([oracle.rng_stream(oracle.RNG_PHILOX7, x, 9, 4, 512, key=(1, 2)) for x in range(64)])
    assert abs(a.mean() - 0.5) < 0.01 and abs(a.var() - 1 / 12) < 0.005
    c = np.corrcoef(a)
    assert np.abs(c - np.eye(64)).max() < 0.25
    assert abs(np.corrcoef(a[:, :-1].ravel(), a[:, 1:].ravel())[0, 1]) < 0.02


def test_rng_float_range(oracle):
    for mode in (0, 1, 2):
        s = oracle.rng_stream(mode, 17, 4, 99, 4096)
        assert s.min() >= 0.0 and s.max() < 1.0


def test_julia_voxeliser(oracle):
    assert np.array_equal(oracle.julia(32), GOLD["julia32"])
    occ = oracle.julia(128).mean() / 255.0
    assert abs(occ - ANCH["julia_occupancy"]) < 2e-4
    g = oracle.julia(32)
    assert set(np.unique(g)) <= {0, 255}


def test_bound_radius(oracle):
    for n, r in ANCH["bound_radius"].items():
        assert oracle.bound_radius(int(n)) == r


@pytest.mark.parametrize("brick", [1, 4])
def test_bounds_against_brute_force(oracle, brick):
    rng = np.random.default_rng(2)
    g = (rng.random((10, 12, 16)) * 255).astype(np.uint8)
    g[rng.random(g.shape) < 0.5] = 0
    r = 2
    got = oracle.bounds(g, r, brick)
    nz, ny, nx = g.shape
    for bk in range(got.shape[0]):
        for bj in range(got.shape[1]):
            for bi in range(got.shape[2]):
                k0, k1 = max(bk * brick - r, 0), min(bk * brick + brick - 1 + r, nz - 1)
                j0, j1 = max(bj * brick - r, 0), min(bj * brick + brick - 1 + r, ny - 1)
                i0, i1 = max(bi * brick - r, 0), min(bi * brick + brick - 1 + r, nx - 1)
                w = g[k0:k1 + 1, j0:j1 + 1, i0:i1 + 1]
                assert got[bk, bj, bi, 0] == w.max() and got[bk, bj, bi, 1] == w.min()
    gf = g.astype(np.float32) / 255
    assert np.array_equal(oracle.bounds(gf, r, brick)[..., 0], got[..., 0].astype(np.float32) / 255)


def test_golden_renders_regression(oracle):
    grid = GOLD["julia32"]
    env = scenes.synthetic_env()
    for est, name in ((oracle.EST_DECOMP, "decomp"), (oracle.EST_GLOBAL, "global"), (oracle.EST_BOUNDED, "bounded")):
        for rng, rname in ((oracle.RNG_SAMPLERH, "samplerh"), (oracle.RNG_PHILOX, "philox")):
            sc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=est,
                                    rng_mode=rng, seed=(123, 456))
            sc.precompute_opacity()
            if est == oracle.EST_DECOMP and rng == oracle.RNG_SAMPLERH:
                assert np.array_equal(sc.opacity, GOLD["opacity32"])
                assert np.array_equal(sc.bounds, GOLD["bounds32_r1"])
            P = oracle.default_param(64, 48)
            acc = None
            for f in range(14):
                acc, _ = sc.render_frame(P, f, acc)
            assert np.array_equal(acc, GOLD[f"{name}_{rname}_f0_13"]), (name, rname)
    for tag, kw in (("mis", dict(env_mis=True)), ("scalar", dict(track_mode=1)), ("multichannel", dict(track_mode=2))):
        sc = oracle.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=oracle.EST_DECOMP,
                                rng_mode=oracle.RNG_PHILOX, seed=(123, 456), **kw)
        sc.precompute_opacity()
        P = oracle.default_param(64, 48, density=150.0, g=0.6, albedo=(0.9, 0.8, 0.7), sigma_t=(1.0, 0.7, 0.45))
        acc = None
        for f in range(8, 14):
            acc, _ = sc.render_frame(P, f, acc)
        assert np.array_equal(acc, GOLD[f"decomp_philox_{tag}_f8_13"]), tag


def test_thread_count_invariance(oracle):
    grid = GOLD["julia32"]
    sc = oracle.OracleScene(grid, scenes.synthetic_env(), scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    P = oracle.default_param(64, 48)
    a, _ = sc.render_frame(P, 3, threads=1)
    b, _ = sc.render_frame(P, 3, threads=4)
    assert np.array_equal(a, b)


def test_work_counters_match_reference_probe(oracle):
    """per-sample work of the reference's live kernel on Julia 256^3 @ 800x600."""
    ref = ANCH["work_counters_julia256_800x600"]
    g = oracle.julia(256)
    sc = oracle.OracleScene(g, scenes.synthetic_env(), scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    P = oracle.default_param(800, 600)
    acc, c = sc.render_frame(P, 0)
    n = c.samples
    assert abs(c.density_lookups / n / ref["density_lookups"] - 1) < 0.03
    assert abs(c.bound_lookups / n / ref["bound_lookups"] - 1) < 0.01
    assert c.env_lookups == n
    assert abs((acc[..., 3] == 0).mean() - ref["zero_scatter_pixel_fraction"]) < 0.01


def test_scatter_distribution_matches_reference_probe(oracle):
    """SURVEY section 7 ("Hard parts"): the reference's live kernel, Julia 256^3 at 800x600, frames 0..7, per-pixel mean
    scatters per sample: 88 % zeros, p90 13.9, p99 48.9, max 205.  The oracle on the same frames with the same (sampler.h)
    streams and the baked default sky -- body and tail of the distribution agree; trajectories cannot (libm differs)."""
    from volpath import host
    ref = ANCH["work_counters_julia256_800x600"]
    env, sun_dir, sun_power = host.bake_sunsky(0.5, 0.2, 1024, 512)
    sc = oracle.OracleScene(oracle.julia(256), env, sun_dir, sun_power, rng_mode=oracle.RNG_SAMPLERH)
    P = oracle.default_param(800, 600)
    acc = None
    for f in range(8):
        acc, _ = sc.render_frame(P, f, acc)
    heat = acc[..., 3].astype(np.float64) / 8
    assert abs((heat == 0).mean() - ref["zero_scatter_pixel_fraction"]) < 0.01
    assert abs(np.percentile(heat, 90) - ref["p90_scatters"]) < 0.5
    assert abs(np.percentile(heat, 99) - ref["p99_scatters"]) < 1.5
    assert abs(heat.max() / ref["max_scatters"] - 1) < 0.15       # a maximum over 480 000 pixels: noisy, still within 15 %
    # the survey's "mean 1.58 scatters/sample" contradicts its own percentiles: 10 % of the pixels are >= p90, 1 % >= p99
    bound = 0.10 * ref["p90_scatters"] + 0.01 * (ref["p99_scatters"] - ref["p90_scatters"])
    assert bound > 1.58 and heat.mean() > bound


def test_white_furnace(oracle):
    """albedo 1, constant environment, no sun: every sample returns the environment constant."""
    g = oracle.julia(32)
    env = np.zeros((8, 16, 4), np.float32)
    env[..., :3] = 0.5
    for est in (oracle.EST_DECOMP, oracle.EST_GLOBAL, oracle.EST_BOUNDED):
        sc = oracle.OracleScene(g, env, scenes.DEFAULT_SUN_DIR, (0.0, 0.0, 0.0), estimator=est)
        P = oracle.default_param(48, 36)
        acc = None
        for f in range(4):
            acc, _ = sc.render_frame(P, f, acc)
        rgb = acc[..., :3] / 4
        # the only loss is the max_depth=800 cut (kernel.cu:34) and float drift of the unit weights
        assert np.abs(rgb - 0.5).max() < 2e-3


def test_majorant_invariance(oracle):
    """global-majorant and decomposition tracking estimate the same integral (frames <= 10: no opacity table)."""
    g = oracle.julia(32)
    env = scenes.synthetic_env()
    imgs = []
    for est in (oracle.EST_DECOMP, oracle.EST_GLOBAL, oracle.EST_BOUNDED):
        sc = oracle.OracleScene(g, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=est,
                                rng_mode=oracle.RNG_PHILOX, seed=(5, est))
        P = oracle.default_param(32, 24, density=40.0)
        acc = None
        for f in range(10):
            acc, _ = sc.render_frame(P, f, acc)
        imgs.append(acc[..., :3].mean(axis=(0, 1)) / 10)
    assert np.allclose(imgs[0], imgs[1], rtol=0.08)
    assert np.allclose(imgs[0], imgs[2], rtol=0.08)


def test_env_tables_and_mis_estimate(oracle):
    """!PASSIVE_ENVMAP: the luminance CDFs (kernel.cu:1144-1210) and one-sample MIS (:2220-2297) estimate the same image."""
    g = oracle.julia(32)
    env = scenes.synthetic_env()
    sc = oracle.OracleScene(g, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, env_mis=True,
                            rng_mode=oracle.RNG_PHILOX, seed=(3, 1))
    h, w = env.shape[:2]
    # tables against a float64 restatement
    phi = np.pi * (np.arange(h) + 0.5) / h
    lum = (env[..., 0].astype(np.float64) * 0.2126 + env[..., 1] * 0.7152 + env[..., 2] * 0.0722) * np.sin(phi)[:, None]
    assert np.allclose(sc.cdf_x, np.cumsum(lum, 1) / lum.sum(1, keepdims=True), atol=2e-5)
    assert np.allclose(sc.cdf_y, np.cumsum(lum.sum(1)) / lum.sum(), atol=2e-5)
    assert np.all(sc.cdf_x[:, -1] == 1.0) and sc.cdf_y[-1] == 1.0
    assert np.isclose(sc.pdfnorm_alt, w * h / (2 * np.pi * np.pi) / lum.sum(), rtol=1e-5)
    # the two builds agree in the mean
    P = oracle.default_param(32, 24, density=60.0)
    imgs = []
    for mis in (True, False):
        s2 = sc if mis else oracle.OracleScene(g, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER,
                                               rng_mode=oracle.RNG_PHILOX, seed=(3, 1))
        acc = None
        for f in range(10):
            acc, c = s2.render_frame(P, f, acc)
        imgs.append(acc[..., :3].mean(axis=(0, 1)) / 10)
        assert (c.env_lookups > c.samples) == mis
    assert np.allclose(imgs[0], imgs[1], rtol=0.05)


def test_scalar_tracking_modes_estimate_the_same_image(oracle):
    """SPECTRAL_TRACKING 0 and MULTI_CHANNEL 1 (compiled out in the reference) against the shipped spectral build: for an
    achromatic medium all three are estimators of the same image."""
    g = oracle.julia(32)
    env = scenes.synthetic_env()
    for est in (oracle.EST_DECOMP, oracle.EST_GLOBAL, oracle.EST_BOUNDED):
        imgs = []
        for tm in (0, 1, 2):
            sc = oracle.OracleScene(g, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=est,
                                    rng_mode=oracle.RNG_PHILOX, seed=(4, tm), track_mode=tm)
            sc.precompute_opacity()   # frames beyond 10 read it in the live kernel (quirk Q5)
            P = oracle.default_param(64, 48, density=60.0)
            acc = None
            for f in range(40):
                acc, _ = sc.render_frame(P, f, acc)
            imgs.append(acc[..., :3].mean(axis=(0, 1)) / 40)
        assert np.allclose(imgs[0], imgs[1], rtol=0.04) and np.allclose(imgs[0], imgs[2], rtol=0.06), (est, imgs)


def test_math_accuracy(oracle):
    rng = np.random.default_rng(1)
    u = rng.random(200000).astype(np.float32)
    u = u[u > 0]

    def ulps(a, exact):
        sp = np.abs(np.spacing(exact.astype(np.float32))).astype(np.float64)
        return np.max(np.abs(a.astype(np.float64) - exact) / sp)

    assert ulps(oracle.math_array(0, u), np.log(u.astype(np.float64))) < 1.0
    xe = (-80 * u).astype(np.float32)
    assert ulps(oracle.math_array(1, xe), np.exp(xe.astype(np.float64))) < 1.5
    xa = (u * 6.2831855).astype(np.float32)
    assert np.abs(oracle.math_array(2, xa) - np.sin(xa.astype(np.float64))).max() < 2e-7
    assert np.abs(oracle.math_array(3, xa) - np.cos(xa.astype(np.float64))).max() < 2e-7
    xc = (u * 2 - 1).astype(np.float32)
    assert ulps(oracle.math_array(4, xc), np.arccos(xc.astype(np.float64))) < 2.0
    xt = np.tan((u - 0.5) * 3.1).astype(np.float32)
    assert ulps(oracle.math_array(5, xt), np.arctan(xt.astype(np.float64))) < 4.0
    assert oracle.math_array(0, np.zeros(1, np.float32))[0] == -np.inf
    assert oracle.math_array(5, np.array([np.inf, -np.inf, np.nan], np.float32)).tolist() == \
        [np.float32(np.pi / 2), -np.float32(np.pi / 2), 0.0]


def test_scale_gamma_and_mat(oracle):
    import ctypes as C
    src = np.random.default_rng(0).random((5, 4), dtype=np.float32)
    dst = np.empty_like(src)
    oracle.lib().vpo_scale(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), 5, 0.25)
    assert np.array_equal(dst, src * np.float32(0.25))
    oracle.lib().vpo_gamma_correct(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), 5, 0.5, 2.2)
    assert np.allclose(dst[:, :3], (src[:, :3] * 0.5) ** (1 / 2.2), rtol=2e-6) and np.all(dst[:, 3] == 1)
    P = oracle.mat(oracle.default_param(4, 4), *scenes.PRESET1)  # host.cpp:1296
    st = np.array([2.29 + 0.0030, 2.39 + 0.0034, 1.97 + 0.046])
    assert np.allclose(list(P.sigma_t), st / st.max(), rtol=1e-6)
    assert np.allclose(list(P.albedo), np.array([2.29, 2.39, 1.97]) / st, rtol=1e-6)


def test_oracle_shadow_rays_draw_from_their_own_substream(oracle):
    """Counter-based streams (oracle rng_enter_shadow): a shadow ray draws pairs 0x80000000 + (id << 20) + 0, 1, ... and the
    path's own stream goes on where it stood.  With the sun's power at zero the image is the environment seen by the escaping
    paths: identical for two sun directions (shadow rays of different lengths) with Philox, different with the reference's
    sequential sampler.h stream."""
    import scenes
    O = oracle
    grid = O.julia(32)
    env = scenes.synthetic_env()
    P = O.default_param(48, 36)
    for est in (O.EST_GLOBAL, O.EST_DECOMP, O.EST_BOUNDED):
        same = {}
        for rng_mode in (O.RNG_PHILOX7, O.RNG_SAMPLERH):
            imgs = []
            for sun in ((-0.0, 0.951057, -0.309017), (0.0, 0.6, 0.8)):
                sc = O.OracleScene(grid, env, sun, (0.0, 0.0, 0.0), estimator=est, rng_mode=rng_mode, seed=(5, 6))
                acc = None
                for f in range(3):
                    acc, cnt = sc.render_frame(P, f, acc)
                imgs.append(acc)
            assert imgs[0][..., 3].max() > 0
            same[rng_mode] = np.array_equal(imgs[0], imgs[1])
        assert same[O.RNG_PHILOX7] and not same[O.RNG_SAMPLERH]


def _ref_silhouette():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_julia_silhouette.npz"))
    H, W = (int(v) for v in z["shape"])
    mask = np.unpackbits(z["mask_bits"])[:H * W].reshape(H, W).astype(bool)
    return mask, z["camera"], z["pose"], z["centre"]


def _iou(a, b):
    return (a & b).sum() / max((a | b).sum(), 1)


def test_julia_silhouette_matches_the_references_own_screenshot(oracle):
    """A pin the REFERENCE holds: its repository ships one render of the procedural Julia-set scene (2.jpg, 960x512 = the
    reference's default window).  tests/golden/ref_julia_silhouette.npz is that image's silhouette (pixels differing from the
    uniform background) and the orbit pose fitted to it with the camera distance HELD at the reference's default 4.0
    (tests/golden/fit_julia_pose.py).  The oracle's silhouette -- pixels where a sample scattered; the medium is opaque -- must
    coincide with it: intersection over union >= 0.95 at half resolution (0.975 when fitted; the rest is JPEG edge blur and
    sub-voxel wisps).  That pins FractalJuliaSet + its voxelisation (A12), the volume box (A14), the camera matrix (H4), the field
    of view and pixel-to-ray map (kernel.cu:1977-1987) and intersectBox (A5) against the reference's own output -- not its
    radiometry: the screenshot's environment is a uniform grey that the current source no longer has.  The scale is pinned, not
    absorbed by the fit: 5 % off in distance (= field of view, box size or Julia radius) costs more than 0.07 of IoU."""
    mask, cam, pose, centre = _ref_silhouette()
    assert mask.shape == (512, 960) and abs(mask.mean() - 0.18) < 0.01
    assert pose[3] == 4.0 and abs(abs(pose[2]) - np.pi) < 0.01      # distance held at the default; the roll came out as the row order
    O = oracle
    grid = O.julia(256)
    m2 = mask.reshape(256, 2, 480, 2).mean(axis=(1, 3)) >= 0.5          # half resolution

    def silhouette(camera):
        osc = O.OracleScene(grid, scenes.synthetic_env(), O.DEFAULT_SUN_DIR, O.DEFAULT_SUN_POWER, estimator=O.EST_GLOBAL,
                            rng_mode=O.RNG_PHILOX7, seed=(1, 2), inv_view=camera, radius=1)
        P = O.default_param(480, 256)
        acc = None
        for f in range(2):
            acc, _ = osc.render_frame(P, f, acc)
        return acc[..., 3] > 0

    good = _iou(m2, silhouette(cam))
    assert good >= 0.95, good
    # the same orbit 5 % farther away: the camera position moves along its own z axis (third column of the 3x4 matrix)
    far = np.array(cam, np.float32).reshape(3, 4).copy()
    far[:, 3] += 0.2 * far[:, 2]
    assert _iou(m2, silhouette(far.ravel())) < good - 0.07


def _interior_stats(img, mask, ref_lum, ref_cnt, B):
    """block-mean luminance of `img` (linear, per sample) inside `mask`, clamped at 1 like the display, against the screenshot's:
    Pearson correlation at the exposure scale that minimises the residual, that scale, the rms residual"""
    H, W = mask.shape
    by, bx = H // B, W // B
    cnt = mask.reshape(by, B, bx, B).sum((1, 3))
    use = (ref_cnt >= ref_cnt.max() // 2) & (cnt > 0)
    w = np.array([0.2126, 0.7152, 0.0722])

    def block_lum(scale):
        lum = np.minimum(img * scale, 1.0) @ w
        return ((lum * mask).reshape(by, B, bx, B).sum((1, 3)) / np.maximum(cnt, 1))[use]
    r = ref_lum[use].astype(np.float64)
    scales = np.geomspace(0.25, 4.0, 241)
    err = [float(((block_lum(sc) - r) ** 2).mean()) for sc in scales]
    k = int(np.argmin(err))
    return float(np.corrcoef(block_lum(scales[k]), r)[0, 1]), float(scales[k]), float(np.sqrt(err[k])), int(use.sum())


def test_julia_interior_matches_the_references_own_screenshot(oracle):
    """The first RADIOMETRIC pin the reference holds: what its Julia screenshot 2.jpg shows INSIDE the silhouette.
    tests/golden/ref_julia_interior.npz = per 16x16 block the mean of the screenshot's pixels, linearised through the display's
    gamma 2.2 (made by tests/golden/make_julia_interior.py), and the sun position fitted to them (tests/golden/fit_julia_sun.py: the
    viewer's sun, like its camera, is moved with the mouse and not recorded -- TWO parameters against 345 blocks; the environment is
    the uniform 0.03 grey of the screenshot's background, host.cpp:1374-1385; the exposure is NOT fitted).  With the reference's default
    medium (density 800, g 0.877, albedo 1: host.cpp:1286-1292) the oracle's render of that scene reproduces the screenshot's shading:
    Pearson correlation of the block luminances >= 0.98 (0.996 on the GPU at full resolution and 128 spp) at an exposure scale of
    0.87 -- the absolute radiance, with the reference's own solar radiance and no free scale, is 13 % above the screenshot's.  That
    is sun power x phase function x albedo x transmittance x multiple scattering x display transform against a reference OUTPUT; the
    GPU test of the same name shows what it excludes (g = 0: rms residual 4.5x; density 80: 8x; albedo 0.8: 8x).
    Half resolution, frames 0-9 (the live kernel before its switch to the optical-depth table): a few seconds."""
    import os
    O = oracle
    mask, cam, pose, centre = _ref_silhouette()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_julia_interior.npz"))
    B = int(z["block"])
    grid = O.julia(256)
    env = np.full((8, 16, 4), 0.03, np.float32)
    env[..., 3] = 1.0
    osc = O.OracleScene(grid, env, tuple(float(v) for v in z["sun_dir"]), tuple(float(v) for v in z["sun_power"]), estimator=O.EST_DECOMP,
                        rng_mode=O.RNG_PHILOX7, seed=(1, 2), inv_view=cam)
    P = O.default_param(480, 256)
    acc, frames = None, 10
    for f in range(frames):
        acc, _ = osc.render_frame(P, f, acc)
    m2 = mask.reshape(256, 2, 480, 2).mean(axis=(1, 3)) >= 0.5
    pear, scale, rms, nb = _interior_stats(acc[..., :3].astype(np.float64) / frames, m2, z["luminance"], z["count"], B // 2)
    assert nb > 300
    assert pear >= 0.98, (pear, scale, rms)
    assert 0.78 <= scale <= 0.97, scale
    assert rms <= 0.035, rms


def test_julia_interior_prefers_the_reference_as_read(oracle):
    """WHICH reading of the reference does its own screenshot support?  The radiometric pin again, with the oracle's what-if switches
    (vpo_debug_set_what_if: variants of the restatement, used by this test only): the restatement as it is against the same integrator
    WITHOUT the "Hyperion" reduction of quirk Q9 (kernel.cu:2039-2045, :2168-2172: from the sixth collision on the phase function is
    made more isotropic and the density is reduced -- a deliberate bias of the reference).  Same pose, same fitted sun, same
    samples (two keys x frames 0-10 at half resolution).  The screenshot sides with the reference as read: without the reduction
    the block luminances miss it by more than twice the residual (0.045 against 0.020 at six keys) at an exposure scale of 1.3
    instead of 0.84 -- deep paths stay deep and the body comes out darker.  So quirk Q9, the one quirk with a first-order effect on
    the image, is pinned by an output of the reference; the Henyey-Greenstein clamp of quirk Q1 changes the residual by a tenth
    (0.022 unclamped against 0.020), inside what these sample counts resolve, and is asserted only not to fit better by much."""
    import os
    O = oracle
    mask, cam, pose, centre = _ref_silhouette()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_julia_interior.npz"))
    grid = O.julia(256)
    env = np.full((8, 16, 4), 0.03, np.float32)
    env[..., 3] = 1.0
    m2 = mask.reshape(256, 2, 480, 2).mean(axis=(1, 3)) >= 0.5
    sun_dir, sun_power = tuple(float(v) for v in z["sun_dir"]), tuple(float(v) for v in z["sun_power"])

    def fit(what_if, keys):
        O.lib().vpo_debug_set_what_if(what_if)
        try:
            tot, n = None, 0
            for key in range(keys):
                osc = O.OracleScene(grid, env, sun_dir, sun_power, estimator=O.EST_DECOMP, rng_mode=O.RNG_PHILOX7, seed=(1 + key, 2), inv_view=cam)
                P = O.default_param(480, 256)
                acc = None
                for f in range(11):
                    acc, _ = osc.render_frame(P, f, acc)
                tot, n = (acc if tot is None else tot + acc), n + 11
        finally:
            O.lib().vpo_debug_set_what_if(0)
        return _interior_stats(tot[..., :3].astype(np.float64) / n, m2, z["luminance"], z["count"], int(z["block"]) // 2)

    pear, scale, rms, _ = fit(0, 2)
    pear9, scale9, rms9, _ = fit(1, 2)
    assert pear >= 0.985 and rms <= 0.03, (pear, scale, rms)
    assert rms9 >= 1.6 * rms and pear9 <= pear - 0.015 and scale9 >= 1.15, (pear9, scale9, rms9, rms)
    pear1, scale1, rms1, _ = fit(2, 2)
    assert rms1 >= 0.85 * rms, (rms1, rms)


def test_julia_interior_cannot_separate_q4_q7_q8(oracle):
    """what-if switches for the three quirks no reference-held output separates yet -- (4) the sun's shadow ray
    on the GLOBAL majorant instead of the local segment's (what Q4 is not, kernel.cu:2172-2178), (8) a control component that carries
    the medium's albedo instead of being carved whole from the scattering coefficient (what Q7 is not, :2108-2110), (16) collision
    weights with the majorant the flight was sampled with instead of the total one (what Q8 is not, :2126-2133) -- tried against the
    one radiometric reference output there is, the interior of the Julia screenshot.  REPORT, not a fit: on that scene all three
    variants render the SAME BITS as the restatement (np.array_equal, every pixel, frames 0-2 at half resolution), so the screenshot
    supports them exactly as much as it supports the reference-as-read and cannot tell them apart.  Why, quirk by quirk:
      * Q4: the grid is binary {0, 255}.  A collision needs a non-zero texel next to it, the segment that found it is at most 0.05
        long and its bound window is ceil(0.05 / cell) voxels wide, so the window of every segment that scatters holds a 255:
        d_max = 1 = the global maximum at every scatter point -- local and global majorant are one number.
      * Q7 / Q8 act only in segments with d_min > 0, i.e. (binary grid) d_min = d_max = 1; the medium is achromatic (min sigma_t =
        max sigma_t), so the residual majorant is max(sigma_t' - sigma_c, 1e-20) = 1e-20, the free flight is 1e20 long and every such
        segment ends AT its control collision: the spectral collision weights are never evaluated there.  (The counter shows such
        segments exist -- 0.5 % of the collisions -- and that nothing in them depends on the three switches.)
    What remains unpinned is therefore stated, not hidden: Q4, Q7 and Q8 matter for soft (non-binary) or chromatic media -- BASELINE
    configs[3] and [4] -- for which the reference holds no output at all (test_whatif_switches_are_live_on_a_soft_chromatic_volume
    shows the switches are not dead code).  DESIGN.md section 3 says the same."""
    import os
    O = oracle
    mask, cam, pose, centre = _ref_silhouette()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_julia_interior.npz"))
    grid = O.julia(256)
    env = np.full((8, 16, 4), 0.03, np.float32)
    env[..., 3] = 1.0
    sun_dir, sun_power = tuple(float(v) for v in z["sun_dir"]), tuple(float(v) for v in z["sun_power"])

    def render(what_if):
        O.lib().vpo_debug_set_what_if(what_if)
        try:
            osc = O.OracleScene(grid, env, sun_dir, sun_power, estimator=O.EST_DECOMP, rng_mode=O.RNG_PHILOX7, seed=(1, 2), inv_view=cam)
            P = O.default_param(480, 256)
            acc, ctrl, sca = None, 0, 0
            for f in range(3):
                acc, c = osc.render_frame(P, f, acc)
                ctrl += c.control_segments
                sca += c.scatters
        finally:
            O.lib().vpo_debug_set_what_if(0)
        return acc, ctrl, sca

    base, ctrl, sca = render(0)
    assert 0 < ctrl < 0.02 * sca, (ctrl, sca)            # segments with a control component exist, and are rare
    for w in (4, 8, 16, 4 | 8 | 16):
        img, c2, s2 = render(w)
        assert np.array_equal(img, base) and (c2, s2) == (ctrl, sca), w
    other, _, s9 = render(1)                              # (the switch that IS separated by the screenshot: quirk Q9)
    assert not np.array_equal(other, base) and s9 > 2 * sca


def test_whatif_switches_are_live_on_a_soft_chromatic_volume(oracle):
    """The what-if switches of the test above are not dead code: on a soft (non-binary) grid with a chromatic, absorbing medium each
    of them changes the image -- there the local majorant at a scatter point is below the volume maximum (Q4), control components
    exist beside a positive residual majorant (Q7, Q8: sigma_r > 0 because min sigma_t < max sigma_t and d_min < d_max) and the albedo
    is below one (Q7).  That is the regime of BASELINE configs[3]/[4] (soft cloud data, chromatic preset), for which the reference
    holds no output: the three quirks are restated from the source (kernel.cu:2048-2134, :2168-2178) and pinned by nothing else."""
    O = oracle
    rs = np.random.default_rng(5)
    g = rs.random((24, 24, 24), dtype=np.float32)
    for _ in range(2):                                    # smooth: neighbouring voxels correlated, minima of windows positive inside
        g = (g + np.roll(g, 1, 0) + np.roll(g, 1, 1) + np.roll(g, 1, 2)) / 4.0
    grid = np.ascontiguousarray((40 + 215 * (g - g.min()) / (g.max() - g.min())).astype(np.uint8))
    W, H = 48, 36

    def render(what_if):
        O.lib().vpo_debug_set_what_if(what_if)
        try:
            osc = O.OracleScene(grid, scenes.synthetic_env(), O.DEFAULT_SUN_DIR, O.DEFAULT_SUN_POWER, estimator=O.EST_DECOMP,
                                rng_mode=O.RNG_PHILOX7, seed=(3, 4))
            P = O.default_param(W, H)
            O.mat(P, 2.29, 2.39, 1.97, 0.30, 0.34, 0.46)  # chromatic, albedo ~0.85 (preset #1's scattering with more absorption)
            P.density = 40.0
            acc, ctrl = None, 0
            for f in range(4):
                acc, c = osc.render_frame(P, f, acc)
                ctrl += c.control_segments
        finally:
            O.lib().vpo_debug_set_what_if(0)
        return acc, ctrl

    base, ctrl = render(0)
    assert ctrl > 0
    for w in (4, 8, 16):
        img, _ = render(w)
        assert not np.array_equal(img[..., :3], base[..., :3]), w
