"""Oracle-INDEPENDENT pins of the HIP path.

Nothing here calls the CPU oracle.  The reference holds no test vectors, so besides the bit-exact
comparison with the independently written oracle (tests/test_parity_gpu.py) the HIP path is checked against what the
physics and the reference's formulas say in float64:

  * component known answers through the vp_test_* hooks: HGPhaseFunction::sample / ::evaluate incl. quirk Q1
    (kernel.cu:575-619), intersectBox (kernel.cu:654-680), tex3D's documented 8-bit-weight trilinear rule
    (kernel.cu:173-178, :682-695) restated in numpy, eval_envmap's direction -> texel mapping (kernel.cu:882-973);
  * white furnace: albedo 1, constant environment, no sun => every pixel is the environment constant, for the three
    estimators, passive and MIS environment lighting, spectral / scalar / multi-channel tracking;
  * homogeneous slab: E[unscattered fraction] = exp(-sigma_t * rho * chord) for the primary free flight
    (kernel.cu:2082-2142 / :1416-1452), and the single-scatter sun radiance -- first-collision density x phase function x
    exp(-sigma_t * rho * distance to the box along the sun) -- for the shadow ray (Tr_spectral, kernel.cu:754-808).

Tolerances are stated at each assert (DESIGN.md section 2 lists them).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PI = np.pi
CAM = np.array([0.0, 0.207912, 0.978148, 3.922986, 0.0, 0.978148, -0.207912, -0.782739, -1.0, 0.0, 0.0, 0.03])  # H4


def _camera_rays(W, H):
    """kernel.cu:1977-1987 in float64: origin and unit direction per pixel, arrays [H, W, 3]"""
    M = CAM.reshape(3, 4)
    x = np.arange(W)[None, :].repeat(H, 0).astype(np.float64)
    y = np.arange(H)[:, None].repeat(W, 1).astype(np.float64)
    u = (x * 2 - W) / W
    v = (y * 2 - H) / W
    z = -1.0 / np.tan(54.43 * 0.00872664626)
    dv = np.stack([u, v, np.full_like(u, z)], -1)
    d = dv @ M[:, :3].T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(M[:, 3], d.shape)
    return o, d


def _slab(o, d, bmin=-1.0, bmax=1.0):
    """intersectBox kernel.cu:654-680 in float64"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        tb = inv * (bmin - o)
        tt = inv * (bmax - o)
    tmin = np.minimum(tt, tb).max(-1)
    tmax = np.maximum(tt, tb).min(-1)
    return (tmax > tmin) & (tmax >= 1e-3), tmin, tmax


# ------------------------------------------------------------------------------------------ component known answers
def test_hg_sample_and_evaluate_match_closed_forms(vp):
    rng = np.random.default_rng(21)
    n = 20000
    g = rng.choice(np.array([0.877, 0.5, -0.6, 0.0, 1e-7, 0.99, -0.3, 0.2], np.float32), n)
    r0 = rng.random(n, dtype=np.float32)
    r1 = rng.random(n, dtype=np.float32)
    r0[:8] = [0.0, 0.99999994, 0.5, 0.25, 0.0, 0.99999994, 1e-6, 0.75]
    nrm = rng.normal(size=(n, 3))
    nrm[:200, 0] = rng.uniform(-0.12, 0.12, 200)  # both branches of Frame's axis choice (|n.x| vs 0.1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm.astype(np.float32)
    cq = rng.uniform(-1, 1, n).astype(np.float32)
    d, ev = vp.test_hg(g, r0, r1, nrm, cq)

    G, R0, R1, N = g.astype(np.float64), r0.astype(np.float64), r1.astype(np.float64), nrm.astype(np.float64)
    s = 2 * R0 - 1
    with np.errstate(divide="ignore", invalid="ignore"):
        f = (1 - G * G) / (1 + G * s)
        ct = np.where(np.abs(G) > 1e-6, np.clip(0.5 / G * (1 + G * G - f * f), 0.0, 1.0), s)  # quirk Q1: [0,1], not [-1,1]
    st = np.sqrt(np.maximum(1 - ct * ct, 0))
    phi = 2 * PI * R1
    a = np.where((np.abs(N[:, 0]) >= np.float32(0.1))[:, None], [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
    t = np.cross(a, N)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(N, t)
    w = t * (np.cos(phi) * st)[:, None] + b * (np.sin(phi) * st)[:, None] + N * ct[:, None]
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    # tolerance: 2e-5 absolute per component (binary32 with <= 3-ulp sin/cos; the frame amplifies by < 4), plus the
    # conditioning of the formula itself next to the pole: 1 + g^2 - f^2 cancels, so binary32 leaves cos(theta) with an
    # absolute error of a few 1e-7 (times 0.5/|g|), and sin(theta) = sqrt(1 - cos^2) turns an error e of cos(theta) ~ 1
    # into sqrt(sin^2 + 2e) - sin
    near_axis = np.abs(np.abs(N[:, 0]) - 0.1) < 1e-6
    e_cos = 4e-7 * np.maximum(1.0, 0.5 / np.maximum(np.abs(G), 1e-3))
    tol = 2e-5 + np.sqrt(st * st + 2 * e_cos) - st
    assert np.all((np.abs(d - w) < tol[:, None])[~near_axis])
    assert np.median(tol) < 3e-5
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-6
    # Q1: with |g| > 1e-6 the sampled direction never points into the back hemisphere of the incoming direction
    cos_out = (d.astype(np.float64) * N).sum(1)
    assert cos_out[np.abs(G) > 1e-6].min() > -2e-6
    assert (np.abs(cos_out[(np.abs(G) > 1e-6)]) < 2e-6).sum() > 50      # ... and the clamp is really hit
    assert cos_out[np.abs(G) <= 1e-6].min() < -0.5                       # the isotropic branch does sample backwards
    # evaluate: (1 - g^2) / (4 pi (1 + g^2 - 2 g cos)^1.5), 1e-5 relative
    # 1e-5 relative, plus the cancellation in x = 1 + g^2 - 2 g cos for a forward peak (g = 0.99, cos ~ 1: x ~ 1e-4):
    # three binary32 roundings of magnitude ~2 (4e-7 absolute) enter x^1.5 as 1.5 * 4e-7 / x
    x = 1 + G * G - 2 * G * cq.astype(np.float64)
    want = (1 - G * G) / (4 * PI * x ** 1.5)
    assert np.all(np.abs(ev / want - 1) < 1e-5 + 6e-7 / x)
    assert np.median(6e-7 / x) < 2e-6


def test_intersect_box_matches_float64_slab_test(vp):
    rng = np.random.default_rng(5)
    grid = np.zeros((8, 12, 16), np.uint8)             # box = +-(1, 12/16, 8/16), kernel.cu:373-378
    vp.init_volume(grid)
    bmin, bmax = np.array([-1, -0.75, -0.5]), np.array([1, 0.75, 0.5])
    n = 1024
    o = rng.uniform(-3, 3, (n, 3))
    o[:200] = rng.uniform(-0.4, 0.4, (200, 3))         # origins inside the box
    d = rng.normal(size=(n, 3))
    d[200:300, 0] = 0.0                                # axis-parallel: invR = +-inf (quirk Q13)
    d[300:360, 1] = 0.0
    d[360:400, [0, 2]] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    hit, tn, tf = vp.test_intersect_box(o32, d32)
    O, D = o32.astype(np.float64), d32.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / D
        tb, tt = inv * (bmin - O), inv * (bmax - O)
    tmin, tmax = np.minimum(tt, tb).max(1), np.maximum(tt, tb).min(1)
    want = (tmax > tmin) & (tmax >= 1e-3)
    decided = (np.abs(tmax - tmin) > 1e-4) & (np.abs(tmax - 1e-3) > 1e-5)   # away from the two decision boundaries
    assert np.array_equal(hit[decided], want[decided]) and decided.mean() > 0.97
    assert hit[:200].all()                                                    # inside-box rays always hit
    fin = np.isfinite(tmin) & np.isfinite(tmax)
    # tolerance 2e-6 relative + 2e-6 absolute (three binary32 operations per slab)
    assert np.allclose(tn[fin], tmin[fin], rtol=2e-6, atol=2e-6) and np.allclose(tf[fin], tmax[fin], rtol=2e-6, atol=2e-6)
    assert np.array_equal(np.isinf(tn), np.isinf(tmin)) and np.array_equal(np.isinf(tf), np.isinf(tmax))


def _tex3d_u8_numpy(grid, pos, bmin, bmax, linear=True):
    """CUDA programming guide, "Texture Fetching": normalised coordinates, clamp addressing, texel-centre convention
    xB = x*N - 0.5, linear filtering with 8-bit fractional weights; uchar texels read as t/255.  Stated with exact integers:
    value = sum_t t * wx * wy * wz / (255 * 2^24), weights in 0..256."""
    f32, f64 = np.float32, np.float64
    nz, ny, nx = grid.shape
    dims = (nx, ny, nz)
    linv = [f32(1.0) / f32(f32(bmax[a]) - f32(bmin[a])) for a in range(3)]
    idx, wgt = [], []
    for a in range(3):
        p = ((pos[:, a].astype(f32) - f32(bmin[a])) * linv[a]).astype(f32)
        if linear:
            xb = (p.astype(f64) * dims[a] - 0.5).astype(f32)     # the unit's own scaling: one rounding
            xb = np.maximum(xb, f32(0))                          # below the first texel centre: both taps clamp to texel 0
            fl = np.floor(xb)
            fr = (xb - fl).astype(f32)
            w = np.floor((fr.astype(f64) * 256 + 0.5).astype(f32)).astype(np.int64)   # round to nearest, ties up
            i0 = np.minimum(fl.astype(np.int64), dims[a] - 1)
            i1 = np.minimum(i0 + 1, dims[a] - 1)
        else:
            i0 = np.clip(np.floor((p * f32(dims[a])).astype(f32)).astype(np.int64), 0, dims[a] - 1)
            i1, w = i0, np.zeros_like(i0)
        idx.append((i0, i1))
        wgt.append(w)
    g = grid.astype(np.int64)
    v = np.zeros(pos.shape[0], np.int64)
    for cz in (0, 1):
        for cy in (0, 1):
            for cx in (0, 1):
                t = g[idx[2][cz], idx[1][cy], idx[0][cx]]
                wx = wgt[0] if cx else 256 - wgt[0]
                wy = wgt[1] if cy else 256 - wgt[1]
                wz = wgt[2] if cz else 256 - wgt[2]
                v += t * wx * wy * wz
    # exact integer v <= 255 * 2^24; the kernel forms float(v) / 2^24 with one rounding, then one multiply by fl(2^24 / (255 * 2^24))
    x = (v.astype(f64) / 2.0 ** 24).astype(f32)
    return (x * f32(f32(2.3374372e-10) * f32(16777216.0))).astype(f32), v


@pytest.mark.parametrize("linear", [True, False])
def test_sample_density_is_the_documented_tex3d_rule(vp, linear):
    rng = np.random.default_rng(17)
    grid = rng.integers(0, 256, (11, 13, 17), dtype=np.uint8)
    grid[rng.random(grid.shape) < 0.3] = 0
    bmin, bmax = (-1.0, -0.6, -0.9), (1.0, 0.7, 0.4)
    vp.init_volume(grid, box=(bmin, bmax), linear=linear)
    n = 50000
    pos = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)                 # inside and outside the box (clamp addressing)
    pos[:2000] = np.stack([rng.uniform(bmin[a], bmax[a], 2000) for a in range(3)], 1).astype(np.float32)
    got = vp.test_sample_density(pos)
    want, v = _tex3d_u8_numpy(grid, pos, bmin, bmax, linear)
    assert np.array_equal(got, want)                                        # tolerance 0: the rule is integer arithmetic
    assert got.max() <= 1.0 and got.min() >= 0.0
    full = np.full((4, 4, 4), 255, np.uint8)
    vp.init_volume(full, linear=linear)
    assert np.all(vp.test_sample_density(rng.uniform(-1, 1, (1000, 3)).astype(np.float32)) == np.float32(1.0))


def test_eval_envmap_direction_to_texel(vp):
    """dir_to_uv kernel.cu:882-895 + point-sampled tex2D :971: theta = atan(z/x) + pi/2 (+ pi if x < 0), phi = acos(y)"""
    rng = np.random.default_rng(3)
    h, w = 32, 64
    env = np.zeros((h, w, 4), np.float32)
    env[..., 0] = np.arange(w)[None, :]
    env[..., 1] = np.arange(h)[:, None]
    env[..., 2] = 7.0
    vp.init_envmap(env)
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d32 = d.astype(np.float32)
    got = vp.test_eval_envmap(d32)
    D = d32.astype(np.float64)
    theta = np.arctan(D[:, 2] / D[:, 0]) + PI / 2 + np.where(D[:, 0] < 0, PI, 0.0)
    u, v = theta / (2 * PI), np.arccos(np.clip(D[:, 1], -1, 1)) / PI
    fu, fv = u * w, v * h
    decided = (np.abs(fu - np.round(fu)) > 1e-3) & (np.abs(fv - np.round(fv)) > 1e-3)   # away from texel edges
    iu, iv = np.clip(np.floor(fu), 0, w - 1), np.clip(np.floor(fv), 0, h - 1)
    assert decided.mean() > 0.9
    assert np.array_equal(got[decided, 0], iu[decided].astype(np.float32))
    assert np.array_equal(got[decided, 1], iv[decided].astype(np.float32))
    assert np.all(got[:, 2] == 7.0)


# ------------------------------------------------------------------------------------------ white furnace
def _furnace_scene(vp, est, env_mode, track, sigma_t, key):
    import scenes
    W, H = 48, 36
    c = np.array([0.7, 0.45, 0.9], np.float32)
    env = np.zeros((16, 32, 4), np.float32)
    env[..., :3] = c
    env[..., 3] = 1
    vp.set_tracking(track)
    vp.set_envmap_sampling(env_mode)
    vp.init_volume(scenes.blob_volume_u8(24), brick=1, linear=True)
    vp.init_envmap(env)
    vp.set_sun((1.0, 0.0, 0.0), (0.0, 0.0, 0.0))       # behind the camera (rays have d.x < 0) and black
    vp.set_camera()
    vp.set_estimator(est)
    vp.set_rng(vp.RNG_PHILOX, key)
    vp.set_shard(0, 1)
    P = vp.make_param(W, H, density=6.0, g=0.5, albedo=(1, 1, 1), sigma_t=sigma_t)
    return P, c, W, H


@pytest.mark.parametrize("est", [0, 1, 2])
@pytest.mark.parametrize("env_mode,track,sigma_t", [(0, 0, (1, 1, 1)), (0, 0, (1.0, 0.7, 0.4)), (1, 0, (1, 1, 1)), (1, 0, (1.0, 0.7, 0.4)),
                                                    (0, 1, (1, 1, 1)), (0, 2, (1.0, 0.7, 0.4))])
def test_white_furnace(vp, est, env_mode, track, sigma_t):
    """albedo 1 + constant environment + no sun: the radiance is the environment constant at every pixel, whatever the
    density, phase function, majorant scheme or light-sampling strategy.  Exact (3e-4: binary32 drift of a weight that is 1 in exact arithmetic) where the estimator's weights are
    identically 1 (achromatic spectral tracking and scalar tracking, passive environment); otherwise within 5 standard
    errors per 8x8-pixel block and 0.5 % on the image mean."""
    frames = 96
    try:
        P, c, W, H = _furnace_scene(vp, est, env_mode, track, sigma_t, key=(31, est * 5 + env_mode * 3 + track))
        buf = vp.DeviceBuffer(W, H)
        if est == 1:
            vp.precompute_opacity((1.0, 0.0, 0.0))
        per = np.empty((frames, H, W, 4), np.float32)
        for f in range(frames):
            buf.reset()
            vp.render_frames(buf.ptr, f, 1, P)
            per[f] = buf.download()
        buf.free()
    finally:
        vp.set_tracking(0)
        vp.set_envmap_sampling(0)
    rgb = per[..., :3].astype(np.float64)
    assert np.isfinite(rgb).all()
    # no path may have been cut at the depth limit (800 scatters / segments): that loss is the one allowed deviation
    depth = per[..., 3] if est == 1 else per[..., 3] * 1000.0
    assert depth.max() < 790
    assert (per[..., 3] > 0).mean() > 0.05                  # the medium really scatters
    mean = rgb.mean(0)
    exact = env_mode == 0 and (track == 1 or (track == 0 and len(set(sigma_t)) == 1))
    if exact:
        assert np.abs(rgb / c - 1).max() < 3e-4
        return
    assert np.abs(mean.mean((0, 1)) / c - 1).max() < 5e-3
    blk = rgb.reshape(frames, H // 4, 4, W // 4, 4, 3).mean(axis=(2, 4))      # 4x4-pixel blocks per frame
    se = blk.std(0) / np.sqrt(frames) + 1e-4
    z = np.abs(blk.mean(0) - c) / se
    assert (z > 5).mean() < 0.005, float((z > 5).mean())


# ------------------------------------------------------------------------------------------ homogeneous slab
def _slab_scene(vp, est, albedo, env_c, sun_dir, sun_power, density, key, W=40, H=30):
    env = np.zeros((8, 16, 4), np.float32)
    env[..., :3] = env_c
    env[..., 3] = 1
    vp.set_tracking(0)
    vp.set_envmap_sampling(0)
    vp.init_volume(np.full((16, 16, 16), 255, np.uint8), brick=1, linear=True)      # density01 == 1 everywhere
    vp.init_envmap(env)
    vp.set_sun(sun_dir, sun_power)
    vp.set_camera()
    vp.set_estimator(est)
    vp.set_rng(vp.RNG_PHILOX, key)
    vp.set_shard(0, 1)
    return vp.make_param(W, H, density=density, g=0.0, albedo=(albedo,) * 3, sigma_t=(1, 1, 1))


@pytest.mark.parametrize("est", [0, 2])
def test_free_flight_transmittance_of_a_homogeneous_slab(vp, est):
    """Pure absorber (albedo 0) in front of a constant environment: radiance = c * P(no collision along the camera ray)
    = c * exp(-rho * chord).  Primary free flight of the global-majorant kernel (kernel.cu:1416-1452) and of the
    restart-segment kernel (:1782-1813; the chord is crossed in 0.05 pieces).  The decomposition kernel is excluded: on a
    constant grid its control component takes the whole extinction out of the scattering coefficient (quirk Q7), which is
    only an estimator of this integral for albedo 1 -- the furnace test covers it."""
    W, H, frames, rho, c = 40, 30, 400, 1.2, 0.8
    P = _slab_scene(vp, est, 0.0, c, (1.0, 0.0, 0.0), (0, 0, 0), rho, key=(5, est), W=W, H=H)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, frames, P)
    img = buf.download()[..., 0].astype(np.float64) / frames
    buf.free()
    o, d = _camera_rays(W, H)
    hit, tmin, tmax = _slab(o, d)
    chord = np.where(hit, tmax - np.maximum(tmin, 0), 0.0)
    want = c * np.exp(-rho * chord)
    assert hit.mean() > 0.3 and (chord > 1.5).any()
    T = want / c
    se = c * np.sqrt(np.maximum(T * (1 - T), 1e-6) / frames)
    z = np.abs(img - want) / se
    assert np.allclose(img[~hit], c, rtol=2e-5)                         # rays that miss the box: the environment itself
    assert (z[hit] > 4.5).mean() < 0.01, float((z[hit] > 4.5).mean())   # binomial, 4.5 sigma per pixel
    assert abs(img[hit].mean() / want[hit].mean() - 1) < 0.01           # 1 % on the mean over the box


@pytest.mark.parametrize("est", [0, 2])
def test_single_scatter_sun_radiance_of_a_homogeneous_slab(vp, est):
    """Black environment, directional sun, albedo a << 1: the radiance is the single-scatter integral
        a * E_sun * p(cos) * int_0^chord rho e^{-rho t} e^{-rho s(t)} dt,   s(t) = distance from x(t) to the box along the sun
    (second order is O(a) smaller).  Exercises the first-collision density, HGPhaseFunction::evaluate, the directional
    sun power (kernel.cu:1269-1283) and the shadow ray's transmittance estimate Tr_spectral (kernel.cu:754-808)."""
    W, H, frames, rho, a = 40, 30, 1500, 1.5, 0.01
    sun_dir = np.array([0.0, 0.8, 0.6])
    power = np.array([4.0e5, 3.0e5, 2.0e5])
    P = _slab_scene(vp, est, a, 0.0, tuple(sun_dir), tuple(power), rho, key=(9, est), W=W, H=H)
    P.g = 0.3
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, frames, P)
    img = buf.download()[..., :3].astype(np.float64) / frames
    buf.free()
    o, d = _camera_rays(W, H)
    hit, tmin, tmax = _slab(o, d)
    t0 = np.maximum(tmin, 0)
    chord = np.where(hit, tmax - t0, 0.0)
    # quadrature over the chord (256 midpoints)
    q = (np.arange(256) + 0.5) / 256
    t = t0[..., None] + chord[..., None] * q
    x = o[..., None, :] + d[..., None, :] * t[..., None]
    _, _, s_exit = _slab(x, np.broadcast_to(sun_dir, x.shape))
    integrand = rho * np.exp(-rho * (t - t0[..., None])) * np.exp(-rho * np.maximum(s_exit, 0))
    integral = integrand.mean(-1) * chord
    g = 0.3
    cos = (d * sun_dir).sum(-1)
    ph = (1 - g * g) / (4 * PI * (1 + g * g - 2 * g * cos) ** 1.5)
    e_sun = power * PI * (0.45 / 94.0) ** 2
    want = a * ph[..., None] * integral[..., None] * e_sun
    assert np.all(img[~hit] == 0)
    # image level: 2 % (second-order scattering adds < a * 1 = 1 %, the statistical error of the sum is ~0.5 %)
    ratio = img[hit].sum(0) / want[hit].sum(0)
    assert np.all(ratio > 0.985) and np.all(ratio < 1.03), ratio
    # per 5x5-pixel block, red channel: within 6 standard errors (per-sample values are 0 or ~a*ph*E) + 3 %
    blk = lambda im: im[..., 0].reshape(H // 5, 5, W // 5, 5).mean(axis=(1, 3))
    pmax = (a * ph * e_sun[0]).max()
    se = np.sqrt(np.maximum(blk(want), 1e-12) * pmax / (frames * 25))
    bad = np.abs(blk(img) - blk(want)) > 6 * se + 0.03 * blk(want)
    assert bad.mean() < 0.02, float(bad.mean())


def test_scatter_statistics_of_the_live_kernel_at_full_size(vp):
    """SURVEY.md section 6 recorded, for the reference's live kernel on Julia-256^3 at 800x600, frames 0..7 (sampler.h
    streams): 88 % of the pixels with 0 scatters, p90 = 13.9, p99 = 48.9, max = 205 scatters per sample and pixel.  The HIP
    path on the same frames: every one of them within the spread of an 8-frame estimate -- including the tail statistic
    (max), which VERDICT r1 asked for as a second pin besides the work counters.  (The survey's "mean 1.58" cannot be a
    per-sample mean of this distribution: its own percentiles imply mean >= 0.10*13.9 + 0.01*(48.9-13.9) = 1.74.)"""
    from volpath import scene as vscene
    P, info = vscene.setup("c3ref", rng_mode=vp.RNG_SAMPLERH, last_frame=8)
    buf = vp.DeviceBuffer(800, 600)
    vp.render_frames(buf.ptr, 0, 8, P)
    heat = buf.download()[..., 3].astype(np.float64) / 8
    buf.free()
    assert abs((heat == 0).mean() - 0.88) < 0.01
    assert abs(np.percentile(heat, 90) - 13.9) < 0.8
    assert abs(np.percentile(heat, 99) - 48.9) < 2.5
    assert abs(heat.max() / 205.0 - 1) < 0.15, heat.max()      # a maximum over 480 000 pixels: noisy, still within 15 %
    assert heat.mean() > 1.74                     # consistent with the percentiles, not with "1.58"
    assert abs(heat.mean() - 3.15) < 0.1


# ------------------------------------------------------------------------------------------ the per-pixel certificates
@pytest.mark.parametrize("est", [0, 1])
def test_pixel_table_certificates_hold_in_float64(vp, est):
    """The tables that let paths skip work are claims about geometry; checked here in float64 against
    the raw volume, without the oracle: (a) before the certified-empty distance every point of the camera ray lies in a cell
    whose 2x2x2 texels are all zero; (b) class 2 = the ray misses the box, class 1 = the whole chord is empty; (c) for the
    local-majorant estimators the recorded origin is the camera origin moved by `segments` steps of 0.05 along the ray, all
    of them in front of the box, and no draw beyond one per segment (the Julia grid has no brick with a positive minimum)."""
    import scenes
    rng = np.random.default_rng(8)
    n = 48
    grid = vp.julia_volume(n)
    vp.init_volume(grid, brick=4 if est else 1, linear=True)
    vp.init_envmap(scenes.synthetic_env())
    vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_estimator(est)
    vp.set_tracking(0)
    vp.set_shard(0, 1)
    W, H = 160, 120
    P = vp.make_param(W, H)
    t = vp.pixel_table(P)
    o, d = _camera_rays(W, H)
    hit, tmin, tmax = _slab(o, d)
    cls = t[..., 5].astype(int)
    t_left = t[..., 4].astype(np.float64)
    packed = t[..., 3].view(np.uint32)
    segs, draws = (packed & 0xffff).astype(int), (packed >> 16).astype(int)
    assert set(np.unique(cls)) == {0, 1, 2}
    decided = np.abs(tmax - tmin) > 1e-4
    assert np.array_equal((cls == 2)[decided], ~hit[decided])
    # non-empty cells of the volume: any texel of the clamped 2x2x2 neighbourhood non-zero
    g = np.pad(grid, ((0, 1), (0, 1), (0, 1)), mode="edge") != 0
    cell_nonempty = np.zeros((n, n, n), bool)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                cell_nonempty |= g[dz:dz + n, dy:dy + n, dx:dx + n]
    if est:
        # the restart crawl in front of the volume: origin moved by segs * 0.05, still outside by more than a segment
        walked = segs * 0.05
        want = o + d * walked[..., None]
        assert np.abs(t[..., :3] - want).max() < 2e-4
        assert np.all(draws == segs)
        front = hit & (tmin > 0)
        assert np.all((tmin - walked)[front] <= 0.05 + 1e-4) and np.all((tmin - walked)[front] > -1e-4)
        assert segs[hit].mean() > 30 and np.all(segs[~hit & decided] == 0)
        start = walked
    else:
        assert np.all(segs == 0)
        start = np.zeros_like(tmin)
    # (a): dense sampling (1/20 cell) of [box entry, certified distance)
    ys, xs = np.nonzero(hit & (t_left > 0))
    pick = rng.choice(len(ys), 400, replace=False)
    cell = 2.0 / n
    checked = 0
    for y, x in zip(ys[pick], xs[pick]):
        t0 = max(tmin[y, x], 0.0)
        t1 = min(start[y, x] + t_left[y, x], tmax[y, x])
        if t1 <= t0:
            continue
        tt = np.arange(t0, t1, cell / 20)
        p = (o[y, x] + d[y, x] * tt[:, None] + 1.0) / 2.0 * n - 0.5
        idx = np.clip(np.floor(np.maximum(p, 0)).astype(int), 0, n - 1)
        assert not cell_nonempty[idx[:, 2], idx[:, 1], idx[:, 0]].any(), (y, x)
        checked += len(tt)
        if cls[y, x] == 1:
            assert t_left[y, x] > 1e29
    assert checked > 100000
    # the certificate is not vacuous: most box-hitting rays get one, and a good share of them is certified to the end
    assert (t_left[hit] > 0).mean() > 0.8 and (cls[hit] == 1).mean() > 0.3


@pytest.mark.parametrize("kind", ["julia", "ragged"])
@pytest.mark.parametrize("sun", [(-0.0, 0.951057, -0.309017), (0.6, -0.3, 0.74), (0.0, 0.0, -1.0)])
def test_sun_clip_certificate_holds_in_float64(vp, sun, kind):
    """Counter-based streams end a sun shadow ray where it has only empty cells left (vp_kernels.hip sun_clip_k).  The table is
    a claim about geometry, checked here in float64 against the raw volume, without the oracle: from random start points in
    random non-empty cells, every point of the ray toward the sun beyond table[cell] * step lies in a cell whose 2x2x2 texels
    are all zero.  Also: every non-empty cell has an entry, empty cells are marked unknown, and the table is not vacuous.
    `ragged`: a non-cubic grid in an off-centre, non-cubic box (cells of different edge lengths per axis)."""
    import scenes
    rng = np.random.default_rng(11)
    if kind == "julia":
        grid = vp.julia_volume(48)
        bmin, bmax = np.array([-1.0, -1.0, -1.0]), np.array([1.0, 1.0, 1.0])
        vp.init_volume(grid, brick=1, linear=True)
    else:
        g = scenes.blob_volume_u8(28, seed=5)[:20, :, :]
        grid = np.ascontiguousarray(np.pad(g, ((0, 0), (0, 0), (0, 8)))[:, :28, :36])      # nz, ny, nx = 20, 28, 36
        bmin, bmax = np.array([-0.7, -1.3, 0.1]), np.array([1.6, 0.2, 1.4])
        vp.init_volume(grid, box=(tuple(bmin), tuple(bmax)), brick=1, linear=True)
    nz, ny, nx = grid.shape
    N = np.array([nx, ny, nz], np.float64)
    vp.init_envmap(scenes.synthetic_env())
    sun = np.asarray(sun, np.float64) / np.linalg.norm(sun)
    vp.set_sun(tuple(sun), scenes.DEFAULT_SUN_POWER)
    vp.set_camera()
    vp.set_rng(vp.RNG_PHILOX7, (1, 2))
    table, step = vp.sun_clip_table((nz, ny, nx))
    cell_edges = (bmax - bmin) / N
    assert abs(step - 0.25 * cell_edges.min()) < 1e-6
    g = np.pad(grid, ((0, 1), (0, 1), (0, 1)), mode="edge") != 0
    cell_nonempty = np.zeros((nz, ny, nx), bool)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                cell_nonempty |= g[dz:dz + nz, dy:dy + ny, dx:dx + nx]
    assert np.array_equal(table == 0xffff, ~cell_nonempty)
    ks, js, is_ = np.nonzero(cell_nonempty)
    pick = rng.choice(len(ks), min(600, len(ks)), replace=False)
    checked = 0
    for k, j, i in zip(ks[pick], js[pick], is_[pick]):
        # a random point of the cell in continuous cell coordinates xb = p * N - 0.5 (cell 0 also takes xb in [-0.5, 0), the last
        # cell ends at N - 0.5), as a world position
        c = np.array([i, j, k])
        lo = np.where(c == 0, -0.5, c.astype(np.float64))
        hi = np.where(c == N - 1, N - 0.5, c + 1.0)
        xb = lo + (hi - lo) * rng.random(3)
        p0 = bmin + (xb + 0.5) / N * (bmax - bmin)
        t0 = float(table[k, j, i]) * step
        # to the box exit (slab test), sampled at 1/20 of the smallest cell edge
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = np.where(sun > 0, (bmax - p0) / sun, np.where(sun < 0, (bmin - p0) / sun, np.inf))
        t1 = float(tt.min())
        if t1 <= t0:
            continue
        ts = np.arange(t0, t1, cell_edges.min() / 20)
        q = ((p0 + sun * ts[:, None]) - bmin) / (bmax - bmin) * N - 0.5
        idx = np.clip(np.floor(np.maximum(q, 0)).astype(int), 0, (N - 1).astype(int))
        assert not cell_nonempty[idx[:, 2], idx[:, 1], idx[:, 0]].any(), (k, j, i)
        checked += len(ts)
    assert checked > (50000 if kind == "julia" else 5000)
    # not vacuous: some of the non-empty cells see the sun within a few cells, and on average a ray ends well before the box does
    assert (table[cell_nonempty] * step < 4 * cell_edges.max()).mean() > 0.04
    assert (table[cell_nonempty] * step).mean() < 0.5 * np.linalg.norm(bmax - bmin)


@pytest.mark.parametrize("kind", ["julia", "ragged"])
def test_exit_table_certificates_hold_in_float64(vp, kind):
    """Exit flights (vp_kernels.hip render_k / exit_dir_slice_k): a path that can meet empty cells only on its way out of the box is
    ended at once.  The table that says so is a claim about geometry, checked here in float64 against the raw volume, without the
    oracle: for random cells and random directions whose class bit is set, from a random start point in the cell, every point of the
    ray up to the box exit lies in a cell whose 2x2x2 texels are all zero.  Also: the table is not vacuous (most empty cells far
    from matter certify most directions) and never certifies a direction out of a non-empty cell.
    `ragged`: a non-cubic grid in an off-centre, non-cubic box (the dominant axis is that of the direction in CELL units)."""
    import scenes
    rng = np.random.default_rng(23)
    if kind == "julia":
        grid = vp.julia_volume(48)
        bmin, bmax = np.array([-1.0, -1.0, -1.0]), np.array([1.0, 1.0, 1.0])
        vp.init_volume(grid, brick=1, linear=True)
    else:
        g = scenes.blob_volume_u8(28, seed=5)[:20, :, :]
        grid = np.ascontiguousarray(np.pad(g, ((0, 0), (0, 0), (0, 8)))[:, :28, :36])      # nz, ny, nx = 20, 28, 36
        bmin, bmax = np.array([-0.7, -1.3, 0.1]), np.array([1.6, 0.2, 1.4])
        vp.init_volume(grid, box=(tuple(bmin), tuple(bmax)), brick=1, linear=True)
    nz, ny, nx = grid.shape
    N = np.array([nx, ny, nz], np.float64)
    table = vp.exit_table((nz, ny, nx))
    g = np.pad(grid, ((0, 1), (0, 1), (0, 1)), mode="edge") != 0
    cell_nonempty = np.zeros((nz, ny, nx), bool)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                cell_nonempty |= g[dz:dz + nz, dy:dy + ny, dx:dx + nx]
    assert not table[:, cell_nonempty].any()                      # no direction is certified out of a non-empty cell
    cell_edges = (bmax - bmin) / N
    checked = rays = 0
    for _ in range(4000):
        c = np.array([rng.integers(0, nx), rng.integers(0, ny), rng.integers(0, nz)])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        e = d * N / (bmax - bmin)                                 # the direction in cell units
        ae = np.abs(e)
        A = 0 if (ae[0] >= ae[1] and ae[0] >= ae[2]) else (1 if ae[1] >= ae[2] else 2)
        B, Cx = [a for a in range(3) if a != A]
        cls = int(e[A] > 0) | int(e[B] > 0) << 1 | int(e[Cx] > 0) << 2
        if not (table[A, c[2], c[1], c[0]] >> cls) & 1:
            continue
        lo = np.where(c == 0, -0.5, c.astype(np.float64))
        hi = np.where(c == N - 1, N - 0.5, c + 1.0)
        xb = lo + (hi - lo) * rng.random(3)
        p0 = bmin + (xb + 0.5) / N * (bmax - bmin)
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = np.where(d > 0, (bmax - p0) / d, np.where(d < 0, (bmin - p0) / d, np.inf))
        t1 = float(tt.min())
        ts = np.arange(0.0, max(t1, 0.0) + cell_edges.min() / 20, cell_edges.min() / 20)
        q = ((p0 + d * ts[:, None]) - bmin) / (bmax - bmin) * N - 0.5
        idx = np.clip(np.floor(np.maximum(q, 0)).astype(int), 0, (N - 1).astype(int))
        assert not cell_nonempty[idx[:, 2], idx[:, 1], idx[:, 0]].any(), (c, d)
        checked += len(ts)
        rays += 1
    assert rays > 800 and checked > 100000
    # not vacuous: cells on the faces of the box certify the directions that leave through their face at once
    assert (table[0, :, :, nx - 1] & 0xaa).any() and (table[0, :, :, 0] & 0x55).any()
    assert (table != 0).mean() > 0.3


def test_shadow_rays_draw_from_their_own_substream(vp):
    """Counter-based streams: what a path draws after a light estimate does not depend on the shadow ray.  With the sun's power
    at zero the image is the environment seen by the escaping paths, so two sun directions (shadow rays of different lengths,
    neither in the camera's view) give the same image bit for bit -- and different ones with the reference's sequential
    sampler.h stream, where a shadow ray's draws move everything behind it."""
    import scenes
    grid = vp.julia_volume(32)
    W, H = 64, 48
    out = {}
    for rng_mode in (vp.RNG_PHILOX7, vp.RNG_PHILOX, vp.RNG_SAMPLERH):
        for est in (vp.EST_GLOBAL, vp.EST_DECOMP):
            imgs = []
            for sun in ((-0.0, 0.951057, -0.309017), (0.0, 0.6, 0.8)):
                vp.init_volume(grid, brick=1, linear=True)
                vp.init_envmap(scenes.synthetic_env())
                vp.set_sun(sun, (0.0, 0.0, 0.0))
                vp.set_camera()
                vp.set_estimator(est)
                vp.set_tracking(0)
                vp.set_shard(0, 1)
                vp.set_rng(rng_mode, (5, 6))
                P = vp.make_param(W, H)
                buf = vp.DeviceBuffer(W, H)
                vp.render_frames(buf.ptr, 0, 6, P)
                imgs.append(buf.download())
                buf.free()
            out[(rng_mode, est)] = np.array_equal(imgs[0], imgs[1])
            assert imgs[0][..., 3].max() > 0  # paths do scatter
    for est in (vp.EST_GLOBAL, vp.EST_DECOMP):
        assert out[(vp.RNG_PHILOX7, est)] and out[(vp.RNG_PHILOX, est)]
        assert not out[(vp.RNG_SAMPLERH, est)]


def test_null_collision_table_is_the_float32_recurrence(vp):
    """The light kernel of the global-majorant estimator looks a path's throughput up by its number of null collisions in empty
    space (vp_kernels.hip thr_table_k).  The table restated here in numpy binary32, operation by operation, from the reference's
    expressions with density +0 (kernel.cu:1355-1366, :1419-1443: Ps = +0, c = Pn = |s t| + |s t| + |s t|, t *= s * ((c / s') / c)
    in the order the reference evaluates them).  No oracle involved.  The weight is exactly 1 for most media (800, the default,
    among them) and one ulp off for about one in five (209, 246 here): there the throughput really depends on the step count."""
    f = np.float32
    moved = 0
    for sigma_t, density, g in (((1, 1, 1), 800.0, 0.877), ((1, 1, 1), 333.3, 0.0), ((0.953, 1.0, 0.843), 800.0, 0.877),
                                ((0.3, 0.7, 0.9), 57.3, -0.4), ((1, 1, 1), 1.0e-3, 0.5), ((1, 1, 1), 209.0, 0.877),
                                ((0.3, 0.7, 1.0), 246.0, -0.4)):
        P = vp.make_param(8, 8, density=density, g=g, sigma_t=sigma_t)
        n = 6000   # beyond the 4096 entries the kernel keeps: same recurrence either way
        got = vp.null_collision_table(P, n)
        st = [f(P.sigma_t.x), f(P.sigma_t.y), f(P.sigma_t.z)]
        s = f(max(f(0), min(f(1), f(-5) * f(0.066666666666666666667))))
        cur = (f(1) - s) * f(P.density) + s * f(P.density) * (f(1) - f(P.g))
        sp = max(st) * cur
        inv = f(1) / sp
        t = f(1)
        ref = np.empty(n, f)
        for k in range(n):
            ref[k] = t
            m = abs(sp * t)
            pn = (m + m) + m
            t = t * (sp * ((inv * pn) / pn))
        assert np.array_equal(got, ref), (sigma_t, density, g, int(np.argmax(got != ref)))
        assert abs(float(got[-1]) - 1.0) < 1e-3        # a rounding drift, not a physical attenuation
        moved += int(got[-1] != 1.0)
    assert moved >= 2


def test_julia_silhouette_matches_the_references_own_screenshot_on_the_gpu(vp):
    """The same reference-held pin as tests/test_oracle_cpu.py::test_julia_silhouette_matches_the_references_own_screenshot, for
    the HIP path alone (no oracle): GPU Julia voxeliser at 256^3, the fitted orbit pose at the reference's default distance 4.0,
    960x512 = the reference's window; the pixels where a sample scattered against the silhouette of the reference's own
    screenshot 2.jpg: IoU >= 0.95 (0.970 when fitted), and 5 % more distance costs more than 0.07."""
    import os
    import scenes
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_julia_silhouette.npz"))
    H, W = (int(v) for v in z["shape"])
    mask = np.unpackbits(z["mask_bits"])[:H * W].reshape(H, W).astype(bool)
    cam = np.asarray(z["camera"], np.float32)
    grid = vp.julia_volume(256)
    iou = lambda a, b: (a & b).sum() / max((a | b).sum(), 1)

    def silhouette(camera):
        vp.init_volume(grid, brick=1, linear=True)
        vp.init_envmap(scenes.synthetic_env())
        vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
        vp.set_camera(tuple(float(v) for v in camera))
        vp.set_estimator(vp.EST_GLOBAL)
        vp.set_tracking(0)
        vp.set_shard(0, 1)
        vp.set_rng(vp.RNG_PHILOX7, (1, 2))
        P = vp.make_param(W, H)
        buf = vp.DeviceBuffer(W, H)
        vp.render_frames(buf.ptr, 0, 4, P)
        out = buf.download()[..., 3] > 0
        buf.free()
        return out

    try:
        good = iou(mask, silhouette(cam))
        assert good >= 0.95, good
        far = cam.reshape(3, 4).copy()
        far[:, 3] += 0.2 * far[:, 2]
        assert iou(mask, silhouette(far.ravel())) < good - 0.07
    finally:
        vp.set_camera()


def test_julia_interior_matches_the_references_own_screenshot_on_the_gpu(vp):
    """The radiometric pin of tests/test_oracle_cpu.py::test_julia_interior_matches_the_references_own_screenshot for the HIP path
    alone (no oracle), at the screenshot's full 960x512 and 128 spp, live estimator across its frame-11 switch: the block-mean
    luminances inside the silhouette of the reference's own render 2.jpg against this library's render of the fitted pose under the
    fitted sun (two parameters; exposure not fitted).  Pearson >= 0.99 (0.996), rms residual <= 0.025 of a mean of 0.39 (0.016), at
    an exposure scale of 0.87.  AND what the pin excludes: the same scene with an isotropic phase function, a tenth of the density
    or albedo 0.8 misses the screenshot by several times that residual -- the reference's defaults (host.cpp:1286-1292) are what
    reproduces its output, through this integrator."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_cpu import _interior_stats
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    zs, z = np.load(os.path.join(g, "ref_julia_silhouette.npz")), np.load(os.path.join(g, "ref_julia_interior.npz"))
    H, W = (int(v) for v in zs["shape"])
    mask = np.unpackbits(zs["mask_bits"])[:H * W].reshape(H, W).astype(bool)
    grid = vp.julia_volume(256)
    env = np.full((8, 16, 4), 0.03, np.float32)
    env[..., 3] = 1.0
    sun_dir, sun_power = tuple(float(v) for v in z["sun_dir"]), tuple(float(v) for v in z["sun_power"])
    spp = 128

    def stats(**medium):
        vp.init_volume(grid, brick=1, linear=True)
        vp.init_envmap(env)
        vp.set_sun(sun_dir, sun_power)
        vp.set_camera(tuple(float(v) for v in zs["camera"]))
        vp.set_estimator(vp.EST_DECOMP)
        vp.set_tracking(0)
        vp.set_shard(0, 1)
        vp.set_rng(vp.RNG_PHILOX7, (1, 2))
        vp.precompute_opacity(sun_dir)
        P = vp.make_param(W, H, **medium)
        buf = vp.DeviceBuffer(W, H)
        vp.render_frames(buf.ptr, 0, spp, P)
        img = buf.download()[..., :3].astype(np.float64) / spp
        buf.free()
        return _interior_stats(img, mask, z["luminance"], z["count"], int(z["block"]))

    try:
        pear, scale, rms, nb = stats()
        assert nb > 300
        assert pear >= 0.99 and rms <= 0.025 and 0.80 <= scale <= 0.95, (pear, scale, rms)
        for medium, factor in ((dict(g=0.0), 3.0), (dict(density=80.0), 5.0), (dict(albedo=(0.8, 0.8, 0.8)), 5.0)):
            p2, s2, r2, _ = stats(**medium)
            assert r2 >= factor * rms and p2 < pear - 0.05, (medium, p2, s2, r2)
    finally:
        vp.set_camera()


@pytest.mark.parametrize("shape,box", [((29, 22, 37), ((-1.0, -0.6, -1.3), (1.0, 0.9, 1.2))), ((8, 16, 24), None), ((3, 5, 2), None), ((64, 64, 64), None),
                                       ((130, 17, 9), ((0.2, -0.1, 0.3), (1.7, 0.05, 0.4)))])
def test_opacity_march_through_lds_tiles_is_the_definition_bit_for_bit(vp, shape, box):
    """north_star's "density grid staged through LDS" where the path's rays are coherent: precompute_opacity (A10, kernel.cu:483-553)
    marches EVERY voxel along one direction with one step, so a workgroup's 8x8x8 voxels are a rigid body moving through the grid and
    the texels it filters over a chunk of steps are a 16^3 box -- opacity_lds_k stages that box (4 KiB, coalesced) and filters from
    LDS.  opacity_k (one packed-cell gather per step, the kernel the oracle comparisons of test_c4_gpu.py were written against)
    stays the definition: here both build the whole table for ragged, flat, tiny and off-centre grids, for directions along an axis,
    against every face, grazing and steep, with both filter modes -- np.array_equal.  (The tile's position is a performance matter
    only: a sample outside it reads the packed cell from global memory; the tiny and flat grids take that path for their clamped
    taps, the others do not.)"""
    import os
    nz, ny, nx = shape
    rs = np.random.default_rng(nx * 1000 + ny)
    g = rs.random((nz, ny, nx), dtype=np.float32)
    g[rs.random(g.shape) < 0.4] = 0
    grid = np.ascontiguousarray((g * 255).astype(np.uint8))
    dirs = [(0.0, 0.951057, -0.309017), (1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0), (-0.57735, 0.57735, -0.57735),
            (0.999, 0.04, 0.02), (-0.3, -0.2, 0.93)]

    def tables(linear):
        out = []
        vp.init_volume(grid, box=box, brick=1, linear=linear)
        for d in dirs:
            vp.precompute_opacity(d)
            out.append(vp.opacity_table(shape))
        return out

    for linear in (True, False):
        lds = tables(linear)
        old = os.environ.get("VP_NO_OPACITY_LDS")
        os.environ["VP_NO_OPACITY_LDS"] = "1"
        ctx = vp.Context(0)
        try:
            with ctx:
                ref = tables(linear)
        finally:
            ctx.destroy()
            if old is None:
                del os.environ["VP_NO_OPACITY_LDS"]
            else:
                os.environ["VP_NO_OPACITY_LDS"] = old
        for d, a, b in zip(dirs, lds, ref):
            assert np.array_equal(a, b), (linear, d, int((a != b).sum()), float(np.abs(a - b).max()))
            assert (b > 0).any()
