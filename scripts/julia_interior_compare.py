"""The first radiometric comparison with a reference OUTPUT: the interior of the reference's Julia screenshot
2.jpg (tests/golden/ref_julia_interior.npz: block means of its linearised pixels inside the silhouette) against this library's
render of the fitted pose under a lighting HYPOTHESIS -- the screenshot's lighting is not recorded anywhere.  Reports rank correlation
of the block luminances and one fitted exposure scale; nothing is tuned.   python scripts/julia_interior_compare.py [SPP]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import volpath as vp
from volpath import scene as vscene


_GRID = None


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)).astype(np.float64), np.argsort(np.argsort(b)).astype(np.float64)
    ra -= ra.mean(); rb -= rb.mean()
    return float((ra * rb).sum() / np.sqrt((ra * ra).sum() * (rb * rb).sum()))


def compare(env, sun_dir, sun_power, spp=256, est=None):
    g = os.path.join(ROOT, "tests", "golden")
    z, ref = np.load(os.path.join(g, "ref_julia_silhouette.npz")), np.load(os.path.join(g, "ref_julia_interior.npz"))
    H, W = (int(v) for v in z["shape"])
    mask = np.unpackbits(z["mask_bits"])[:H * W].reshape(H, W).astype(bool)
    global _GRID
    if _GRID is None:
        _GRID = vp.julia_volume(256)
    vp.init_volume(_GRID, brick=1, linear=True)
    vp.init_envmap(env)
    vp.set_sun(sun_dir, sun_power)
    vp.set_camera(tuple(float(v) for v in z["camera"]))
    vp.set_estimator(vp.EST_DECOMP if est is None else est)        # the reference's live kernel
    vp.set_tracking(0); vp.set_shard(0, 1)
    vp.set_rng(vp.RNG_PHILOX7, (1, 2))
    if est is None or est == vp.EST_DECOMP:
        vp.precompute_opacity(sun_dir)
    P = vp.make_param(W, H)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, spp, P)
    img = buf.download()[..., :3].astype(np.float64) / spp
    buf.free()
    vp.set_camera()
    B = int(ref["block"]); by, bx = H // B, W // B
    cnt = ref["count"]; use = cnt >= 128
    w = np.array([0.2126, 0.7152, 0.0722])

    def block_lum(scale):
        lum = np.minimum(img * scale, 1.0) @ w                        # the display clamps at 1 before its gamma
        s = (lum * mask).reshape(by, B, bx, B).sum((1, 3))
        return (s / np.maximum(cnt, 1))[use]
    r = ref["luminance"][use].astype(np.float64)
    raw = block_lum(1.0)
    scales = np.geomspace(1e-3, 1e3, 601)
    err = [float(((block_lum(s) - r) ** 2).mean()) for s in scales]
    k = int(np.argmin(err))
    fit = block_lum(scales[k])
    return {"blocks": int(use.sum()), "spearman_unscaled": spearman(raw, r), "spearman_at_fit": spearman(fit, r),
            "exposure_scale": float(scales[k]), "rms_residual": float(np.sqrt(err[k])), "ref_mean": float(r.mean()), "ref_std": float(r.std()),
            "render_mean_unscaled": float(raw.mean()), "pearson_at_fit": float(np.corrcoef(fit, r)[0, 1]),
            "render_unclamped_mean_radiance": float((img @ w)[mask].mean())}


if __name__ == "__main__":
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    vp.set_device(0)
    grey = np.full((8, 16, 4), 0.03, np.float32); grey[..., 3] = 1.0
    two = grey.copy(); two[:5, :, 0] = 0.03; two[:5, :, 1] = 0.07; two[:5, :, 2] = 0.23           # host.cpp:1374-1385, the disabled branch
    sunsky = vscene.default_sunsky()
    hyps = {
        "uniform 0.03 grey environment + the default sun (setup_sunsky(0.5, 0.2))": (grey, sunsky[1], sunsky[2]),
        "the source's disabled two-tone environment (host.cpp:1374-1385) + the default sun": (two, sunsky[1], sunsky[2]),
        "uniform grey environment, no sun": (grey, sunsky[1], (0.0, 0.0, 0.0)),
        "the default Hosek sky + sun (the current source's live branch)": sunsky,
    }
    import json
    for name, (env, sd, sp) in hyps.items():
        print(name, json.dumps(compare(env, sd, sp, spp)), flush=True)
