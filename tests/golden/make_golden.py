#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.  Run in the build container:
    python tests/golden/make_golden.py

Sources, by authority:
  ref_anchors.json   values recorded FROM THE REFERENCE (SURVEY.md section 4 / section 6 probes) and the
                     published Random123 Philox known-answer vectors -- typed in and kept by hand; this script never
                     writes it.
  hosek_ref.npz      outputs of the reference's own Hosek sky sources (oracle/_ref/libhosek_ref.so,
                     built by oracle/Makefile from /root/reference/src/sunsky/hosek where they lie).
  oracle_*.npz       outputs of the CPU oracle (regression vectors for the GPU path; oracle-made).
  ref_julia_silhouette.npz   the silhouette of the reference's OWN render of its procedural Julia-set scene -- the screenshot
                     /root/reference/2.jpg, 960x512 = the reference's default window -- as a bit mask (pixels that differ from the
                     uniform background), and the camera pose recovered for it by tests/golden/fit_julia_pose.py (orbit direction,
                     roll and pan fitted; the distance held at the reference's default 4.0).  Pins the geometry chain -- Julia
                     voxeliser, volume box, camera matrix, field of view, pixel-to-ray map, box intersection -- against an output
                     the reference itself holds.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import scenes  # noqa: E402


def hosek_ref():
    """Sky and solar radiance from the reference's Hosek-Wilkie implementation (alien-world init,
    sky_tungsten.cpp:416-429: intensity 100, T=5777 K, turbidity 2, albedo 0.2)."""
    path = os.path.join(ROOT, "oracle", "_ref", "libhosek_ref.so")
    L = C.CDLL(path)
    init = L._Z42arhosekskymodelstate_alienworld_alloc_initddddd
    init.restype = C.c_void_p
    init.argtypes = [C.c_double] * 5
    rad = L._Z24arhosekskymodel_radianceP20ArHosekSkyModelStateddd
    rad.restype = C.c_double
    rad.argtypes = [C.c_void_p] + [C.c_double] * 3
    srad = L._Z30arhosekskymodel_solar_radianceP20ArHosekSkyModelStateddd
    srad.restype = C.c_double
    srad.argtypes = [C.c_void_p] + [C.c_double] * 3
    elevs = np.array([0.05, 0.3, 0.62831853, 1.0, 1.2566370614359172, 1.5], np.float64)  # incl. default (pi*0.4)
    thetas = np.linspace(0.0, 1.55, 9)
    gammas = np.linspace(0.0, 3.1, 9)
    lambdas = np.array([360.0 + i * (830.0 - 360.0) / 9 for i in range(7)], np.float64)  # sky_tungsten.cpp:385-390
    sky = np.zeros((len(elevs), len(thetas), len(gammas), len(lambdas)))
    sun = np.zeros((len(elevs), len(lambdas)))
    for a, el in enumerate(elevs):
        st = init(float(el), 100.0, 5777.0, 2.0, 0.2)
        for b, th in enumerate(thetas):
            for c, ga in enumerate(gammas):
                for d, lam in enumerate(lambdas):
                    sky[a, b, c, d] = rad(st, float(th), float(ga), float(lam))
        for d, lam in enumerate(lambdas):
            sun[a, d] = srad(st, float(np.pi / 2 - el), 0.0, float(lam))
    np.savez_compressed(os.path.join(HERE, "hosek_ref.npz"), elevations=elevs, thetas=thetas, gammas=gammas,
                        lambdas=lambdas, sky_radiance=sky, solar_radiance=sun)


def oracle_renders():
    """Oracle accumulators for a tiny scene: Julia 32^3 @ 64x48, synthetic 64x32 sky, default sun/camera."""
    grid = O.julia(32)
    env = scenes.synthetic_env()
    out = {"julia32": grid}
    for est, name in ((O.EST_DECOMP, "decomp"), (O.EST_GLOBAL, "global"), (O.EST_BOUNDED, "bounded")):
        for rng, rname in ((O.RNG_SAMPLERH, "samplerh"), (O.RNG_PHILOX, "philox")):
            sc = O.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=est,
                               rng_mode=rng, seed=(123, 456))
            sc.precompute_opacity()
            P = O.default_param(64, 48)
            acc = None
            for f in range(14):
                acc, _ = sc.render_frame(P, f, acc)
            out[f"{name}_{rname}_f0_13"] = acc
            if est == O.EST_DECOMP and rng == O.RNG_SAMPLERH:
                out["opacity32"] = sc.opacity.astype(np.float32)
                out["bounds32_r1"] = sc.bounds
    # the compiled-out builds of the reference's switches: active environment sampling (MIS) and scalar tracking
    for tag, kw in (("mis", dict(env_mis=True)), ("scalar", dict(track_mode=1)), ("multichannel", dict(track_mode=2))):
        sc = O.OracleScene(grid, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, estimator=O.EST_DECOMP,
                           rng_mode=O.RNG_PHILOX, seed=(123, 456), **kw)
        sc.precompute_opacity()
        P = O.default_param(64, 48, density=150.0, g=0.6, albedo=(0.9, 0.8, 0.7), sigma_t=(1.0, 0.7, 0.45))
        acc = None
        for f in range(8, 14):
            acc, _ = sc.render_frame(P, f, acc)
        out[f"decomp_philox_{tag}_f8_13"] = acc
    # chromatic preset #1 with the brick table
    g64 = O.julia(64)
    sc = O.OracleScene(g64, env, scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER, brick=8)
    P = O.mat(O.default_param(64, 48), *scenes.PRESET1)
    acc = None
    for f in range(4):
        acc, _ = sc.render_frame(P, f, acc)
    out["julia64_brick8_preset1_f0_3"] = acc
    out["bounds64_brick8"] = sc.bounds
    np.savez_compressed(os.path.join(HERE, "oracle_renders.npz"), **out)


def ref_julia_silhouette(pose=None):
    """tests/golden/ref_julia_silhouette.npz from the reference's screenshot; `pose` = the six numbers fit_julia_pose.py prints
    (default: keep the pose already stored in the fixture)."""
    sys.path.insert(0, HERE)
    import fit_julia_pose as F
    path = os.path.join(HERE, "ref_julia_silhouette.npz")
    if pose is None:
        pose = np.load(path)["pose"]
    m = F.reference_mask()
    np.savez_compressed(path, mask_bits=np.packbits(m), shape=np.array(m.shape), pose=np.asarray(pose, np.float64),
                        camera=np.asarray(F.camera(pose), np.float32), centre=F.CENTRE)


if __name__ == "__main__":
    O.build()
    hosek_ref()
    oracle_renders()
    ref_julia_silhouette()
    print("golden fixtures written to", HERE)
