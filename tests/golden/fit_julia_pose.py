#!/usr/bin/env python3
"""How the camera pose of tests/golden/ref_julia_silhouette.npz was found (run once, in the build container; ~10 minutes).

The reference repository holds ONE output of its procedural Julia-set scene: the screenshot /root/reference/2.jpg (960x512, the
reference's default window, host.cpp:1287-1288).  The viewer is interactive, so the camera of the screenshot is unknown: it orbits
its focus point at cam_focus_dist = 4 (host.cpp:112, :819-831).  This script recovers the orbit (direction, roll, pan; the distance
is HELD at the reference's default 4.0) by maximising the intersection-over-union of the screenshot's silhouette with the silhouette
the CPU oracle renders (pixels where a sample scattered): a coarse search over 900 directions x 60 image rotations at quarter
resolution, then Nelder-Mead at half resolution on the 256^3 grid.  Result (stored in the fixture): IoU 0.975 at half resolution,
0.970 at full resolution, roll = -3.1428 (pi to 0.001: the screenshot's row order), pan (0.056, 0.070).
Only the silhouette is compared: the screenshot's environment (a uniform grey) is not the sun/sky of the current source.
"""
import sys
import os
import numpy as np
from PIL import Image
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import oracle_lib as O  # noqa: E402
import scenes  # noqa: E402
from volpath import host  # noqa: E402

# focus point of the reference's default camera: cam_position + cam_forward * cam_focus_dist (host.cpp:108-112)
CENTRE = np.array([3.922986, -0.782739, 0.03]) + 4.0 * np.array([-0.978148, 0.207912, 0.0])


def reference_mask(path="/root/reference/2.jpg"):
    ref = np.asarray(Image.open(path).convert("RGB")).astype(np.float32)
    bg = np.median(ref.reshape(-1, 3), axis=0)          # the uniform background
    return np.abs(ref - bg).max(axis=2) > 24


def camera(p):
    """p = (polar angle of the camera direction from +y, azimuth, roll, distance, pan right, pan up) -> 3x4 camera-to-world"""
    th, ph, roll, dist, px, py = p
    d = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
    up0 = np.array([0.0, 1.0, 0.0]) - d * d[1]
    up0 /= np.linalg.norm(up0)
    right0 = np.cross(up0, d)
    up = np.cos(roll) * up0 + np.sin(roll) * right0
    right = np.cross(up, d)
    pos = CENTRE + px * right + py * up + dist * d
    return host.camera_matrix(pos.astype(np.float32), (-d).astype(np.float32), up.astype(np.float32))


def oracle_mask(grid, p, W, H, frames=2):
    osc = O.OracleScene(grid, scenes.synthetic_env(), O.DEFAULT_SUN_DIR, O.DEFAULT_SUN_POWER, estimator=O.EST_GLOBAL,
                        rng_mode=O.RNG_PHILOX7, seed=(1, 2), inv_view=camera(p), radius=1)
    P = O.default_param(W, H)
    acc = None
    for f in range(frames):
        acc, _ = osc.render_frame(P, f, acc)
    return acc[..., 3] > 0


def iou(a, b):
    return (a & b).sum() / max((a | b).sum(), 1)


def main():
    full = reference_mask()
    small = lambda S: np.asarray(Image.fromarray((full * 255).astype(np.uint8)).resize((960 // S, 512 // S), Image.BILINEAR)) > 127
    g128 = O.julia(128)
    best, m4 = (0.0, None), small(4)
    for i in range(900):                                  # Fibonacci sphere of camera directions, image rotations for the roll
        y = 1 - 2 * (i + 0.5) / 900
        ph = i * np.pi * (3 - np.sqrt(5))
        th = np.arccos(y)
        m = oracle_mask(g128, (th, ph, 0.0, 4.0, 0.0, 0.0), 240, 128, frames=1)
        for ang in range(0, 360, 6):
            r = np.asarray(Image.fromarray((m * 255).astype(np.uint8)).rotate(ang, resample=Image.BILINEAR)) > 127
            s = iou(m4, r)
            if s > best[0]:
                best = (s, (th, ph, -np.deg2rad(ang)))
    print("coarse", best)
    g256, m2 = O.julia(256), small(2)
    q0 = np.array([best[1][0], best[1][1], best[1][2], 0.0, 0.0])
    to_p = lambda q: np.array([q[0], q[1], q[2], 4.0, q[3], q[4]])
    r = minimize(lambda q: -iou(m2, oracle_mask(g256, to_p(q), 480, 256)), q0, method="Nelder-Mead",
                 options=dict(maxiter=400, xatol=5e-4, fatol=1e-4, initial_simplex=q0 + np.vstack([np.zeros(5), np.diag([0.05] * 5)])))
    p = to_p(r.x)
    print("refined", -r.fun, list(p), "full resolution", iou(full, oracle_mask(g256, p, 960, 512, frames=3)))


if __name__ == "__main__":
    main()
