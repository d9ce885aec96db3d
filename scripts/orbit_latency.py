"""The reference's interactive use: the camera orbits, every move resets the accumulation (fb->reset(), src/volumeRender.cpp:617-625, :769) and
the host goes on calling render_kernel frame by frame.  Per move: the per-camera tables + pixel lists are rebuilt on the GPU and the frame
look-ahead starts over.  usage: orbit_latency.py [workload] [moves] [frames per move]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import numpy as np
import volpath as vp
from volpath import scene, host
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fpm = int(sys.argv[3]) if len(sys.argv) > 3 else 32
vp.set_device(0)
P, info = scene.setup(wl, last_frame=fpm + 4, rng_mode=vp.RNG_PHILOX7)
buf = vp.DeviceBuffer(P.width, P.height)
first, rest, prep = [], [], []
t_all = time.perf_counter()
for m in range(moves):
    a = 2.0 * np.pi * m / moves
    pos = (3.9 * np.cos(a), -0.78, 3.9 * np.sin(a))
    fwd = (-np.cos(a), 0.2, -np.sin(a))
    vp.set_camera(tuple(float(v) for v in host.camera_matrix(pos, fwd, (0.0, 1.0, 0.0))))
    buf.reset()
    vp.synchronize()
    t = time.perf_counter(); vp.prepare(P); prep.append((time.perf_counter() - t) * 1e3)
    for f in range(fpm):
        t = time.perf_counter(); vp.render_kernel(buf.ptr, f, P); vp.synchronize()
        (first if f == 0 else rest).append((time.perf_counter() - t) * 1e3)
wall = (time.perf_counter() - t_all) * 1e3
print(f"{wl}: {moves} camera moves x {fpm} frames: wall {wall:.0f} ms ({P.width * P.height * moves * fpm / wall / 1e3:.1f} Msamples/s); per move: "
      f"tables + pixel lists (vp_prepare) median {np.median(prep):.2f} ms max {np.max(prep):.2f}; first frame median {np.median(first):.2f} ms; "
      f"later frames median {np.median(rest):.3f} ms mean {np.mean(rest):.3f}")
