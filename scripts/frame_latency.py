"""The reference's call pattern: one render_kernel per frame with a sync after each (host.cpp:631-632).  usage: frame_latency.py [workload] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp
from volpath import scene
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
vp.set_device(0)
P, info = scene.setup(wl, last_frame=N + 4, rng_mode=int(os.environ.get("VP_PERF_RNG", vp.RNG_PHILOX)))   # (Philox2x32-10 unless VP_PERF_RNG says otherwise)
buf = vp.DeviceBuffer(P.width, P.height)
for f in range(4):
    vp.render_kernel(buf.ptr, f, P); vp.synchronize()
vp.render_time_ms()
t0 = time.perf_counter(); per = []
for f in range(4, 4 + N):
    t = time.perf_counter(); vp.render_kernel(buf.ptr, f, P); vp.synchronize(); per.append((time.perf_counter() - t) * 1e3)
wall = (time.perf_counter() - t0) * 1e3
ms, n = vp.render_time_ms()
per = np.array(per)
print(f"{wl}: {N} single-frame calls: wall {wall:.1f} ms ({P.width * P.height * N / wall / 1e3:.1f} Msamples/s), kernel {ms:.1f} ms over {n} launches, "
      f"per-frame median {np.median(per):.2f} ms, min {per.min():.2f}, max {per.max():.2f}")
