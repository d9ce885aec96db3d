import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import volpath as vp
from volpath import scene
mode = sys.argv[1]
N = 1200
torch.cuda.set_device(0)
vp.set_device(0)
if mode in ("torchstream", "torchstream_hi"):
    st = torch.cuda.Stream(priority=-1 if mode.endswith("hi") else 0)
    vp.set_stream(st.cuda_stream)
P, info = scene.setup("c2", last_frame=N + 4, rng_mode=vp.RNG_PHILOX7)
buf = vp.DeviceBuffer(P.width, P.height)
for f in range(4):
    vp.render_kernel(buf.ptr, f, P); vp.synchronize()
for rep in range(2):
    buf.reset(); vp.synchronize()
    t0 = time.perf_counter()
    for f in range(N):
        vp.render_kernel(buf.ptr, f, P); vp.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(mode, f"{N} frames: {P.width * P.height * N / wall / 1e3:.1f} Msamples/s")
