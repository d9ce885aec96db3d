"""Image means of the three kernels and the compiled-out builds at BASELINE size (debug aid for the convergence test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp
from volpath import scene as vscene
vp.set_device(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 192
for name, est, track, envm in (("decomp", 1, 0, 0), ("global", 0, 0, 0), ("bounded", 2, 0, 0), ("decomp_mis", 1, 0, 1),
                               ("decomp_scalar", 1, 1, 0), ("decomp_multichannel", 1, 2, 0)):
    vp.set_tracking(track); vp.set_envmap_sampling(envm)
    P, info = vscene.setup("c3ref", rng_mode=vp.RNG_PHILOX, key=(11, est * 7 + track * 3 + envm), last_frame=frames)
    vp.set_estimator(est)
    buf = vp.DeviceBuffer(800, 600)
    vp.render_frames(buf.ptr, 0, frames, P)
    im = buf.download()[..., :3].astype(np.float64) / frames
    print(name, im.mean(axis=(0, 1)), im.max(), np.isfinite(im).all())
    buf.free()
