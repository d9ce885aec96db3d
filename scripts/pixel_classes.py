import os, sys
sys.path.insert(0, "/root/repo/cuda-volpath_amd")
import numpy as np, volpath as vp
from volpath import scene
vp.set_device(0)
for wl in ("c2", "c3", "c4s"):
    P, info = scene.setup(wl, rng_mode=vp.RNG_PHILOX, last_frame=0)
    t = vp.pixel_table(P)
    te = t[..., 4]
    segs = t[..., 3].view(np.uint32) & 0xffff
    inf = te > 1e29
    zero = te == 0
    print(wl, "pixels", te.size, "t_empty=inf %.3f" % inf.mean(), "t_empty=0 %.3f" % zero.mean(), "finite>0 %.3f" % ((~inf & ~zero).mean()), "crawl segs mean %.1f" % segs.mean())
    # tiles all-inf or all (inf or zero-with-no-hit)
    H, W = te.shape
    th, tw = (H + 7) // 8, (W + 7) // 8
    pad = np.ones((th * 8, tw * 8), bool); pad[:H, :W] = inf
    tiles = pad.reshape(th, 8, tw, 8).all(axis=(1, 3))
    print("   tiles with every pixel t_empty=inf: %.3f" % tiles.mean())
