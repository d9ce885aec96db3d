"""Per-shard render time of one workload for world sizes 2/4/8, one process, shards rendered one after another:
max/mean of the shard times is the load-balance bound on multi-GPU efficiency.  usage: shard_balance.py [workload] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp
from volpath import scene
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
vp.set_device(0)
P, info = scene.setup(wl, last_frame=frames)
buf = vp.DeviceBuffer(P.width, P.height)
for world in (1, 2, 4, 8):
    t = []
    for r in range(world):
        vp.set_shard(r, world)
        vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize(); vp.render_time_ms()
        vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
        ms, n = vp.render_time_ms()
        t.append(ms)
    t = np.array(t)
    print(f"{wl} world={world}: shard ms {np.round(t, 1).tolist()}  max/mean {t.max() / t.mean():.3f}  sum/serial {t.sum():.1f}")
