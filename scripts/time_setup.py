"""Times the one-off set-up stages (volume upload + bound table, opacity precompute) per volume size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import volpath as vp
from volpath import scene
vp.set_device(0)
for n in (128, 256, 512):
    t = time.time(); g = vp.julia_volume(n); t_j = time.time() - t
    for brick in (1, 8):
        t = time.time(); vp.init_volume(g, brick=brick); vp.synchronize(); t_i = time.time() - t
        print(f"N={n} brick={brick}: julia {t_j:.3f} s, init_cuda {t_i:.3f} s", flush=True)
    t = time.time(); vp.precompute_opacity(scene.DEFAULT_SUN_DIR); vp.synchronize(); t_o = time.time() - t
    print(f"N={n}: precompute_opacity {t_o:.3f} s", flush=True)
