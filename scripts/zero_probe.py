import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp, scenes
vp.set_device(0)
n = int(sys.argv[1]); rng = int(sys.argv[2])
grid = np.zeros((n, n, n), np.uint8)
vp.init_volume(grid); vp.init_envmap(scenes.synthetic_env(1024, 512)); vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
vp.set_camera(); vp.set_estimator(0); vp.set_rng(rng, (1, 2))
P = vp.make_param(800, 600); buf = vp.DeviceBuffer(800, 600)
vp.render_frames(buf.ptr, 0, 2, P); vp.synchronize(); vp.render_time_ms()
vp.render_frames(buf.ptr, 0, 32, P); vp.synchronize(); ms, _ = vp.render_time_ms()
print(f"zeros{n} rng={rng} bpc={os.environ.get('VP_BLOCKS_PER_CU','5')}: {550.6*800*600*32/ms/1e6:.1f} G steps/s")
