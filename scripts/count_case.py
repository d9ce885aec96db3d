import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import volpath as vp, scenes
est, brick, rng, frames = (int(a) for a in sys.argv[1:5])
vp.set_device(0)
grid = vp.julia_volume(256)
vp.init_volume(grid, brick=brick); vp.init_envmap(scenes.synthetic_env(1024, 512))
vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER); vp.set_camera(); vp.set_estimator(est); vp.set_rng(rng, (1, 2))
P = vp.make_param(800, 600); buf = vp.DeviceBuffer(800, 600)
vp.enable_counters(True); vp.read_counters()
vp.render_frames(buf.ptr, 0, frames, P)
c = vp.read_counters(); print({k: v / c["samples"] for k, v in c.items()})
