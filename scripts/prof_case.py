"""One render configuration, for rocprofv3 runs: EST BRICK RNG FRAMES [N] from argv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import volpath as vp, scenes
est, brick, rng, frames = (int(a) for a in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 256
vp.set_device(0)
grid = vp.julia_volume(N)
vp.init_volume(grid, brick=brick); vp.init_envmap(scenes.synthetic_env(1024, 512))
vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER); vp.set_camera(); vp.set_estimator(est); vp.set_rng(rng, (1, 2))
if est == 1 and frames > 11: vp.precompute_opacity(scenes.DEFAULT_SUN_DIR)
P = vp.make_param(800, 600); buf = vp.DeviceBuffer(800, 600)
vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
ms, n = vp.render_time_ms(); print("render ms", ms, "Ms/s", 800 * 600 * frames / ms / 1e3)
