"""Is the tracking step limited by memory or by instruction issue?  Same estimator on an all-zero grid of
different sizes (every fetch an L1/L2 hit for the small one): steps/s should match if issue-bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp, scenes
vp.set_device(0)
rng = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name, grid in (("julia256", None), ("zeros256", np.zeros((256, 256, 256), np.uint8)), ("zeros16", np.zeros((16, 16, 16), np.uint8))):
    if grid is None: grid = vp.julia_volume(256)
    vp.init_volume(grid); vp.init_envmap(scenes.synthetic_env(1024, 512)); vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    vp.set_camera(); vp.set_estimator(0); vp.set_rng(rng, (1, 2))
    P = vp.make_param(800, 600); buf = vp.DeviceBuffer(800, 600)
    vp.enable_counters(True); vp.read_counters(); vp.render_frames(buf.ptr, 0, 4, P); c = vp.read_counters(); vp.enable_counters(False)
    vp.render_time_ms(); vp.render_frames(buf.ptr, 0, 32, P); vp.synchronize(); ms, _ = vp.render_time_ms()
    L = c["density_lookups"] / c["samples"]
    print(f"{name}: {800*600*32/ms/1e3:.1f} Msamples/s, {L:.1f} lookups/sample -> {L*800*600*32/ms/1e6:.1f} G steps/s")
    buf.free()
