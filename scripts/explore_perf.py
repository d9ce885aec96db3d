"""Exploratory timing of the render kernel at BASELINE sizes (not the bench; prints raw numbers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp, scenes

vp.set_device(0)
N = int(os.environ.get("N", 256)); W, H = 800, 600
t = time.time(); grid = vp.julia_volume(N); print("julia", time.time() - t, grid.mean() / 255, flush=True)
env = scenes.synthetic_env(1024, 512)
CASES = eval(os.environ.get('CASES', '[(0,1,0,64),(0,1,1,64),(1,1,0,64),(1,8,0,64),(1,8,1,64)]'))
for est, brick, rng, frames in CASES:
    t = time.time(); vp.init_volume(grid, brick=brick); print("init_volume", time.time() - t, flush=True)
    vp.init_envmap(env); vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER); vp.set_camera()
    vp.set_estimator(est); vp.set_rng(rng, (1, 2))
    P = vp.make_param(W, H)
    buf = vp.DeviceBuffer(W, H)
    vp.render_frames(buf.ptr, 0, 1, P); vp.synchronize(); vp.render_time_ms()
    vp.enable_counters(True); vp.read_counters()
    vp.render_frames(buf.ptr, 0, 2, P); c = vp.read_counters(); vp.enable_counters(False); vp.render_time_ms()
    t = time.time(); vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize(); dt = time.time() - t
    ms, n = vp.render_time_ms()
    ns = W * H * frames
    per = {k: v / c["samples"] for k, v in c.items()}
    print(f"est={est} brick={brick} rng={rng}: {ns/dt/1e6:.1f} Msamples/s wall, kernel {ms:.1f} ms/{n} launches -> {ns/ms/1e3:.1f} Ms/s;"
          f" lookups/sample den={per['density_lookups']:.1f} bnd={per['bound_lookups']:.1f} sca={per['scatters']:.2f}"
          f" -> {per['density_lookups']*ns/ms/1e6:.1f} G lookups/s", flush=True)
    buf.free()
