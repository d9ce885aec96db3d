"""Times one BASELINE workload: python scripts/prof_workload.py WORKLOAD RNG FRAMES"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import volpath as vp
from volpath import scene
wl, rng, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vp.set_device(0)
P, info = scene.setup(wl, rng_mode=rng, last_frame=frames)
buf = vp.DeviceBuffer(P.width, P.height)
vp.render_frames(buf.ptr, 0, 2, P); vp.synchronize(); vp.render_time_ms()
vp.enable_counters(True); vp.read_counters(); vp.render_frames(buf.ptr, 0, min(frames, 16), P); c = vp.read_counters(); vp.enable_counters(False)
buf.reset(); vp.render_time_ms()
vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
ms, n = vp.render_time_ms()
ns = P.width * P.height * frames
per = {k: v / max(c["samples"], 1) for k, v in c.items()}
bps = 8 * per["density_lookups"] + 2 * per["bound_lookups"] + 32 * per["opacity_lookups"] + 16 * per["env_lookups"] + 32
print(f"{info['name']} rng={rng}: {ns/ms/1e3:.1f} Msamples/s ({ms:.1f} ms, {n} launches); per sample: den {per['density_lookups']:.1f} bnd {per['bound_lookups']:.1f} "
      f"opa {per['opacity_lookups']:.2f} sca {per['scatters']:.2f}; {bps:.0f} B/sample -> {bps*ns/ms/1e6:.0f} GB/s algorithmic")
