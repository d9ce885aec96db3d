"""C2 with the chromatic medium of preset #1 (development: block split of the global-majorant kernels for a non-achromatic medium)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import volpath as vp
from volpath import scene
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vp.set_device(0)
P, info = scene.setup("c2", rng_mode=vp.RNG_PHILOX7, last_frame=frames)
vp.mat(P, *scene.PRESET1)
buf = vp.DeviceBuffer(P.width, P.height)
vp.render_frames(buf.ptr, 0, 2, P); vp.synchronize(); vp.render_time_ms()
best = 0
for r in range(2):
    buf.reset(); vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
    ms, n = vp.render_time_ms(); best = max(best, P.width * P.height * frames / ms / 1e3)
print(f"c2 chromatic {frames} frames: {best:.1f} Msamples/s  image {hashlib.sha1(buf.download().tobytes()).hexdigest()[:12]}")
