"""Kernel-time throughput of bench workloads, one launch each (development tool; the bench is bench.py):
   python scripts/perf_workloads.py c2,c3,c3ref,c4s [FRAMES] [REPEAT]
Prints Msamples/s by HIP-event kernel time and a hash of the accumulator (identical bits across kernel variants)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import volpath as vp
from volpath import scene
wls = sys.argv[1].split(",")
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
vp.set_device(0)
sky = scene.default_sunsky()
for wl in wls:
    P, info = scene.setup(wl, rng_mode=int(os.environ.get("VP_PERF_RNG", vp.RNG_PHILOX)), last_frame=frames, sunsky=sky)
    buf = vp.DeviceBuffer(P.width, P.height)
    vp.render_frames(buf.ptr, 0, 2, P); vp.synchronize(); vp.render_time_ms()
    best = 0
    for r in range(rep):
        buf.reset()
        vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
        ms, n = vp.render_time_ms()
        best = max(best, P.width * P.height * frames / ms / 1e3)
    h = hashlib.sha1(buf.download().tobytes()).hexdigest()[:12]
    print(f"{wl:6s} {frames} frames: {best:8.1f} Msamples/s  ({ms:.1f} ms, {n} launches)  image {h}", flush=True)
    buf.free()
