"""Where the lane slots go: block tallies of the counting kernel variant (VP_DEBUG_COUNTERS=1).
   python scripts/block_profile.py c3 [FRAMES]"""
import os, sys
os.environ["VP_DEBUG_COUNTERS"] = "1"
# the tallies live in the PROFILING form of the counting kernels: cd cuda-volpath_amd && make dev DEVNAME=prof DEVFLAGS=-DVP_PROFILE_BLOCKS=1
_prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-volpath_amd", "libvolpath_hip_prof.so")
if "VOLPATH_LIB" not in os.environ:
    if not os.path.exists(_prof):
        sys.exit("block_profile.py needs " + _prof + ": cd cuda-volpath_amd && make dev DEVNAME=prof DEVFLAGS=-DVP_PROFILE_BLOCKS=1")
    os.environ["VOLPATH_LIB"] = _prof
os.environ.setdefault("VP_COUNT_APPROACH", "1")   # tallies of what the timed launch executes (approach_k ahead of the kernel)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import volpath as vp
from volpath import scene
wl = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
vp.set_device(0)
P, info = scene.setup(wl, rng_mode=vp.RNG_PHILOX, last_frame=frames)
buf = vp.DeviceBuffer(P.width, P.height)
vp.enable_counters(True); vp.read_counters()
vp.render_frames(buf.ptr, 0, frames, P)
c = vp.read_counters()
print(wl, {k: round(v / c["samples"], 2) for k, v in c.items()})
