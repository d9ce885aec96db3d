"""Does a knob change the bits?  Renders a bench workload twice (child processes: knobs are read when the library starts), with and
without an environment setting, and lists the pixels whose accumulators differ.
   python scripts/knob_diff.py c4s 32 VP_NO_APPROACH_LOCAL=1"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
    import volpath as vp
    from volpath import scene
    wl, frames, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    vp.set_device(0)
    P, info = scene.setup(wl, rng_mode=int(os.environ.get("VP_PERF_RNG", vp.RNG_PHILOX7)), last_frame=frames)
    buf = vp.DeviceBuffer(P.width, P.height)
    vp.render_frames(buf.ptr, 0, frames, P); vp.synchronize()
    np.save(out, buf.download())
    sys.exit(0)
wl, frames, knob = sys.argv[1], sys.argv[2], sys.argv[3]
imgs = []
for i, extra in enumerate(({}, dict([knob.split("=")]))):
    out = f"/tmp/knob_diff_{i}.npy"
    subprocess.check_call([sys.executable, __file__, "--child", wl, frames, out], env={**os.environ, **extra})
    imgs.append(np.load(out))
a, b = imgs
d = np.argwhere((a != b).any(axis=2))
print(f"{wl} {frames} frames, {knob}: {len(d)} of {a.shape[0] * a.shape[1]} pixels differ")
for y, x in d[:12]:
    print(f"  pixel ({x},{y}): {a[y, x]} vs {b[y, x]}")
