"""debug: the sequence of tests/test_fuzz_gpu.py::test_call_sequences_with_batches_in_flight, checked after every segment"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import volpath as vp
from volpath import scene as vscene, host
workload, seed = sys.argv[1], int(sys.argv[2])
check_each = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nseg = int(sys.argv[4]) if len(sys.argv) > 4 else 28
vp.set_device(0)
rng = np.random.default_rng(4200 + seed)
P0, info = vscene.setup(workload, rng_mode=vp.RNG_PHILOX7 if seed != 1 else vp.RNG_SAMPLERH, last_frame=400)
W, H = P0.width, P0.height
bufs = [vp.DeviceBuffer(W, H), vp.DeviceBuffer(W, H)]
ref = vp.DeviceBuffer(W, H)
cams = [info["camera"]]
for a in rng.uniform(0, 2 * np.pi, 5):
    cams.append(tuple(float(v) for v in host.camera_matrix((3.9 * np.cos(a), -0.78, 3.9 * np.sin(a)), (-np.cos(a), 0.2, -np.sin(a)), (0.0, 1.0, 0.0))))
segs, cur, cam, density, frame = [], 0, 0, 100.0, 0
la = 256
vp.set_lookahead(la)
vp.synchronize()
for seg in range(nseg):
    u = rng.random()
    what = "-"
    if u < 0.45:
        cam = int(rng.integers(0, len(cams))); vp.set_camera(cams[cam]); frame = 0; what = f"camera {cam}"
    elif u < 0.60:
        density = float(np.float32(rng.uniform(60, 300))); what = "density"
    elif u < 0.70:
        la = int(rng.choice([0, 8, 64, 256])); vp.set_lookahead(la); what = f"lookahead {la}"
    elif u < 0.80:
        cur = 1 - cur; what = "buffer"
    elif u < 0.90:
        frame = int(rng.integers(0, 300)); what = "jump"
    P = vp.make_param(W, H, density=density)
    n = int(rng.choice([1, 2, 5, 12, 30, 45, 70]))
    sync = rng.random() < 0.7
    for f in range(frame, frame + n):
        vp.render_kernel(bufs[cur].ptr, f, P)
        if sync:
            vp.synchronize()
    segs.append((cur, cam, density, frame, n))
    frame += n
    if check_each:
        got = bufs[cur].download()
        vp.set_lookahead(0)
        ref.reset()
        for (b, cm, dn, f0, m) in segs:
            if b == cur:
                vp.set_camera(cams[cm]); vp.render_frames(ref.ptr, f0, m, vp.make_param(W, H, density=dn))
        ok = np.array_equal(got, ref.download(), equal_nan=True)
        vp.set_camera(cams[cam]); vp.set_lookahead(la)
        print(seg, what, segs[-1], "sync" if sync else "nosync", "OK" if ok else "MISMATCH", flush=True)
        if not ok:
            d = got != ref.download()
            print("  differing pixels", int(d.any(-1).sum()), "of", W * H)
            break
if not check_each:
    got = [b.download() for b in bufs]
    vp.set_lookahead(0)
    for i in range(2):
        ref.reset()
        for (b, cm, dn, f0, m) in segs:
            if b == i:
                vp.set_camera(cams[cm]); vp.render_frames(ref.ptr, f0, m, vp.make_param(W, H, density=dn))
        r = ref.download()
        print("buffer", i, "OK" if np.array_equal(got[i], r, equal_nan=True) else f"MISMATCH in {int((got[i] != r).any(-1).sum())} pixels")
    print(segs)
if not check_each and os.environ.get("VP_DEBUG_SEG"):
    # which segment's contribution is off?  replay buffer 0 leaving out one segment at a time... cheaper: render each segment alone
    i = 0
    ref.reset()
    for (b, cm, dn, f0, m) in segs:
        if b == i:
            vp.set_camera(cams[cm]); vp.render_frames(ref.ptr, f0, m, vp.make_param(W, H, density=dn))
    r = ref.download()
    bad = (got[i] != r).any(-1)
    ys, xs = np.nonzero(bad); print("bad pixels", int(bad.sum()), "e.g.", xs[:3], ys[:3])
    delta = got[i].astype(np.float64) - r.astype(np.float64)
    for k, (b, cm, dn, f0, m) in enumerate(segs):
        if b != i: continue
        for f in range(f0, f0 + m):
            ref.reset(); vp.set_camera(cams[cm]); vp.render_frames(ref.ptr, f, 1, vp.make_param(W, H, density=dn))
            one = ref.download().astype(np.float64)
            # does the error look like "this frame missing" or "this frame twice"?
            e_missing = np.abs(delta + one)[bad].max(); e_twice = np.abs(delta - one)[bad].max()
            if min(e_missing, e_twice) < 1e-3 * np.abs(one)[bad].max() + 1e-6:
                print("segment", k, segs[k], "frame", f, "missing" if e_missing < e_twice else "added twice")
