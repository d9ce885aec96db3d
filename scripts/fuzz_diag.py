"""diagnose a failing seed of tests/test_fuzz_gpu.py: python scripts/fuzz_diag.py SEED [key=value ...] (est= rng_mode= linear= brick= cubic=1 u8=1 nframes=)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import numpy as np
import oracle_lib as oracle
import volpath as vp
from volpath import host
import test_fuzz_gpu as T
seed = int(sys.argv[1])
ov = dict(a.split("=") for a in sys.argv[2:])
c = T._case(seed, host)
for k in ("est", "rng_mode", "brick", "nframes", "first", "W", "H"):
    if k in ov: c[k] = int(ov[k])
if "linear" in ov: c["linear"] = bool(int(ov["linear"]))
if "u8" in ov: c["grid"] = np.ascontiguousarray((np.clip(c["grid"], 0, 1) * 255).astype(np.uint8)) if c["grid"].dtype != np.uint8 else c["grid"]
if "cubic" in ov:
    n = min(c["grid"].shape); c["grid"] = np.ascontiguousarray(c["grid"][:n, :n, :n])
if "density" in ov: c["kw"]["density"] = float(ov["density"])
if "achromatic" in ov: c["kw"].pop("sigma_t", None); c["kw"].pop("albedo", None)
if "defcam" in ov: c["cam"] = np.array(vp.DEFAULT_CAMERA, np.float32)
if "defbox" in ov: c["box"] = None
c["late"] = c["late"] and c["est"] == 1
c["env_mis"] = bool(int(ov.get("env_mis", 0))); c["track"] = int(ov.get("track", 0))
ref, cnt = T._oracle_render(oracle, c)
vp.set_device(0)
vP = vp.make_param(c["W"], c["H"], **c["kw"])
vp.init_volume(c["grid"], box=c["box"], brick=c["brick"], linear=c["linear"]); vp.init_envmap(c["env"]); vp.set_sun(c["sun_dir"], c["sun_power"])
vp.set_tracking(c["track"]); vp.set_envmap_sampling(1 if c["env_mis"] else 0); vp.set_camera(c["cam"]); vp.set_estimator(c["est"]); vp.set_rng(c["rng_mode"], c["key"]); vp.set_shard(0, 1)
if c["late"]: vp.precompute_opacity(c["sun_dir"])
buf = vp.DeviceBuffer(c["W"], c["H"])
vp.render_frames(buf.ptr, c["first"], c["nframes"], vP)
got = buf.download()
bad = np.argwhere(np.any(got != ref, axis=-1))
print(f"seed {seed} {ov}: grid {c['grid'].shape} {c['grid'].dtype} est {c['est']} rng {c['rng_mode']} linear {c['linear']} brick {c['brick']} box {c['box'] is not None} "
      f"{c['W']}x{c['H']} frames {c['first']}+{c['nframes']}: {len(bad)} differing of {c['W'] * c['H']}, max abs {np.abs(got - ref).max():.3g}; sca/smp {cnt['scatters'] / cnt['samples']:.2f}")
for y, x in bad[:4]:
    print("   pixel", (x, y), "got", got[y, x], "ref", ref[y, x])
# component check: the density fetch at random points of the box (and a little outside)
osc = oracle.OracleScene(c["grid"], c["env"], c["sun_dir"], c["sun_power"], box=c["box"], brick=c["brick"], linear=c["linear"], estimator=c["est"], rng_mode=c["rng_mode"], seed=c["key"], inv_view=c["cam"])
import ctypes as C
rs = np.random.default_rng(1)
nz, ny, nx = c["grid"].shape
bmin = np.array(c["box"][0] if c["box"] else (-1.0, -ny / nx, -nz / nx)); bmax = np.array(c["box"][1] if c["box"] else (1.0, ny / nx, nz / nx))
pts = (bmin + (bmax - bmin) * rs.uniform(-0.05, 1.05, (200000, 3))).astype(np.float32)
g = vp.test_sample_density(pts)
L = oracle.lib()
o = np.array([L.vpo_sample_density(C.byref(osc.S), (C.c_float * 3)(*p)) for p in pts[:20000]], np.float32)
d = np.flatnonzero(g[:20000] != o)
print(f"   density fetch: {len(d)} of 20000 points differ" + (f"; first: p {pts[d[0]]} gpu {g[d[0]]!r} oracle {o[d[0]]!r}" if len(d) else ""))
