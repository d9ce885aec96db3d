"""wall time of the C-ABI entry points on a tiny scene (development tool): python scripts/time_api_ops.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'cuda-volpath_amd'))
import numpy as np, volpath as vp, scenes, oracle_lib as oracle
vp.set_device(0)
grid = oracle.julia(16)
vp.init_volume(grid, brick=1); vp.init_envmap(scenes.synthetic_env()); vp.set_sun(scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER); vp.set_camera(); vp.set_estimator(1); vp.set_rng(1,(1,2)); vp.set_shard(0,1)
def T(name, fn, n=1):
    vp.synchronize(); t=time.time()
    for _ in range(n): fn()
    vp.synchronize(); print(f"{name}: {(time.time()-t)/n*1e3:.2f} ms")
T("precompute_opacity", lambda: vp.precompute_opacity(scenes.DEFAULT_SUN_DIR))
W,H=40,24
P=vp.make_param(W,H,density=200.0,g=0.6)
b=vp.DeviceBuffer(W,H)
fr=[0]
def rk():
    fr[0]+=1; vp.render_kernel(b.ptr, fr[0], P)
T("render_kernel consecutive x50", rk, 50)
def jump():
    fr[0]+=1000; vp.render_kernel(b.ptr, fr[0], P)
T("render_kernel jump x10", jump, 10)
T("render_frames 3", lambda: vp.render_frames(b.ptr, 5, 3, P), 5)
T("set_camera+render", lambda: (vp.set_camera(), rk()), 5)
T("set_estimator+render", lambda: (vp.set_estimator(0), rk(), vp.set_estimator(1)), 3)
T("init_envmap", lambda: vp.init_envmap(scenes.synthetic_env(seed=5)), 3)
T("download", lambda: b.download(), 5)
T("set_lookahead", lambda: vp.set_lookahead(8), 3)
