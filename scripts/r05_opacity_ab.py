"""Round 5: precompute_opacity with the density grid staged through LDS (opacity_lds_k) against the gather form (opacity_k):
wall time of the call (synchronised), the same table?   python scripts/r05_opacity_ab.py [N ...]      (through gpurun)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-volpath_amd"))
import numpy as np
import volpath as vp
from volpath import scene
vp.set_device(0)
sizes = [int(a) for a in sys.argv[1:]] or [256, 512]
for kind in ("julia", "cloud"):
    for n in sizes:
        grid = vp.julia_volume(n) if kind == "julia" else np.ascontiguousarray((np.clip(vp.cloud_volume(n, 1), 0, 1) * np.float32(255)).astype(np.uint8))
        for d in ((0.0, 0.951057, -0.309017), (0.6, -0.5, 0.62)):
            row = []
            for env in ("1", "0"):
                os.environ["VP_NO_OPACITY_LDS"] = env
                ctx = vp.Context(0)
                with ctx:
                    vp.init_volume(grid, brick=16 if n >= 512 else 8, linear=True)
                    vp.precompute_opacity(d); vp.synchronize()
                    best = 1e9
                    for r in range(3):
                        t0 = time.perf_counter(); vp.precompute_opacity(d); vp.synchronize(); best = min(best, time.perf_counter() - t0)
                    h = hashlib.sha1(vp.opacity_table((n, n, n)).tobytes()).hexdigest()[:12]
                ctx.destroy()
                row.append((best * 1e3, h))
            print(f"{kind} {n}^3 toward {d}: opacity_k {row[0][0]:8.2f} ms, opacity_lds_k {row[1][0]:8.2f} ms ({row[0][0] / row[1][0]:.2f}x)  tables {'identical' if row[0][1] == row[1][1] else 'DIFFER'} {row[1][1]}", flush=True)
        del grid
