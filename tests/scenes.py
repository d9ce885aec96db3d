"""Small deterministic scenes shared by the parity tests (inputs only; no oracle, no product)."""
import numpy as np

DEFAULT_SUN_DIR = (-0.0, 0.951057, -0.309017)          # SURVEY.md section 4 anchor, setup_sunsky(0.5, 0.2)
DEFAULT_SUN_POWER = (51797.34, 42480.11, 32578.49)     # sunColor * 0.02
PRESET1 = (2.29, 2.39, 1.97, 0.0030, 0.0034, 0.046)    # host.cpp:1296


def synthetic_env(w=64, h=32, seed=7):
    """A smooth-ish positive lat-long map; values of the order of the baked sky."""
    rng = np.random.default_rng(seed)
    env = np.zeros((h, w, 4), np.float32)
    yy = np.linspace(0, 1, h, dtype=np.float32)[:, None]
    base = np.stack([0.09 + 0.3 * (1 - yy), 0.12 + 0.3 * (1 - yy), 0.2 + 0.4 * (1 - yy)], -1)
    env[..., :3] = base + 0.05 * rng.random((h, w, 3), dtype=np.float32)
    env[..., 3] = 1.0
    return env


def blob_volume_f32(n=24, seed=3):
    """A float density volume in [0,1] with empty regions (for quantized=false paths)."""
    rng = np.random.default_rng(seed)
    z, y, x = np.mgrid[0:n, 0:n, 0:n].astype(np.float32)
    c = (n - 1) / 2
    r = np.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2) / c
    v = np.clip(1.2 - r * 1.6, 0, 1) * (0.6 + 0.4 * rng.random((n, n, n), dtype=np.float32))
    v[r > 0.8] = 0
    return v.astype(np.float32)


def blob_volume_u8(n=24, seed=3):
    return (blob_volume_f32(n, seed) * 255.0).astype(np.uint8)
