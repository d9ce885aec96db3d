"""ctypes binding of the TEST-ONLY CPU oracle (oracle/libvp_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

# OpenMP's own default is one thread per logical CPU of the HOST; a container that may use 8 or 16 of them spends its time
# switching between 200 threads.  The CPUs this process may run on, at most 16.
try:
    DEFAULT_THREADS = max(1, min(16, len(os.sched_getaffinity(0))))
except AttributeError:
    DEFAULT_THREADS = max(1, min(16, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

RNG_SAMPLERH, RNG_PHILOX, RNG_PHILOX7 = 0, 1, 2
EST_GLOBAL, EST_DECOMP, EST_BOUNDED = 0, 1, 2


class Param(C.Structure):
    """param.h:4-12"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("density", C.c_float), ("brightness", C.c_float),
                ("albedo", C.c_float * 3), ("g", C.c_float), ("sigma_t", C.c_float * 3)]


class Scene(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("grid_u8", C.c_void_p), ("grid_f32", C.c_void_p),
                ("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("linear", C.c_int),
                ("brick", C.c_int), ("bnx", C.c_int), ("bny", C.c_int), ("bnz", C.c_int),
                ("bounds_u8", C.c_void_p), ("bounds_f32", C.c_void_p),
                ("opacity", C.c_void_p),
                ("env", C.c_void_p), ("env_w", C.c_int), ("env_h", C.c_int),
                ("sun_dir", C.c_float * 3), ("sun_power", C.c_float * 3), ("sun_power_original", C.c_float * 3),
                ("inv_view", C.c_float * 12),
                ("estimator", C.c_int), ("rng_mode", C.c_int), ("seed", C.c_uint32 * 2),
                ("env_mis", C.c_int), ("env_cdf_y", C.c_void_p), ("env_cdf_x", C.c_void_p),
                ("env_pdfnorm_alt", C.c_float), ("track_mode", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "density_lookups", "bound_lookups", "opacity_lookups",
                                          "env_lookups", "scatters", "rng_draws", "control_segments")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libvp_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libvp_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.vpo_hash.restype = C.c_uint32
        L.vpo_hash.argtypes = [C.c_uint32]
        L.vpo_bound_radius.restype = C.c_int
        L.vpo_bound_radius.argtypes = [C.c_int, C.c_float]
        L.vpo_hg_eval.restype = C.c_float
        L.vpo_hg_eval.argtypes = [C.c_float, C.c_float]
        L.vpo_sample_density.restype = C.c_float
        L.vpo_sample_opacity.restype = C.c_float
        L.vpo_mat.argtypes = [C.c_void_p] + [C.c_float] * 6
        L.vpo_build_env_tables.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vpo_debug_mis_zero_pdf.restype = C.c_uint64
        L.vpo_debug_shadow_overflow.restype = C.c_uint64
        L.vpo_debug_set_what_if.argtypes = [C.c_int]
        L.vpo_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.vpo_gamma_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def julia(n):
    g = np.empty((n, n, n), np.uint8)  # [k][j][i], x fastest
    lib().vpo_julia_voxelize(C.c_int(n), _p(g))
    return g


def cloud(n, seed=1):
    g = np.empty((n, n, n), np.float32)  # [k][j][i], x fastest
    lib().vpo_cloud_voxelize(C.c_int(n), C.c_uint32(seed), _p(g))
    return g


def bound_radius(nx, search_radius=0.05):
    return lib().vpo_bound_radius(nx, search_radius)


def bounds(grid, radius, brick=1):
    nz, ny, nx = grid.shape
    b = lambda n: (n + brick - 1) // brick
    if grid.dtype == np.uint8:
        out = np.empty((b(nz), b(ny), b(nx), 2), np.uint8)
        lib().vpo_bounds_u8(_p(grid), nx, ny, nz, radius, brick, _p(out))
    else:
        out = np.empty((b(nz), b(ny), b(nx), 2), np.float32)
        lib().vpo_bounds_f32(_p(grid), nx, ny, nz, radius, brick, _p(out))
    return out


def default_param(width, height, density=800.0, g=0.877, brightness=1.0, albedo=(1, 1, 1), sigma_t=(1, 1, 1)):
    """host.cpp:1286-1292 with preset #13 (host.cpp:1308)"""
    P = Param()
    P.width, P.height, P.density, P.brightness, P.g = width, height, density, brightness, g
    P.albedo[:] = albedo
    P.sigma_t[:] = sigma_t
    return P


def mat(P, X, Y, Z, R, G, B):
    lib().vpo_mat(C.byref(P), X, Y, Z, R, G, B)
    return P


class OracleScene:
    """Keeps numpy buffers alive next to the C struct."""

    def __init__(self, grid, env, sun_dir, sun_power, box=None, brick=1, radius=None, linear=True,
                 estimator=EST_DECOMP, rng_mode=RNG_SAMPLERH, seed=(0, 0), inv_view=None, extra_dilate=None,
                 env_mis=False, track_mode=0):
        L = lib()
        self.grid = np.ascontiguousarray(grid)
        nz, ny, nx = self.grid.shape
        S = Scene()
        S.nx, S.ny, S.nz = nx, ny, nz
        if self.grid.dtype == np.uint8:
            S.grid_u8 = _p(self.grid).value
        else:
            self.grid = self.grid.astype(np.float32)
            S.grid_f32 = _p(self.grid).value
        if box is None:
            box = ((-1.0, -ny / nx, -nz / nx), (1.0, ny / nx, nz / nx))  # kernel.cu:373-378
        S.box_min[:] = box[0]
        S.box_max[:] = box[1]
        S.linear = int(linear)
        if radius is None:
            radius = bound_radius(nx)
            if brick > 1:
                radius += 1 if extra_dilate is None else extra_dilate  # trilinear support
        self.radius = radius
        self.bounds = bounds(self.grid, radius, brick)
        S.brick = brick
        S.bnz, S.bny, S.bnx = self.bounds.shape[:3]
        if self.bounds.dtype == np.uint8:
            S.bounds_u8 = _p(self.bounds).value
        else:
            S.bounds_f32 = _p(self.bounds).value
        self.env = np.ascontiguousarray(env, np.float32)
        S.env = _p(self.env).value
        S.env_h, S.env_w = self.env.shape[:2]
        d = (C.c_float * 3)(*sun_dir)
        p = (C.c_float * 3)(*sun_power)
        L.vpo_set_sun(C.byref(S), d, p)
        if inv_view is None:
            L.vpo_default_camera(S.inv_view)
        else:
            S.inv_view[:] = list(np.asarray(inv_view, np.float32).ravel())
        S.estimator, S.rng_mode = estimator, rng_mode
        S.track_mode = track_mode
        S.seed[:] = seed
        self.S = S
        self.opacity = None
        if env_mis:
            self.enable_env_mis()

    def enable_env_mis(self):
        """!PASSIVE_ENVMAP: luminance CDF tables (kernel.cu:1144-1210) + one-sample MIS in the integrator"""
        S = self.S
        self.cdf_y = np.empty(S.env_h, np.float32)
        self.cdf_x = np.empty((S.env_h, S.env_w), np.float32)
        norm = C.c_float()
        lib().vpo_build_env_tables(_p(self.env), S.env_w, S.env_h, _p(self.cdf_y), _p(self.cdf_x), C.byref(norm))
        S.env_mis, S.env_cdf_y, S.env_cdf_x = 1, _p(self.cdf_y).value, _p(self.cdf_x).value
        S.env_pdfnorm_alt = norm.value
        self.pdfnorm_alt = norm.value

    def precompute_opacity(self, threads=0):
        threads = threads or DEFAULT_THREADS
        S = self.S
        self.opacity = np.empty((S.nz, S.ny, S.nx), np.float32)
        lib().vpo_precompute_opacity(C.byref(S), S.sun_dir, _p(self.opacity), threads)
        S.opacity = _p(self.opacity).value
        return self.opacity

    def opacity_voxels(self, ijk, light_dir=None, threads=0):
        """single voxels of the optical-depth table (the march of kernel.cu:497-523 per voxel), ijk = (n, 3) ints (i, j, k);
        light_dir: the direction handed to precompute_opacity (default: the scene's sun, as host.cpp:341 does)"""
        threads = threads or DEFAULT_THREADS
        ijk = np.ascontiguousarray(ijk, np.int32)
        out = np.empty(len(ijk), np.float32)
        d = self.S.sun_dir if light_dir is None else (C.c_float * 3)(*[float(v) for v in light_dir])
        lib().vpo_opacity_voxels(C.byref(self.S), d, _p(ijk), len(ijk), _p(out), threads)
        return out

    def render_frame(self, P, frame, accum=None, rows=None, threads=0):
        threads = threads or DEFAULT_THREADS
        if accum is None:
            accum = np.zeros((P.height, P.width, 4), np.float32)
        y0, y1 = rows if rows else (0, P.height)
        cnt = Counters()
        lib().vpo_render_frame(C.byref(self.S), C.byref(P), frame, _p(accum), y0, y1, threads, C.byref(cnt))
        return accum, cnt

    def render_sample(self, P, x, y, frame):
        out = (C.c_float * 4)()
        cnt = Counters()
        lib().vpo_render_sample(C.byref(self.S), C.byref(P), x, y, frame, out, C.byref(cnt))
        return np.array(out[:], np.float32), cnt


def rng_stream(mode, x, y, frame, n, key=(0, 0)):
    out = np.empty(n, np.float32)
    lib().vpo_rng_stream(mode, x, y, frame, key[0], key[1], n, _p(out))
    return out


def philox(ctr, key, rounds=10):
    """Philox2x32-10 / -7: ctr = (c0, c1), key = one word -> two words"""
    c = (C.c_uint32 * 2)(*ctr)
    o = (C.c_uint32 * 2)()
    (lib().vpo_philox2x32_10 if rounds == 10 else lib().vpo_philox2x32_7)(c, C.c_uint32(key), o)
    return list(o)


def math_array(which, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().vpo_math_array(which, _p(x), _p(out), x.size)
    return out


# SURVEY.md section 4 anchors (default sun x=.5, y=.2)
DEFAULT_SUN_DIR = (-0.0, 0.951057, -0.309017)
DEFAULT_SUN_POWER = (51797.34, 42480.11, 32578.49)
