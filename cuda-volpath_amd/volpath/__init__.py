"""ctypes binding of libvolpath_hip.so (include/volpath.h).

This is the host-side mirror of the reference's kernel-TU interface (src/volumeRender.cpp:117-128,
:347-356): the same entry points under the same names, called the way the reference host calls
them.  Everything here runs on the GPU through the C ABI; there is no CPU fallback -- if the
library or a gfx950 device is missing the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOLPATH_LIB", os.path.join(os.path.dirname(_HERE), "libvolpath_hip.so"))  # override: A/B builds

EST_GLOBAL, EST_DECOMP, EST_BOUNDED = 0, 1, 2
RNG_SAMPLERH, RNG_PHILOX, RNG_PHILOX7 = 0, 1, 2
ENV_PASSIVE, ENV_MIS = 0, 1
TRACK_SPECTRAL, TRACK_SCALAR, TRACK_MULTI_CHANNEL = 0, 1, 2

# every symbol include/volpath.h declares (tests check the library exports each one)
PART1_SYMBOLS = ["init_cuda", "set_texture_filter_mode", "free_cuda_buffers", "precompute_opacity", "init_envmap",
                 "free_envmap", "set_sun", "copy_inv_view_matrix", "copy_inv_model_matrix", "init_rng", "free_rng",
                 "render_kernel", "scale", "gamma_correct"]
PART2_SYMBOLS = ["vp_last_error", "vp_version", "vp_device_count", "vp_set_device", "vp_set_stream", "vp_get_stream", "vp_synchronize",
                 "vp_set_estimator", "vp_set_rng", "vp_set_envmap_sampling", "vp_get_env_tables", "vp_set_lookahead", "vp_set_tracking", "vp_set_bound_brick", "vp_set_shard", "vp_render_frames",
                 "vp_enable_counters", "vp_read_counters", "vp_render_time_ms", "vp_get_bound_table", "vp_get_opacity", "vp_get_pixel_table", "vp_get_null_collision_table", "vp_get_sun_clip_table", "vp_get_exit_table", "vp_set_exit_flights", "vp_render_class_time_ms", "vp_last_approach_mode", "vp_last_approach_table", "vp_last_light_const", "vp_last_lds_form", "vp_lookahead_stats", "vp_prepare", "vp_reserve_frames", "vp_get_pixel_lists",
                 "vp_julia_voxelize", "vp_cloud_voxelize", "vp_test_math", "vp_test_rng", "vp_test_sample_density", "vp_test_hg", "vp_test_intersect_box",
                 "vp_test_eval_envmap", "vp_ctx_create", "vp_ctx_destroy", "vp_ctx_set_current", "vp_ctx_get_current", "vp_ctx_device",
                 "vp_accumulate", "vp_tile_owner", "vp_malloc", "vp_free", "vp_memset",
                 "vp_upload", "vp_download"]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class Extent(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t)]


class Param(C.Structure):
    """src/param.h:4-12"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("density", C.c_float), ("brightness", C.c_float),
                ("albedo", Float3), ("g", C.c_float), ("sigma_t", Float3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "density_lookups", "density_loads", "bound_lookups",
                                          "opacity_lookups", "env_lookups", "scatters", "rng_draws")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class VolpathError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (no GPU is touched until the first call that needs one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VolpathError(f"{LIB_PATH} is missing: run `make -C cuda-volpath_amd` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.vp_last_error.restype = C.c_char_p
        L.vp_version.restype = C.c_char_p
        L.vp_malloc.restype = C.c_void_p
        L.vp_malloc.argtypes = [C.c_size_t]
        L.vp_free.argtypes = [C.c_void_p]
        L.vp_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.vp_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vp_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vp_set_stream.argtypes = [C.c_void_p]
        L.vp_get_stream.restype = C.c_void_p
        L.vp_set_rng.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.vp_get_env_tables.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.vp_render_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Param)]
        L.vp_read_counters.argtypes = [C.POINTER(Counters), C.c_int]
        L.vp_render_time_ms.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]
        L.vp_get_bound_table.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 5
        L.vp_get_opacity.argtypes = [C.c_void_p, C.c_size_t]
        L.vp_get_pixel_table.argtypes = [C.POINTER(Param), C.c_void_p, C.c_size_t]
        L.vp_get_null_collision_table.argtypes = [C.POINTER(Param), C.c_void_p, C.c_size_t]
        L.vp_get_sun_clip_table.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
        L.vp_render_class_time_ms.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]
        L.vp_prepare.argtypes = [C.POINTER(Param)]
        L.vp_get_pixel_lists.argtypes = [C.POINTER(Param), C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.vp_julia_voxelize.argtypes = [C.c_int, C.c_void_p]
        L.vp_cloud_voxelize.argtypes = [C.c_int, C.c_uint32, C.c_void_p]
        L.vp_test_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.vp_test_rng.argtypes = [C.c_int] + [C.c_uint32] * 5 + [C.c_int, C.c_void_p]
        L.vp_test_sample_density.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vp_test_hg.argtypes = [C.c_void_p] * 7 + [C.c_int]
        L.vp_test_intersect_box.argtypes = [C.c_void_p] * 5 + [C.c_int]
        L.vp_test_eval_envmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vp_ctx_create.restype = C.c_void_p
        L.vp_ctx_create.argtypes = [C.c_int]
        L.vp_ctx_destroy.argtypes = [C.c_void_p]
        L.vp_ctx_set_current.argtypes = [C.c_void_p]
        L.vp_ctx_get_current.restype = C.c_void_p
        L.vp_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vp_tile_owner.argtypes = [C.c_uint, C.c_uint, C.c_int]
        L.init_cuda.argtypes = [C.c_void_p, Extent, C.c_bool, C.POINTER(Float3), C.POINTER(Float3)]
        L.init_cuda.restype = None
        L.set_texture_filter_mode.argtypes = [C.c_bool]
        L.precompute_opacity.argtypes = [C.POINTER(C.c_float)]
        L.init_envmap.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.set_sun.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.copy_inv_view_matrix.argtypes = [C.POINTER(C.c_float), C.c_size_t]
        L.copy_inv_model_matrix.argtypes = [C.POINTER(C.c_float), C.c_size_t]
        L.init_rng.argtypes = [Dim3, Dim3, C.c_int, C.c_int]
        L.render_kernel.argtypes = [Dim3, Dim3, C.c_void_p, C.c_int, C.POINTER(Param)]
        L.scale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.gamma_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        for name in ("set_texture_filter_mode", "free_cuda_buffers", "precompute_opacity", "init_envmap",
                     "free_envmap", "set_sun", "copy_inv_view_matrix", "copy_inv_model_matrix", "init_rng",
                     "free_rng", "render_kernel", "scale", "gamma_correct"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise VolpathError(f"volpath error {rc}: {lib().vp_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().vp_device_count()


def make_param(width, height, density=800.0, g=0.877, brightness=1.0, albedo=(1, 1, 1), sigma_t=(1, 1, 1)):
    """host.cpp:1286-1292 defaults with preset #13 (host.cpp:1308)."""
    P = Param()
    P.width, P.height, P.density, P.brightness, P.g = width, height, density, brightness, g
    P.albedo = Float3(*albedo)
    P.sigma_t = Float3(*sigma_t)
    return P


def mat(P, X, Y, Z, R, G, B):
    """Mat(), host.cpp:44-57: sigma_s, sigma_a -> sigma_t normalised by its max, albedo = sigma_s/sigma_t."""
    f = np.float32
    st = [f(X) + f(R), f(Y) + f(G), f(Z) + f(B)]
    al = [f(X) / st[0], f(Y) / st[1], f(Z) / st[2]]
    m = max(st)
    st = [s / m for s in st]
    P.albedo = Float3(*[float(a) for a in al])
    P.sigma_t = Float3(*[float(s) for s in st])
    return P


# H4: the default camera of the reference (host.cpp:108-115 through lookAt/inverse/transpose, :617-623)
DEFAULT_CAMERA = (0.0, 0.207912, 0.978148, 3.922986, 0.0, 0.978148, -0.207912, -0.782739, -1.0, 0.0, 0.0, 0.03)


class DeviceBuffer:
    """A caller-owned float4 accumulator in HBM (CudaFrameBuffer, host.cpp:358-389)."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.nbytes = width * height * 16
        self.ptr = lib().vp_malloc(self.nbytes)
        if not self.ptr:
            raise VolpathError(lib().vp_last_error().decode())
        self.reset()

    def reset(self):
        _chk(lib().vp_memset(self.ptr, 0, self.nbytes))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        assert arr.nbytes == self.nbytes
        _chk(lib().vp_upload(self.ptr, _p(arr), self.nbytes))

    def download(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _chk(lib().vp_download(_p(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().vp_free(self.ptr)
            self.ptr = None


def set_device(i):
    _chk(lib().vp_set_device(i))


def set_stream(stream_ptr):
    _chk(lib().vp_set_stream(stream_ptr))


def synchronize():
    _chk(lib().vp_synchronize())


def init_volume(grid, box=None, brick=1, linear=True):
    """init_cuda + set_texture_filter_mode as host.cpp:1336-1344 calls them. grid[k][j][i], uint8 or float32."""
    L = lib()
    grid = np.ascontiguousarray(grid)
    quantized = grid.dtype == np.uint8
    if not quantized:
        grid = np.ascontiguousarray(grid, np.float32)
    nz, ny, nx = grid.shape
    _chk(L.vp_set_bound_brick(brick))
    ext = Extent(nx, ny, nz)
    if box is None:
        L.init_cuda(_p(grid), ext, quantized, None, None)
    else:
        L.init_cuda(_p(grid), ext, quantized, C.byref(Float3(*box[0])), C.byref(Float3(*box[1])))
    L.set_texture_filter_mode(bool(linear))


def init_envmap(env):
    env = np.ascontiguousarray(env, np.float32)
    h, w = env.shape[:2]
    lib().init_envmap(_p(env), w, h)


def set_sun(direction, power):
    d = (C.c_float * 3)(*direction)
    p = (C.c_float * 3)(*power)
    lib().set_sun(d, p)


def set_camera(m=DEFAULT_CAMERA):
    a = (C.c_float * 12)(*m)
    lib().copy_inv_view_matrix(a, 48)


def precompute_opacity(direction):
    d = (C.c_float * 3)(*direction)
    lib().precompute_opacity(d)


def set_estimator(est):
    _chk(lib().vp_set_estimator(est))


def set_rng(mode, key=(0, 0)):
    _chk(lib().vp_set_rng(mode, key[0], key[1]))


def set_tracking(mode):
    """TRACK_SPECTRAL (shipped) / TRACK_SCALAR (SPECTRAL_TRACKING 0) / TRACK_MULTI_CHANNEL (MULTI_CHANNEL 1)"""
    _chk(lib().vp_set_tracking(mode))


LOOKAHEAD_DEFAULT = 256


def set_lookahead(max_frames=LOOKAHEAD_DEFAULT):
    """frames render_kernel may render ahead per launch (0/1: one launch per call; default 256)"""
    _chk(lib().vp_set_lookahead(max_frames))


def set_envmap_sampling(mode):
    """ENV_PASSIVE (the reference's shipped build) or ENV_MIS (its !PASSIVE_ENVMAP alternative)"""
    _chk(lib().vp_set_envmap_sampling(mode))


def env_tables(width, height):
    cdf_y = np.empty(height, np.float32)
    cdf_x = np.empty((height, width), np.float32)
    norm = C.c_float()
    _chk(lib().vp_get_env_tables(_p(cdf_y), _p(cdf_x), C.byref(norm)))
    return cdf_y, cdf_x, norm.value


def set_shard(rank, world):
    _chk(lib().vp_set_shard(rank, world))


def render_kernel(buf_ptr, spp, P):
    """The reference's per-frame call (host.cpp:631): one sample per pixel of frame `spp`."""
    g = Dim3((P.width + 7) // 8, (P.height + 7) // 8, 1)
    b = Dim3(8, 8, 1)
    lib().render_kernel(g, b, buf_ptr, spp, C.byref(P))


def render_frames(buf_ptr, first, n, P):
    _chk(lib().vp_render_frames(buf_ptr, first, n, C.byref(P)))


def enable_counters(on=True):
    _chk(lib().vp_enable_counters(int(on)))


def read_counters(reset=True):
    c = Counters()
    _chk(lib().vp_read_counters(C.byref(c), int(reset)))
    return c.as_dict()


def render_time_ms(reset=True):
    t = C.c_double()
    n = C.c_int()
    _chk(lib().vp_render_time_ms(C.byref(t), C.byref(n), int(reset)))
    return t.value, n.value


def render_class_time_ms(reset=True):
    """({"general": ms, "light": ms, "misses_box": ms}, {class: pixels}): kernel time per pixel class since the last reset"""
    ms = (C.c_double * 3)()
    px = (C.c_uint32 * 3)()
    _chk(lib().vp_render_class_time_ms(ms, px, int(reset)))
    names = ("general", "light", "misses_box")
    return dict(zip(names, ms[:])), dict(zip(names, px[:]))


def last_approach_mode():
    """0 / 1 / 2: how the last render launch took its general pixels' camera rays to the medium (vp_last_approach_mode)"""
    return int(lib().vp_last_approach_mode())


def last_approach_table():
    """1 if that walk read the per-view table of restart segments (vp_last_approach_table)"""
    return int(lib().vp_last_approach_table())


def reserve_frames(P, nframes):
    """size the sample staging for a coming render_frames job of nframes now (vp_reserve_frames)"""
    _chk(lib().vp_reserve_frames(C.byref(P), int(nframes)))


def lookahead_stats():
    """(look-ahead batches launched, batches told to stop while still running) of the current context (vp_lookahead_stats)"""
    a, b = C.c_uint(0), C.c_uint(0)
    _chk(lib().vp_lookahead_stats(C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def last_lds_form():
    """0 / 1 / 2: how the last launch of the decomposition estimator read its brick table (vp_last_lds_form)"""
    return int(lib().vp_last_lds_form())


def last_light_const():
    """True if the last render call wrote its light pixel class as per-pixel constants (vp_last_light_const)"""
    return bool(lib().vp_last_light_const())


def prepare(P):
    """build the per-camera tables, pixel lists and sun table of the current state now (vp_prepare)"""
    _chk(lib().vp_prepare(C.byref(P)))


def pixel_lists(P):
    """(general, light, misses_box): uint32 arrays of y << 16 | x, the pixel lists of the current shard / camera (test hook)"""
    cnt = (C.c_uint32 * 3)()
    _chk(lib().vp_get_pixel_lists(C.byref(P), None, 0, cnt))
    n = sum(cnt[:])
    out = np.empty(max(n, 1), np.uint32)
    _chk(lib().vp_get_pixel_lists(C.byref(P), _p(out), n, cnt))
    a, b = cnt[0], cnt[0] + cnt[1]
    return out[:a], out[a:b], out[b:n]


def bound_table(quantized=True):
    bnx, bny, bnz, brick, radius = (C.c_int() for _ in range(5))
    _chk(lib().vp_get_bound_table(None, 0, bnx, bny, bnz, brick, radius))
    out = np.empty((bnz.value, bny.value, bnx.value, 2), np.uint8 if quantized else np.float32)
    _chk(lib().vp_get_bound_table(_p(out), out.nbytes, bnx, bny, bnz, brick, radius))
    return out, brick.value, radius.value


def sun_clip_table(shape):
    """(uint16[nz, ny, nx], step): per cell, the distance in units of `step` beyond which a ray from anywhere in the cell toward
    the sun meets empty cells only; 0xffff = unknown (counter-based streams; include/volpath.h vp_get_sun_clip_table)"""
    out = np.empty(shape, np.uint16)
    step = C.c_float()
    _chk(lib().vp_get_sun_clip_table(_p(out), out.size, C.byref(step)))
    return out, step.value


def set_exit_flights(mode):
    """0 off, 1 global-majorant estimator only (default), 2 also the decomposition estimator (include/volpath.h)"""
    _chk(lib().vp_set_exit_flights(mode))


def exit_table(shape):
    """uint8[3, nz, ny, nx]: the direction table of the exit flights (include/volpath.h vp_get_exit_table)"""
    out = np.empty((3,) + tuple(shape), np.uint8)
    _chk(lib().vp_get_exit_table(_p(out), out.size))
    return out


def pixel_table(P):
    """(H, W, 8) float32: crawl end xyz, packed counts (view as uint32), certified-empty distance, pixel class (0 general,
    1 the whole chord is certified empty: light kernel, 2 the camera ray misses the box), 2 unused"""
    out = np.empty((P.height, P.width, 8), np.float32)
    _chk(lib().vp_get_pixel_table(C.byref(P), _p(out), out.size))
    return out


def null_collision_table(P, count):
    """float32[count]: throughput of an unscattered global-majorant path after n null collisions in empty space"""
    out = np.empty(count, np.float32)
    _chk(lib().vp_get_null_collision_table(C.byref(P), _p(out), out.size))
    return out


def opacity_table(shape):
    out = np.empty(shape, np.float32)
    _chk(lib().vp_get_opacity(_p(out), out.size))
    return out


def test_math(which, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    _chk(lib().vp_test_math(which, _p(x), _p(out), x.size))
    return out


def test_rng(mode, x, y, frame, n, key=(0, 0)):
    out = np.empty(n, np.float32)
    _chk(lib().vp_test_rng(mode, x, y, frame, key[0], key[1], n, _p(out)))
    return out


def test_sample_density(pos):
    pos = np.ascontiguousarray(pos, np.float32)
    out = np.empty(pos.shape[0], np.float32)
    _chk(lib().vp_test_sample_density(_p(pos), _p(out), pos.shape[0]))
    return out


def test_hg(g, r0, r1, normal, cos_query):
    """(direction after HGPhaseFunction::sample through Frame(normal), HGPhaseFunction::evaluate(cos_query)) on the device"""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    g, r0, r1, normal, cos_query = f(g), f(r0), f(r1), f(normal), f(cos_query)
    n = g.size
    d = np.empty((n, 3), np.float32)
    e = np.empty(n, np.float32)
    _chk(lib().vp_test_hg(_p(g), _p(r0), _p(r1), _p(normal), _p(cos_query), _p(d), _p(e), n))
    return d, e


def test_intersect_box(origin, direction):
    o = np.ascontiguousarray(origin, np.float32)
    d = np.ascontiguousarray(direction, np.float32)
    n = o.shape[0]
    hit = np.empty(n, np.int32)
    tn = np.empty(n, np.float32)
    tf = np.empty(n, np.float32)
    _chk(lib().vp_test_intersect_box(_p(o), _p(d), _p(hit), _p(tn), _p(tf), n))
    return hit.astype(bool), tn, tf


def test_eval_envmap(direction):
    d = np.ascontiguousarray(direction, np.float32)
    out = np.empty_like(d)
    _chk(lib().vp_test_eval_envmap(_p(d), _p(out), d.shape[0]))
    return out


class Context:
    """One scene on one GPU (include/volpath.h "Contexts").  `with ctx:` makes it the calling thread's current context."""

    def __init__(self, device=0):
        self.h = lib().vp_ctx_create(device)
        if not self.h:
            raise VolpathError(lib().vp_last_error().decode())
        self._prev = []

    def __enter__(self):
        self._prev.append(lib().vp_ctx_get_current())
        _chk(lib().vp_ctx_set_current(self.h))
        return self

    def __exit__(self, *exc):
        _chk(lib().vp_ctx_set_current(self._prev.pop()))

    def destroy(self):
        if self.h:
            _chk(lib().vp_ctx_destroy(self.h))
            self.h = None


def accumulate(dst_ptr, src_ptr, n_float4):
    _chk(lib().vp_accumulate(dst_ptr, src_ptr, n_float4))


def tile_owner(tx, ty, world):
    return lib().vp_tile_owner(tx, ty, world)


def julia_volume(n):
    """FractalJuliaSet (kernel.cu:84-140) voxelised on the GPU -> uint8 [k][j][i]."""
    out = np.empty((n, n, n), np.uint8)
    _chk(lib().vp_julia_voxelize(n, _p(out)))
    return out


def cloud_volume(n, seed=1):
    """the flagged synthetic cloud (vp_cloud_voxelize) voxelised on the GPU -> float32 [k][j][i] in [0,1]"""
    out = np.empty((n, n, n), np.float32)
    _chk(lib().vp_cloud_voxelize(n, seed, _p(out)))
    return out


def scale(dst_ptr, src_ptr, n, s):
    lib().scale(dst_ptr, src_ptr, n, s)


def gamma_correct(dst_ptr, src_ptr, n, s, gamma):
    lib().gamma_correct(dst_ptr, src_ptr, n, s, gamma)
