"""ctypes binding of libvolpath_host.so: the C++ host side above the C ABI (sun/sky bake, image writers,
volume ingest, camera, material presets) -- the pieces of src/volumeRender.cpp, src/image.cpp, src/sunsky and
vdbloader that feed or drain the hot path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libvolpath_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C cuda-volpath_amd`")
        L = C.CDLL(LIB_PATH)
        L.vph_bake_sunsky.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vph_hosek.argtypes = [C.c_double] * 8 + [C.c_void_p]
        L.vph_sky_color.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
        L.vph_sun_color.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.vph_write_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.vph_image_ops.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.vph_load_binary.restype = C.c_void_p
        L.vph_load_binary.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.vph_load_raw.restype = C.c_void_p
        L.vph_load_raw.argtypes = [C.c_char_p, C.c_size_t]
        L.vph_load_vdb.restype = C.c_void_p
        L.vph_load_vdb.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.vph_dump_dense.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.vph_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
        L.vph_free.argtypes = [C.c_void_p]
        L.vph_camera_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.vph_material_preset.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bake_sunsky(x=0.5, y=0.2, width=1024, height=512):
    """update_sunsky(baked=true) for setup_sunsky(x, y) (host.cpp:268-333) -> (env[h,w,4], sun_dir[3], sun_power[3])"""
    env = np.empty((height, width, 4), np.float32)
    d = np.empty(3, np.float32)
    p = np.empty(3, np.float32)
    lib().vph_bake_sunsky(x, y, width, height, _p(env), _p(d), _p(p))
    return env, d, p


def hosek(elevation, theta, gamma, lam, intensity=100.0, kelvin=5777.0, turbidity=2.0, albedo=0.2):
    out = np.empty(2, np.float64)
    rc = lib().vph_hosek(elevation, intensity, kelvin, turbidity, albedo, theta, gamma, lam, _p(out))
    if rc:
        raise ValueError("no coefficients for this turbidity")
    return out[0], out[1]


def sun_color(theta, phi):
    rgb = np.empty(3, np.float32)
    d = np.empty(3, np.float32)
    lib().vph_sun_color(theta, phi, _p(rgb), _p(d))
    return rgb, d


def sky_color(theta, phi, direction, cel=False):
    rgb = np.empty(3, np.float32)
    dd = np.ascontiguousarray(direction, np.float32)
    lib().vph_sky_color(theta, phi, _p(dd), int(cel), _p(rgb))
    return rgb


def write_image(rgba, path, hdr=False, tonemap=0, gamma=2.2, scale=1.0):
    a = np.ascontiguousarray(rgba, np.float32)
    h, w = a.shape[:2]
    lib().vph_write_image(_p(a), w, h, path.encode(), int(hdr), tonemap, gamma, scale)


def image_op(rgba, op, arg=0.0):
    a = np.ascontiguousarray(rgba, np.float32).copy()
    h, w = a.shape[:2]
    if lib().vph_image_ops(_p(a), w, h, {"scale": 0, "flip": 1, "gamma": 2, "reinhard": 3}[op], arg):
        raise ValueError(op)
    return a


def load_binary(path, quantized=True):
    w, h, d = C.c_int(), C.c_int(), C.c_int()
    ptr = lib().vph_load_binary(path.encode(), C.byref(w), C.byref(h), C.byref(d), int(quantized))
    if not ptr:
        return None
    n = w.value * h.value * d.value
    ct = C.c_uint8 if quantized else C.c_float
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy().reshape(d.value, h.value, w.value)
    lib().vph_free(ptr)
    return arr


def dump_dense(path, vol):
    v = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = v.shape
    return lib().vph_dump_dense(path.encode(), _p(v), nx, ny, nz) == 0


def quantize(vol, max_value=0.0):
    v = np.ascontiguousarray(vol, np.float32)
    out = np.empty(v.shape, np.uint8)
    lib().vph_quantize(_p(v), v.size, max_value, _p(out))
    return out


def load_vdb(path, quantized=True):
    w, h, d = C.c_int(), C.c_int(), C.c_int()
    ptr = lib().vph_load_vdb(path.encode(), C.byref(w), C.byref(h), C.byref(d), int(quantized))
    return ptr or None


def camera_matrix(pos=None, fwd=None, up=None, focus=0.0):
    m = np.empty(12, np.float32)
    a = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (pos, fwd, up)]
    lib().vph_camera_matrix(*[None if v is None else _p(v) for v in a], focus, _p(m))
    return m


def material_preset(index):
    st = np.empty(3, np.float32)
    al = np.empty(3, np.float32)
    if lib().vph_material_preset(index, _p(st), _p(al)):
        raise IndexError(index)
    return st, al
