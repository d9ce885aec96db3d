"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/volpath.h declares;
without a GPU every entry point that needs one fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "volpath.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef __cplusplus\s*\nvoid render_kernel[^;]*;\s*#else", "", text)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text))
    return {n for n in names if n not in ("defined",)}


def test_header_symbols_are_exported():
    import volpath
    L = volpath.lib()
    decl = _declared_symbols()
    assert {"init_cuda", "render_kernel", "init_envmap", "set_sun", "precompute_opacity", "scale", "gamma_correct",
            "vp_render_frames", "vp_set_shard"} <= decl
    missing = [s for s in sorted(decl) if not hasattr(L, s)]
    assert not missing, missing
    assert set(volpath.PART1_SYMBOLS + volpath.PART2_SYMBOLS) == decl


def test_param_layout_matches_reference():
    import volpath
    assert C.sizeof(volpath.Param) == 44  # src/param.h:4-12
    assert volpath.Param.g.offset == 28 and volpath.Param.sigma_t.offset == 32


def test_no_gpu_fails_loudly():
    import volpath
    if volpath.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(volpath.VolpathError):
        volpath.set_device(0)
    assert b"HIP device" in volpath.lib().vp_last_error() or volpath.lib().vp_last_error() != b""
    with pytest.raises(volpath.VolpathError):
        volpath.render_frames(None, 0, 1, volpath.make_param(8, 8))
    with pytest.raises(volpath.VolpathError):
        volpath.julia_volume(8)


def test_argument_validation_without_gpu():
    import volpath
    with pytest.raises(volpath.VolpathError):
        volpath.set_estimator(7)
    with pytest.raises(volpath.VolpathError):
        volpath.set_shard(3, 2)
    L = volpath.lib()
    assert L.vp_set_bound_brick(3) != 0 and L.vp_set_bound_brick(8) == 0 and L.vp_set_bound_brick(1) == 0
    volpath.set_shard(0, 1)


def test_mat_matches_reference_formula():
    import numpy as np
    import volpath
    P = volpath.mat(volpath.make_param(4, 4), 2.29, 2.39, 1.97, 0.0030, 0.0034, 0.046)
    st = np.array([2.29 + 0.0030, 2.39 + 0.0034, 1.97 + 0.046])
    assert np.allclose([P.sigma_t.x, P.sigma_t.y, P.sigma_t.z], st / st.max(), rtol=1e-6)
