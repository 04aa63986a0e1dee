"""The oracle's deterministic sin/cos/acos/pow against libm (numpy float64): they stand in for the
platform libm behind f32::sin, f32::cos, f32::acos, f32::powf in the reference (geo.rs:106-143,
common.rs:57), so the distance to a correctly rounded result is stated here: <= 0.5001 ulp."""
import ctypes as C

import numpy as np

from oracle import binding as ob


def _eval(f, xs):
    return np.array([f(C.c_float(float(x))) for x in xs], np.float32)


def _max_ulp(got, ref64):
    ulp = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - ref64) / ulp))


def test_sin_cos_on_the_sampler_range():
    L = ob.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.linspace(-0.8, 6.3, 8001), rng.uniform(0, 2 * np.pi, 8000)]).astype(np.float32)
    assert _max_ulp(_eval(L.orc_math_sin, xs), np.sin(xs.astype(np.float64))) <= 0.5001
    assert _max_ulp(_eval(L.orc_math_cos, xs), np.cos(xs.astype(np.float64))) <= 0.5001
    assert L.orc_math_sin(C.c_float(0.0)) == 0.0 and L.orc_math_cos(C.c_float(0.0)) == 1.0


def test_sin_cos_larger_arguments_used_by_the_mesh_generator():
    L = ob.lib()
    xs = np.random.default_rng(1).uniform(-50, 400, 4000).astype(np.float32)
    assert _max_ulp(_eval(L.orc_math_sin, xs), np.sin(xs.astype(np.float64))) <= 0.51
    assert _max_ulp(_eval(L.orc_math_cos, xs), np.cos(xs.astype(np.float64))) <= 0.51


def test_acos():
    L = ob.lib()
    xs = np.concatenate([np.linspace(-1, 1, 8001), np.random.default_rng(2).uniform(-1, 1, 8000)]).astype(np.float32)
    assert _max_ulp(_eval(L.orc_math_acos, xs), np.arccos(xs.astype(np.float64))) <= 0.5001
    assert L.orc_math_acos(C.c_float(1.0)) == 0.0
    assert L.orc_math_acos(C.c_float(-1.0)) == np.float32(np.pi)
    assert np.isnan(L.orc_math_acos(C.c_float(1.5)))


def test_pow_gamma():
    L = ob.lib()
    g = float(np.float32(1.0) / np.float32(2.2))
    xs = np.concatenate([np.linspace(0, 1, 8001), np.random.default_rng(3).uniform(0, 1, 8000), [1e-30, 1e-38, 1e-44]]).astype(np.float32)
    got = np.array([L.orc_math_pow(C.c_float(float(x)), C.c_float(g)) for x in xs], np.float32)
    ref = np.power(xs.astype(np.float64), g)
    nz = ref > 0
    assert _max_ulp(got[nz], ref[nz]) <= 0.5001
    assert L.orc_math_pow(C.c_float(0.0), C.c_float(g)) == 0.0 and L.orc_math_pow(C.c_float(1.0), C.c_float(g)) == 1.0
