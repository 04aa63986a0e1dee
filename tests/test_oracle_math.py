"""The oracle's deterministic sin/cos/acos/pow against libm (numpy float64): they stand in for the
platform libm behind f32::sin, f32::cos, f32::acos, f32::powf in the reference (geo.rs:106-143,
common.rs:57), so the distance to the exact result is stated here: <= 0.5001 ulp for the double-precision
definitions (ScreenParams, tone map), < 0.9 ulp for the samplers' single-precision ones."""
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


def _batch(op, xs):
    xs = np.ascontiguousarray(xs, np.float32)
    out = np.zeros_like(xs)
    ob.lib().orc_math_batch(op, xs.ctypes.data_as(C.c_void_p), xs.shape[0], out.ctypes.data_as(C.c_void_p))
    return out


def test_sampler_trig_on_every_reachable_input():
    """f32::sin / cos / acos as random_sphere and random_hemisphere call them (geo.rs:101-145): the single-precision
    definitions stay below 0.9 ulp of the exact value on every input the samplers can produce (xi = i * 2^-23, i < 2^23)
    and on a dense sweep of the domains; tools/mathcheck.cpp is the exhaustive form (every float)."""
    xi = np.arange(1 << 23, dtype=np.float32) * np.float32(2.0 ** -23)
    theta = (np.float32(2.0) * np.float32(np.pi)) * xi
    u = np.float32(1.0) - np.float32(2.0) * xi
    phi = _batch(7, u)
    assert _max_ulp(phi, np.arccos(u.astype(np.float64))) < 0.9
    for x in (theta, phi, np.linspace(0, 6.2831855, 1_000_001).astype(np.float32)):
        assert _max_ulp(_batch(5, x), np.sin(x.astype(np.float64))) < 0.9
        assert _max_ulp(_batch(6, x), np.cos(x.astype(np.float64))) < 0.9
    sweep = np.linspace(-1, 1, 1_000_001).astype(np.float32)
    assert _max_ulp(_batch(7, sweep), np.arccos(sweep.astype(np.float64))) < 0.9
    assert _batch(7, np.array([1.0, -1.0], np.float32)).tolist() == [0.0, float(np.float32(np.pi))]
    assert _batch(5, np.zeros(1, np.float32))[0] == 0.0 and _batch(6, np.zeros(1, np.float32))[0] == 1.0
    assert np.isnan(_batch(7, np.array([1.5, -1.5, np.nan], np.float32))).all()


def test_pow_gamma():
    L = ob.lib()
    g = float(np.float32(1.0) / np.float32(2.2))
    xs = np.concatenate([np.linspace(0, 1, 8001), np.random.default_rng(3).uniform(0, 1, 8000), [1e-30, 1e-38, 1e-44]]).astype(np.float32)
    got = np.array([L.orc_math_pow(C.c_float(float(x)), C.c_float(g)) for x in xs], np.float32)
    ref = np.power(xs.astype(np.float64), g)
    nz = ref > 0
    assert _max_ulp(got[nz], ref[nz]) <= 0.5001
    assert L.orc_math_pow(C.c_float(0.0), C.c_float(g)) == 0.0 and L.orc_math_pow(C.c_float(1.0), C.c_float(g)) == 1.0


def test_single_precision_trig_renders_like_the_double_precision_definition(teapot_obj):
    """Goldens 04/05/07 were regenerated when the samplers' sin / cos / acos went from double-precision evaluations (<= 0.5001 ulp) to the
    single-precision forms (< 0.9 ulp), so they cannot vouch for that change.  This does: the oracle renders the same frame with either
    definition (orc_set_double_trig), and the images must be the same picture: individual shadow rays flip where a sample point moves by an
    ulp, but the mean absolute difference of the 8-bit frames stays far below one level and the means of the two frames agree."""
    import ctypes as C
    from oracle import binding as ob
    L = ob.lib()
    L.orc_set_double_trig.argtypes = [C.c_int]
    L.orc_set_double_trig.restype = None
    osc = ob.OracleScene.named("teapot", teapot_obj)
    p = ob.make_params(width=48, height=48, spp=64, light_samples=4, bounces=3, seed=0x5EED)
    single = osc.render(p)["rgba"].astype(np.float64)
    L.orc_set_double_trig(1)
    try:
        double = osc.render(p)["rgba"].astype(np.float64)
    finally:
        L.orc_set_double_trig(0)
    assert np.array_equal(osc.render(p)["rgba"], single.astype(np.uint8))                # the switch is off again
    mad = np.abs(single - double)[:, :, :3].mean()
    assert mad < 0.5, mad
    assert abs(single[:, :, :3].mean() - double[:, :, :3].mean()) < 0.05
