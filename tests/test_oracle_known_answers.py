"""Known-answer tests for the CPU oracle.  The reference ships no golden vectors
(its only test, src/scene/objects.rs:323-343, asserts nothing), so each case here is derived by
hand from the cited reference source.
"""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import binding as ob

f32 = np.float32


def _arr(*v):
    return np.array(v, np.float32)


def tri_hit(p9, org, d):
    t = C.c_float(0)
    ok = ob.lib().orc_tri_intersect(ob._p(_arr(*p9)), ob._p(_arr(*org)), ob._p(_arr(*d)), C.byref(t))
    return bool(ok), t.value


def sph_hit(c4, org, d):
    t = C.c_float(0)
    ok = ob.lib().orc_sphere_intersect(ob._p(_arr(*c4)), ob._p(_arr(*org)), ob._p(_arr(*d)), C.byref(t))
    return bool(ok), t.value


# 1. camera (raytracer.rs:110-133)
def test_camera_screen_params_and_corner_ray():
    p = ob.make_params(600, 600)
    assert p.start == f32(-0.70710677) and p.total == f32(1.4142135) and p.z == f32(1.7)
    assert p.inv_w == f32(1.0) / f32(600.0) and p.aspect_ratio == 1.0
    out = np.zeros(3, np.float32)
    ob.lib().orc_screen_to_world(C.byref(p), 0, 0, ob._p(out))
    iw = (f32(0) + f32(0.5)) * p.inv_w
    xi = (f32(p.start) + f32(iw) * f32(p.total)) * f32(1.0)
    yi = -f32(p.start) - f32(iw) * f32(p.total)
    v = np.array([xi, yi, -f32(1.7)], np.float32)
    n = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    assert np.array_equal(out, (v / n).astype(np.float32))
    assert out[0] < 0 and out[1] > 0 and out[2] < 0          # pixel (0,0) is top-left, world +y is up
    p2 = ob.make_params(1920, 1080)
    assert p2.aspect_ratio == f32(1920.0) / f32(1080.0)


# 2. central ray and the diagonal tie-break (mod.rs:289, 311-312, 427-437)
def test_central_ray_hits_back_wall_at_68_on_the_shared_diagonal():
    sc = ob.OracleScene.named("specular")
    ids, ts = sc.intersect_rays([[0, 0, 0]], [[0, 0, -1]])
    assert ts[0] == 68.0 and ids[0] in (4, 5)
    p = ob.make_params(600, 600)
    pid, pt, _ = sc.cast_primary(p)
    diag = pid[np.arange(600), np.arange(600)]
    # on the diagonal both back-wall triangles are hit at the same t; the winner must be the first one the
    # reference's DFS reaches (strict `<`, mod.rs:437) — the same id along the whole visible diagonal
    on_wall = [int(v) for v in diag[205:395] if v in (4, 5)]
    assert len(on_wall) > 100 and len(set(on_wall)) == 1


# 3. triangle accept / reject (objects.rs:38-57)
UNIT = (0, 0, 0, 1, 0, 0, 0, 1, 0)


def test_triangle_inside_edges_vertices():
    assert tri_hit(UNIT, (0.25, 0.25, 1), (0, 0, -1)) == (True, 1.0)
    for pt in [(0.5, 0.0), (0.0, 0.5), (0.5, 0.5), (0, 0), (1, 0), (0, 1)]:      # edges and vertices are inside
        ok, t = tri_hit(UNIT, (pt[0], pt[1], 2), (0, 0, -1))
        assert ok and t == 2.0
    assert not tri_hit(UNIT, (0.6, 0.6, 1), (0, 0, -1))[0]
    assert not tri_hit(UNIT, (-0.01, 0.5, 1), (0, 0, -1))[0]
    assert tri_hit(UNIT, (0.25, 0.25, -1), (0, 0, 1)) == (True, 1.0)              # two-sided


def test_triangle_b1_plus_b2_boundary_is_one_plus_eps():
    # 1.0 + EPS rounds to 1 + 2^-23 in f32: b1 + b2 == 1 + 2^-23 is accepted, 1 + 2^-22 is not
    one_ulp = float(np.nextafter(f32(0.5), f32(1)))                                 # 0.5 + 2^-24
    assert tri_hit(UNIT, (0.5, one_ulp, 1), (0, 0, -1))[0]                          # sum rounds to <= 1 + 2^-23
    assert not tri_hit(UNIT, (0.5, 0.5 + 2.0 ** -22, 1), (0, 0, -1))[0]


def test_triangle_t_threshold_and_parallel_ray():
    assert not tri_hit(UNIT, (0.25, 0.25, 5e-8), (0, 0, -1))[0]                     # t < EPS
    assert tri_hit(UNIT, (0.25, 0.25, 2e-7), (0, 0, -1))[0]
    assert not tri_hit(UNIT, (0.25, 0.25, -1), (0, 0, -1))[0]                       # behind
    ok, t = tri_hit(UNIT, (0.25, 0.25, 1), (1, 0, 0))                               # parallel: 1/0 = inf
    assert (not ok) or not (t < np.inf)                                             # never a closer hit (mod.rs:437)
    ok, t = tri_hit(UNIT, (0.25, 0.25, 0), (1, 0, 0))                               # in-plane: 0*inf = NaN
    assert (not ok) or math.isnan(t) or not (t < np.inf)


# 4. sphere (objects.rs:58-76)
def test_sphere_outside_inside_tangent_behind():
    s = (0, 0, -10, 2)
    assert sph_hit(s, (0, 0, 0), (0, 0, -1)) == (True, 8.0)
    ok, t = sph_hit(s, (0, 0, -10), (0, 0, -1))                                     # origin inside: NEGATIVE root
    assert ok and t == -2.0
    ok, t = sph_hit(s, (2, 0, 0), (0, 0, -1))                                       # tangent: d2 == r^2
    assert ok and t == 10.0
    assert not sph_hit(s, (0, 0, -20), (0, 0, -1))[0]                               # both roots negative
    assert not sph_hit(s, (2.001, 0, 0), (0, 0, -1))[0]


def test_negative_sphere_distance_wins_closest_hit():
    """Quirk (1): no d > 0 guard in Scene::intersect (mod.rs:437)."""
    sc = ob.OracleScene.named("specular")
    c = [20.0 / 3.0, -14.0, -48.0 - 20.0 / 3.0]
    ids, ts = sc.intersect_rays([c], [[0, 0, -1]])
    assert ids[0] == 12 and ts[0] < 0


# 5. the reference's own test triangle (objects.rs:331-342)
def test_reference_triangle_surface_normal():
    p9 = _arr(-5, -5, -20, 5, -5, -20, 5, 5, -20)
    n9 = np.concatenate([_arr(-0.4, 0, 1) / np.sqrt(f32(1.16)), _arr(0.4, 0, 1) / np.sqrt(f32(1.16)), _arr(0, 0, 1)]).astype(np.float32)
    out = np.zeros(3, np.float32)
    pt = _arr(4.173316, 3.258237, -20.0)
    ob.lib().orc_tri_normal(ob._p(p9), ob._p(n9), ob._p(pt), ob._p(out))
    # barycentrics by hand: v0=(10,0,0) v1=(10,10,0) v2=p-p1 -> w = y/10, v = (x - y)/10, u = 1 - v - w
    v2 = pt - p9[:3]
    w = v2[1] / 10.0
    v = (v2[0] - v2[1]) / 10.0
    u = 1.0 - v - w
    k = 1.0 / math.sqrt(1.16)
    exp = np.array([(-0.4 * k) * u + (0.4 * k) * v, 0.0, k * u + k * v + w])
    assert np.allclose(out, exp, atol=2e-6)
    assert abs(np.linalg.norm(out) - 1.0) > 1e-4                                    # quirk (7): not re-normalised


# 6/7. object ordering, light list, teapot load (mod.rs:83-97, 229-264)
def test_specular_object_order_and_lights():
    sc = ob.OracleScene.named("specular")
    e = sc.export()
    assert (sc.T, sc.S, sc.nlights) == (10, 3, 1) and list(e["lights"]) == [10]
    assert np.allclose(e["sph"][0], [0, 24.2, -58, 7])                              # lamp pokes through the ceiling
    bsdf = e["mats"][:, 0]
    assert bsdf[e["sph_mat"][1]] == 1 and bsdf[e["sph_mat"][2]] == 0                # mirror, then red
    assert sc.nnodes == 2 * 13 - 1


def test_teapot_load(teapot_obj):
    sc = ob.OracleScene.named("teapot", teapot_obj)
    assert (sc.T, sc.S) == (1034, 2) and list(sc.export()["lights"]) == [1034]
    e = sc.export()
    # first face `f 7//7 6//6 1//1` (obj/teapot.obj:1067), scale 0.13, offset (-20/3 - 2, -15, -48 - 2.5*20/3)
    v7 = _arr(37.0303, 30.4443, 14.2938)
    off = _arr(f32(-20.0) / f32(3.0) - f32(2.0), -15.0, f32(-48.0) - f32(2.5) * f32(20.0) / f32(3.0))
    assert np.array_equal(e["tri_pos"][10, :3], (off + v7 * f32(0.13)).astype(np.float32))
    nr = np.linalg.norm(e["tri_nrm"][10:].reshape(-1, 3), axis=1)
    assert np.allclose(nr, 1.0, atol=1e-5)                                          # Triangle::new normalises vn (objects.rs:249)


def test_triangle_lights_are_an_error_not_a_panic():
    tri = _arr(-1, -1, -5, 1, -1, -5, 0, 1, -5)
    mats = np.zeros((1, 8), np.uint32)
    mats[0, 4:7] = _arr(1, 1, 1).view(np.uint32)
    sc = ob.OracleScene.raw(tri, _arr(0, 0, 1, 0, 0, 1, 0, 0, 1), [0], np.zeros((0, 4)), [], mats)
    with pytest.raises(RuntimeError):
        sc.render(ob.make_params(8, 8, spp=1, light_samples=1, bounces=1))


# 8. tone map (common.rs:46-58)
def test_tone_map():
    tc = lambda v: ob.lib().orc_to_color(C.c_float(v))  # noqa: E731
    assert [tc(0.0), tc(1.0), tc(0.5), tc(7.0), tc(-3.0), tc(float("nan"))] == [0, 255, 186, 255, 0, 0]
    assert tc(float("inf")) == 255
    assert tc(0.999999) == 254                                                      # truncation, not rounding


# 9. RNG (fastrand 1.4.0 generator restated; stream layout build-defined)
def test_rng_generator_and_stream():
    raw = np.zeros(4, np.uint32)
    ob.lib().orc_rng_raw(0, 4, ob._p(raw))
    # PCG-XSH-RR 64/32 by hand: output(s) = rotr32((s ^ s>>18) >> 27, s >> 59), s' = s*M + C
    s, M, Cc, exp = 0, 6364136223846793005, 1442695040888963407, []
    for _ in range(4):
        x = ((s ^ (s >> 18)) >> 27) & 0xFFFFFFFF
        r = s >> 59
        exp.append(((x >> r) | (x << ((32 - r) & 31))) & 0xFFFFFFFF)
        s = (s * M + Cc) & (2 ** 64 - 1)
    assert list(raw) == exp
    a = np.zeros(1000, np.float32)
    ob.lib().orc_rng_stream(0x5EED, 42, 3, 1000, ob._p(a))
    assert (a >= 0).all() and (a < 1).all() and np.array_equal(a * 2 ** 23, np.floor(a * 2 ** 23))   # multiples of 2^-23
    b = np.zeros(1000, np.float32)
    ob.lib().orc_rng_stream(0x5EED, 42, 4, 1000, ob._p(b))
    assert not np.array_equal(a, b) and abs(a.mean() - 0.5) < 0.05


def test_draw_count_formula_row_R():
    """Per vertex: 2*nl*L light draws, +1 roulette when bounces_left >= 2 — checked
    through the ray counters: occluded calls = vertices * nl * L."""
    sc = ob.OracleScene.named("diffuse")
    p = ob.make_params(32, 32, spp=2, light_samples=3, bounces=1)
    r = sc.render(p)
    # bounces = 1: exactly one vertex per sample that hits something; every camera ray is one intersect call
    assert r["intersect_calls"] == 32 * 32 * 2
    assert r["occluded_calls"] % 3 == 0 and r["occluded_calls"] <= 32 * 32 * 2 * 3
    p0 = ob.make_params(32, 32, spp=2, light_samples=3, bounces=0)
    r0 = sc.render(p0)
    assert r0["occluded_calls"] == 0 and r0["intersect_calls"] == 32 * 32 * 2


def test_samplers_shapes():
    """random_sphere covers the whole sphere, random_hemisphere the +z half (geo.rs:101-145)."""
    v = np.zeros(3, np.float32)
    zs, zh = [], []
    for i in range(400):
        ob.lib().orc_random_sphere(0x5EED, i, 0, ob._p(v))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        zs.append(v[2])
        ob.lib().orc_random_hemisphere(0x5EED, i, 0, ob._p(v))
        assert abs(np.linalg.norm(v) - 1) < 1e-5 and v[2] >= 0
        zh.append(v[2])
    assert min(zs) < -0.8 and max(zs) > 0.8 and abs(np.mean(zh) - 0.5) < 0.08
    out = np.zeros(3, np.float32)
    ob.lib().orc_to_coord_space(ob._p(_arr(0, 0, 1)), ob._p(_arr(0, 0, -1)), ob._p(out))   # Frisvad singular branch
    assert np.array_equal(out, _arr(0, 0, -1))
    ob.lib().orc_to_coord_space(ob._p(_arr(1, 0, 0)), ob._p(_arr(0, 0, 1)), ob._p(out))
    assert np.array_equal(out, _arr(1, 0, 0))


def test_sharded_oracle_render_equals_whole():
    sc = ob.OracleScene.named("diffuse")
    whole = sc.render(ob.make_params(40, 24, spp=2, light_samples=2, bounces=3))["rgba"]
    acc = np.zeros_like(whole)
    for s in range(3):
        part = sc.render(ob.make_params(40, 24, spp=2, light_samples=2, bounces=3, tile=(16, 8), shard=(s, 3)))["rgba"]
        assert not (acc.astype(bool) & part.astype(bool)).any()
        acc |= part
    assert np.array_equal(acc, whole)
