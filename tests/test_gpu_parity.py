"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bars: primary-hit IDs, distances, ray counts and RGBA8 images bit-exact
(the north-star gate is IDs bit-exact and PSNR >= 40 dB; the shared arithmetic is built so that
the stronger bar holds).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def oracle_params(**kw):
    from oracle import binding as ob
    return ob.make_params(**kw)


def to_rt_params(op):
    """oracle RenderParams -> rt_render_params (same byte layout)."""
    from rustracer_amd._lib import RenderParams
    p = RenderParams()
    assert C.sizeof(p) == C.sizeof(op)
    C.memmove(C.byref(p), C.byref(op), C.sizeof(p))
    return p


def gpu_render(scene, p):
    from rustracer_amd._lib import check, lib, ptr
    rgba = np.zeros((p.height, p.width, 4), np.uint8)
    check(lib().rt_render(scene.handle, C.byref(p), ptr(rgba), None, None))
    return rgba, scene.counters()


def gpu_cast(scene, p):
    from rustracer_amd._lib import check, lib, ptr
    ids = np.full((p.height, p.width), 0xFFFFFFFF, np.uint32)
    ts = np.full((p.height, p.width), np.inf, np.float32)
    check(lib().rt_cast_primary(scene.handle, C.byref(p), ptr(ids), ptr(ts)))
    return ids, ts


@pytest.fixture(scope="module")
def scenes(gpu, teapot_obj):
    """(oracle scene, HIP scene) pairs built from the same constructors."""
    from oracle import binding as ob
    import rustracer_amd as ra
    cache = {}

    def get(name):
        if name not in cache:
            if name.startswith("dragon"):
                nu, nv = (500, 100) if name == "dragon_small" else (120, 40)
                vp, vn, f = ra.dragon_standin_mesh(nu, nv)
                cache[name] = (ob.OracleScene.dragon_mesh(vp, vn, f), ra.Scene.from_dragon_mesh(vp, vn, f))
            else:
                cache[name] = (ob.OracleScene.named(name, teapot_obj), ra.Scene._named(name, teapot_obj))
        return cache[name]
    return get


# ---------------------------------------------------------------------------------------------
def test_device_math_bit_exact(gpu):
    """sin/cos/acos/pow/to_color evaluated on the GPU == the oracle's, bit for bit."""
    from oracle import binding as ob
    from rustracer_amd._lib import check, lib, ptr
    L = ob.lib()
    rng = np.random.default_rng(1)
    x = np.concatenate([np.linspace(-0.8, 6.3, 20001), rng.uniform(0, 6.2831855, 20000)]).astype(np.float32)
    xa = np.concatenate([np.linspace(-1, 1, 20001), rng.uniform(-1, 1, 20000), [1.5, -1.5, np.nan]]).astype(np.float32)
    xp = np.concatenate([np.linspace(0, 1, 20001), rng.uniform(0, 1, 20000), [0.0, 1.0, np.nan]]).astype(np.float32)
    yp = np.full(xp.shape, np.float32(1.0) / np.float32(2.2), np.float32)
    xc = np.concatenate([np.linspace(-0.5, 1.5, 20001), [np.nan, np.inf, -np.inf, 0.0, 1.0, 0.5]]).astype(np.float32)
    for op, arr, y, f in ((0, x, None, L.orc_math_sin), (1, x, None, L.orc_math_cos), (2, xa, None, L.orc_math_acos)):
        out = np.zeros_like(arr)
        check(lib().rt_debug_math(op, ptr(arr), None, arr.shape[0], ptr(out)))
        ref = np.array([f(C.c_float(float(v))) for v in arr], np.float32)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), "op %d differs" % op
    # the samplers' single-precision sin / cos / acos on EVERY input they can produce: theta = 2 pi xi, u = 1 - 2 xi, phi = acos(u)
    # for all 2^23 values of xi (fastrand f32), plus the domain edges
    xi = np.arange(1 << 23, dtype=np.float32) * np.float32(2.0 ** -23)
    theta = (np.float32(2.0) * np.float32(np.pi)) * xi
    u = np.concatenate([np.float32(1.0) - np.float32(2.0) * xi, np.array([-1.0, 1.5, -1.5, np.nan], np.float32)])
    phi = np.zeros_like(u)
    check(lib().rt_debug_math(7, ptr(u), None, u.shape[0], ptr(phi)))
    ref = np.zeros_like(u)
    L.orc_math_batch(7, ptr(u), u.shape[0], ptr(ref))
    # outside [-1, 1] both sides return a NaN (sign and payload are the platform's: sqrt of a negative number)
    assert np.array_equal(np.isnan(phi), np.isnan(ref)) and np.isnan(phi[-3:]).all() and not np.isnan(phi[:-3]).any()
    bad = np.flatnonzero(phi[:-3].view(np.uint32) != ref[:-3].view(np.uint32))
    assert bad.size == 0, "acos_s differs at %d inputs, first u=%r: gpu %r oracle %r" % (bad.size, u[bad[0]], phi[bad[0]], ref[bad[0]])
    for arr in (theta, np.ascontiguousarray(phi[: 1 << 23])):
        for op in (5, 6):
            out = np.zeros_like(arr)
            check(lib().rt_debug_math(op, ptr(arr), None, arr.shape[0], ptr(out)))
            ref = np.zeros_like(arr)
            L.orc_math_batch(op, ptr(arr), arr.shape[0], ptr(ref))
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), "op %d differs" % op
    out = np.zeros_like(xp)
    check(lib().rt_debug_math(3, ptr(xp), ptr(yp), xp.shape[0], ptr(out)))
    ref = np.array([L.orc_math_pow(C.c_float(float(v)), C.c_float(float(yp[0]))) for v in xp], np.float32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    out = np.zeros_like(xc)
    check(lib().rt_debug_math(4, ptr(xc), None, xc.shape[0], ptr(out)))
    ref = np.array([L.orc_to_color(C.c_float(float(v))) for v in xc], np.float32)
    assert np.array_equal(out, ref)


def test_device_rng_stream(gpu):
    from oracle import binding as ob
    from rustracer_amd._lib import check, lib, ptr
    for seed, pix, s in ((0x5EED, 0, 0), (0x5EED, 123456, 7), (1, 2073599, 255), (2 ** 63 + 5, 4_000_000_000, 1023)):
        got = np.zeros(64, np.float32)
        ref = np.zeros(64, np.float32)
        check(lib().rt_debug_rng(seed, pix, s, 64, ptr(got)))
        ob.lib().orc_rng_stream(seed, pix, s, 64, ref.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got, ref)
        assert (got >= 0).all() and (got < 1).all()


def test_known_rays(scenes):
    """SURVEY.md §8c items 2-4 on the GPU: central-ray tie-break, negative sphere t, occlusion rules."""
    osc, gsc = scenes("specular")
    org = np.zeros((4, 3), np.float32)
    d = np.array([[0, 0, -1], [0.3, 0.1, -1], [0, 1, 0], [0, 0, 1]], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    gi, gt = gsc.intersect_rays(org, d)
    oi, ot = osc.intersect_rays(org, d)
    assert np.array_equal(gi, oi) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert gt[0] == 68.0 and gi[0] in (4, 5)           # back wall, on the shared diagonal
    assert gi[3] == 0xFFFFFFFF                           # through the open front
    # origin inside the red sphere (object 12): the reference returns the negative root (objects.rs:70-75)
    c = np.array([[20.0 / 3.0, -14.0, -48.0 - 20.0 / 3.0]], np.float32)
    gi, gt = gsc.intersect_rays(c, np.array([[0, 0, -1]], np.float32))
    oi, ot = osc.intersect_rays(c, np.array([[0, 0, -1]], np.float32))
    assert gi[0] == oi[0] == 12 and gt[0] == ot[0] and gt[0] < 0
    # occlusion: emissive sphere never blocks; walls do; max_dist respected
    o = np.array([[0, 0, -40], [0, 0, -40], [0, 0, -40], [0, -19, -58]], np.float32)
    dd = np.array([[0, 1, 0], [0, -1, 0], [0, -1, 0], [0, 1, 0]], np.float32)
    md = np.array([100, 100, 10, 100], np.float32)
    assert np.array_equal(gsc.occluded_rays(o, dd, md), osc.occluded_rays(o, dd, md))


@pytest.mark.parametrize("name", ["specular", "diffuse", "teapot", "triangle", "sphere_on_plane", "dragon_tiny"])
def test_random_rays_match_oracle(scenes, name):
    """Scene::intersect / is_occluded on random rays from random origins: ids, t and occlusion bit-exact."""
    osc, gsc = scenes(name)
    rng = np.random.default_rng(7)
    n = 20000
    org = rng.uniform([-19, -19, -66], [19, 19, 0], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50, 0] = 0.0                                     # zero components: inf / NaN paths of the slab test
    d[50:100, 1] = 0.0
    d[100:130, 2] = np.float32(1e-13)                    # |1/d| beyond 2^40: is_occluded falls back from the 8-wide tree to the binary one
    d[130:160, 0] = np.float32(-3e-12)                   # |1/d| just inside the 8-wide tree's range
    d[160:170, 1] = np.float32(1e-42)                    # denormal component: 1/d = inf
    gi, gt = gsc.intersect_rays(org, d)
    oi, ot = osc.intersect_rays(org, d)
    assert np.array_equal(gi, oi)
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    md = rng.uniform(0.5, 60, n).astype(np.float32)
    assert np.array_equal(gsc.occluded_rays(org, d, md), osc.occluded_rays(org, d, md))


@pytest.mark.parametrize("name,w,h", [("specular", 600, 600), ("teapot", 512, 512), ("teapot", 333, 177), ("dragon_small", 480, 270),
                                      ("triangle", 128, 128)])
def test_primary_ids_bit_exact(scenes, name, w, h):
    """BASELINE config 3 gate: 100 % of primary-hit object ids (and distances) identical."""
    osc, gsc = scenes(name)
    op = oracle_params(width=w, height=h)
    oi, ot, _ = osc.cast_primary(op)
    gi, gt = gpu_cast(gsc, to_rt_params(op))
    assert np.array_equal(gi, oi), "%d pixels differ" % int((gi != oi).sum())
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    if name == "specular":                              # i == j looks along the back wall's shared diagonal (tie-break)
        diag = gi[np.arange(200, 400), np.arange(200, 400)]
        assert set(np.unique(diag)) <= {4, 5, 11, 12}


def test_debug_render_matches_oracle(scenes):
    osc, gsc = scenes("teapot")
    op = oracle_params(width=200, height=150, mode=0)
    ref = osc.render(op)["rgba"]
    got, _ = gpu_render(gsc, to_rt_params(op))
    assert np.array_equal(got, ref)


CASES = [
    # name, w, h, spp, L, bounces, estimator
    ("teapot", 96, 96, 8, 4, 1, 0),              # config 2 shape: direct lighting, light importance sampling
    ("teapot", 64, 64, 4, 4, 8, 0),              # GI, depth 8
    ("specular", 96, 64, 8, 4, 8, 0),            # mirror sphere: specular branch, non-square aspect
    ("diffuse", 64, 64, 4, 2, 50, 0),            # reference default depth 50
    ("dragon_tiny", 64, 64, 4, 8, 8, 0),         # config 4 shape at toy size
    ("specular_teapot", 64, 64, 4, 4, 8, 0),     # config 5 shape at toy size
    ("sphere_on_plane", 64, 64, 16, 8, 1, 1),    # config 1: hemisphere estimator
    ("teapot", 64, 64, 4, 4, 3, 1),              # hemisphere estimator inside GI
    ("triangle", 64, 64, 4, 4, 4, 0),            # specular triangle with blended normals
    ("teapot", 48, 48, 3, 5, 0, 0),              # bounces = 0: emission only
]


@pytest.mark.parametrize("name,w,h,spp,L,b,est", CASES)
def test_render_matches_oracle(scenes, name, w, h, spp, L, b, est):
    osc, gsc = scenes(name)
    op = oracle_params(width=w, height=h, spp=spp, light_samples=L, bounces=b, direct_estimator=est, seed=0x5EED)
    ref = osc.render(op)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    assert cnt["rays_closest"] == ref["intersect_calls"]
    assert cnt["rays_shadow"] == ref["occluded_calls"]
    assert cnt["samples"] == w * h * spp
    ndiff = int((got != ref["rgba"]).sum())
    assert psnr(got, ref["rgba"]) >= 40.0                # north-star gate
    assert ndiff == 0, "%d of %d bytes differ (PSNR %.1f dB)" % (ndiff, got.size, psnr(got, ref["rgba"]))


def _two_light_scene():
    """A raw object list (rt_scene_create / Scene::new): the teapot scene's objects plus a second, coloured lamp and a mirror
    sphere, so that one_bounce_radiance_importance loops over two lights (raytracer.rs:266) and an emissive sphere sits inside
    the box (never a blocker, mod.rs:481)."""
    from oracle import binding as ob
    import rustracer_amd as ra
    e = ra.Scene.new_specular().export()
    mats = np.concatenate([e["mats"], np.zeros((2, 8), np.uint32)])
    lamp2 = np.array([0, 0, 0, 0, 0.9, 0.5, 0.2, 0], np.float32)          # diffuse, black reflectance, emittance (0.9, 0.5, 0.2)
    mirror = np.array([0, 0.9, 0.9, 0.9, 0, 0, 0, 0], np.float32)
    mats[-2] = lamp2.view(np.uint32); mats[-2][0] = 0
    mats[-1] = mirror.view(np.uint32); mats[-1][0] = 1
    sph = np.concatenate([e["sph"], np.array([[-12.0, -4.0, -40.0, 3.0], [9.0, 6.0, -52.0, 4.0]], np.float32)])
    sph_mat = np.concatenate([e["sph_mat"], np.array([len(mats) - 2, len(mats) - 1], np.uint32)])
    args = (e["tri_pos"], e["tri_nrm"], e["tri_mat"], sph, sph_mat, mats)
    return ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)


@pytest.mark.parametrize("w,h,spp,L,b,cam", [(72, 56, 4, 3, 6, (0.0, 0.0, 0.0)), (64, 64, 3, 5, 4, (2.0, -4.0, -6.0)), (40, 40, 2, 32, 3, (0.0, 0.0, 0.0)),
                                             (33, 47, 2, 70, 2, (0.0, 0.0, 0.0)), (24, 20, 1, 300, 2, (0.0, 0.0, 0.0))])
def test_two_lights_odd_sample_counts_and_moved_camera(gpu, w, h, spp, L, b, cam):
    """lights x light_samples = 6, 10, 64, 140, 600 slots per vertex (not powers of two, more than a wave), camera off the origin
    (Raytracer::move_camera, raytracer.rs:477-481): image and ray counts identical to the oracle."""
    osc, gsc = _two_light_scene()
    assert list(gsc.light_indexes()) == [10, 13]
    op = oracle_params(width=w, height=h, spp=spp, light_samples=L, bounces=b, cam=cam, seed=99)
    ref = osc.render(op)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
    assert np.array_equal(got, ref["rgba"]), "%d bytes differ" % int((got != ref["rgba"]).sum())
    for opt in ("persist", "wide"):                      # the one-thread-per-ray kernel and the binary tree give the same image
        gsc.set_option(opt, 0)
        alt, _ = gpu_render(gsc, to_rt_params(op))
        gsc.set_option(opt, 1)
        assert np.array_equal(alt, ref["rgba"]), opt


def test_occluder_cache_and_scheduling_switches_do_not_change_a_pixel(gpu):
    """The per-light occluder cache ("hot", fixed at the first render), the root-node test in k_light, the own-object test and the
    scheduling constants of k_trace_p and k_extend_p (and whether closest-hit rays take the persistent kernel at all) are search-order choices only: every combination gives the oracle's image and ray counts."""
    op = oracle_params(width=56, height=48, spp=3, light_samples=6, bounces=5, seed=7)
    osc, gsc = _two_light_scene()
    ref = osc.render(op)
    p = to_rt_params(op)
    for opts in ({}, {"self_test": 0}, {"refill_min": 1, "phase_bias": 1}, {"refill_min": 64, "phase_bias": 4, "shadow_waves": 5},
                 {"light_waves": 8}, {"extend_persist": 0}, {"extend_persist": 2, "extend_refill": 1, "extend_inner": 1},
                 {"extend_persist": 2, "extend_refill": 64, "extend_inner": 64, "extend_grid": 1}, {"extend_refill": 7, "extend_inner": 40, "extend_grid": 3},
                 {"shaft": 0}, {"shaft": 1}, {"shaft": 3, "refill_min": 1}, {"shaft": 16}, {"camera_once": 0}, {"camera_once": 0, "light_sum": 0, "extend_persist": 0}, {"light_grid": 8}, {"light_grid": 1, "trace_grid": 1}):
        for k, v in opts.items():
            gsc.set_option(k, v)
        got, cnt = gpu_render(gsc, p)
        assert np.array_equal(got, ref["rgba"]), opts
        assert cnt["rays_shadow"] == ref["occluded_calls"] and cnt["rays_closest"] == ref["intersect_calls"], opts
        for k, v in {"self_test": 1, "refill_min": 16, "phase_bias": 16, "shadow_waves": 6, "light_waves": 4, "extend_persist": 1, "extend_refill": 24,
                     "extend_inner": 20, "extend_grid": 8, "shaft": 6, "camera_once": 1, "light_sum": 1, "light_grid": 0, "trace_grid": 8}.items():
            gsc.set_option(k, v)
    with pytest.raises(Exception):
        gsc.set_option("hot", 0)                           # the cache is chosen at the first render
    _, cold = _two_light_scene()
    cold.set_option("hot", 0)
    got, _ = gpu_render(cold, p)
    assert np.array_equal(got, ref["rgba"]), "hot=0"
    cold.set_option("stats", 1)
    _, c_cold = gpu_render(cold, p)
    gsc.set_option("stats", 1)
    _, c_hot = gpu_render(gsc, p)
    gsc.set_option("stats", 0)
    assert 0 < c_hot["trace_steps"][5] < c_cold["trace_steps"][5]      # rays handed to k_trace_p: fewer with the cache


def _floaters_scene(n_tri=260, n_sph=60, seed=5, extra_lamps=0):
    """new_specular's room with a second, small lamp low in it (so that path vertices lie beside, below, above and inside the box of a
    light: every sign class of a beam's direction interval) and many small diffuse triangles and balls floating in the air: beams that
    are clear, beams a floater barely touches, beams it fills."""
    from oracle import binding as ob
    import rustracer_amd as ra
    e = ra.Scene.new_specular().export()
    rng = np.random.default_rng(seed)
    mats = np.concatenate([e["mats"], np.zeros((2, 8), np.uint32)])
    mats[-2] = np.array([0, 0, 0, 0, 2.5, 2.0, 1.5, 0], np.float32).view(np.uint32); mats[-2][0] = 0          # second lamp
    mats[-1] = np.array([0, 0.7, 0.7, 0.6, 0, 0, 0, 0], np.float32).view(np.uint32); mats[-1][0] = 0          # floaters
    c = rng.uniform([-18, -18, -66], [18, 16, -30], (n_tri, 3))
    tri = np.concatenate([c + rng.normal(scale=rng.uniform(0.05, 1.2, (n_tri, 1)), size=(n_tri, 3)) for _ in range(3)], axis=1).astype(np.float32)
    nrm = np.tile(np.array([0, 1, 0, 0, 1, 0, 0, 1, 0], np.float32), (n_tri, 1))
    balls = np.concatenate([rng.uniform([-18, -18, -66], [18, 16, -30], (n_sph, 3)), rng.uniform(0.05, 1.5, (n_sph, 1))], axis=1).astype(np.float32)
    lamps = np.concatenate([np.array([[6.0, -12.0, -45.0, 0.6]]), np.concatenate([rng.uniform([-15, -15, -60], [15, 15, -35], (extra_lamps, 3)),
                                                                                   rng.uniform(0.3, 1.2, (extra_lamps, 1))], axis=1)]).astype(np.float32)
    sph = np.concatenate([e["sph"], lamps, balls])
    sph_mat = np.concatenate([e["sph_mat"], np.full(1 + extra_lamps, len(mats) - 2, np.uint32), np.full(n_sph, len(mats) - 1, np.uint32)])
    args = (np.concatenate([e["tri_pos"], tri]), np.concatenate([e["tri_nrm"], nrm]), np.concatenate([e["tri_mat"], np.full(n_tri, len(mats) - 1, np.uint32)]),
            sph, sph_mat, mats)
    return ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)


def test_beams_settle_rays_without_changing_a_pixel(scenes):
    """k_light walks the occlusion BVH once per (vertex, light) with the hull of the vertex and the light's box ("shaft"); a beam that can
    meet no primitive record settles its unblocked rays as visible and they never reach k_trace_p.  One-sided by construction, so image and
    ray counts equal the oracle's for every node budget, and the counters show beams found clear and fewer rays traced."""
    cases = [(_floaters_scene(), dict(width=112, height=84, spp=3, light_samples=6, bounces=4, seed=17)),
             (_floaters_scene(n_tri=40, n_sph=4, seed=9), dict(width=96, height=96, spp=2, light_samples=16, bounces=3, seed=3, cam=(3.0, -6.0, -8.0))),
             (_two_light_scene(), dict(width=80, height=60, spp=3, light_samples=8, bounces=5, seed=23)),
             (_floaters_scene(n_tri=30, n_sph=6, seed=13, extra_lamps=5), dict(width=72, height=54, spp=2, light_samples=3, bounces=3, seed=41)),   # 7 lights: more than carry an occluder cache
             (scenes("dragon_small"), dict(width=128, height=96, spp=2, light_samples=8, bounces=4, seed=29)),
             (scenes("specular_teapot"), dict(width=96, height=72, spp=2, light_samples=8, bounces=6, seed=31))]
    for (osc, gsc), kw in cases:
        op = oracle_params(**kw)
        ref = osc.render(op)
        p = to_rt_params(op)
        gsc.set_option("stats", 1)
        traced = {}
        try:
            for budget in (0, 2, 8, 16):
                gsc.set_option("shaft", budget)
                got, cnt = gpu_render(gsc, p)
                assert np.array_equal(got, ref["rgba"]), (kw, budget, int((got != ref["rgba"]).sum()))
                assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"], (kw, budget)
                traced[budget] = cnt["trace_steps"][5]
                walked, clear, nodes, settled = cnt["shaft"]
                if budget == 0:
                    assert walked == 0 and settled == 0
                else:
                    assert walked > 0 and nodes <= walked * budget and (clear > 0) == (settled > 0), (kw, budget, cnt["shaft"])
                    if budget >= 8:
                        assert clear > 0, (kw, budget, cnt["shaft"])
            assert traced[16] <= traced[8] <= traced[2] <= traced[0] and traced[16] < traced[0], traced       # a larger budget proves more beams clear
            # The settled rays' terms of the sum (raytracer.rs:279) are added by k_light itself, in sample order, up to the first sample of the
            # (vertex, light) that leaves a record; "light_sum" 0 leaves every visible ray a record for k_shade_b instead.  "force_binary" hands
            # every second / eighth / every sample to k_binary, so that records turn up in the middle of clear beams: the sum must close there and k_shade_b carry on.
            gsc.set_option("shaft", 8)
            summed = {}
            for light_sum, force_binary in ((1, 0), (0, 0), (1, 1), (1, 7), (0, 3), (1, 0x80000000)):
                gsc.set_option("light_sum", light_sum)
                gsc.set_option("force_binary", force_binary)
                got, cnt = gpu_render(gsc, p)
                assert np.array_equal(got, ref["rgba"]), (kw, light_sum, force_binary, int((got != ref["rgba"]).sum()))
                assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"], (kw, light_sum, force_binary)
                summed[(light_sum, force_binary)] = cnt["trace_steps"][7]
                assert cnt["trace_steps"][7] <= cnt["shaft"][3]
            assert summed[(1, 0)] > 0 and summed[(0, 0)] == 0 and summed[(0, 3)] == 0 and summed[(1, 0x80000000)] == 0, summed
            assert 0 < summed[(1, 1)] < summed[(1, 7)] < summed[(1, 0)], summed      # the earlier the first forced record, the shorter the summed prefix
        finally:
            gsc.set_option("stats", 0)
            gsc.set_option("shaft", 6)
            gsc.set_option("light_sum", 1)
            gsc.set_option("force_binary", 0)


def test_random_scenes_frames_and_option_settings(gpu):
    """tools/fuzz_parity.py in small: random scenes (the room with floaters and extra lamps, dragon-class meshes, the reference's named scenes),
    frame sizes, sample counts, depths, cameras, estimators and settings of eleven options; image and both ray counts equal the oracle's."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"), "11", "40", "mesh"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "cases 40 x 3 renders, mismatches 0" in r.stdout


def test_render_lanes_do_not_change_a_pixel(scenes):
    """rt_render without progress / interrupt words cuts the owned pixels into up to 4 lanes that render concurrently on streams of their own
    (pass buffers and traversal-stack rows per lane).  Pixels are independent, so image and ray counts must equal the oracle's for any number
    of lanes, any pass size, sharded or not, with the importance and the hemisphere estimator, and with a polling caller (one lane)."""
    import rustracer_amd as ra
    osc, gsc = scenes("dragon_small")
    gsc.set_option("lane_min_pixels", 64)
    try:
        for est, shard, (w, h) in ((0, (0, 1), (150, 90)), (1, (0, 1), (97, 61)), (0, (1, 3), (160, 128))):
            op = oracle_params(width=w, height=h, spp=5, light_samples=4, bounces=4, seed=123, direct_estimator=est, tile=(32, 32), shard=shard)
            ref = osc.render(op)
            p = to_rt_params(op)
            for lanes, max_paths in ((1, 0), (2, 0), (3, 4096), (4, 1000), (4, 0)):
                gsc.set_option("lanes", lanes)
                gsc.set_option("max_paths", max_paths)
                gsc.set_option("light_token", 1 if lanes == 3 else 0)      # the lanes' k_light launches chained through events
                got, cnt = gpu_render(gsc, p)
                assert np.array_equal(got, ref["rgba"]), (est, shard, lanes, max_paths, int((got != ref["rgba"]).sum()))
                assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"], (est, shard, lanes, max_paths)
    finally:
        gsc.set_option("lanes", 2)
        gsc.set_option("max_paths", 0)
        gsc.set_option("lane_min_pixels", 65536)
        gsc.set_option("light_token", 0)


def test_coplanar_neighbours_of_a_tessellated_wall(gpu):
    """k_light flags a ray whose start triangle has a coplanar neighbour the ray provably cannot pass (the flat axis of the bvh slab test ends
    with ray_max <= (c - o) / d <= 0), and k_trace_p then skips that neighbour's record as it skips the start triangle's.  Held against the
    oracle where neighbours abound: a room whose floor, ceiling and back wall are grids of coplanar triangles (each triangle has several
    neighbours in its plane, only one of which may be skipped), with a lamp and a blocker box; image and ray counts identical, and the
    counters show the skip at work (far fewer primitive tests than traced rays)."""
    from oracle import binding as ob
    import rustracer_amd as ra
    tris = []

    def grid(axis, c, n, lo=-5.0, hi=5.0, flip=False):
        xs = np.linspace(lo, hi, n + 1, dtype=np.float32)
        o = [(axis + 1) % 3, (axis + 2) % 3]
        for i in range(n):
            for j in range(n):
                q = []
                for (u, v) in ((xs[i], xs[j]), (xs[i + 1], xs[j]), (xs[i + 1], xs[j + 1]), (xs[i], xs[j + 1])):
                    pt = [0.0, 0.0, 0.0]
                    pt[axis], pt[o[0]], pt[o[1]] = c, u, v
                    q.append(pt)
                a, b, cc, d = q
                tris.append(a + b + cc if not flip else a + cc + b)
                tris.append(a + cc + d if not flip else a + d + cc)
    grid(1, -5.0, 5)            # floor y = -5
    grid(1, 5.0, 4, flip=True)  # ceiling y = 5
    grid(2, -15.0, 3)           # back wall z = -15 (x, y in [-5, 5])
    tri = np.array(tris, np.float32)
    tri[:, [2, 5, 8]] += np.where(np.isin(np.arange(tri.shape[0]), np.arange(0, 82)), -10.0, 0.0).astype(np.float32)[:, None]   # floor and ceiling in front of the camera
    blocker = np.array([[-1, -5, -9, 1, -5, -9, 0, -2, -10], [-1, -5, -11, 0, -2, -10, 1, -5, -11]], np.float32)
    tri = np.concatenate([tri, blocker])
    nrm = np.tile(np.array([0, 1, 0, 0, 1, 0, 0, 1, 0], np.float32), (tri.shape[0], 1))
    mats = np.zeros((2, 8), np.uint32)
    mats[0, 1:4] = np.float32(0.6).view(np.uint32)
    mats[1, 4:7] = np.float32(40.0).view(np.uint32)
    sph = np.array([[0.0, 4.6, -10.0, 0.8]], np.float32)       # lamp poking through the ceiling grid
    args = (tri, nrm, np.zeros(tri.shape[0], np.uint32), sph, np.array([1], np.uint32), mats)
    osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)
    op = oracle_params(width=96, height=80, spp=4, light_samples=8, bounces=3, seed=31)
    ref = osc.render(op)
    gsc.set_option("stats", 1)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    assert np.array_equal(got, ref["rgba"]), "%d bytes differ" % int((got != ref["rgba"]).sum())
    assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
    steps = cnt["trace_steps"]
    assert steps[5] > 1000 and steps[3] < steps[5]      # primitive tests (lanes of prim steps) < rays handed to k_trace_p


def test_occluder_cache_with_a_sphere_next_to_the_lamp(gpu):
    """The cache's sphere arm: a diffuse ball hangs right under the lamp of the sphere-on-plane scene (their boxes overlap, so it is
    the lamp's cached occluder) and shadows most of the plane; images with the cache on and off equal the oracle's."""
    from oracle import binding as ob
    import rustracer_amd as ra
    e = ra.Scene._named("sphere_on_plane", None).export()
    sph = np.concatenate([e["sph"], np.array([[0.5, 14.0, -57.0, 5.0]], np.float32)])
    sph_mat = np.concatenate([e["sph_mat"], e["sph_mat"][-1:]])                      # the grey of the other ball
    args = (e["tri_pos"], e["tri_nrm"], e["tri_mat"], sph, sph_mat, e["mats"])
    osc = ob.OracleScene.raw(*args)
    op = oracle_params(width=64, height=48, spp=4, light_samples=8, bounces=3, seed=11)
    ref = osc.render(op)
    for hot in (1, 0):
        gsc = ra.Scene.from_arrays(*args)
        gsc.set_option("hot", hot)
        gsc.set_option("stats", 1)
        got, cnt = gpu_render(gsc, to_rt_params(op))
        assert np.array_equal(got, ref["rgba"]), "hot=%d" % hot
        assert cnt["rays_shadow"] == ref["occluded_calls"]
        traced = cnt["trace_steps"][5]
        if hot:
            traced_hot = traced
    assert traced_hot < traced                                                       # the cached ball settles rays before traversal


def test_axis_aligned_shadow_rays_take_the_binary_tree(gpu):
    """Light sampling almost never produces a direction with a zero component, so Scene::is_occluded is driven directly:
    rays with zero, denormal and tiny direction components (the 0 * inf = NaN cases of the slab test, which leave the 8-wide
    tree for the binary one) next to ordinary ones, some starting exactly on wall planes."""
    osc, gsc = _two_light_scene()
    rng = np.random.default_rng(3)
    n = 4096
    org = rng.uniform([-19, -19, -66], [19, 19, 0], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[0::4, rng.integers(3)] = 0.0
    d[1::8, 2] = np.float32(1e-44)
    d[2::8, 0] = np.float32(-2e-13)
    org[3::16] = np.float32(20.0) * np.sign(org[3::16])              # origins exactly on wall planes
    md = rng.uniform(0.5, 80, n).astype(np.float32)
    assert np.array_equal(gsc.occluded_rays(org, d, md), osc.occluded_rays(org, d, md))


def test_pass_size_shrinks_to_what_the_device_holds_and_stays_there(scenes):
    """When the pass buffers of the configured size cannot be allocated the render gives the workspace back, halves the pass size until it
    fits, and starts the next frame from the size that fitted (re-allocating a hundred GB per frame costs seconds).  The "alloc_limit" hook
    makes any buffer beyond 8 MB fail like a full device: same image as the oracle's, twice, the remembered size below the default."""
    from rustracer_amd._lib import check, lib
    osc, _ = scenes("teapot")
    import rustracer_amd as ra
    from conftest import TEAPOT_OBJ
    op = oracle_params(width=200, height=150, spp=8, light_samples=4, bounces=3, seed=77)
    ref = osc.render(op)
    p = to_rt_params(op)
    small = ra.Scene._named("teapot", TEAPOT_OBJ)         # a scene handle of its own: the workspace of the shared one is already large
    gpu_render(small, to_rt_params(oracle_params(width=8, height=8, spp=1, light_samples=1, bounces=1)))     # scene resident, workspace tiny
    assert small.get_option("max_paths_fit") == 0
    check(lib().rt_set_global_option(b"alloc_limit", 8 << 20))
    try:
        for _ in range(2):
            got, cnt = gpu_render(small, p)               # 240,000 paths: 15 MB of ray records, 23 MB of path stack
            assert np.array_equal(got, ref["rgba"])
            assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
            fit = small.get_option("max_paths_fit")
            assert 0 < fit < (1 << 26) and fit * 4 * 16 <= (8 << 20), fit
        small.set_option("max_paths", 0)                  # setting the option forgets the remembered size
        assert small.get_option("max_paths_fit") == 0
    finally:
        check(lib().rt_set_global_option(b"alloc_limit", 0))


def test_render_independent_of_pass_size_and_shards(scenes):
    """Chunking samples/pixels into passes and splitting tiles over shards never changes a pixel."""
    osc, gsc = scenes("teapot")
    op = oracle_params(width=80, height=72, spp=6, light_samples=3, bounces=5)
    ref = osc.render(op)["rgba"]
    p = to_rt_params(op)
    for max_paths in (1 << 22, 80 * 72 * 2, 1000, 64):
        gsc.set_option("max_paths", max_paths)
        got, _ = gpu_render(gsc, p)
        assert np.array_equal(got, ref), "max_paths=%d" % max_paths
    gsc.set_option("max_paths", 0)                        # back to the default
    from rustracer_amd._lib import check, lib, ptr
    frame = np.zeros((72, 80, 4), np.uint8)
    total_samples = 0
    for shard in range(3):
        q = p.copy()
        q.tile_w, q.tile_h, q.shard_index, q.shard_count = 16, 8, shard, 3
        check(lib().rt_render(gsc.handle, C.byref(q), ptr(frame), None, None))
        total_samples += gsc.counters()["samples"]
    assert np.array_equal(frame, ref)
    assert total_samples == 80 * 72 * 6


def test_stats_option_counts_and_same_image(scenes):
    """The instrumented kernels (algorithmic-bytes counters) produce the same image and sane counts."""
    osc, gsc = scenes("teapot")
    op = oracle_params(width=64, height=64, spp=2, light_samples=2, bounces=2)
    p = to_rt_params(op)
    plain, c0 = gpu_render(gsc, p)
    gsc.set_option("stats", 1)
    inst, c1 = gpu_render(gsc, p)
    gsc.set_option("stats", 0)
    assert np.array_equal(plain, inst)
    assert c1["rays_closest"] == c0["rays_closest"] and c1["rays_shadow"] == c0["rays_shadow"]
    assert c1["closest_inner"] > c1["rays_closest"] and c1["closest_tris"] > 0 and c1["closest_sphs"] > 0
    assert c1["shadow_wide"] > 0 and c1["shadow_inner"] == 0 and c1["shadow_tris"] > 0     # is_occluded rays walk the 8-wide occlusion BVH
    gsc.set_option("wide", 0)
    gsc.set_option("stats", 1)
    binary, c2 = gpu_render(gsc, p)
    gsc.set_option("stats", 0)
    gsc.set_option("wide", 1)
    assert np.array_equal(plain, binary)                                                     # ... and the binary tree gives the same image
    assert c2["shadow_wide"] == 0 and c2["shadow_inner"] > c2["rays_shadow"] and c2["rays_shadow"] == c0["rays_shadow"]
    assert c0["closest_inner"] == 0 and c0["shadow_inner"] == 0
    assert c1["closest_hits_tri"] + c1["closest_hits_sph"] > 0
    # closest-hit rays: ordered traversal with pruning never does more work than the reference's
    # unpruned search (shadow rays exit early in either order, so only primary rays are compared)
    gsc.set_option("stats", 1)
    gi, _ = gpu_cast(gsc, p)
    c2 = gsc.counters()
    gsc.set_option("stats", 0)
    oi, _, ocnt = osc.cast_primary(op)
    assert np.array_equal(gi, oi)
    assert c2["rays_closest"] == 64 * 64 == int(ocnt[0])
    assert c2["closest_inner"] + c2["closest_tris"] + c2["closest_sphs"] <= int(ocnt[2])


def test_golden_fixtures_on_gpu(gpu, teapot_obj):
    """The committed golden vectors (made by tests/golden/make_golden.py from the oracle) on the HIP path."""
    import json
    import rustracer_amd as ra
    from oracle import binding as ob
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for case in manifest["cases"]:
        data = np.load(os.path.join(GOLDEN, case["file"]))
        if case["scene"] == "dragon_tiny":
            vp, vn, f = ra.dragon_standin_mesh(120, 40)
            sc = ra.Scene.from_dragon_mesh(vp, vn, f)
        else:
            sc = ra.Scene._named(case["scene"], teapot_obj)
        op = ob.make_params(**case["params"])
        p = to_rt_params(op)
        if case["kind"] == "ids":
            gi, gt = gpu_cast(sc, p)
            assert np.array_equal(gi, data["ids"]) and np.array_equal(gt.view(np.uint32), data["t"].view(np.uint32)), case["file"]
        else:
            got, _ = gpu_render(sc, p)
            assert np.array_equal(got, data["rgba"]), case["file"]
        sc.close()


def test_interrupt_and_progress(scenes):
    """interrupt is polled between passes (raytracer.rs:184-204); progress reaches 100 (raytracer.rs:220)."""
    from rustracer_amd._lib import RT_ERR_INTERRUPTED, lib, ptr
    _, gsc = scenes("teapot")
    p = to_rt_params(oracle_params(width=64, height=64, spp=4, light_samples=2, bounces=2))
    rgba = np.zeros((64, 64, 4), np.uint8)
    prog, intr = C.c_uint32(0), C.c_uint32(0)
    assert lib().rt_render(gsc.handle, C.byref(p), ptr(rgba), C.byref(prog), C.byref(intr)) == 0
    assert prog.value == 100
    intr.value = 1
    assert lib().rt_render(gsc.handle, C.byref(p), ptr(rgba), C.byref(prog), C.byref(intr)) == RT_ERR_INTERRUPTED


def test_error_paths(gpu):
    """The ABI returns codes where the reference panics (triangle lights: objects.rs:98-100)."""
    import rustracer_amd as ra
    from rustracer_amd._lib import RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RtError, lib, ptr
    tri = np.array([[-1, -1, -5, 1, -1, -5, 0, 1, -5]], np.float32)
    nrm = np.array([[0, 0, 1] * 3], np.float32)
    mats = np.zeros((1, 8), np.uint32)
    mats[0, 4:7] = np.array([1, 1, 1], np.float32).view(np.uint32)        # emissive triangle
    sc = ra.Scene.from_arrays(tri, nrm, [0], np.zeros((0, 4)), [], mats)
    p = to_rt_params(oracle_params(width=16, height=16, spp=1, light_samples=1, bounces=1))
    rgba = np.zeros((16, 16, 4), np.uint8)
    assert lib().rt_render(sc.handle, C.byref(p), ptr(rgba), None, None) == RT_ERR_UNSUPPORTED
    p.spp = 0
    assert lib().rt_render(sc.handle, C.byref(p), ptr(rgba), None, None) == RT_ERR_INVALID
    with pytest.raises(RtError):
        ra.Scene.from_arrays(tri, nrm, [3], np.zeros((0, 4)), [], mats)    # material index out of range


def test_tile_pack_unpack_kernels(gpu):
    """rt_tiles_pack / rt_tiles_unpack (device) agree with the index-map version used for the gather."""
    import torch
    from rustracer_amd import tiles
    from rustracer_amd._lib import check, lib
    w, h, tw, th, world = 100, 70, 32, 16, 3
    p = to_rt_params(oracle_params(width=w, height=h))
    p.tile_w, p.tile_h, p.shard_count = tw, th, world
    frame = torch.arange(w * h, dtype=torch.int32, device="cuda").reshape(h, w, 1).contiguous()
    out = torch.zeros_like(frame)
    for r in range(world):
        p.shard_index = r
        n = lib().rt_tiles_per_shard(C.byref(p)) * tw * th
        compact = torch.zeros(n, dtype=torch.int32, device="cuda")
        check(lib().rt_tiles_pack(C.byref(p), frame.data_ptr(), compact.data_ptr(), 4, None))
        torch.cuda.synchronize()
        ref = tiles.pack_tiles(frame.reshape(-1, 1), tiles.tile_index_map(w, h, tw, th, r, world)).reshape(-1)
        assert torch.equal(compact, ref)
        check(lib().rt_tiles_unpack(C.byref(p), r, compact.data_ptr(), out.data_ptr(), 4, None))
    torch.cuda.synchronize()
    assert torch.equal(out, frame)


def test_python_api_mirror_renders(gpu, teapot_obj, tmp_path, monkeypatch):
    """Scene/Raytracer/Canvas mirror: headless image mode writes ./dump/<secs>.png (canvas.rs:41-110)."""
    import rustracer_amd as ra
    monkeypatch.chdir(tmp_path)
    cfg = ra.Config(screen_width=64, screen_height=48, samples_per_pixel=2, light_samples=2, bounces=3, image_mode=True)
    rt = ra.Raytracer.new(cfg, ra.Scene.new_teapot(teapot_obj))
    assert rt.inner.rendering_mode == ra.RenderingMode.Debug      # raytracer.rs:445
    rt.start()
    assert rt.inner.render_progress == 100 and not rt.inner.is_rendering
    path = rt.inner.canvas.last_path
    assert path and os.path.exists(path) and open(path, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    buf = rt.inner.pixel_buffer.data.reshape(48, 64, 4)
    assert (buf[:, :, 3] == 255).all() and buf[:, :, :3].max() > 0
    rt.move_camera((0.0, 0.0, -1.0))
    assert rt.inner.camera_position == [0.0, 0.0, -2.0]            # CAMERA_SPEED, raytracer.rs:18, 479
    rt.toggle_rendering_mode()
    assert rt.inner.rendering_mode == ra.RenderingMode.Debug
    rt.render(True)
    rt.reset_camera()
    assert rt.inner.camera_position == [0.0, 0.0, 0.0]


# ---------------------------------------------------------------------------------------------
# BASELINE's full sizes: 1920x1080 over the 871,200-triangle dragon-class mesh (configs 3 and 4).
@pytest.fixture(scope="module")
def full_dragon(gpu):
    from oracle import binding as ob
    import rustracer_amd as ra
    vp, vn, f = ra.dragon_standin_mesh(*ra.DRAGON_STANDIN_FULL)
    return ob.OracleScene.dragon_mesh(vp, vn, f), ra.Scene.from_dragon_mesh(vp, vn, f)


def test_full_size_primary_ids_bit_exact(full_dragon):
    """Config 3 as BASELINE states it: every one of the 1920x1080 primary rays finds the oracle's object, at the oracle's distance."""
    osc, gsc = full_dragon
    op = oracle_params(width=1920, height=1080, spp=1, light_samples=1, bounces=1)
    oi, ot, ocnt = osc.cast_primary(op)
    gi, gt = gpu_cast(gsc, to_rt_params(op))
    assert int(ocnt[0]) == 1920 * 1080
    assert np.array_equal(gi, oi), "%d ids differ" % int((gi != oi).sum())
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert len(np.unique(gi)) > 20000                       # the mesh fills a good part of the frame


def test_full_size_gi_tiles_match_oracle_and_shards_compose(full_dragon):
    """Config 4's frame (1080p, 32 light samples, depth 8) at 2 spp: the oracle renders every 97th 32x32 tile and those pixels must be
    bit-identical in the GPU frame; the frame must not depend on how it is cut into passes or shards (size-independent properties:
    pixels are independent, RNG streams are keyed by the global pixel index)."""
    osc, gsc = full_dragon
    W, H, K = 1920, 1080, 97
    op = oracle_params(width=W, height=H, spp=2, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(0, K))
    ref = osc.render(op)
    full = oracle_params(width=W, height=H, spp=2, light_samples=32, bounces=8, seed=0x5EED)
    p = to_rt_params(full)
    got, cnt = gpu_render(gsc, p)
    assert cnt["samples"] == W * H * 2 and cnt["rays_shadow"] % 32 == 0
    tiles_x = W // 32
    n_tiles = tiles_x * ((H + 31) // 32)
    checked = 0
    for t in range(0, n_tiles, K):
        x0, y0 = (t % tiles_x) * 32, (t // tiles_x) * 32
        a, b = got[y0:y0 + 32, x0:x0 + 32], ref["rgba"][y0:y0 + 32, x0:x0 + 32]
        assert np.array_equal(a, b), "tile %d: %d bytes differ" % (t, int((a != b).sum()))
        checked += a.shape[0] * a.shape[1]
    assert checked >= 20 * 1024
    # the same frame from 5 shards of 64x16 tiles, in smaller passes
    from rustracer_amd._lib import check, lib, ptr
    gsc.set_option("max_paths", 1 << 20)
    frame = np.zeros((H, W, 4), np.uint8)
    rays = 0
    for shard in range(5):
        q = p.copy()
        q.tile_w, q.tile_h, q.shard_index, q.shard_count = 64, 16, shard, 5
        check(lib().rt_render(gsc.handle, C.byref(q), ptr(frame), None, None))
        c = gsc.counters()
        rays += c["rays_closest"] + c["rays_shadow"]
    gsc.set_option("max_paths", 0)
    assert np.array_equal(frame, got)
    assert rays == cnt["rays_closest"] + cnt["rays_shadow"]


@pytest.mark.parametrize("name,w,h,spp,L,b,est,stride", [
    ("sphere_on_plane", 512, 512, 16, 8, 1, 1, 1),          # config 1 whole: hemisphere estimator, direct lighting
    ("teapot", 1024, 1024, 64, 16, 1, 0, 64),               # config 2 at its own spp: every 64th tile
    ("specular_teapot", 3840, 2160, 16, 32, 8, 0, 500),     # config 5's 4K frame at 16 of 1024 spp: every 500th tile
])
def test_baseline_configs_at_full_resolution(scenes, name, w, h, spp, L, b, est, stride):
    """The other BASELINE configs at their full resolution: the oracle renders every `stride`-th 32x32 tile, the GPU the whole
    frame; the oracle's tiles must be bit-identical in it."""
    osc, gsc = scenes(name)
    ref = osc.render(oracle_params(width=w, height=h, spp=spp, light_samples=L, bounces=b, direct_estimator=est, tile=(32, 32), shard=(0, stride)))
    got, cnt = gpu_render(gsc, to_rt_params(oracle_params(width=w, height=h, spp=spp, light_samples=L, bounces=b, direct_estimator=est)))
    assert cnt["samples"] == w * h * spp
    tiles_x = (w + 31) // 32
    n_tiles = tiles_x * ((h + 31) // 32)
    for t in range(0, n_tiles, stride):
        x0, y0 = (t % tiles_x) * 32, (t // tiles_x) * 32
        a, r = got[y0:y0 + 32, x0:x0 + 32], ref["rgba"][y0:y0 + 32, x0:x0 + 32]
        assert np.array_equal(a, r), "tile %d: %d bytes differ" % (t, int((a != r).sum()))
    if stride == 1:
        assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]


def _own_tiles(w, h, K):
    tiles_x = (w + 31) // 32
    return [((t % tiles_x) * 32, (t // tiles_x) * 32) for t in range(0, tiles_x * ((h + 31) // 32), K)]


def test_config4_at_its_full_spp_on_strided_tiles(full_dragon):
    """BASELINE config 4 exactly as stated (1920x1080, 256 spp, 32 light samples, depth 8) on every 97th 32x32 tile: both sides render
    shard (0, 97) of the frame, so every sample index 0..255 and the accumulation across sample chunks (k_resolve's s0 > 0 path) are
    compared at full size.  RGBA8 bit-identical, ray counts equal; once in one pass per chunk of the default size, once cut into >= 4
    sample chunks."""
    osc, gsc = full_dragon
    W, H, K = 1920, 1080, 97
    op = oracle_params(width=W, height=H, spp=256, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(0, K))
    ref = osc.render(op)
    tiles = _own_tiles(W, H, K)
    n_pix = sum(min(32, W - x0) * min(32, H - y0) for x0, y0 in tiles)
    for max_paths in (0, n_pix * 50):                      # default pass size; then 50 of the 256 samples per pass: 6 chunks, the last one short
        gsc.set_option("max_paths", max_paths)
        got, cnt = gpu_render(gsc, to_rt_params(op))
        gsc.set_option("max_paths", 0)
        assert cnt["samples"] == n_pix * 256
        assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
        for x0, y0 in tiles:
            a, b = got[y0:y0 + 32, x0:x0 + 32], ref["rgba"][y0:y0 + 32, x0:x0 + 32]
            assert np.array_equal(a, b), "tile at (%d, %d): %d bytes differ (max_paths %d)" % (x0, y0, int((a != b).sum()), max_paths)
    assert len(tiles) >= 20


def test_config5_at_its_full_spp_on_strided_tiles(scenes):
    """BASELINE config 5 exactly as stated (3840x2160, 1024 spp, 32 light samples, depth 8, spheres + teapot) on every 1600th tile."""
    osc, gsc = scenes("specular_teapot")
    W, H, K = 3840, 2160, 1600
    op = oracle_params(width=W, height=H, spp=1024, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(0, K))
    ref = osc.render(op)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    tiles = _own_tiles(W, H, K)
    assert cnt["samples"] == sum(min(32, W - x0) * min(32, H - y0) for x0, y0 in tiles) * 1024
    assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
    for x0, y0 in tiles:
        a, b = got[y0:y0 + 32, x0:x0 + 32], ref["rgba"][y0:y0 + 32, x0:x0 + 32]
        assert np.array_equal(a, b), "tile at (%d, %d): %d bytes differ" % (x0, y0, int((a != b).sum()))
    assert len(tiles) >= 5


def test_degenerate_scenes_and_frames(gpu):
    """Empty object list, a lamp alone (no blocker: no occlusion BVH), one wall under a lamp, a 1x1 frame and a frame whose size is no
    multiple of the screen tile: images and ray counts equal the oracle's."""
    from oracle import binding as ob
    import rustracer_amd as ra
    mats = np.zeros((2, 8), np.uint32)
    mats[0, 1:4] = np.array([0.78, 0.78, 0.78], np.float32).view(np.uint32)          # grey diffuse
    mats[1, 4:7] = np.array([1, 1, 1], np.float32).view(np.uint32)                   # lamp
    none3, none4 = np.zeros((0, 9), np.float32), np.zeros((0, 4), np.float32)
    lamp = np.array([[0.0, 6.0, -30.0, 3.0]], np.float32)
    wall = np.array([[-30, -8, -10, 30, -8, -10, 0, -8, -70]], np.float32)
    wall_n = np.array([[0, 1, 0] * 3], np.float32)
    cases = [
        (none3, none3, [], none4, [], mats),                                          # nothing at all
        (none3, none3, [], lamp, [1], mats),                                          # the lamp alone
        (wall, wall_n, [0], lamp, [1], mats),                                         # one triangle + lamp
        (wall, wall_n, [0], none4, [], mats),                                         # one triangle, no light
    ]
    for k, args in enumerate(cases):
        osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)
        for (w, h, spp, L, b) in ((1, 1, 3, 2, 4), (45, 37, 2, 3, 3)):
            op = oracle_params(width=w, height=h, spp=spp, light_samples=L, bounces=b, seed=5)
            ref = osc.render(op)
            got, cnt = gpu_render(gsc, to_rt_params(op))
            assert np.array_equal(got, ref["rgba"]), (k, w, h)
            assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"], (k, w, h)
            oi, ot, _ = osc.cast_primary(op)
            gi, gt = gpu_cast(gsc, to_rt_params(op))
            assert np.array_equal(gi, oi) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)), (k, w, h)


def test_coincident_and_degenerate_triangles(gpu):
    """2000 copies of one triangle (every closest hit is a 2000-way tie in t, settled by the reference's traversal order; the builder
    falls back to halving), zero-area triangles and slivers, under a lamp: ids, distances, images and ray counts equal the oracle's."""
    from oracle import binding as ob
    import rustracer_amd as ra
    rng = np.random.default_rng(12)
    one = np.array([-9, -6, -30, 9, -6, -30, 0, 8, -34], np.float32)
    tri = np.tile(one, (2000, 1))
    zero_area = np.tile(np.array([1, 1, -20, 1, 1, -20, 2, 2, -21], np.float32), (50, 1))
    p = rng.uniform([-15, -10, -50], [15, 10, -15], (300, 3)).astype(np.float32)
    sliver = np.concatenate([p, p + np.float32(1e-4) * rng.normal(size=(300, 3)).astype(np.float32), p + rng.normal(size=(300, 3)).astype(np.float32)], axis=1)
    floor = np.array([[-40, -12, 0, 40, -12, 0, 0, -12, -90]], np.float32)
    pos = np.concatenate([tri, zero_area, sliver, floor]).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1] * 3, np.float32), (pos.shape[0], 1))
    mats = np.zeros((2, 8), np.uint32)
    mats[0, 1:4] = np.array([0.7, 0.6, 0.5], np.float32).view(np.uint32)
    mats[1, 4:7] = np.array([1, 1, 1], np.float32).view(np.uint32)
    args = (pos, nrm, np.zeros(pos.shape[0], np.uint32), np.array([[3.0, 14.0, -25.0, 4.0]], np.float32), [1], mats)
    osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)
    op = oracle_params(width=96, height=80, spp=2, light_samples=4, bounces=4, seed=3)
    oi, ot, _ = osc.cast_primary(op)
    gi, gt = gpu_cast(gsc, to_rt_params(op))
    assert np.array_equal(gi, oi) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert ((gi < 2000).sum()) > 500                                   # the stack of copies is in view
    ref = osc.render(op)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    assert np.array_equal(got, ref["rgba"])
    assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]


@pytest.mark.parametrize("scale_by", [3.0e18, 2.0e8, 1.0e-18])
def test_scene_at_extreme_scale(gpu, scale_by):
    """The specular scene blown up past the 8-wide tree's coordinate range (is_occluded falls back to the binary tree; products in
    the triangle test overflow to inf and NaN) and shrunk towards the denormals: same IEEE operations, same ids, distances and image."""
    from oracle import binding as ob
    import rustracer_amd as ra
    e = ra.Scene.new_specular().export()
    s = np.float32(scale_by)
    sph = e["sph"] * s
    args = (e["tri_pos"] * s, e["tri_nrm"], e["tri_mat"], sph, e["sph_mat"], e["mats"])
    osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)
    assert (gsc.info.n_wide_nodes == 0) == (scale_by > 1.0e18)
    op = oracle_params(width=72, height=64, spp=2, light_samples=3, bounces=4, seed=21)
    oi, ot, _ = osc.cast_primary(op)
    gi, gt = gpu_cast(gsc, to_rt_params(op))
    assert np.array_equal(gi, oi) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    ref = osc.render(op)
    got, cnt = gpu_render(gsc, to_rt_params(op))
    assert cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
    assert np.array_equal(got, ref["rgba"]), "%d bytes differ" % int((got != ref["rgba"]).sum())


def test_render_from_a_second_thread(scenes):
    """Raytracer::render(false) renders on a thread of its own (raytracer.rs:527): hipSetDevice binds a thread, not the process,
    so every entry point selects the scene's device again.  Same pixels from a fresh thread as from the calling one."""
    import threading
    osc, gsc = scenes("teapot")
    op = oracle_params(width=80, height=60, spp=2, light_samples=2, bounces=3)
    ref = osc.render(op)["rgba"]
    box = {}

    def work():
        box["rgba"], box["cnt"] = gpu_render(gsc, to_rt_params(op))
        box["ids"], _ = gpu_cast(gsc, to_rt_params(op))

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert np.array_equal(box["rgba"], ref)
    assert np.array_equal(box["ids"], osc.cast_primary(op)[0])


@pytest.mark.parametrize("op,what", [(0, "1/x on every float"), (1, "sqrt on every float"), (2, "a/b on 2^35 pairs"), (3, "normalise + 1/d on 2^32 vectors")])
def test_fast_ieee_forms_are_exact(gpu, op, what):
    """The kernels replace hipcc's 11- and 17-instruction expansions of IEEE division and square root by shorter sequences for operands in
    [2^-96, 2^96] (rt_kernels.h: rcp_mid, div_mid, sqrt_mid).  Those must give the compiler's (= the CPU's, = the reference's) bits on every
    input: compared on the GPU itself, exhaustively where the domain allows."""
    from rustracer_amd._lib import check, lib
    out = np.zeros(2, np.uint64)
    check(lib().rt_debug_exact(op, out.ctypes.data_as(C.c_void_p)))
    assert out[0] == 0, "%s: %d inputs differ, e.g. 0x%x" % (what, int(out[0]), int(out[1]))


def _oracle_intersect_parallel(osc, org, d, workers=8):
    """The oracle's ray batch call is a plain loop; ctypes releases the GIL, so chunks run on threads."""
    from concurrent.futures import ThreadPoolExecutor
    n = org.shape[0]
    cuts = np.linspace(0, n, workers + 1).astype(int)
    with ThreadPoolExecutor(workers) as ex:
        parts = list(ex.map(lambda k: osc.intersect_rays(org[cuts[k]:cuts[k + 1]], d[cuts[k]:cuts[k + 1]]), range(workers)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def test_pruned_closest_hit_equals_the_unpruned_search_on_slivers_and_grazing_rays(gpu):
    """Scene::intersect in the reference is unordered and unpruned: any leaf the ray reaches may win with whatever distance its triangle test
    returns, however ill-conditioned (mod.rs:436-441).  The kernels visit near-first and skip subtrees whose box entry lies beyond the best hit
    (rt_kernels.h prune_limit).  The rule is held here against the oracle's unpruned search where it is weakest: 120,000 sliver triangles
    (edge ratios 1e-3 .. 1e-7, sizes 1 .. 1e4, at distances up to 1e4) between walls, and 1.2 million rays of which a third graze a sliver or a
    wall almost in its plane, a third aim at sliver edges, a third are random.  ids and distances bit-equal."""
    from oracle import binding as ob
    import rustracer_amd as ra
    rng = np.random.default_rng(11)
    T = 120_000

    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    size = 10.0 ** rng.uniform(0, 4, T)
    ratio = 10.0 ** rng.uniform(-7, -3, T)
    p1 = rng.uniform(-1, 1, (T, 3)) * (10.0 ** rng.uniform(0, 4, (T, 1)))
    u = unit(rng.normal(size=(T, 3)))
    v = unit(np.cross(u, rng.normal(size=(T, 3))))
    e1 = u * size[:, None]
    e2 = u * (size * rng.uniform(0.05, 0.95, T))[:, None] + v * (size * ratio)[:, None]
    tri = np.concatenate([p1, p1 + e1, p1 + e2], axis=1)
    # walls: 12 large triangles of a box of half side 2e4
    Lw = 2.0e4
    c = np.array([[sx, sy, sz] for sx in (-Lw, Lw) for sy in (-Lw, Lw) for sz in (-Lw, Lw)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    walls = np.array([np.concatenate([c[a], c[b], c[cc]]) for q in quads for a, b, cc in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))])
    tri = np.concatenate([walls, tri]).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1, 0, 0, 1, 0, 0, 1], np.float32), (tri.shape[0], 1))
    mats = np.zeros((1, 8), np.uint32)
    mats[0, 1:4] = np.float32(0.5).view(np.uint32)
    args = (tri, nrm, np.zeros(tri.shape[0], np.uint32), np.zeros((0, 4), np.float32), np.zeros(0, np.uint32), mats)
    osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)

    N = 400_000
    t64 = tri.astype(np.float64)
    # (a) grazing: towards a point of a triangle (sliver or wall), along a direction almost inside its plane
    k = rng.integers(0, tri.shape[0], N)
    a, b, cc = t64[k, 0:3], t64[k, 3:6], t64[k, 6:9]
    w = rng.dirichlet((1, 1, 1), N)
    q = a * w[:, :1] + b * w[:, 1:2] + cc * w[:, 2:]
    nn = unit(np.cross(b - a, cc - a) + 1e-300)
    inplane = unit((b - a) * rng.normal(size=(N, 1)) + (cc - a) * rng.normal(size=(N, 1)) * 1e3 + 1e-300)
    d_a = unit(inplane + nn * (10.0 ** rng.uniform(-7, -2, (N, 1))) * rng.choice([-1.0, 1.0], (N, 1)))
    o_a = q - d_a * (10.0 ** rng.uniform(0, 4, (N, 1)))
    # (b) at sliver edges and vertices from afar
    k = rng.integers(12, tri.shape[0], N)
    a, b, cc = t64[k, 0:3], t64[k, 3:6], t64[k, 6:9]
    s = rng.uniform(0, 1, (N, 1))
    q = np.where(rng.uniform(size=(N, 1)) < 0.5, a + (b - a) * s, a + (cc - a) * s)
    o_b = rng.uniform(-1.5e4, 1.5e4, (N, 3))
    d_b = unit(q - o_b)
    # (c) random
    o_c = rng.uniform(-1.5e4, 1.5e4, (N, 3))
    d_c = unit(rng.normal(size=(N, 3)))
    org = np.concatenate([o_a, o_b, o_c]).astype(np.float32)
    d = np.concatenate([d_a, d_b, d_c]).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gi, gt = gsc.intersect_rays(org, d)
    oi, ot = _oracle_intersect_parallel(osc, org, d, workers=min(len(os.sched_getaffinity(0)), 16))
    bad = np.flatnonzero((gi != oi) | (gt.view(np.uint32) != ot.view(np.uint32)))
    assert bad.size == 0, "%d of %d rays differ; first: ray %d gpu (%d, %r) oracle (%d, %r)" % (bad.size, org.shape[0], bad[0], gi[bad[0]], gt[bad[0]], oi[bad[0]], ot[bad[0]])
    assert (gi[:N] != 0xFFFFFFFF).mean() > 0.9 and len(np.unique(gi)) > 20000


def _grazing_pairs(n_pairs, rays_per_pair, seed, theta_lo=-7.0, theta_hi=-3.0):
    """The configuration in which distance pruning is weakest: a large, well-shaped triangle T met almost in its own plane (theta_lo ..
    theta_hi decades of radians off it) by rays that hit a small triangle S first, S lying in front of the point where the ray enters T's
    box.  The reference searches unpruned and takes whatever distance T's test returns (mod.rs:436-441), and at such angles 1 / (s1 . e1)
    has few correct bits.  Every pair sits in a cell of its own, far from the others.  Returns (triangles, origins, directions), float32."""
    rng = np.random.default_rng(seed)

    def unit(v):
        return v / np.linalg.norm(v, axis=1, keepdims=True)
    P, R = n_pairs, rays_per_pair
    cell = 4000.0
    g = int(np.ceil(P ** (1 / 3)))
    centre = np.stack(np.unravel_index(np.arange(P), (g, g, g)), axis=1).astype(np.float64) * cell
    size = 10.0 ** rng.uniform(1, 2.7, P)                                  # T: side 10 .. 500, angles well away from 0
    u = unit(rng.normal(size=(P, 3)))
    v = unit(np.cross(u, rng.normal(size=(P, 3))))
    a = centre + rng.uniform(-50, 50, (P, 3))
    b = a + u * size[:, None]
    c = a + (u * rng.uniform(0.2, 0.8, (P, 1)) + v * rng.uniform(0.5, 1.0, (P, 1))) * size[:, None]
    nn = unit(np.cross(b - a, c - a))
    # rays: towards a point q inside T, direction = in-plane direction tilted by theta out of the plane
    pi = np.repeat(np.arange(P), R)
    w = rng.dirichlet((1.5, 1.5, 1.5), P * R)
    q = a[pi] * w[:, :1] + b[pi] * w[:, 1:2] + c[pi] * w[:, 2:]
    phi = rng.uniform(0, 2 * np.pi, (P * R, 1))
    inplane = u[pi] * np.cos(phi) + v[pi] * np.sin(phi)
    theta = 10.0 ** rng.uniform(theta_lo, theta_hi, (P * R, 1)) * rng.choice([-1.0, 1.0], (P * R, 1))
    d = unit(inplane * np.cos(theta) + nn[pi] * np.sin(theta))
    dist = 10.0 ** rng.uniform(1.5, 3.0, (P * R, 1))
    o = q - d * dist
    # where the ray enters T's box (slab test in double), and S across the ray in front of that point: one S per pair, placed on the
    # pair's first ray, large enough that the pair's other rays (which fan out from nearby origins) mostly hit it too
    lo, hi = np.minimum(np.minimum(a, b), c)[pi], np.maximum(np.maximum(a, b), c)[pi]
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    t_in = np.max(np.minimum(t0, t1), axis=1)
    first = np.arange(P) * R
    # all rays of a pair share the first ray's origin region: re-aim them from one origin so that one small S covers them
    o = np.repeat(o[first], R, axis=0) + rng.normal(scale=1e-3, size=(P * R, 3))
    d = unit(q - o)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    t_in = np.max(np.minimum(t0, t1), axis=1)
    t_s = np.maximum(t_in[first], 1.0) * rng.uniform(0.2, 0.8, P)          # S in front of the box entry of the pair's first ray
    mid = o[first] + d[first] * t_s[:, None]
    e1 = unit(np.cross(d[first], rng.normal(size=(P, 3))))
    e2 = unit(np.cross(d[first], e1))
    half = (size * 1.5)[:, None] * (t_s / dist[first, 0])[:, None] + 0.05   # covers the fan of the pair's rays at that distance
    s_tri = np.concatenate([mid - e1 * half - e2 * half, mid + e1 * 2 * half - e2 * half, mid - e1 * half + e2 * 2 * half], axis=1)
    # three small decoys beside every T, off its plane (the rays never touch them): T then sits in a subtree of its own with them, and the search
    # reaches that subtree's node only after it has found S, which is when its children's boxes are weighed against the best distance
    cen = (a + b + c) / 3.0
    decoys = []
    for k in range(3):
        dc = cen + nn * rng.uniform(6, 15, (P, 1)) * rng.choice([-1.0, 1.0], (P, 1)) + u * rng.uniform(-0.2, 0.2, (P, 1)) * size[:, None]
        decoys.append(np.concatenate([dc, dc + u * 0.8, dc + v * 0.8], axis=1))
    tri = np.concatenate([np.concatenate([a, b, c], axis=1), s_tri] + decoys).astype(np.float32)
    return tri, o.astype(np.float32), unit(d).astype(np.float32)


def test_pruning_differs_from_the_unpruned_search_only_where_the_reference_returns_noise(gpu):
    """The dangerous configuration for distance pruning: 1,500 (far well-shaped triangle, nearer small triangle) pairs, the far
    one in a subtree of its own, and a million rays that run within 1e-7 .. 1e-3 rad of the far triangle's plane and hit the near one first.
    What the pruning rule guarantees, and what is held here: a ray on which the kernels and the reference's unpruned search disagree is a ray
    for which the reference's own answer is arithmetic noise: the triangle it returns is not hit at all in exact arithmetic, or the distance
    it returns for it is off from the true one by more than the pruning slack (2^-10 relative).  (Proof: the kernels skip a subtree only when
    the ray enters its box beyond the best distance by more than the slack; a triangle in it that is truly hit is hit inside that box, so a
    returned distance below the best one is short of the true one by more than the slack.)  Everywhere else ids and distances are bit-equal;
    the disagreeing rays number a few in a million even here, where every ray is aimed to provoke them."""
    from oracle import binding as ob
    import rustracer_amd as ra
    P = 1500
    tri, org, d = _grazing_pairs(P, 700, seed=21)
    nrm = np.tile(np.array([0, 0, 1, 0, 0, 1, 0, 0, 1], np.float32), (tri.shape[0], 1))
    mats = np.zeros((1, 8), np.uint32)
    mats[0, 1:4] = np.float32(0.5).view(np.uint32)
    args = (tri, nrm, np.zeros(tri.shape[0], np.uint32), np.zeros((0, 4), np.float32), np.zeros(0, np.uint32), mats)
    osc, gsc = ob.OracleScene.raw(*args), ra.Scene.from_arrays(*args)
    gi, gt = gsc.intersect_rays(org, d)
    oi, ot = _oracle_intersect_parallel(osc, org, d, workers=min(len(os.sched_getaffinity(0)), 16))
    gsc.set_option("graze_guard", 1)       # the per-ray guard (off by default: DESIGN.md section 4): a ray that has met a triangle almost in its plane goes on unpruned
    ui, ut = gsc.intersect_rays(org, d)
    gsc.set_option("graze_guard", 0)
    near_first = (oi >= P) & (oi < 2 * P)
    bad = np.flatnonzero((gi != oi) | (gt.view(np.uint32) != ot.view(np.uint32)))
    print("\nwinner: the near triangle %.3f, the far one %.3f, miss %.3f; rays that differ from the oracle: %d (%d with the per-ray guard)"
          % (near_first.mean(), (oi < P).mean(), (oi == 0xFFFFFFFF).mean(), bad.size, int(((ui != oi) | (ut.view(np.uint32) != ot.view(np.uint32))).sum())))
    assert near_first.mean() > 0.3 and (oi < P).mean() > 0.005
    assert bad.size <= 20, "%d of %d rays differ" % (bad.size, org.shape[0])
    assert int(((ui != oi) | (ut.view(np.uint32) != ot.view(np.uint32))).sum()) <= bad.size      # the guard never makes it worse
    # every disagreement: the reference's winner, examined in double precision on the same float32 inputs
    t64, o64, d64 = tri.astype(np.float64), org.astype(np.float64), d.astype(np.float64)
    for r in bad:
        assert oi[r] != 0xFFFFFFFF and gi[r] != 0xFFFFFFFF and ot[r] < gt[r], (r, gi[r], gt[r], oi[r], ot[r])    # the reference found something nearer than the kernels' hit
        A, B, Cc = t64[oi[r], 0:3], t64[oi[r], 3:6], t64[oi[r], 6:9]
        n = np.cross(B - A, Cc - A)
        den = float(np.dot(n, d64[r]))
        t_true = float(np.dot(n, A - o64[r])) / den if den != 0.0 else np.inf
        hit = o64[r] + t_true * d64[r] - A
        M = np.array([[np.dot(B - A, B - A), np.dot(B - A, Cc - A)], [np.dot(B - A, Cc - A), np.dot(Cc - A, Cc - A)]])
        b1, b2 = np.linalg.solve(M, np.array([np.dot(hit, B - A), np.dot(hit, Cc - A)])) if np.isfinite(t_true) else (np.nan, np.nan)
        inside = np.isfinite(t_true) and t_true > 0 and b1 >= 0 and b2 >= 0 and b1 + b2 <= 1
        noise = (not inside) or abs(float(ot[r]) - t_true) > 0.9 * 2.0 ** -10 * abs(t_true)
        assert noise, "ray %d: the reference's hit (%d, %r) is sound (true distance %r) and the kernels returned (%d, %r)" % (r, oi[r], ot[r], t_true, gi[r], gt[r])


def test_native_rccl_gather_at_world_1(scenes):
    """The multi-GPU frame path (rt_comm_init_rank + rt_gather_tiles = pack -> ncclAllGather -> unpack, include/rustracer_hip.h) executed
    on this GPU with a one-rank communicator: RCCL initialises, the collective runs on the render stream, and the frame comes back as
    rendered (RGBA8 and a 16-byte-per-pixel payload); a shard description that does not match the communicator is refused."""
    import torch
    from rustracer_amd import tiles
    from rustracer_amd._lib import RtError, check, lib
    osc, gsc = scenes("teapot")
    comm = tiles.Comm(1, 0, lambda ident: ident)
    try:
        op = oracle_params(width=200, height=150, spp=2, light_samples=2, bounces=2, tile=(32, 32), shard=(0, 1))
        p = to_rt_params(op)
        frame = torch.zeros((150, 200, 4), dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):                                   # the second frame reuses the staging buffers
            check(lib().rt_render_device(gsc.handle, C.byref(p), frame.data_ptr(), stream))
            comm.gather(p, frame.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), osc.render(op)["rgba"])
        wide = torch.arange(150 * 200 * 4, dtype=torch.int32, device="cuda:0").reshape(150, 200, 4)
        keep = wide.clone()
        comm.gather(p, wide.data_ptr(), stream, bytes_per_pixel=16)
        torch.cuda.synchronize()
        assert torch.equal(wide, keep)
        q = p.copy()
        q.shard_index, q.shard_count = 1, 2
        with pytest.raises(RtError):
            comm.gather(q, frame.data_ptr(), stream)
        # the host-buffer form the C++ Raytracer uses
        rgba = np.zeros((150, 200, 4), np.uint8)
        check(lib().rt_render_comm(gsc.handle, C.byref(p), comm._h, rgba.ctypes.data_as(C.c_void_p), None, None))
        assert np.array_equal(rgba, osc.render(op)["rgba"])
        assert comm.last_gather_ms() > 0.0 and "RCCL" in comm.describe() and "librccl" in comm.describe()
    finally:
        comm.destroy()


@pytest.mark.parametrize("lanes", [1, 3])
def test_a_failure_in_mid_frame_joins_the_lanes_and_still_enters_the_collective(scenes, lanes):
    """Every exit of the render joins the lane streams back into the caller's stream, and a rank whose render fails still takes part in the
    frame's collective (its peers would wait for ever otherwise) before it reports the error.  The failure is injected after N passes
    ("fail_after_passes"): the call returns RT_ERR_HIP, the caller's stream and the communicator stay usable, and the next frame is right."""
    import torch
    from rustracer_amd import tiles
    from rustracer_amd._lib import RT_ERR_HIP, RtError, check, lib
    osc, gsc = scenes("dragon_small")
    op = oracle_params(width=160, height=120, spp=6, light_samples=3, bounces=3, tile=(32, 32), shard=(0, 1), seed=77)
    p = to_rt_params(op)
    ref = osc.render(op)["rgba"]
    comm = tiles.Comm(1, 0, lambda ident: ident)
    gsc.set_option("lanes", lanes)
    gsc.set_option("lane_min_pixels", 64)
    gsc.set_option("max_paths", 160 * 120 // lanes)          # 6 passes per lane
    try:
        frame = torch.zeros((120, 160, 4), dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        rgba = np.zeros((120, 160, 4), np.uint8)
        for fail_at in (1, 2 * lanes):
            gsc.set_option("fail_after_passes", fail_at)
            rc = lib().rt_render_device(gsc.handle, C.byref(p), frame.data_ptr(), stream)
            assert rc == RT_ERR_HIP and b"injected failure" in lib().rt_last_error()
            torch.cuda.synchronize()                          # nothing is left running on a lane stream: the frame is the caller's again
            probe = (frame.to(torch.int32).sum() >= 0).item()
            assert probe
            with pytest.raises(RtError) as ei:
                check(lib().rt_render_comm(gsc.handle, C.byref(p), comm._h, rgba.ctypes.data_as(C.c_void_p), None, None))
            assert ei.value.code == RT_ERR_HIP and "injected failure" in str(ei.value)
            gather_after_failure = comm.last_gather_ms()
            assert gather_after_failure > 0.0                 # the collective ran although the render had failed
            gsc.set_option("fail_after_passes", 0)
            check(lib().rt_render_comm(gsc.handle, C.byref(p), comm._h, rgba.ctypes.data_as(C.c_void_p), None, None))
            assert np.array_equal(rgba, ref), (lanes, fail_at)
            check(lib().rt_render_device(gsc.handle, C.byref(p), frame.data_ptr(), stream))
            torch.cuda.synchronize()
            assert np.array_equal(frame.cpu().numpy(), ref), (lanes, fail_at)
    finally:
        gsc.set_option("fail_after_passes", 0)
        gsc.set_option("lanes", 2)
        gsc.set_option("lane_min_pixels", 65536)
        gsc.set_option("max_paths", 0)
        comm.destroy()


@pytest.mark.parametrize("lanes", [1, 2])
def test_partial_frames_refine_and_equal_the_oracle_at_the_samples_done(scenes, lanes):
    """Control surface (raytracer.rs:203-214, canvas.rs:179): while rt_render runs, the host buffer a canvas polls shows the frame fill and
    refine.  With the samples spread over passes of 3, the buffer after k passes is the mean of the first 3k samples of every pixel, which is
    bit for bit the oracle's image at 3k spp; an interrupt leaves that partial frame in place; the finished frame equals the oracle at full spp.
    With two render lanes a round is one pass of each lane over its half of the pixels: the same partial frames."""
    import threading
    from rustracer_amd._lib import RT_ERR_INTERRUPTED, lib, ptr
    osc, gsc = scenes("teapot")
    W, H, SPP, CH = 96, 64, 12, 3
    kw = dict(width=W, height=H, light_samples=2, bounces=3)
    refs = {k: osc.render(oracle_params(spp=k, **kw))["rgba"] for k in range(CH, SPP + 1, CH)}
    p = to_rt_params(oracle_params(spp=SPP, **kw))
    gsc.set_option("max_paths", W * H // lanes * CH)
    gsc.set_option("lanes", lanes)
    gsc.set_option("lane_min_pixels", 64)
    try:
        rgba = np.zeros((H, W, 4), np.uint8)
        prog, intr = C.c_uint32(0), C.c_uint32(0)
        seen, done = [], threading.Event()

        def poll():                                   # the canvas: snapshots while the render runs
            last = -1
            while not done.is_set():
                v = prog.value
                if v != last and 0 < v < 100:
                    seen.append((v, rgba.copy()))
                    last = v
        t = threading.Thread(target=poll)
        t.start()
        rc = lib().rt_render(gsc.handle, C.byref(p), ptr(rgba), C.byref(prog), C.byref(intr))
        done.set()
        t.join()
        assert rc == 0 and np.array_equal(rgba, refs[SPP])
        # every snapshot taken at a pass boundary is one of the partial images (a snapshot may straddle a refresh: skip torn ones)
        whole = [s for _, s in seen if any(np.array_equal(s, r) for r in refs.values())]
        assert len(seen) == 0 or len(whole) >= 1
        # deterministic form: interrupt after the first pass has been reported.  One lane looks at the flag between bounce levels, so raising it
        # after the third of four passes still stops the fourth; several lanes look between rounds only, and the last round may get through
        for stop_at in ((1, 2, 3) if lanes == 1 else (1, 2)):
            rgba[:] = 0
            prog.value, intr.value = 0, 0
            box = {}

            def stopper():
                while not done2.is_set():
                    if prog.value >= int(100 * stop_at / (SPP // CH)):
                        intr.value = 1
                        return
            done2 = threading.Event()
            t = threading.Thread(target=stopper)
            t.start()
            rc = lib().rt_render(gsc.handle, C.byref(p), ptr(rgba), C.byref(prog), C.byref(intr))
            done2.set()
            t.join()
            assert rc == RT_ERR_INTERRUPTED
            k_done = [k for k, r in refs.items() if np.array_equal(rgba, r)]
            assert k_done and stop_at * CH <= k_done[0] < SPP, "the interrupted frame is not the oracle's image at a multiple of %d spp" % CH
    finally:
        gsc.set_option("max_paths", 0)
        gsc.set_option("lanes", 2)
        gsc.set_option("lane_min_pixels", 65536)


@pytest.mark.parametrize("name", ["teapot", "specular", "dragon_small", "dragon_full", "chain"])
def test_device_built_trees_equal_the_host_built(gpu, teapot_obj, name):
    """BVH::build (mod.rs:99) and the 8-wide occlusion tree built on the GPU are the host builders' trees word for word
    (node words, pre-order numbering, ranks; wide nodes and primitive records), from 13 objects to the 871,210 of the full stand-in."""
    import rustracer_amd as ra
    from rustracer_amd._lib import check, lib

    def make():
        if name == "dragon_small":
            return ra.Scene.from_dragon_mesh(*ra.dragon_standin_mesh(500, 100))
        if name == "dragon_full":
            return ra.Scene.from_dragon_mesh(*ra.dragon_standin_mesh(*ra.DRAGON_STANDIN_FULL))
        if name == "chain":                                   # a 31-deep chain of spheres: halved splits, leaves at every level
            n = 74
            z = -(1e-4 * 2.0 ** np.arange(n))
            sph = np.stack([np.zeros(n) + 0.25, np.zeros(n) + 0.5, z, 0.02 * np.abs(z)], axis=1).astype(np.float32)
            mats = np.zeros((1, 8), np.uint32)
            return ra.Scene.from_arrays(np.zeros((0, 9)), np.zeros((0, 9)), [], sph, np.zeros(n, np.uint32), mats)
        return ra.Scene._named(name, teapot_obj)

    L = lib()
    try:
        check(L.rt_set_global_option(b"device_build_min", 2))
        check(L.rt_set_global_option(b"device_build", 0))
        host = make()
        check(L.rt_set_global_option(b"device_build", 1))
        dev = make()
    finally:
        check(L.rt_set_global_option(b"device_build", 1))
        check(L.rt_set_global_option(b"device_build_min", 4096))
    hi, di = host.build_info(), dev.build_info()
    assert not hi["binary_on_device"] and di["binary_on_device"], di
    print("\n%s: %d objects; binary tree host %.1f ms, GPU %.1f ms (%.1f between events); 8-wide tree host %.1f ms, %s %.1f ms (%.1f between events)"
          % (name, dev.info.n_tri + dev.info.n_sph, hi["binary_ms"], di["binary_ms"], di["binary_device_ms"], hi["wide_ms"],
             "GPU" if di["wide_on_device"] else "host", di["wide_ms"], di["wide_device_ms"]))
    hw, hr = host.bvh()
    dw, dr = dev.bvh()
    assert host.info.bvh_depth == dev.info.bvh_depth and host.info.n_inner_nodes == dev.info.n_inner_nodes
    assert np.array_equal(hr, dr)
    assert np.array_equal(hw, dw), "%d node words differ, first node %d" % (int((hw != dw).sum()), int(np.argwhere(hw != dw)[0][0]))
    hn, hp = host.wide_bvh()
    dn, dp = dev.wide_bvh()
    assert host.info.wide_depth == dev.info.wide_depth
    assert np.array_equal(hn, dn) and np.array_equal(hp.view(np.uint32), dp.view(np.uint32))
    # ... and the GPU-built binary tree against the ORACLE's own recursive builder (oracle_rt.cpp, bvh 0.3.2 BVHNode::build restated), directly:
    # every inner node's two child boxes and child references, in pre-order, and the leaf ranks = the reference's visiting order
    from oracle import binding as ob
    from test_host_scene import _oracle_inner_words
    if name == "dragon_small":
        osc = ob.OracleScene.dragon_mesh(*ra.dragon_standin_mesh(500, 100))
    elif name == "dragon_full":
        osc = ob.OracleScene.dragon_mesh(*ra.dragon_standin_mesh(*ra.DRAGON_STANDIN_FULL))
    elif name == "chain":
        e = dev.export()
        osc = ob.OracleScene.raw(e["tri_pos"], e["tri_nrm"], e["tri_mat"], e["sph"], e["sph_mat"], e["mats"])
    else:
        osc = ob.OracleScene.named(name, teapot_obj)
    ow = _oracle_inner_words(osc)
    assert ow.shape == dw.shape and np.array_equal(ow[:, :14], dw[:, :14]), "GPU-built tree differs from the oracle's in %d node words" % int((ow[:, :14] != dw[:, :14]).sum())
    order, stack = [], [int(dev.info.n_inner_nodes and 0)]
    if dev.info.n_inner_nodes:
        while stack:
            ref = stack.pop()
            if ref & 0x80000000:
                order.append(ref & 0x7FFFFFFF)
                continue
            stack.append(int(ow[ref, 12]))          # left pushed, right pushed: right popped first (mod.rs:427-432)
            stack.append(int(ow[ref, 13]))
        assert np.array_equal(dr[np.array(order, np.int64)], np.arange(len(order), dtype=dr.dtype))
