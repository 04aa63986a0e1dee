"""The 8-wide occlusion BVH that Scene::is_occluded rays traverse on the device (host_scene.h WideBVH).

It has no reference counterpart, so what is pinned here is the property the drop-in claim rests on:
the wide tree is a conservative candidate filter.  (1) structure: every non-emissive object is in it
exactly once, emissive ones never (src/scene/mod.rs:481), and every quantised child box contains, in
exact arithmetic, everything below it; (2) candidates: for random rays, every object whose own box the
reference's slab test passes (the objects Scene::is_occluded reaches, mod.rs:462-477) is reached by an
emulation of the kernel's padded single-precision node test.  No GPU needed."""
import numpy as np
import pytest

import rustracer_amd as ra

F = np.float32
PRIM = 0x80000000


def _scene(name, teapot_obj):
    if name == "dragon_tiny":
        return ra.Scene.from_dragon_mesh(*ra.dragon_standin_mesh(60, 20))
    return ra.Scene._named(name, teapot_obj)


def _own_box(rec):
    """Bounded for Triangle / Sphere (src/scene/objects.rs:293-316) from a primitive record."""
    if rec[3:4].view(np.uint32)[0] == 0:
        v = np.stack([rec[0:3], rec[4:7], rec[8:11]])
        return v.min(axis=0), v.max(axis=0), int(rec[7:8].view(np.uint32)[0])
    c, r = rec[0:3], rec[4]
    return (c - r).astype(F), (c + r).astype(F), int(rec[5:6].view(np.uint32)[0])


def _decode(node):
    p = node[0:3].view(F).astype(np.float64)
    meta = int(node[3])
    e = np.array([(meta >> (8 * a)) & 0xFF for a in range(3)], np.int64) - 128
    n_inner, n_child = (meta >> 24) & 0xF, meta >> 28
    q = node[8:20].view(np.uint8).reshape(6, 8).astype(np.float64)     # lo_x lo_y lo_z hi_x hi_y hi_z
    step = np.ldexp(1.0, e)
    lo = p[:, None] + q[0:3] * step[:, None]
    hi = p[:, None] + q[3:6] * step[:, None]
    return p, e, n_inner, n_child, int(node[4]), int(node[5]), lo, hi, q


@pytest.mark.parametrize("name", ["specular", "teapot", "specular_teapot", "sphere_on_plane", "dragon_tiny"])
def test_wide_bvh_is_a_conservative_cover_of_the_blockers(name, teapot_obj):
    sc = _scene(name, teapot_obj)
    nodes, prims = sc.wide_bvh()
    e = sc.export()
    T = sc.info.n_tri
    emissive = set(int(i) for i in sc.light_indexes())
    blockers = set(range(T + sc.info.n_sph)) - emissive
    assert nodes.shape[0] > 0 and prims.shape[0] == len(blockers)

    seen_nodes, seen_objs = set(), []

    def walk(ni):
        """Returns the exact bounds (float64) of everything below node ni."""
        assert ni not in seen_nodes
        seen_nodes.add(ni)
        p, ex, n_inner, n_child, child_base, prim_base, lo, hi, q = _decode(nodes[ni])
        assert 1 <= n_child <= 8 and n_inner <= n_child
        assert (q[0:3, n_child:] == 255).all() and (q[3:6, n_child:] == 0).all()
        mn_all, mx_all = np.full(3, np.inf), np.full(3, -np.inf)
        for s in range(n_child):
            if s < n_inner:
                mn, mx = walk(child_base + s)
            else:
                bmn, bmx, obj = _own_box(prims[prim_base + s - n_inner])
                seen_objs.append(obj)
                mn, mx = bmn.astype(np.float64), bmx.astype(np.float64)
                # the record holds the object's own floats
                if obj < T:
                    assert np.array_equal(prims[prim_base + s - n_inner][[0, 1, 2, 4, 5, 6, 8, 9, 10]].view(np.uint32), e["tri_pos"][obj].view(np.uint32))
                else:
                    assert np.array_equal(prims[prim_base + s - n_inner][[0, 1, 2, 4]].view(np.uint32), e["sph"][obj - T].view(np.uint32))
            assert (lo[:, s] <= mn).all() and (hi[:, s] >= mx).all(), (ni, s)
            mn_all, mx_all = np.minimum(mn_all, mn), np.maximum(mx_all, mx)
        assert (p <= mn_all).all() and (p + 255.0 * np.ldexp(1.0, ex) >= mx_all).all()
        return mn_all, mx_all

    walk(0)
    assert seen_nodes == set(range(nodes.shape[0]))
    assert sorted(seen_objs) == sorted(blockers)


def _slab_exact(o, inv, neg, mn, mx):
    """bvh 0.3.2 Ray::intersects_aabb in float32 (finite 1/d): max of the near planes <= min of the far planes, min far > 0."""
    near = np.where(neg, mx, mn)
    far = np.where(neg, mn, mx)
    tn = ((near - o).astype(F) * inv).astype(F)
    tf = ((far - o).astype(F) * inv).astype(F)
    return tn.max() <= tf.min() and tf.min() > 0


def _reached_by_kernel_test(nodes, prims, o, inv, neg, lim):
    """Emulation of occluded() in rt_kernels.h: padded fma form per child slot, all hit slots followed."""
    out, stack = set(), [0]
    while stack:
        ni = stack.pop()
        p, ex, n_inner, n_child, child_base, prim_base, lo, hi, q = _decode(nodes[ni])
        s = np.ldexp(inv.astype(np.float64), ex).astype(F)                       # exact scaling
        c = ((nodes[ni][0:3].view(F) - o).astype(F) * inv).astype(F)
        pad = ((F(256.0) * np.abs(s).astype(np.float64) + np.abs(c).astype(np.float64)).astype(F) * F(2.0 ** -20)).astype(F)
        cn, cf = (c - pad).astype(F), (c + pad).astype(F)
        for sl in range(n_child):
            qn = np.where(neg, q[3:6, sl], q[0:3, sl])
            qf = np.where(neg, q[0:3, sl], q[3:6, sl])
            tn = (qn * s.astype(np.float64) + cn.astype(np.float64)).astype(F)   # fma: one rounding
            tf = (qf * s.astype(np.float64) + cf.astype(np.float64)).astype(F)
            if max(tn.max(), F(0.0)) <= min(tf.min(), lim):
                if sl < n_inner:
                    stack.append(child_base + sl)
                else:
                    out.add(prim_base + sl - n_inner)
    return out


@pytest.mark.parametrize("name", ["teapot", "dragon_tiny"])
def test_every_object_the_reference_reaches_is_a_candidate(name, teapot_obj):
    sc = _scene(name, teapot_obj)
    nodes, prims = sc.wide_bvh()
    boxes = [_own_box(prims[i]) for i in range(prims.shape[0])]
    rng = np.random.default_rng(7)
    n_checked = 0
    for k in range(60):
        # origins on and around the geometry (shadow rays start on surfaces), directions anywhere, some axis-aligned-ish
        rec = prims[rng.integers(prims.shape[0])]
        o = (rec[0:3] + (rng.standard_normal(3) * (0.0 if k % 3 == 0 else 2.0))).astype(F)
        d = rng.standard_normal(3)
        if k % 7 == 0:
            d[rng.integers(3)] *= 1e-6
        d = (d / np.linalg.norm(d)).astype(F)
        dn = (d / np.sqrt((d * d).sum(dtype=F), dtype=F)).astype(F)      # bvh::Ray::new normalises again (mod.rs:506-512)
        inv = (F(1.0) / dn).astype(F)
        neg = dn < 0
        reached = _reached_by_kernel_test(nodes, prims, o, inv, neg, F(np.inf))
        for i, (mn, mx, _) in enumerate(boxes):
            if _slab_exact(o, inv, neg, mn, mx):
                n_checked += 1
                assert i in reached, (k, i)
    assert n_checked > 200
