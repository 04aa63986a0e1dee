"""Host side of the product library (no GPU needed): the C ABI exports, the scene constructors and the
BVH flattener, checked against the oracle's independent implementation of the same reference code."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rustracer_amd as ra
from oracle import binding as ob
from rustracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rustracer_hip.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_between_oracle_and_abi():
    assert C.sizeof(_lib.RenderParams) == C.sizeof(ob.RenderParams) == 88
    for (n1, _), (n2, _) in zip(_lib.RenderParams._fields_, ob.RenderParams._fields_):
        assert n1 == n2 and getattr(_lib.RenderParams, n1).offset == getattr(ob.RenderParams, n2).offset
    assert C.sizeof(_lib.Material) == 32


def test_screen_params_equal_oracle():
    for w, h in ((600, 600), (1920, 1080), (3840, 2160), (333, 177)):
        p = _lib.RenderParams()
        fov = float(np.float32(90.0) * (np.float32(np.pi) / np.float32(180.0)))
        _lib.lib().rt_screen_params(w, h, C.c_float(fov), C.byref(p))
        q = ob.make_params(w, h)
        for f in ("width", "height", "inv_w", "inv_h", "start", "total", "aspect_ratio", "z"):
            assert getattr(p, f) == getattr(q, f), f


def _oracle_inner_words(osc):
    """The oracle's node array (with leaf nodes) in the product's inner-only pre-order layout."""
    ow = osc.bvh()
    is_leaf = ow[:, 0] == 1
    inner_idx = np.cumsum(~is_leaf) - 1

    def ref(ci):
        return np.where(is_leaf[ci], 0x80000000 | ow[ci, 3], inner_idx[ci]).astype(np.uint32)
    oi = ow[~is_leaf]
    om = np.zeros((oi.shape[0], 16), np.uint32)
    om[:, 0:12] = oi[:, 4:16]
    om[:, 12] = ref(oi[:, 1])
    om[:, 13] = ref(oi[:, 2])
    return om


@pytest.mark.parametrize("name", ["triangle", "specular", "diffuse", "teapot", "specular_teapot", "sphere_on_plane"])
def test_named_scenes_and_bvh_equal_oracle(name, teapot_obj):
    o = ob.OracleScene.named(name, teapot_obj)
    p = ra.Scene._named(name, teapot_obj)
    eo, ep = o.export(), p.export()
    for k in eo:
        assert np.array_equal(eo[k].view(np.uint32), ep[k].view(np.uint32)), k
    words, rank = p.bvh()
    assert np.array_equal(words, _oracle_inner_words(o))
    assert sorted(rank.tolist()) == list(range(p.info.n_tri + p.info.n_sph))
    assert p.info.n_inner_nodes == max(p.info.n_tri + p.info.n_sph - 1, 0)


def test_rank_is_the_reference_dfs_order():
    """rank = visiting order of the reference traversal: left pushed, right pushed, right popped first (mod.rs:427-432)."""
    p = ra.Scene.new_specular()
    words, rank = p.bvh()
    order, stack = [], [0]
    while stack:
        ref = stack.pop()
        if ref & 0x80000000:
            order.append(ref & 0x7FFFFFFF)
            continue
        stack.append(int(words[ref, 12]))
        stack.append(int(words[ref, 13]))
    assert [int(rank[o]) for o in order] == list(range(len(order)))


def test_dragon_standin_is_deterministic_and_bvh_equals_oracle():
    vp, vn, f = ra.dragon_standin_mesh(120, 40)
    vp2, _, _ = ra.dragon_standin_mesh(120, 40)
    assert np.array_equal(vp, vp2) and f.shape == (9600, 3) and f.max() == 120 * 40 - 1
    import zlib
    assert zlib.crc32(vp.tobytes()) == zlib.crc32(vp2.tobytes())
    assert np.allclose(np.linalg.norm(vn, axis=1), 1.0, atol=1e-5)
    o = ob.OracleScene.dragon_mesh(vp, vn, f)
    p = ra.Scene.from_dragon_mesh(vp, vn, f)
    assert np.array_equal(p.bvh()[0], _oracle_inner_words(o))
    e = p.export()
    tp = e["tri_pos"][10:].reshape(-1, 3)          # placed by scale 2, offset (-20/3, -20, -48 - 40/3): inside the Cornell box
    assert tp[:, 0].min() > -20 and tp[:, 0].max() < 20 and tp[:, 1].min() > -20.01 and tp[:, 2].min() > -68
    assert list(p.light_indexes()) == [10 + 9600]


def test_threaded_builders_equal_the_sequential_ones_and_the_oracle(monkeypatch):
    """51,200 triangles: large enough for build_bvh to hand subtrees to other threads and for build_wide_bvh to split its levels.
    Node indices (pre-order), ranks (right-first DFS) and the 8-wide tree must not depend on the thread count."""
    vp, vn, f = ra.dragon_standin_mesh(400, 64)
    built = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RT_HOST_THREADS", threads)
        p = ra.Scene.from_dragon_mesh(vp, vn, f)
        words, rank = p.bvh()
        nodes, prims = p.wide_bvh()
        built[threads] = (words.copy(), rank.copy(), nodes.copy(), prims.copy(), p.info.bvh_depth, p.info.wide_depth)
    for threads in ("3", "8"):
        for a, b in zip(built["1"][:4], built[threads][:4]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), threads
        assert built["1"][4:] == built[threads][4:]
    words, rank = built["8"][0], built["8"][1]
    o = ob.OracleScene.dragon_mesh(vp, vn, f)
    assert np.array_equal(words, _oracle_inner_words(o))
    order, stack = [], [0]                                  # rank = the reference's visiting order (mod.rs:427-432)
    while stack:
        ref = stack.pop()
        if ref & 0x80000000:
            order.append(ref & 0x7FFFFFFF)
            continue
        stack.append(int(words[ref, 12]))
        stack.append(int(words[ref, 13]))
    assert np.array_equal(rank[np.array(order)], np.arange(len(order), dtype=rank.dtype))


def test_obj_loader_polygons_first_model_and_missing_normals(tmp_path):
    """Scene::load_obj (mod.rs:135-189): fan triangulation, models[0] only, flat normals when `vn` is absent."""
    path = tmp_path / "dragon.obj"
    path.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 5\ng first\nf 1 2 3 4\ng second\nf 1 2 5\n")
    p = ra.Scene._named("dragon", str(path))
    o = ob.OracleScene.named("dragon", str(path))
    assert p.info.n_tri == 10 + 2 == o.T
    ep, eo = p.export(), o.export()
    assert np.array_equal(ep["tri_pos"].view(np.uint32), eo["tri_pos"].view(np.uint32))
    assert np.array_equal(ep["tri_nrm"].view(np.uint32), eo["tri_nrm"].view(np.uint32))
    a = ep["tri_pos"][10].reshape(3, 3)
    off = np.array([np.float32(-20.0) / np.float32(3.0), -20.0, np.float32(-48.0) - np.float32(2.0) * np.float32(20.0) / np.float32(3.0)], np.float32)
    assert np.array_equal(a[1], off + np.array([2, 0, 0], np.float32))      # scale 2.0 (mod.rs:200)
    assert np.allclose(ep["tri_nrm"][10][:3], [0, 0, 1])
    with pytest.raises(_lib.RtError):
        ra.Scene._named("dragon", str(tmp_path / "nope.obj"))                # the reference panics (mod.rs:136); the ABI returns a code


def test_scene_create_validates_arguments():
    mats = np.zeros((1, 8), np.uint32)
    with pytest.raises(_lib.RtError):
        ra.Scene.from_arrays(np.zeros((1, 9)), np.zeros((1, 9)), [1], np.zeros((0, 4)), [], mats)
    bad = mats.copy()
    bad[0, 0] = 7
    with pytest.raises(_lib.RtError):
        ra.Scene.from_arrays(np.zeros((1, 9)), np.zeros((1, 9)), [0], np.zeros((0, 4)), [], bad)
    s = ra.Scene.from_arrays(np.zeros((0, 9)), np.zeros((0, 9)), [], np.zeros((0, 4)), [], np.zeros((0, 8), np.uint32))
    assert s.info.n_inner_nodes == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librustracer_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_tile_index_map_partitions_the_frame():
    from rustracer_amd import tiles
    w, h, tw, th, world = 100, 70, 32, 16, 3
    seen = np.zeros(w * h, np.int32)
    for r in range(world):
        idx = tiles.tile_index_map(w, h, tw, th, r, world)
        assert idx.shape[0] == tiles.tiles_per_shard(w, h, tw, th, world) * tw * th
        np.add.at(seen, idx[idx >= 0], 1)
    assert (seen == 1).all()
    p = _lib.RenderParams()
    p.width, p.height, p.tile_w, p.tile_h, p.shard_count = w, h, tw, th, world
    assert _lib.lib().rt_tiles_per_shard(C.byref(p)) == tiles.tiles_per_shard(w, h, tw, th, world)
