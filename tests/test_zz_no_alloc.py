"""rt_cast_primary_noalloc (include/rustracer_hip.h): the primary cast for callers that cannot allocate.  Its kernel's
scratch (work counters + traversal-stack overflow rows) is a static device array of the library; ids, distances and
counters must equal rt_cast_primary_device's, bit for bit, on two calls in a row, the second with a smaller input."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cast(fn, scene, p, torch, must_not_allocate=False):
    from rustracer_amd._lib import check
    ids = torch.full((p.height, p.width), 0x7FFFFFFF, dtype=torch.int32, device="cuda:0")
    ts = torch.full((p.height, p.width), -1.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]              # hipMemGetInfo: any hipMalloc / hipFree inside the call would move it
    check(fn(scene.handle, C.byref(p), ids.data_ptr(), ts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    if must_not_allocate:
        assert torch.cuda.mem_get_info()[0] == free_before, "the no-alloc entry point changed the device's free memory"
    cnt = scene.counters()
    return ids.cpu().numpy().view(np.uint32), ts.cpu().numpy(), cnt


def _chain_scene(n, ratio, z0):
    """Spheres whose centres are spaced geometrically along -z: the SAH buckets peel them off one or two at a time, so the
    tree is a chain about n / 2 deep, and a ray down the axis keeps one postponed subtree per level on its stack."""
    import rustracer_amd as ra
    z = -(z0 * ratio ** np.arange(n))
    sph = np.stack([np.zeros(n), np.zeros(n), z, 0.02 * np.abs(z)], axis=1).astype(np.float32)
    mats = np.zeros((1, 8), np.uint32)                          # one diffuse, non-emissive material
    mats[0, 1:4] = np.float32(0.5).view(np.uint32)
    return ra.Scene.from_arrays(np.zeros((0, 9)), np.zeros((0, 9)), [], sph, np.zeros(n, np.uint32), mats)


@pytest.mark.parametrize("which", ["dragon_small", "chain"])
def test_no_alloc_cast_equals_device_cast(gpu, which):
    import torch
    import rustracer_amd as ra
    from rustracer_amd._lib import RtError, lib
    from rustracer_amd.raytracer import Config, RenderConfig, make_params
    from rustracer_amd._lib import RT_MODE_DEBUG

    if which == "dragon_small":
        vp, vn, f = ra.dragon_standin_mesh(500, 100)      # 100,000 triangles, BVH depth 23
        scene = ra.Scene.from_dragon_mesh(vp, vn, f)
    else:
        scene = _chain_scene(74, 2.0, 1e-4)               # BVH depth 31 of the 32 the scratch holds: axial rays fill the overflow rows
        assert 24 <= scene.info.bvh_depth <= 32
    L = lib()

    def params(w, h, shard=(0, 1)):
        cfg = Config(screen_width=w, screen_height=h)
        return make_params(RenderConfig(cfg), (0.0, 0.0, 0.0), RT_MODE_DEBUG, tile=(32, 32), shard=shard)

    # before the scene is resident the entry point must refuse (uploading would allocate)
    with pytest.raises(RtError):
        _cast(L.rt_cast_primary_noalloc, scene, params(64, 64), torch)

    scene.set_option("stats", 1)                            # traversal counters on: they live in the scratch too
    for w, h, shard in ((1280, 720, (0, 1)), (333, 177, (0, 1)), (640, 360, (1, 3))):   # second and third inputs are smaller
        p = params(w, h, shard)
        ref_ids, ref_t, ref_cnt = _cast(L.rt_cast_primary_device, scene, p, torch)
        got_ids, got_t, got_cnt = _cast(L.rt_cast_primary_noalloc, scene, p, torch, must_not_allocate=True)
        assert np.array_equal(got_ids, ref_ids), "%d ids differ at %dx%d" % (int((got_ids != ref_ids).sum()), w, h)
        assert np.array_equal(got_t.view(np.uint32), ref_t.view(np.uint32))
        for k in ("rays_closest", "samples", "closest_inner", "closest_tris", "closest_sphs"):
            assert got_cnt[k] == ref_cnt[k], (k, got_cnt[k], ref_cnt[k])
        assert got_cnt["closest_inner"] > 0
    # and again without the counters
    scene.set_option("stats", 0)
    p = params(200, 120)
    ref_ids, ref_t, _ = _cast(L.rt_cast_primary_device, scene, p, torch)
    got_ids, got_t, _ = _cast(L.rt_cast_primary_noalloc, scene, p, torch, must_not_allocate=True)
    assert np.array_equal(got_ids, ref_ids) and np.array_equal(got_t.view(np.uint32), ref_t.view(np.uint32))
    # what the entry point exists for: it can be recorded into a HIP graph (stream capture forbids hipMalloc / hipFree and every
    # synchronising call), and the replayed graph gives the same ids and distances
    from rustracer_amd._lib import check
    ids = torch.full((p.height, p.width), 0x7FFFFFFF, dtype=torch.int32, device="cuda:0")
    ts = torch.full((p.height, p.width), -1.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        check(L.rt_cast_primary_noalloc(scene.handle, C.byref(p), ids.data_ptr(), ts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    ids.fill_(0x7FFFFFFF)
    ts.fill_(-1.0)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy().view(np.uint32), ref_ids) and np.array_equal(ts.cpu().numpy().view(np.uint32), ref_t.view(np.uint32))
    # the counters live in the GPU's shared scratch: once another scene has made a no-alloc call there, this scene's are gone and say so
    other = _chain_scene(20, 2.0, 1e-3)
    q = params(64, 64)
    _cast(L.rt_cast_primary_device, other, q, torch)
    _cast(L.rt_cast_primary_noalloc, scene, p, torch)
    assert scene.counters()["samples"] == p.width * p.height
    _cast(L.rt_cast_primary_noalloc, other, q, torch)
    with pytest.raises(RtError):
        scene.counters()
    assert other.counters()["samples"] == 64 * 64


def test_no_alloc_rejects_a_tree_deeper_than_its_scratch(gpu):
    """A degenerate scene whose SAH tree is a chain (every object nested around the next): deeper than 32 levels."""
    import torch
    import rustracer_amd as ra
    from rustracer_amd._lib import RT_ERR_LIMIT, RT_MODE_DEBUG, RtError, lib
    from rustracer_amd.raytracer import Config, RenderConfig, make_params
    scene = _chain_scene(60, 2.5, 2e-5)
    p = make_params(RenderConfig(Config(screen_width=64, screen_height=64)), (0.0, 0.0, 0.0), RT_MODE_DEBUG, tile=(32, 32), shard=(0, 1))
    ids, ts, _ = _cast(lib().rt_cast_primary_device, scene, p, torch)
    if scene.info.bvh_depth <= 32:
        pytest.skip("this scene's tree is only %d deep" % scene.info.bvh_depth)
    with pytest.raises(RtError) as e:
        _cast(lib().rt_cast_primary_noalloc, scene, p, torch)
    assert e.value.code == RT_ERR_LIMIT
