"""One-off wide parity check of BASELINE config 4 (or 5: third argument) at its full spp: GPU frame against the oracle on every K-th 32x32
tile, shard offset J."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rustracer_amd as ra
from oracle import binding as ob
from test_gpu_parity import oracle_params, to_rt_params, gpu_render, _own_tiles

J, K = int(sys.argv[1]), int(sys.argv[2])
CONFIG = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ra._lib.check(ra._lib.lib().rt_init(0))
if CONFIG == 4:
    vp, vn, f = ra.dragon_standin_mesh(*ra.DRAGON_STANDIN_FULL)
    osc, gsc = ob.OracleScene.dragon_mesh(vp, vn, f), ra.Scene.from_dragon_mesh(vp, vn, f)
    W, H, SPP = 1920, 1080, 256
else:       # config 5: new_specular's spheres + the teapot, 3840x2160, 1024 spp
    from conftest import TEAPOT_OBJ as teapot
    osc, gsc = ob.OracleScene.named("specular_teapot", teapot), ra.Scene._named("specular_teapot", teapot)
    W, H, SPP = 3840, 2160, 1024
op = oracle_params(width=W, height=H, spp=SPP, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(J, K))
t0 = time.time(); ref = osc.render(op); t1 = time.time()
gsc.set_option("lane_min_pixels", 64)
got, cnt = gpu_render(gsc, to_rt_params(op)); t2 = time.time()
tiles_x = (W + 31) // 32
tiles = [((t % tiles_x) * 32, (t // tiles_x) * 32) for t in range(J, tiles_x * ((H + 31) // 32), K)]
bad = sum(int(not np.array_equal(got[y:y + 32, x:x + 32], ref["rgba"][y:y + 32, x:x + 32])) for x, y in tiles)
print("tiles %d, differing %d; rays closest %d == %d, shadow %d == %d; oracle %.1f s, gpu %.1f s" % (
    len(tiles), bad, cnt["rays_closest"], ref["intersect_calls"], cnt["rays_shadow"], ref["occluded_calls"], t1 - t0, t2 - t1))
assert bad == 0 and cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
