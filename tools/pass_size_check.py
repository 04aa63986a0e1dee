"""One-off check of the largest pass the library runs (2^31 light-sample slots: 64 Mi paths x 32 light samples per lane): the whole config-4 frame
at 64 spp with the default pass size against the same frame rendered in passes of 8 Mi paths on three lanes (GPU vs GPU: the image must not
depend on the pass size), and against the oracle on every K-th 32x32 tile."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rustracer_amd as ra
from oracle import binding as ob
from test_gpu_parity import oracle_params, to_rt_params, gpu_render

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ra._lib.check(ra._lib.lib().rt_init(0))
vp, vn, f = ra.dragon_standin_mesh(*ra.DRAGON_STANDIN_FULL)
osc, gsc = ob.OracleScene.dragon_mesh(vp, vn, f), ra.Scene.from_dragon_mesh(vp, vn, f)
W, H, SPP = 1920, 1080, 64
full = oracle_params(width=W, height=H, spp=SPP, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(0, 1))
t0 = time.time(); a, ca = gpu_render(gsc, to_rt_params(full)); t1 = time.time()
print("default pass size: lanes %d, max_paths %d, %.1f s, rays closest %d shadow %d" % (gsc.get_option("lanes"), gsc.get_option("max_paths"), t1 - t0, ca["rays_closest"], ca["rays_shadow"]), flush=True)
gsc.set_option("lanes", 3); gsc.set_option("max_paths", 1 << 23)
b, cb = gpu_render(gsc, to_rt_params(full))
print("8 Mi paths, 3 lanes: bytes differing %d, rays closest %d shadow %d" % (int((a != b).sum()), cb["rays_closest"], cb["rays_shadow"]), flush=True)
assert np.array_equal(a, b) and ca["rays_closest"] == cb["rays_closest"] and ca["rays_shadow"] == cb["rays_shadow"]
sub = oracle_params(width=W, height=H, spp=SPP, light_samples=32, bounces=8, seed=0x5EED, tile=(32, 32), shard=(5, K))
t0 = time.time(); ref = osc.render(sub); t1 = time.time()
tiles_x = (W + 31) // 32
tiles = [((t % tiles_x) * 32, (t // tiles_x) * 32) for t in range(5, tiles_x * ((H + 31) // 32), K)]
bad = sum(int(not np.array_equal(a[y:y + 32, x:x + 32], ref["rgba"][y:y + 32, x:x + 32])) for x, y in tiles)
print("oracle on %d tiles (%.1f s): differing %d" % (len(tiles), t1 - t0, bad))
assert bad == 0
