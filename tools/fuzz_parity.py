"""Randomised parity soak: random scenes (new_specular's room + floaters + extra lamps), frame sizes, sample counts, depths and option
settings, GPU frame and ray counts against the oracle.  python tools/fuzz_parity.py <seed> <cases> [mesh] [big]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rustracer_amd as ra
from test_gpu_parity import oracle_params, to_rt_params, gpu_render, _floaters_scene
from oracle import binding as ob
from conftest import TEAPOT_OBJ

KINDS = ("floaters", "dragon", "dragon", "teapot", "specular_teapot", "specular", "diffuse")      # fourth argument "mesh": scenes with meshes too

seed, cases = int(sys.argv[1]), int(sys.argv[2])
ra._lib.check(ra._lib.lib().rt_init(0))
rng = np.random.default_rng(seed)
OPTS = {"shaft": (0, 1, 4, 8, 16), "light_sum": (0, 1), "camera_once": (0, 1), "force_binary": (0, 0, 0, 1, 7, 0x80000000), "lanes": (1, 2, 3, 4), "hot_first": (0,),
        "max_paths": (0, 64, 1000, 5000, 100000), "extend_persist": (0, 1, 2), "light_grid": (0, 1, 8), "self_test": (0, 1), "refill_min": (1, 16, 64), "lane_min_pixels": (64, 65536)}
bad = 0
t0 = time.time()
for c in range(cases):
    n_tri, n_sph, lamps = int(rng.integers(0, 300)), int(rng.integers(0, 40)), int(rng.integers(0, 4))
    kind = KINDS[int(rng.integers(len(KINDS)))] if len(sys.argv) > 3 else "floaters"
    if kind == "floaters":
        osc, gsc = _floaters_scene(n_tri=max(n_tri, 1), n_sph=max(n_sph, 1), seed=int(rng.integers(1 << 30)), extra_lamps=lamps)
    elif kind == "dragon":
        vp, vn, f = ra.dragon_standin_mesh(int(rng.integers(20, 160)), int(rng.integers(10, 60)))      # 400 .. 19,200 triangles
        osc, gsc = ob.OracleScene.dragon_mesh(vp, vn, f), ra.Scene.from_dragon_mesh(vp, vn, f)
    else:
        osc, gsc = ob.OracleScene.named(kind, TEAPOT_OBJ), ra.Scene._named(kind, TEAPOT_OBJ)
    n_tri = kind
    if rng.integers(4) == 0:
        gsc.set_option("hot", 0)
    for rep in range(3):
        BIG = len(sys.argv) > 4      # fifth argument: frames up to 700 x 500 at up to 12 spp (several lanes, several passes)
        kw = dict(width=int(rng.integers(1, 700 if BIG else 140)), height=int(rng.integers(1, 500 if BIG else 100)), spp=int(rng.integers(1, 13 if BIG else 7)), light_samples=int(rng.choice([1, 2, 3, 5, 8, 16, 33, 70])),
                  bounces=int(rng.integers(0, 7)), seed=int(rng.integers(1 << 30)), cam=tuple(float(v) for v in rng.uniform([-8, -8, -12], [8, 8, 4])))
        est = int(rng.integers(5) == 0)
        op = oracle_params(direct_estimator=est, **kw)
        ref = osc.render(op)
        chosen = {}
        for k, vals in OPTS.items():
            if k == "hot_first":
                continue
            chosen[k] = int(vals[rng.integers(len(vals))])
            gsc.set_option(k, chosen[k])
        got, cnt = gpu_render(gsc, to_rt_params(op))
        ok = np.array_equal(got, ref["rgba"]) and cnt["rays_closest"] == ref["intersect_calls"] and cnt["rays_shadow"] == ref["occluded_calls"]
        if not ok:
            bad += 1
            print("MISMATCH case", c, rep, dict(n_tri=n_tri, n_sph=n_sph, lamps=lamps), kw, "est", est, chosen, "bytes", int((got != ref["rgba"]).sum()),
                  cnt["rays_closest"], ref["intersect_calls"], cnt["rays_shadow"], ref["occluded_calls"], flush=True)
    if c % 5 == 4:
        print("case %d done, %d mismatches, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("cases %d x 3 renders, mismatches %d" % (cases, bad))
sys.exit(1 if bad else 0)
