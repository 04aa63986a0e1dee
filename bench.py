#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: Mrays/s (whole job) and samples/s for
dragon-class global illumination at 1920x1080, 256 spp, 32 light samples, depth 8 (config 4).

A step = one complete frame of that config: every pass of the wavefront path tracer over every owned
screen tile, plus (N > 1) the RCCL all_gather of the tiles into the final frame.  The frame is fixed
as N grows (strong scaling): rank r renders the 32x32 tiles t with t % N == r.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

obj/dragon.obj is absent from the reference mount (.MISSING_LARGE_BLOBS:1): the mesh is the
deterministic dragon-class stand-in (871,200 triangles) unless obj/dragon.obj exists.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--light-samples", type=int, default=32)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--mesh", default="full", choices=["full", "small"], help="871,200 or 100,000 triangle stand-in")
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EED)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample (0 = skip)")
    ap.add_argument("--stats-spp", type=int, default=4, help="spp of the instrumented pass that measures bytes/ray (0 = skip)")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to rt_set_option (tuning experiments)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world

    import rustracer_amd as ra
    from rustracer_amd import tiles
    from rustracer_amd._lib import RT_MODE_FULL, check, lib
    from rustracer_amd.raytracer import Config, RenderConfig, make_params

    torch.cuda.set_device(local_rank)
    check(lib().rt_init(local_rank))
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    # ---- scene (replicated on every rank; the BVH is built on the host) ----
    dragon_obj = os.path.join(ROOT, "obj", "dragon.obj")
    standin = ra.DRAGON_STANDIN_FULL if a.mesh == "full" else ra.DRAGON_STANDIN_SMALL
    t0 = time.perf_counter()
    if os.path.exists(dragon_obj):
        scene = ra.Scene.new_dragon(dragon_obj)
        mesh_label = "obj/dragon.obj"
    else:
        scene = ra.Scene.new_dragon(dragon_obj, standin=standin)
        mesh_label = "synthetic dragon-class mesh"
    scene_s = time.perf_counter() - t0
    n_tri = scene.info.n_tri
    if a.max_paths:
        scene.set_option("max_paths", a.max_paths)
    for kv in a.opt:
        k, val = kv.split("=")
        scene.set_option(k, int(val))

    cfg = Config(screen_width=a.width, screen_height=a.height, samples_per_pixel=a.spp, light_samples=a.light_samples,
                 bounces=a.bounces, seed=a.seed)
    tile = (32, 32)
    params = make_params(RenderConfig(cfg), (0.0, 0.0, 0.0), RT_MODE_FULL, tile=tile, shard=(rank, world))
    frame = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        check(lib().rt_render_device(scene.handle, C.byref(params), frame.data_ptr(), stream))
        if world > 1:
            return tiles.gather_frame(frame, a.width, a.height, tile[0], tile[1], rank, world)
        return frame

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    scene.set_option("timing", 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        final = step()
    fence()
    elapsed = time.perf_counter() - t0
    scene.set_option("timing", 0)

    cnt = scene.counters()   # counts and HIP-event kernel times of the last timed step of this rank
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([cnt["rays_closest"], cnt["rays_shadow"], cnt["samples"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_closest, rays_shadow, samples = [float(v) for v in tot.tolist()]
    rays_per_step = rays_closest + rays_shadow
    ms_per_step = elapsed / a.steps * 1e3

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (k_trace_p: Scene::is_occluded for the light-sample rays that survive k_light) ----
    roofline = None
    if a.stats_spp > 0 and cnt["launches_shadow"] > 0:
        sp = params.copy()
        sp.spp = a.stats_spp
        scene.set_option("stats", 1)
        check(lib().rt_render_device(scene.handle, C.byref(sp), frame.data_ptr(), stream))
        torch.cuda.synchronize(dev)
        sc = scene.counters()
        scene.set_option("stats", 0)
        ts = dict(zip(["node_steps", "node_lanes", "prim_steps", "prim_lanes", "refills", "refill_lanes", "box_miss"], sc["trace_steps"]))
        launches = cnt["launches_shadow"]
        avg_ms = cnt["kernel_ms_shadow"] / launches
        if ts["refill_lanes"] > 0:
            # Algorithmic bytes of one traced ray: 80 B per 8-wide node visited, 48 B per primitive record tested, and
            # 56 B of ray I/O (32 B ray record + 16 B vertex position + 4 B flag read + 4 B flag write-back, the last counted for every ray).
            # k_light settles the other is_occluded rays (test of the object the ray starts on) before this kernel; they are not counted here.
            traced_frac = ts["refill_lanes"] / max(sc["rays_shadow"], 1)
            nodes_per_ray = ts["node_lanes"] / ts["refill_lanes"]
            prims_per_ray = ts["prim_lanes"] / ts["refill_lanes"]
            bytes_per_ray = 80.0 * nodes_per_ray + 48.0 * prims_per_ray + 56.0
            rays_per_launch = traced_frac * cnt["rays_shadow"] / launches
            kernel = "k_trace_p<false,8> (Scene::is_occluded: persistent any-hit traversal of the 8-wide occlusion BVH with lane refill)"
            per_ray = {"wide_nodes_80B": round(nodes_per_ray, 3), "prim_records_48B": round(prims_per_ray, 3), "ray_io_B": 56,
                       "false_candidates_per_ray": round(ts["box_miss"] / ts["refill_lanes"], 3),
                       "traced_fraction_of_is_occluded_rays": round(traced_frac, 4),
                       "lanes_per_node_step": round(ts["node_lanes"] / max(ts["node_steps"], 1), 1),
                       "lanes_per_prim_step": round(ts["prim_lanes"] / max(ts["prim_steps"], 1), 1),
                       "lanes_per_refill": round(ts["refill_lanes"] / max(ts["refills"], 1), 1)}
        else:
            # binary-tree kernel (option wide=0 / persist=0, or a scene outside the 8-wide tree's range): SURVEY.md §8(d) formula
            bytes_per_ray = (80.0 * sc["shadow_wide"] + 64.0 * sc["shadow_inner"] + 48.0 * sc["shadow_tris"] + 16.0 * sc["shadow_sphs"]) \
                / max(sc["rays_shadow"], 1) + 48.0
            rays_per_launch = cnt["rays_shadow"] / launches
            kernel = "k_shadow<false,false> (Scene::is_occluded, one thread per light sample)"
            per_ray = {"wide_nodes_80B": round(sc["shadow_wide"] / max(sc["rays_shadow"], 1), 3), "binary_nodes_64B": round(sc["shadow_inner"] / max(sc["rays_shadow"], 1), 3),
                       "tri_48B": round(sc["shadow_tris"] / max(sc["rays_shadow"], 1), 3)}
        achieved = bytes_per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e9
        bytes_per_closest_ray = (64.0 * sc["closest_inner"] + 48.0 * sc["closest_tris"] + 16.0 * sc["closest_sphs"]) / max(sc["rays_closest"], 1) + 48.0 \
            + (80.0 * sc["closest_hits_tri"] + 32.0 * sc["closest_hits_sph"]) / max(sc["rays_closest"], 1)
        traffic, traffic_src = pmc_traffic(a, "k_trace_p")
        roofline = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kernel, "per_traced_ray": per_ray,
            "bytes_per_ray": round(bytes_per_ray, 1), "rays_per_launch": round(rays_per_launch, 1), "avg_launch_ms": round(avg_ms, 4),
            "launches_per_step": int(launches),
            "kernel_ms_per_step": {"k_trace_p": round(cnt["kernel_ms_shadow"], 2), "k_light+k_binary": round(cnt["kernel_ms_light"], 2),
                                   "k_extend": round(cnt["kernel_ms_closest"], 2), "shade+generate+resolve": round(cnt["kernel_ms_shade"], 2)},
            "k_extend_bytes_per_ray": round(bytes_per_closest_ray, 1),
            "k_extend_GBps": round(bytes_per_closest_ray * cnt["rays_closest"] / max(cnt["kernel_ms_closest"], 1e-9) / 1e6, 2),
            "note": "algorithmic bytes over kernel time; the scene (%.0f MB) is resident in L2 / Infinity Cache, so this rate is served by the caches and may "
                    "exceed the HBM peak, while `traffic` (HBM bytes per launch from the PMC counters) stays far below the algorithmic bytes "
                    "(SURVEY.md §8d, DESIGN.md §6)" % (scene.info.device_bytes / 1e6),
        }

    # ---- CPU baseline: the oracle (reference algorithm, unpruned traversal, recursive integrator) on a bounded sample ----
    cpu = None
    if a.cpu_seconds > 0:
        cpu = cpu_baseline(a, standin if not os.path.exists(dragon_obj) else None, dragon_obj, a.cpu_seconds)

    value = rays_per_step * a.steps / elapsed / 1e6
    out = {
        "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: dragon GI, %dx%d, %d spp, %d light samples, depth %d, 32x32 screen tiles over %d GPU(s)%s"
                   % (a.width, a.height, a.spp, a.light_samples, a.bounces, world, " + RCCL all_gather" if world > 1 else ""),
                   "mesh": "%s, %d triangles" % (mesh_label, n_tri - 10), "seed": a.seed, "scene_build_s": round(scene_s, 2)},
        "samples_per_s": round(samples * a.steps / elapsed, 1),
        "rays_per_step": int(rays_per_step), "rays_closest_per_step": int(rays_closest), "rays_shadow_per_step": int(rays_shadow),
        "rays_per_sample": round(rays_per_step / max(samples, 1), 2),
        "kernel_ms_last_step": {"trace_shadow": round(cnt["kernel_ms_shadow"], 2), "trace_closest": round(cnt["kernel_ms_closest"], 2),
                                "light_sampling": round(cnt["kernel_ms_light"], 2), "shade_generate_resolve": round(cnt["kernel_ms_shade"], 2)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(a, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc.json, written by
    tools/summarize_prof.py from FETCH_SIZE / WRITE_SIZE collected in their own passes and corrected as MI355X_MICROARCH.md
    prescribes).  bench.py cannot collect PMC counters on itself; the figure is only reported for the workload it was measured on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc.json")
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None, None
    w = j.get("workload", {})
    if (w.get("width"), w.get("height"), w.get("light_samples"), w.get("bounces"), w.get("mesh")) != (a.width, a.height, a.light_samples, a.bounces, a.mesh):
        return None, None
    k = j.get("kernels", {}).get(kernel)
    if not k or k.get("hbm_bytes_per_launch") is None:
        return None, None
    return k["hbm_bytes_per_launch"], "profiles/r01_pmc.json (%s)" % j.get("source", "")


def cpu_baseline(a, standin, dragon_obj, target_s):
    """The oracle timed on the host cores: the same frame, a strided subset of its 32x32 tiles
    (shard 0 of `stride`) at reduced spp, sized to about `target_s` seconds."""
    from oracle import binding as ob
    import rustracer_amd as ra
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    if standin is not None:
        vp, vn, f = ra.dragon_standin_mesh(*standin)
        osc = ob.OracleScene.dragon_mesh(vp, vn, f)
    else:
        osc = ob.OracleScene.named("dragon", dragon_obj)
    spp = min(a.spp, 2)

    def run(stride):
        nonlocal spp
        p = ob.make_params(width=a.width, height=a.height, spp=spp, light_samples=a.light_samples, bounces=a.bounces, seed=a.seed,
                           tile=(32, 32), shard=(0, stride))
        t0 = time.perf_counter()
        r = osc.render(p, nthreads=cores)
        dt = time.perf_counter() - t0
        return r["intersect_calls"] + r["occluded_calls"], dt

    n_tiles = ((a.width + 31) // 32) * ((a.height + 31) // 32)
    stride = max(n_tiles // 16, 1)                # calibration: ~16 tiles at 2 spp
    rays, dt = run(stride)
    rate = rays / max(dt, 1e-6)
    per_tile_spp = rays / max(n_tiles // stride, 1) / spp
    want_tile_spp = max(rate * target_s / max(per_tile_spp, 1), 16.0)      # (tiles x spp) that fill target_s
    if want_tile_spp <= n_tiles * spp:
        stride = max(int(n_tiles * spp / want_tile_spp), 1)
    else:
        stride = 1
        spp = int(min(a.spp, max(spp, want_tile_spp // n_tiles)))
    rays, dt = run(stride)
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "oracle (C++ restatement of the reference algorithm, %d threads): every %d-th 32x32 tile of the same %dx%d frame at %d spp, "
                      "%d light samples, depth %d: %d rays in %.1f s" % (cores, stride, a.width, a.height, spp, a.light_samples, a.bounces, rays, dt)}


if __name__ == "__main__":
    main()
