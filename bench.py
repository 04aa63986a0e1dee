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
    if os.environ.get("RT_BENCH_WATCHDOG"):   # seconds: dump every thread's stack and exit if the run is still going (rehearsals)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a.gpus)          # plain `python bench.py --gpus N`: start the N ranks as a child job, before any GPU call
    if world != a.gpus:
        a.gpus = world

    import rustracer_amd as ra
    from rustracer_amd import tiles
    from rustracer_amd._lib import RT_MODE_FULL, check, lib
    from rustracer_amd.raytracer import Config, RenderConfig, make_params

    # Data path: the library's own RCCL communicator (rt_comm_*: pack -> ncclAllGather -> unpack on the render stream).  torch.distributed is
    # only the control plane (gloo on the host: the unique id, the barrier, the reduction of the timings).
    # RT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices, where RCCL refuses to
    # start; the tiles then travel through the host; not a measurement).
    backend = os.environ.get("RT_BENCH_BACKEND", "rccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)   # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal) ranks share devices
    torch.cuda.set_device(dev_index)
    check(lib().rt_init(dev_index))
    dev = torch.device("cuda", dev_index)
    comm, gather_note, comm_desc = None, None, None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if backend != "rccl":
            gather_note = "RT_BENCH_BACKEND=%s: a rehearsal (%d ranks on %d GPU(s)), tiles gathered through the host with gloo; not a measurement" % (
                backend, world, torch.cuda.device_count())
        if backend == "rccl":
            def bcast(ident):
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            try:
                comm = tiles.Comm(world, rank, bcast)
            except Exception as e:   # keep the run alive and say so in the line: a scaling number through the host is still a number
                gather_note = "native RCCL communicator failed (%s): tiles gathered through the host with gloo" % (e,)
            flags = [None] * world
            dist.all_gather_object(flags, gather_note)
            if any(flags):
                gather_note = next(f for f in flags if f)
                if comm is not None:
                    comm.destroy()
                comm = None
            if comm is not None:
                comm_desc = comm.describe()

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
        if comm is not None:
            comm.gather(params, frame.data_ptr(), stream)      # in place: every rank ends up with the whole frame
        elif world > 1:
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
    # HIP events on the launch stream around this rank's own render of every timed step (the lanes fork from and join back into it), and
    # the library's events around the gather: with tiles dealt t % world an uneven frame shows up as a spread of the per-rank times
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    gather_ms = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        check(lib().rt_render_device(scene.handle, C.byref(params), frame.data_ptr(), stream))
        ev[i][1].record()
        if comm is not None:
            comm.gather(params, frame.data_ptr(), stream)      # in place: every rank ends up with the whole frame
        elif world > 1:
            tg = time.perf_counter()
            final = tiles.gather_frame(frame, a.width, a.height, tile[0], tile[1], rank, world)
            gather_ms.append((time.perf_counter() - tg) * 1e3)
    fence()
    elapsed = time.perf_counter() - t0
    scene.set_option("timing", 0)
    render_ms_rank = sum(e0.elapsed_time(e1) for e0, e1 in ev) / max(a.steps, 1)
    if comm is not None:
        gather_ms = [comm.last_gather_ms()]                    # of the last step

    cnt = scene.counters()   # counts and HIP-event kernel times of the last timed step of this rank
    lanes = scene.get_option("lanes")
    el = torch.tensor([elapsed], dtype=torch.float64)
    tot = torch.tensor([cnt["rays_closest"], cnt["rays_shadow"], cnt["samples"]], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    per_rank = [render_ms_rank]
    per_rank_gather = [sum(gather_ms) / len(gather_ms) if gather_ms else 0.0]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, (render_ms_rank, per_rank_gather[0]))
        per_rank = [b[0] for b in box]
        per_rank_gather = [b[1] for b in box]
    rays_closest, rays_shadow, samples = [float(v) for v in tot.tolist()]
    rays_per_step = rays_closest + rays_shadow
    ms_per_step = elapsed / a.steps * 1e3

    if comm is not None:
        comm.destroy()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ----
    # Three kernels carry the frame: k_light (Object::sample_l + the occluder-cache tests, one thread per path vertex looping over the
    # light samples), k_trace_p (Scene::is_occluded for the survivors) and k_extend_p (Scene::intersect).  Each gets its algorithmic
    # bytes per launch over its HIP-event launch time; `roofline` is the one with the largest share of the frame.
    roofline = None
    traced_shadow_per_step = None
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
        n_shadow = max(sc["rays_shadow"], 1)
        nlL = a.light_samples * len(scene.light_indexes())
        kernels = {}
        if ts["refill_lanes"] > 0:
            # k_trace_p, per traced ray: 80 B per 8-wide node visited, 48 B per primitive record tested, 56 B of ray I/O
            # (32 B ray record + 16 B vertex position + 4 B flag read + 4 B flag write-back).  The rays k_light settles are not counted.
            traced_frac = ts["refill_lanes"] / n_shadow
            traced_shadow_per_step = traced_frac * cnt["rays_shadow"]
            nodes_per_ray = ts["node_lanes"] / ts["refill_lanes"]
            prims_per_ray = ts["prim_lanes"] / ts["refill_lanes"]
            b_trace = 80.0 * nodes_per_ray + 48.0 * prims_per_ray + 56.0
            kernels["k_trace_p"] = {
                "what": "Scene::is_occluded: persistent any-hit traversal of the 8-wide occlusion BVH with lane refill",
                "unit": "traced ray", "bytes_per_unit": round(b_trace, 1), "units_per_launch": round(traced_frac * cnt["rays_shadow"] / launches, 1),
                "launches_per_step": int(launches), "ms_per_step": round(cnt["kernel_ms_shadow"], 2),
                "per_unit": {"wide_nodes_80B": round(nodes_per_ray, 3), "prim_records_48B": round(prims_per_ray, 3), "ray_io_B": 56,
                             "false_candidates": round(ts["box_miss"] / ts["refill_lanes"], 3),
                             "traced_fraction_of_is_occluded_rays": round(traced_frac, 4),
                             "lanes_per_node_step": round(ts["node_lanes"] / max(ts["node_steps"], 1), 1),
                             "lanes_per_prim_step": round(ts["prim_lanes"] / max(ts["prim_steps"], 1), 1),
                             "lanes_per_refill": round(ts["refill_lanes"] / max(ts["refills"], 1), 1)}}
            # k_light, per light sample: the vertex record (position 16 B, normal 16 B, path slot 8 B, RNG state 8 B) is read once per
            # vertex and serves nlL samples; every sample writes 1/64 of two row masks (2 x 8 B); a ray it can settle neither as blocked nor,
            # inside a clear beam, as visible and summed writes its 16-B record (direction, distance), and one that still has to be traced 16 B
            # more (1/d of the box test, flags).
            # A settled ray whose term k_light adds itself leaves none; each (vertex, light) writes the 16 B of that sum.
            settled_frac = (sc["shaft"][3] - sc["trace_steps"][7]) / n_shadow
            b_light = 48.0 / max(nlL, 1) + 16.0 / 64.0 + 16.0 / max(a.light_samples, 1) + 16.0 * (traced_frac + settled_frac) + 16.0 * traced_frac
            kernels["k_light"] = {
                "what": "Object::sample_l for every (vertex, light sample) + the test of the object the ray starts on + the per-light occluder cache",
                "unit": "light sample", "bytes_per_unit": round(b_light, 2), "units_per_launch": round(cnt["rays_shadow"] / launches, 1),
                "launches_per_step": int(launches), "ms_per_step": round(cnt["kernel_ms_light"], 2)}
            sh = sc["shaft"]
            if sh[0]:
                # the per-(vertex, light) beam walk: how many beams were walked, how many were clear, what the walk cost
                kernels["k_light"]["shaft"] = {"beams_walked_per_vertex_light": round(sh[0] * a.light_samples / n_shadow, 4), "clear_fraction_of_walked": round(sh[1] / sh[0], 4),
                                               "nodes_per_walk": round(sh[2] / sh[0], 3), "rays_settled_visible_fraction_of_is_occluded_rays": round(sh[3] / n_shadow, 4),
                                               "of_them_summed_in_k_light": round(sc["trace_steps"][7] / max(sh[3], 1), 4)}
        else:
            # binary-tree kernel (option wide=0 / persist=0, or a scene outside the 8-wide tree's range): SURVEY.md §8(d) formula
            b_shadow = (80.0 * sc["shadow_wide"] + 64.0 * sc["shadow_inner"] + 48.0 * sc["shadow_tris"] + 16.0 * sc["shadow_sphs"]) / n_shadow + 48.0
            kernels["k_shadow"] = {
                "what": "Scene::is_occluded, one thread per light sample (sampling and traversal in one kernel)",
                "unit": "is_occluded ray", "bytes_per_unit": round(b_shadow, 1), "units_per_launch": round(cnt["rays_shadow"] / launches, 1),
                "launches_per_step": int(launches), "ms_per_step": round(cnt["kernel_ms_shadow"], 2)}
        # k_extend, per intersect ray: SURVEY.md §8(d): 64 B per binary node, 48 B per triangle, 16 B per sphere, 48 B of ray I/O, plus the
        # shading-side fetch per hit (80 B triangle normals + material, 32 B sphere)
        n_closest = max(sc["rays_closest"], 1)
        b_ext = (64.0 * sc["closest_inner"] + 48.0 * sc["closest_tris"] + 16.0 * sc["closest_sphs"]) / n_closest + 48.0 \
            + (80.0 * sc["closest_hits_tri"] + 32.0 * sc["closest_hits_sph"]) / n_closest
        kernels["k_extend_p"] = {
            "what": "Scene::intersect: closest hit on the binary SAH tree (while-while, rank tie-break): k_camera for the camera rays (one traversal per pixel and pass), the persistent "
                    "k_extend_p with lane refill after the first bounce (8 of 9 launches; the PMC figures are k_extend_p's)",
            "unit": "intersect ray", "bytes_per_unit": round(b_ext, 1), "units_per_launch": round(cnt["rays_closest"] / max(cnt["launches_closest"], 1), 1),
            "launches_per_step": int(cnt["launches_closest"]), "ms_per_step": round(cnt["kernel_ms_closest"], 2)}
        # The timed steps run `lanes` render lanes on streams of their own: kernels of different lanes overlap, so a
        # kernel's HIP-event duration in the timed region includes the time it shared the GPU, and the durations add up to more than the
        # frame.  One more frame with a single lane gives every kernel's duration when it has the GPU to itself: that is the duration the
        # roofline figures use (`frac`), the overlapped one is reported beside it (`timed_region`).
        if lanes > 1:
            scene.set_option("lanes", 1)
            scene.set_option("timing", 1)
            check(lib().rt_render_device(scene.handle, C.byref(params), frame.data_ptr(), stream))
            torch.cuda.synchronize(dev)
            ex = scene.counters()
            scene.set_option("timing", 0)
            scene.set_option("lanes", lanes)
        else:
            ex = cnt
        ms_field = {"k_trace_p": "kernel_ms_shadow", "k_shadow": "kernel_ms_shadow", "k_light": "kernel_ms_light", "k_extend_p": "kernel_ms_closest"}
        n_field = {"k_trace_p": "launches_shadow", "k_shadow": "launches_shadow", "k_light": "launches_shadow", "k_extend_p": "launches_closest"}
        frame_ms = cnt["kernel_ms_shadow"] + cnt["kernel_ms_light"] + cnt["kernel_ms_closest"] + cnt["kernel_ms_shade"]
        ex_frame_ms = ex["kernel_ms_shadow"] + ex["kernel_ms_light"] + ex["kernel_ms_closest"] + ex["kernel_ms_shade"]
        pmc = pmc_kernels(a) if world == 1 else {}   # per-unit counter figures of the newest committed profile of these kernel sources
        mix = isa_mix()
        for name, k in kernels.items():
            ex_launches = max(ex[n_field[name]], 1)
            units_ex = k["units_per_launch"] * k["launches_per_step"] / ex_launches          # the same units in the one-lane frame's launches
            k["timed_avg_launch_ms"] = round(k["ms_per_step"] / max(k["launches_per_step"], 1), 4)
            k["avg_launch_ms"] = round(ex[ms_field[name]] / ex_launches, 4)
            k["launches_one_lane"] = int(ex_launches)
            k["units_per_launch_one_lane"] = round(units_ex, 1)
            k["share_of_kernel_time"] = round(ex[ms_field[name]] / max(ex_frame_ms, 1e-9), 4)
            k["achieved_GBps"] = round(k["bytes_per_unit"] * units_ex / max(k["avg_launch_ms"], 1e-9) / 1e6, 2)
            k["frac_of_hbm_peak"] = round(k["achieved_GBps"] / HBM_PEAK_GBS, 5)
            k["timed_achieved_GBps"] = round(k["bytes_per_unit"] * k["units_per_launch"] / max(k["timed_avg_launch_ms"], 1e-9) / 1e6, 2)
            m = pmc.get(name, {}) if isinstance(pmc.get(name, {}), dict) else {}
            # counters travel per unit (profiles/r*_pmc.json), so they describe THIS launch size whatever size they were profiled at
            k["traffic"] = round(m["hbm_bytes_per_unit"] * units_ex) if "hbm_bytes_per_unit" in m else None
            k["traffic_bytes_per_unit"] = m.get("hbm_bytes_per_unit")
            for f in ("l2_hit_rate", "valu_issue_busy", "valu_active_lanes", "valu_insts_per_unit"):
                if f in m:
                    k[f] = m[f]
            mx = mix.get(name)
            if mx and "valu_insts_per_unit" in m:
                rate = m["valu_insts_per_unit"] * units_ex / max(k["avg_launch_ms"] * 1e-3, 1e-12)      # wave64 VALU instructions per second, whole chip
                k["compute"] = {"bound": "valu_issue", "achieved": round(rate / 1e9, 2), "peak": round(mx["issue_peak_wave_instructions_per_s"] / 1e9, 2),
                                "unit": "G wave64-instructions/s", "frac": round(rate / mx["issue_peak_wave_instructions_per_s"], 4),
                                "mean_cycles_per_instruction": mx["mean_cycles_per_valu_instruction"], "mix": mx["mix"]}
        dom = max(kernels, key=lambda n: kernels[n]["avg_launch_ms"] * kernels[n]["launches_one_lane"])
        d = kernels[dom]
        roofline = {
            "bound": "hbm", "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac_of_hbm_peak"],
            "traffic": d["traffic"], "traffic_source": pmc.get("_source"), "traffic_note": pmc.get("_note"),
            "kernel": "%s (%s)" % (dom, d["what"]), "bytes_per_unit": d["bytes_per_unit"], "unit_of_work": d["unit"],
            "units_per_launch": d["units_per_launch_one_lane"], "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches_one_lane"],
            "definition": "achieved = bytes_per_unit x units_per_launch / avg_launch_ms of the dominant kernel (largest share of the frame's kernel time), the "
                          "launch time being the HIP-event mean over a frame rendered with ONE lane, i.e. with the GPU to itself; traffic = L2<->fabric bytes "
                          "per launch = hbm_bytes_per_unit (FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes, per unit) x units_per_launch",
            "timed_region": {"render_lanes": lanes, "avg_launch_ms": d["timed_avg_launch_ms"], "units_per_launch": d["units_per_launch"],
                             "launches_per_step": d["launches_per_step"], "achieved": d["timed_achieved_GBps"],
                             "frac": round(d["timed_achieved_GBps"] / HBM_PEAK_GBS, 5), "kernel_ms_sum": round(frame_ms, 2), "ms_per_step": round(ms_per_step, 3),
                             "note": "in the timed steps %d lanes run concurrently: a kernel's HIP-event duration includes the time it shared the GPU with the "
                                     "other lanes' kernels, and the durations sum to more than the step" % lanes},
            "frame_ms_one_lane": round(ex_frame_ms, 2),
            "compute": d.get("compute"),
            "compute_note": "the kernel is bound by VALU instruction issue, not by HBM: `compute` sets its measured wave64 VALU instruction rate (SQ_INSTS_VALU "
                            "per unit from the PMC passes x units / launch time) against the issue ceiling of its instruction mix (profiles/r*_isa_mix.json: "
                            "static class mix of its loop; every non-transcendental instruction at the fast class's issue cost unless the slow class alone "
                            "takes longer, transcendentals added in full: the mixed-stream rows of tools/valu_rate.hip, profiles/r02_valu_rate.txt)",
            "kernels": kernels,
            "kernel_ms_per_step_one_lane": {"k_light": round(ex["kernel_ms_light"], 2), "k_trace_p": round(ex["kernel_ms_shadow"], 2),
                                            "k_camera+k_extend_p": round(ex["kernel_ms_closest"], 2), "shade+resolve+k_binary": round(ex["kernel_ms_shade"], 2)},
            "note": "The scene (%.0f MB) is resident in L2 / Infinity Cache: most algorithmic bytes of the traversal kernels (BVH nodes, primitive records) are "
                    "served on-die, so their `achieved` can exceed what HBM delivers; `traffic` is what crossed the fabric" % (scene.info.device_bytes / 1e6),
        }

    # ---- CPU baseline: the oracle (reference algorithm, unpruned traversal, recursive integrator) on a bounded sample ----
    cpu = None
    if a.cpu_seconds > 0 and world == 1:   # reported at N = 1 only (the host cores are shared by the ranks otherwise)
        cpu = cpu_baseline(a, standin if not os.path.exists(dragon_obj) else None, dragon_obj, a.cpu_seconds)

    value = rays_per_step * a.steps / elapsed / 1e6
    out = {
        "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: dragon GI, %dx%d, %d spp, %d light samples, depth %d, 32x32 screen tiles over %d GPU(s)%s"
                   % (a.width, a.height, a.spp, a.light_samples, a.bounces, world, (" + native RCCL all-gather (rt_gather_tiles)" if comm is not None or world == 1 else " + gloo all_gather through the host") if world > 1 else ""),
                   "mesh": "%s, %d triangles" % (mesh_label, n_tri - 10), "seed": a.seed, "scene_build_s": round(scene_s, 2), "render_lanes": lanes,
                   "spp": a.spp, "max_paths": scene.get_option("max_paths")},
        "samples_per_s": round(samples * a.steps / elapsed, 1),
        "rays_per_step": int(rays_per_step), "rays_closest_per_step": int(rays_closest), "rays_shadow_per_step": int(rays_shadow),
        "rays_per_sample": round(rays_per_step / max(samples, 1), 2),
        # the headline counts every Scene::intersect and Scene::is_occluded call, as SURVEY.md 8(d) defines a ray; most is_occluded calls are
        # settled by the test of the object the ray starts on and the per-light occluder cache, without a tree search.  This is the rate of
        # the rays that do search a tree (all intersect rays + the is_occluded rays k_trace_p traverses), world == 1 only.
        "traversed_Mrays_per_s": (round((rays_closest + traced_shadow_per_step) * a.steps / elapsed / 1e6, 2) if traced_shadow_per_step is not None and world == 1 else None),
        # HIP-event kernel times of the last timed step, summed per class; with more than one render lane kernels overlap and the sum exceeds the frame
        "kernel_ms_last_step": {"trace_shadow": round(cnt["kernel_ms_shadow"], 2), "trace_closest": round(cnt["kernel_ms_closest"], 2),
                                "light_sampling": round(cnt["kernel_ms_light"], 2), "shade_generate_resolve": round(cnt["kernel_ms_shade"], 2)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if world > 1:
        out["ranks"] = {"render_ms_per_step": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3), "per_rank": [round(v, 3) for v in per_rank]},
                        "gather_ms": {"max": round(max(per_rank_gather), 3), "per_rank": [round(v, 3) for v in per_rank_gather],
                                      "what": ("HIP events around rt_gather_tiles (pack + ncclAllGather + unpack) of the last step" if comm_desc
                                               else "host wall time of the gloo gather per step")},
                        "rccl": comm_desc}
    if gather_note:
        out["gather_note"] = gather_note
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a
    CHILD process and relay its output (rank 0's JSON line).  This process has not touched the GPU (torch.cuda.device_count() does not
    initialise it) and never replaces itself: a process that has initialised HIP must not exec.  With fewer GPUs than ranks (a rehearsal on
    a one-GPU box) the ranks share devices, RCCL cannot start there, and the tiles travel through the host (RT_BENCH_BACKEND=gloo, said in
    the line's gather_note)."""
    import socket
    import subprocess
    env = dict(os.environ)
    n_dev = torch.cuda.device_count()
    if n_dev < n:
        env.setdefault("RT_BENCH_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def isa_mix():
    """Issue-class mix of the kernels' loops (profiles/r*_isa_mix.json, tools/isa_mix.py), only when it was made from the sources that run."""
    import glob
    from rustracer_amd._lib import kernel_source_hash
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_mix.json")))
    if not paths:
        return {}
    try:
        with open(paths[-1]) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return {}
    return j.get("kernels", {}) if j.get("kernel_source_hash") == kernel_source_hash() else {}


def pmc_kernels(a):
    """Per-kernel PMC figures of the newest committed rocprofv3 passes (profiles/r*_pmc.json, written by tools/summarize_prof.py): HBM-side
    bytes per launch from FETCH_SIZE / WRITE_SIZE (own passes, corrected as MI355X_MICROARCH.md prescribes), L2 hit rate, VALU issue
    utilisation.  bench.py cannot collect PMC counters on itself, so the figures are reported only for the workload AND the kernel sources
    they were measured on: each file carries the hash of the device sources at profile time, and a file whose hash differs from the
    running tree's is ignored (`traffic` null, with a note)."""
    import glob
    from rustracer_amd._lib import kernel_source_hash
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not paths:
        return {"_note": "no profiles/r*_pmc.json"}
    path = paths[-1]
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return {"_note": "%s unreadable" % os.path.basename(path)}
    w = j.get("workload", {})
    if (w.get("width"), w.get("height"), w.get("light_samples"), w.get("bounces"), w.get("mesh")) != (a.width, a.height, a.light_samples, a.bounces, a.mesh):
        return {"_note": "%s was measured on another workload" % os.path.basename(path)}
    if j.get("kernel_source_hash") != kernel_source_hash():
        return {"_note": "%s was measured on other kernel sources (hash %s at commit %s, running %s): PMC figures withheld"
                         % (os.path.basename(path), j.get("kernel_source_hash"), j.get("git_commit"), kernel_source_hash())}
    out = dict(j.get("kernels", {}))
    out["_source"] = "profiles/%s (%s; commit %s)" % (os.path.basename(path), j.get("source", ""), j.get("git_commit"))
    return out


def host_cpus():
    """CPUs this process may really use: its affinity mask, cut by the cgroup CPU quota and by OMP_NUM_THREADS when the box sets one
    (the GPU boxes of this pool expose every core of the machine to a job whose share is 16)."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(int(-(-int(q) // int(per))), 1)
    except (OSError, ValueError):
        pass
    omp = int(os.environ.get("OMP_NUM_THREADS", "0")) or None
    use = min(v for v in (affinity, quota, omp) if v)
    return use, {"os_cpu_count": os.cpu_count(), "sched_getaffinity": affinity, "cgroup_quota": quota, "OMP_NUM_THREADS": omp}


def cpu_baseline(a, standin, dragon_obj, target_s):
    """The oracle timed on the host cores: the same frame, a strided subset of its 32x32 tiles
    (shard 0 of `stride`) at reduced spp, sized to about `target_s` seconds."""
    from oracle import binding as ob
    import rustracer_amd as ra
    cores, cpu_info = host_cpus()
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
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "host_cpus": cpu_info,
            "sample": "oracle (C++ restatement of the reference algorithm, %d threads): every %d-th 32x32 tile of the same %dx%d frame at %d spp, "
                      "%d light samples, depth %d: %d rays in %.1f s" % (cores, stride, a.width, a.height, spp, a.light_samples, a.bounces, rays, dt)}


if __name__ == "__main__":
    main()
