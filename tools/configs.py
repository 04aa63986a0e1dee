#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one GPU (they are parity-test cases, not bench lines): prints one JSON line each.

    python tools/configs.py            # configs 1, 2, 3, 5 (5 at reduced spp: the full one is 0.9 T rays)
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import rustracer_amd as ra  # noqa: E402
from rustracer_amd._lib import RT_DIRECT_HEMISPHERE, RT_MODE_FULL, check, lib  # noqa: E402
from rustracer_amd.raytracer import Config, RenderConfig, make_params  # noqa: E402


def run(label, scene, w, h, spp, L, b, est=0, cast=False, steps=3):
    dev = torch.device("cuda", 0)
    cfg = Config(screen_width=w, screen_height=h, samples_per_pixel=spp, light_samples=L, bounces=b, direct_estimator=est)
    p = make_params(RenderConfig(cfg), (0.0, 0.0, 0.0), RT_MODE_FULL)
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    ids = torch.zeros((h, w), dtype=torch.int32, device=dev)
    ts = torch.zeros((h, w), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        if cast:
            check(lib().rt_cast_primary_device(scene.handle, C.byref(p), ids.data_ptr(), ts.data_ptr(), stream))
        else:
            check(lib().rt_render_device(scene.handle, C.byref(p), frame.data_ptr(), stream))
    step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    c = scene.counters()
    rays = c["rays_closest"] + c["rays_shadow"]
    print(json.dumps({"config": label, "ms_per_frame": round(dt * 1e3, 3), "Mrays_per_s": round(rays / dt / 1e6, 1), "rays_per_frame": rays,
                      "samples_per_s": round(c["samples"] / dt, 1), "objects": scene.info.n_tri + scene.info.n_sph}), flush=True)


def main():
    check(lib().rt_init(0))
    teapot = os.path.join(ROOT, "obj", "teapot.obj")
    run("1: sphere on plane 512x512, direct lighting, hemisphere estimator, 16 spp, L=8", ra.Scene.new_sphere_on_plane(), 512, 512, 16, 8, 1, est=RT_DIRECT_HEMISPHERE)
    run("2: teapot 1024x1024, direct lighting, light importance sampling, 64 spp, L=16", ra.Scene.new_teapot(teapot), 1024, 1024, 64, 16, 1)
    dragon = ra.Scene.new_dragon(os.path.join(ROOT, "obj", "dragon.obj"))
    run("3: dragon-class mesh (871,200 triangles) 1920x1080 primary-ray cast, ids + t", dragon, 1920, 1080, 1, 1, 0, cast=True, steps=20)
    del dragon
    run("5 (at 64 of 1024 spp): specular + diffuse spheres + teapot 3840x2160 GI, L=32, depth 8", ra.Scene.new_specular_teapot(teapot), 3840, 2160, 64, 32, 8, steps=1)


if __name__ == "__main__":
    main()
