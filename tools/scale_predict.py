"""What one rank of N would do, timed on the one GPU at hand: the config-4 frame with shard (r, N) for N = 1, 2, 4, 8 and a few r.
Whole-job prediction = rays of the whole frame / the slowest share's time; it leaves out the RCCL gather (8.3 MB per frame) and
whatever differs between the GPUs of a node.  A prediction for DESIGN.md section 5, not a measurement of scaling."""
import ctypes as C
import sys, time
sys.path.insert(0, ".")
import torch
import rustracer_amd as ra
from rustracer_amd._lib import RT_MODE_FULL, check, lib
from rustracer_amd.raytracer import Config, RenderConfig, make_params

SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
check(lib().rt_init(0))
scene = ra.Scene.new_dragon("obj/dragon.obj", standin=ra.DRAGON_STANDIN_FULL)
cfg = Config(screen_width=1920, screen_height=1080, samples_per_pixel=SPP, light_samples=32, bounces=8, seed=0x5EED)
frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
base = None
for n in (1, 2, 4, 8):
    worst, rays_total = 0.0, 0
    for r in sorted({0, n // 2, n - 1}):
        p = make_params(RenderConfig(cfg), (0.0, 0.0, 0.0), RT_MODE_FULL, tile=(32, 32), shard=(r, n))
        check(lib().rt_render_device(scene.handle, C.byref(p), frame.data_ptr(), stream))      # warm-up (sizes the workspace)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(2):
            check(lib().rt_render_device(scene.handle, C.byref(p), frame.data_ptr(), stream))
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 500.0
        c = scene.counters()
        rays = c["rays_closest"] + c["rays_shadow"]
        print("N=%d rank %d: %.1f ms per frame share, %.0f Mrays in it, %.1f Grays/s on this GPU" % (n, r, ms, rays / 1e6, rays / ms / 1e6), flush=True)
        if ms > worst:
            worst = ms
        rays_total = max(rays_total, rays)
    if n == 1:
        base = worst
    print("  -> N=%d: slowest share %.1f ms; predicted speed-up %.2f of %d (efficiency %.3f), gather not included" % (n, worst, base / worst, n, base / worst / n), flush=True)
