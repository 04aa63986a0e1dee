"""Probe: does rendering two tile shards of one frame concurrently on two HIP streams (two scene handles, two host threads) beat one
stream?  The memory-latency-bound kernels of one shard (k_shade_*, k_extend_p) could fill the gaps of the other's VALU-bound ones."""
import ctypes as C
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import rustracer_amd as ra
from rustracer_amd._lib import RT_MODE_FULL, check, lib
from rustracer_amd.raytracer import Config, RenderConfig, make_params

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
check(lib().rt_init(0))
dev = torch.device("cuda", 0)
cfg = Config(screen_width=1920, screen_height=1080, samples_per_pixel=spp, light_samples=32, bounces=8, seed=24301)
frame = torch.zeros((1080, 1920, 4), dtype=torch.uint8, device=dev)


def run(n_shards, max_paths, reps=3):
    scenes = [ra.Scene.new_dragon("obj/dragon.obj", standin=ra.DRAGON_STANDIN_FULL) for _ in range(n_shards)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_shards)]
    for sc in scenes:
        sc.set_option("max_paths", max_paths)
    params = [make_params(RenderConfig(cfg), (0.0, 0.0, 0.0), RT_MODE_FULL, tile=(32, 32), shard=(k, n_shards)) for k in range(n_shards)]

    def work(k):
        check(lib().rt_render_device(scenes[k].handle, C.byref(params[k]), frame.data_ptr(), streams[k].cuda_stream))

    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(n_shards)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize(dev)
        best = min(best, time.perf_counter() - t0)
    for sc in scenes:
        sc.close()
    return best * 1e3


for n, mp in ((1, 1 << 24), (2, 1 << 23), (2, 1 << 24), (3, 1 << 23), (4, 1 << 22)):
    print("shards/streams %d, max_paths %d: %.1f ms per frame" % (n, mp, run(n, mp)), flush=True)
