"""world_size-2 gloo test of the screen-tile shard + all_gather path (rustracer_amd/tiles.py) on CPU.
Each rank renders its tiles with the ORACLE (the checker stands in for the GPU here; the GPU kernels use
the same tile enumeration, tests/test_gpu_parity.py), gathers, and every rank must hold the unsharded frame."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from rustracer_amd import tiles
    w, h, tw, th = 72, 50, 16, 16       # ragged right and bottom tiles
    sc = ob.OracleScene.named("diffuse")
    kw = dict(width=w, height=h, spp=2, light_samples=2, bounces=3)
    mine = sc.render(ob.make_params(tile=(tw, th), shard=(rank, world), **kw), nthreads=2)["rgba"]
    full = tiles.gather_frame(torch.from_numpy(mine.copy()), w, h, tw, th, rank, world).numpy()
    whole = sc.render(ob.make_params(**kw), nthreads=2)["rgba"]
    ids = np.full((h, w), 0xFFFFFFFF, np.uint32)
    p = ob.make_params(width=w, height=h, tile=(tw, th), shard=(rank, world))
    own_ids, _, _ = sc.cast_primary(p, nthreads=2)
    ids_full = tiles.gather_frame(torch.from_numpy(own_ids.astype(np.int64)).reshape(h, w, 1), w, h, tw, th, rank, world).numpy()
    whole_ids, _, _ = sc.cast_primary(ob.make_params(width=w, height=h), nthreads=2)
    q.put((rank, bool(np.array_equal(full, whole)), bool(np.array_equal(ids_full.reshape(h, w), whole_ids.astype(np.int64)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_gather_reconstructs_the_frame():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, True), (1, True, True)]
