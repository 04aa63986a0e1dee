"""Screen-tile sharding across GPUs and the framebuffer gather.

No reference counterpart: the reference parallelises over image columns on one host
(src/raytracer.rs:197-216).  Here every rank renders the tiles t with t % world == rank into its own
full-size frame, packs them into a compact buffer, and one all_gather (RCCL on GPUs, gloo in the CPU
tests) moves every rank's tiles to all ranks, where they are scattered into the final frame
(`Canvas` reads rank 0's).  The RNG streams are keyed by the global pixel index, so the image does
not depend on the number of ranks.
"""
import ctypes as C

import numpy as np
import torch


class Comm:
    """RCCL communicator of the native gather (rt_comm_*, include/rustracer_hip.h): one per process, made from a 128-byte unique id
    that rank 0 creates and `broadcast` (a callable: bytes-or-None -> bytes, e.g. over a gloo process group or a file) hands to
    every rank.  gather(params, frame_ptr, stream) = pack -> ncclAllGather -> unpack on the stream, nothing else."""

    def __init__(self, world, rank, broadcast):
        from ._lib import check, lib
        self.world, self.rank = world, rank
        ident, failed = None, None
        if rank == 0:
            try:
                buf = (C.c_uint8 * 128)()
                check(lib().rt_comm_unique_id(buf))
                ident = bytes(buf)
            except Exception as e:      # the other ranks are waiting in `broadcast`: hand them the failure (None) before raising it here
                failed = e
        ident = broadcast(ident)
        if failed is not None:
            raise failed
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != 128:
            raise RuntimeError("the RCCL unique id did not arrive (got %r)" % (type(ident),))
        h = C.c_void_p()
        check(lib().rt_comm_init_rank((C.c_uint8 * 128).from_buffer_copy(ident), world, rank, C.byref(h)))
        self._h = h

    def gather(self, params, frame_ptr, stream, bytes_per_pixel=4):
        from ._lib import check, lib
        check(lib().rt_gather_tiles(self._h, C.byref(params), frame_ptr, bytes_per_pixel, stream))

    def describe(self):
        """Which RCCL the communicator resolved to (version code and the path of the shared object)."""
        from ._lib import lib
        return (lib().rt_comm_describe(self._h) or b"").decode(errors="replace")

    def last_gather_ms(self):
        """HIP-event time of the last gather (pack + ncclAllGather + unpack); waits for it."""
        from ._lib import check, lib
        ms = C.c_double(0.0)
        check(lib().rt_comm_last_gather_ms(self._h, C.byref(ms)))
        return ms.value

    def destroy(self):
        from ._lib import lib
        if self._h:
            lib().rt_comm_destroy(self._h)
            self._h = C.c_void_p()


def tiles_per_shard(width, height, tile_w, tile_h, world):
    tiles = ((width + tile_w - 1) // tile_w) * ((height + tile_h - 1) // tile_h)
    return (tiles + world - 1) // world


def tile_index_map(width, height, tile_w, tile_h, shard, world):
    """Pixel index (y*width + x) of every compact slot of `shard`, -1 where the slot is padding
    (tile beyond the image or pixel beyond its edge).  Same enumeration as own_pixel_xy in the kernels."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    n_slots = tiles_per_shard(width, height, tile_w, tile_h, world)
    k = np.arange(n_slots, dtype=np.int64)[:, None, None]
    ty = np.arange(tile_h, dtype=np.int64)[None, :, None]
    tx = np.arange(tile_w, dtype=np.int64)[None, None, :]
    t = shard + k * world
    x = (t % tiles_x) * tile_w + tx
    y = (t // tiles_x) * tile_h + ty
    valid = (t < tiles_x * tiles_y) & (x < width) & (y < height)
    idx = np.where(valid, y * width + x, -1)
    return idx.reshape(-1)


_INDEX_CACHE = {}


def _device_index(width, height, tile_w, tile_h, shard, world, device):
    """The index map of one shard as device tensors (gather source with padding clamped to 0, padding mask, scatter
    destinations, scatter sources), built once per frame geometry: the per-frame gather must not rebuild or re-upload them."""
    key = (width, height, tile_w, tile_h, shard, world, str(device))
    hit = _INDEX_CACHE.get(key)
    if hit is None:
        idx = torch.as_tensor(tile_index_map(width, height, tile_w, tile_h, shard, world), device=device)
        valid = idx >= 0
        hit = (idx.clamp(min=0), ~valid, idx[valid], torch.nonzero(valid).reshape(-1))
        _INDEX_CACHE[key] = hit
    return hit


def pack_tiles(frame, index_map):
    """frame: [H*W, C] tensor -> compact [n_slots*tile_pixels, C] (padding slots zero)."""
    idx = torch.as_tensor(index_map, device=frame.device)
    out = frame[idx.clamp(min=0)]
    out[idx < 0] = 0
    return out


def unpack_tiles(compact, index_map, frame):
    idx = torch.as_tensor(index_map, device=frame.device)
    valid = idx >= 0
    frame[idx[valid]] = compact[valid]
    return frame


def gather_frame(frame, width, height, tile_w, tile_h, rank, world, group=None):
    """frame: this rank's [H, W, C] tensor with its own tiles rendered.  Returns the complete frame
    (on every rank; one all_gather of equal-sized compact buffers)."""
    import torch.distributed as dist
    if world == 1:
        return frame
    c = frame.shape[-1]
    flat = frame.reshape(-1, c)
    src, pad, _, _ = _device_index(width, height, tile_w, tile_h, rank, world, frame.device)
    mine = flat[src]
    mine[pad] = 0
    gathered = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if dist.get_backend(group) == "nccl":                                      # RCCL: one collective, w*h*C bytes in total
        dist.all_gather_into_tensor(gathered, mine.contiguous(), group=group)
    else:                                                                      # gloo (CPU tests, rehearsals): host tensors, list form
        parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous().cpu(), group=group)
        for r in range(world):
            gathered[r].copy_(parts[r])
    out = torch.zeros_like(flat)
    for r in range(world):
        _, _, dst, sel = _device_index(width, height, tile_w, tile_h, r, world, frame.device)
        out[dst] = gathered[r][sel]
    return out.reshape(frame.shape)
