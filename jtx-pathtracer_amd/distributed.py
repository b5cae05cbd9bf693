"""Multi-GPU pixel-tile sharding: one process per GPU, torch.distributed (RCCL on ROCm, gloo on CPU).

Every pixel-sample is an independent path whose RNG depends only on (row, col, sample)
(camera.cpp:101), and the reference already partitions the frame into 32x32 tiles with no inter-tile
communication (camera.cpp:55-64).  Rank r owns the row-major 32x32 tiles k with k % world == r
(interleaved, so sky-heavy and geometry-heavy regions spread over all GPUs) and accumulates ALL
strata of its own pixels in the reference's sample order; the per-pixel float sums are therefore
bit-identical to the 1-GPU result.  The scene is replicated.  The only collective is one reduce
(sum) per frame of the accumulation buffer -- and of the RGB8 image -- in which non-owned pixels
are exactly zero.
"""
import ctypes as C

import numpy as np

from . import _capi as capi

TILE = 32


def tile_owner_mask(width, height, rank, world):
    """Boolean (H, W) mask of the pixels whose 32x32 tile belongs to `rank`."""
    tiles_x = (width + TILE - 1) // TILE
    rows = np.arange(height)[:, None] // TILE
    cols = np.arange(width)[None, :] // TILE
    return ((rows * tiles_x + cols) % max(1, world)) == rank


def owned_tiles(width, height, rank, world):
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return len(range(rank, tiles, max(1, world)))


def render_shard(scene, cam_desc, rank, world, acc, img=None, stream=None, count_rays=False,
                 sample_begin=0, sample_end=0, integrator=0, profile_kernels=False):
    """Launch this rank's share of the frame into device tensors `acc` (H*W*3 f32) / `img` (H*W*3 u8).

    Asynchronous on `stream` (an int hipStream_t, e.g. torch.cuda.current_stream().cuda_stream).
    """
    lib = capi.load()
    o = capi.RenderOpts()
    o.tile_rank, o.tile_world = rank, world
    o.count_rays = 1 if count_rays else 0
    o.sample_begin, o.sample_end = sample_begin, sample_end
    o.integrator = integrator
    o.reserved = 1 if profile_kernels else 0
    capi.check(lib.jtx_mi_render_device(scene.handle, C.byref(cam_desc), C.byref(o),
                                        C.c_void_p(acc.data_ptr()),
                                        C.c_void_p(img.data_ptr()) if img is not None else None,
                                        C.c_void_p(stream) if stream else None))


def reduce_frame(acc, img=None, dst=0, group=None):
    """The per-frame collective: sum the disjoint shards onto rank `dst` (RCCL ring over xGMI)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if img is not None:
        dist.reduce(img, dst=dst, op=dist.ReduceOp.SUM, group=group)
