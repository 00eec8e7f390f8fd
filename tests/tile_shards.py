"""Multi-GPU layer of the path: image tiles shard over ranks, one framebuffer exchange closes a render.

The reference is single-device; this is the only distributed step of the build.  Tiles are 64x64 (the reference's
block-sampler grid in debug builds, Kernel/BlockSampler/IBlockSampler_device.h:6-22); tile t (row-major) belongs to
rank t % world.  Every rank accumulates into a full-size, zero-initialised PixelData frame, so the exchange is a
sum-reduce to rank 0 (disjoint tiles => the sum IS the gather) — one RCCL collective over xGMI on GPUs, gloo on CPU.
"""
import numpy as np

TILE = 64


def tile_owner(width, height, world):
    """(height, width) int array: rank that renders each pixel"""
    tx = (width + TILE - 1) // TILE
    ys, xs = np.mgrid[0:height, 0:width]
    return ((ys // TILE) * tx + (xs // TILE)) % world


def tile_mask(width, height, rank, world):
    return tile_owner(width, height, world) == rank


def local_pixel_count(width, height, rank, world):
    """paths a rank generates per pass, incl. the clipped lanes of border tiles (= shard_pixel_count in kernels.h)"""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    nt = tx * ty
    return (nt // world + (1 if rank < nt % world else 0)) * TILE * TILE


def reduce_framebuffer(fb, dst=0):
    """sum the ranks' PixelData frames into rank `dst` (torch tensor, in place). One collective per render."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and fb.is_cuda:   # gloo only all-reduces device tensors (1-GPU test mode of bench.py)
            dist.all_reduce(fb, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def reduce_framebuffer_to(src, dst=0):
    """out of place (the per-pass gather of a progressive display, ctl_image_reduce_to): returns the sum of the ranks' frames on rank `dst` (None elsewhere);
    every rank's own cumulative frame `src` is left as it is, so the call can be repeated after every pass"""
    import torch.distributed as dist
    out = src.clone()
    reduce_framebuffer(out, dst=dst)
    return out if (not dist.is_initialized() or dist.get_rank() == dst) else None
