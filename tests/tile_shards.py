"""Multi-GPU layer of the path: image tiles shard over ranks, one framebuffer exchange closes a render.

The reference is single-device; this is the only distributed step of the build.  Tiles are 64x64 (the reference's
block-sampler grid in debug builds, Kernel/BlockSampler/IBlockSampler_device.h:6-22); tile t (row-major) belongs to
rank t % world.  Every rank accumulates into a full-size, zero-initialised PixelData frame.  The exchange is ONE gather of
the ranks' packed tiles to rank 0 (ctl_image_gather: ceil(tiles / world) slots of 64 x 64 x 7 floats per rank; pack_tiles /
unpack_tiles below are csrc/comm.cpp's k_tiles in numpy), or — the fallback — one sum-reduce of the whole frames (disjoint
tiles => the sum IS the gather): one RCCL collective over xGMI on GPUs, gloo on CPU.
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


SLOT = TILE + 1   # a packed slot is the 64 x 64 tile plus a one-pixel halo to its right and below (csrc/comm.cpp)


def packed_slots(width, height, world):
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    return (tx * ty + world - 1) // world


def pack_tiles(frame, rank, world):
    """(h, w, 7) frame -> (slots, 65 * 65, 7): slot k = tile k * world + rank, row-major 65 x 65 pixels — the tile, and in row / column 64 its HALO: the frame's
    pixels just right of / below the tile where another rank owns them (zero where the same rank does: they travel with that tile).  The clipped part of a border tile
    and a missing last slot are zero (= ctl_image_pack_tiles, csrc/comm.cpp k_pack_tiles)"""
    h, w, c = frame.shape
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    own = tile_owner(w, h, world)
    out = np.zeros((packed_slots(w, h, world), SLOT, SLOT, c), frame.dtype)
    for k in range(out.shape[0]):
        t = k * world + rank
        if t >= tx * ty:
            continue
        y0, x0 = (t // tx) * TILE, (t % tx) * TILE
        blk = frame[y0:y0 + SLOT, x0:x0 + SLOT].copy()
        foreign = own[y0:y0 + SLOT, x0:x0 + SLOT] != rank
        halo = np.zeros(blk.shape[:2], bool); halo[TILE:, :] = True; halo[:, TILE:] = True
        blk[halo & ~foreign] = 0
        # a foreign pixel at the corner of its tile lies in the halo of up to three tiles, and the rank may own more than one of them: it travels with exactly one —
        # the tile on its left before the one above before the diagonal one (slot_pixel in csrc/comm.cpp)
        if blk.shape[0] > TILE and t % tx != 0 and (t + tx - 1) % world == rank:
            blk[TILE, 0] = 0
        if blk.shape[0] > TILE and blk.shape[1] > TILE and ((t + tx) % world == rank or (t + 1) % world == rank):
            blk[TILE, TILE] = 0
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(out.shape[0], SLOT * SLOT, c)


def unpack_tiles(frame, world, packed_all):
    """write the packed slots of ALL ranks ((world, slots, 65 * 65, 7), rank-major) into the (h, w, 7) frame, in place: every tile is copied, THEN every halo is added
    (= ctl_image_unpack_tiles: k_unpack_tiles + k_add_halos)"""
    h, w, c = frame.shape
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    p = np.asarray(packed_all).reshape(world, -1, SLOT, SLOT, c)
    for phase in (0, 1):
        for r in range(world):
            for k in range(p.shape[1]):
                t = k * world + r
                if t >= tx * ty:
                    continue
                y0, x0 = (t // tx) * TILE, (t % tx) * TILE
                if phase == 0:
                    hh, ww = min(TILE, h - y0), min(TILE, w - x0)
                    frame[y0:y0 + hh, x0:x0 + ww] = p[r, k, :hh, :ww]
                else:
                    hh, ww = min(SLOT, h - y0), min(SLOT, w - x0)
                    add = p[r, k, :hh, :ww].copy(); add[:min(TILE, hh), :min(TILE, ww)] = 0
                    frame[y0:y0 + hh, x0:x0 + ww] += add
    return frame


def gather_framebuffer(frame, dst=0):
    """ONE gather of every rank's packed tiles to rank `dst` (ctl_image_gather_to's contract over torch.distributed): returns the complete (h, w, 7) frame on `dst`
    (None elsewhere); `frame` (this rank's own numpy frame) is left as it is."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = torch.from_numpy(np.ascontiguousarray(pack_tiles(frame, rank, world)))
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst)
    if rank != dst:
        return None
    return unpack_tiles(frame.copy(), world, torch.stack(bufs).numpy())
