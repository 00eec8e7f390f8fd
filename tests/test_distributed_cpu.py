"""N > 1 path on CPU: two processes over gloo run the host logic bench.py uses for multi-GPU — tile ownership, a
per-rank full-size PixelData frame, ONE sum-reduce to rank 0 (tests/tile_shards.py).  The per-rank radiance
comes from the oracle (the GPUs are not here); what is tested is that the shards partition the film and that the single
collective reproduces the one-rank frame exactly."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from cudatracerlib_amd import scenes
    import tile_shards as parallel
    sc = scenes.cornell_box(w, h)
    orc = oracle.Oracle()
    tables = orc.sequence_tables(1)
    full, _ = orc.render(sc.desc, w, h, n_passes=1, tables=tables, max_path_length=4, threads=2)
    mine = full * parallel.tile_mask(w, h, rank, world)[..., None]          # this rank's tiles, zeros elsewhere
    fb = torch.from_numpy(np.ascontiguousarray(mine.reshape(-1)))
    parallel.reduce_framebuffer(fb, dst=0)                                     # the one exchange step
    if rank == 0:
        np.save(out_path, np.stack([fb.numpy().reshape(h, w, 7), full]))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_ownership_partitions_the_film():
    import tile_shards as parallel
    for (w, h, world) in ((192, 128, 2), (1920, 1080, 8), (100, 70, 3)):
        own = parallel.tile_owner(w, h, world)
        assert own.shape == (h, w) and own.min() == 0 and own.max() == min(world, ((w + 63) // 64) * ((h + 63) // 64)) - 1
        masks = [parallel.tile_mask(w, h, r, world) for r in range(world)]
        assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((h, w), int))
        # per-rank path capacity (incl. clipped border lanes) covers the owned pixels
        for r in range(world):
            assert parallel.local_pixel_count(w, h, r, world) >= masks[r].sum()
    # load balance at the benchmark size: 510 tiles over 8 ranks
    counts = [parallel.tile_mask(1920, 1080, r, 8).sum() for r in range(8)]
    assert max(counts) / min(counts) < 1.1


def test_two_rank_gloo_reduce_reproduces_single_rank_frame(tmp_path):
    import torch.multiprocessing as mp
    w, h = 192, 128   # 3 x 2 tiles
    out = str(tmp_path / "fb.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got, full = np.load(out)
    assert np.array_equal(got, full)   # x + 0 == x: bit-identical to the one-rank frame
    assert full[..., 6].sum() == w * h   # one sample per pixel per pass (a jittered sample may land in the neighbouring pixel)


def _worker_progressive(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from cudatracerlib_amd import scenes
    import tile_shards as parallel
    sc = scenes.cornell_box(w, h)
    orc = oracle.Oracle()
    tables = orc.sequence_tables(3)
    mask = parallel.tile_mask(w, h, rank, world)[..., None]
    shown = []
    for k in range(1, 4):   # the rank's cumulative tile frame after pass k; the display frame is gathered after EVERY pass
        full, _ = orc.render(sc.desc, w, h, n_passes=k, tables=tables[:k], max_path_length=3, threads=2)
        mine = torch.from_numpy(np.ascontiguousarray((full * mask).reshape(-1)))
        disp = parallel.reduce_framebuffer_to(mine, dst=0)
        assert np.array_equal(mine.numpy().reshape(h, w, 7), full * mask)        # the source is untouched
        if rank == 0:
            shown.append(disp.numpy().reshape(h, w, 7).copy())
    end = mine.clone(); parallel.reduce_framebuffer(end, dst=0)                 # ONE in-place reduce at the end of the render
    if rank == 0:
        np.save(out_path, np.stack(shown + [end.numpy().reshape(h, w, 7), full]))
    dist.barrier()
    dist.destroy_process_group()


def test_per_pass_gathers_end_with_the_frame_of_one_final_reduce(tmp_path):
    """progressive display (north_star: a gather of the framebuffer at the end of each pass): K out-of-place per-pass gathers (ctl_image_reduce_to's contract) show the
    cumulative frame after every pass and end, bit for bit, with what ONE end-of-render in-place reduce gives — and with the one-rank frame"""
    import torch.multiprocessing as mp
    w, h = 192, 128
    out = str(tmp_path / "prog.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_progressive, args=(r, 2, port, w, h, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    s1, s2, s3, end, full = np.load(out)
    assert np.array_equal(s3, end) and np.array_equal(end, full)
    assert s1[..., 6].sum() == w * h and s2[..., 6].sum() == 2 * w * h and s3[..., 6].sum() == 3 * w * h


def _spilled_frames(w, h, world, seed=11):
    """what the ranks of a render hold: `full` = the one-rank frame; per rank its own tiles of it, where a few samples of pixels on a tile's right / bottom edge were
    accumulated one pixel further — in ANOTHER rank's tile (pixel + jitter rounds up, compaction.h add_sample_ordered) — integer-valued so that every sum is exact"""
    import tile_shards as parallel
    rs = np.random.RandomState(seed)
    own = parallel.tile_owner(w, h, world)
    base = rs.randint(0, 50, size=(h, w, 7)).astype(np.float32)
    frames = [np.where((own == r)[..., None], base, np.float32(0)) for r in range(world)]
    full = base.copy()
    n_spill = 0
    for y in range(h):
        for x in range(w):
            for (dx, dy) in ((1, 0), (0, 1), (1, 1)):
                xx, yy = x + dx, y + dy
                if xx < w and yy < h and own[yy, xx] != own[y, x] and rs.uniform() < 0.25:
                    v = rs.randint(1, 9, size=7).astype(np.float32)
                    frames[own[y, x]][yy, xx] += v; full[yy, xx] += v; n_spill += 1
    assert n_spill > 20
    return full, frames


def _worker_gather(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tile_shards as parallel
    full, frames = _spilled_frames(w, h, world)                                               # the same on every rank (seeded)
    mine = frames[rank]
    before = mine.copy()
    got = parallel.gather_framebuffer(mine, dst=0)                                            # ONE gather of ceil(tiles / world) packed slots per rank
    assert np.array_equal(mine, before)                                                       # the source is untouched: the call serves the per-pass exchange too
    red = torch.from_numpy(mine.reshape(-1).copy()); parallel.reduce_framebuffer(red, dst=0)  # the fallback: one sum-reduce of the whole frames
    again = parallel.gather_framebuffer(mine, dst=0)                                          # out of place: repeatable
    if rank == 0:
        np.save(out_path, np.stack([got, red.numpy().reshape(h, w, 7), full, again]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 192, 160), (3, 320, 100), (3, 1920 // 4, 1080 // 4)])
def test_gather_of_packed_tiles_equals_the_reduce(tmp_path, world, w, h):
    """north_star's exchange — a gather of the framebuffer — on 2 and 3 ranks over gloo: uneven tile counts (9 tiles / 2 ranks, 10 tiles / 3 ranks: the last slot of
    some ranks is padding), clipped border tiles, and samples a rank accumulated one pixel inside ANOTHER rank's tile (they travel in the slot's halo and are added on the
    root); the gathered frame == the reduced frame == the one-rank frame, bit for bit (integer-valued data: every sum is exact); a repeat gives the same frame"""
    import torch.multiprocessing as mp
    import tile_shards as parallel
    nt = ((w + 63) // 64) * ((h + 63) // 64)
    assert nt % world != 0 or h % 64 != 0
    out = str(tmp_path / "g.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_gather, args=(r, world, port, w, h, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got, red, full, again = np.load(out)
    assert np.array_equal(got, full) and np.array_equal(red, full) and np.array_equal(again, full)
    assert parallel.packed_slots(w, h, world) * world >= nt


def test_pack_unpack_round_trip_and_sizes():
    import tile_shards as parallel
    assert parallel.packed_slots(1920, 1080, 8) * 65 * 65 * 28 == 7571200          # 7.6 MB per rank at the benchmark size, against 58 MB for the reduce
    for (w, h, world) in ((100, 70, 3), (64, 64, 1), (130, 65, 4), (192, 128, 8), (200, 150, 2)):
        full, frames = _spilled_frames(w, h, world, seed=w) if world > 1 and (w > 64 or h > 64) else (np.random.RandomState(1).randint(0, 9, size=(h, w, 7)).astype(np.float32), None)
        if frames is None:
            frames = [full]
        packed = np.stack([parallel.pack_tiles(frames[r], r, world) for r in range(world)])
        assert packed.shape == (world, parallel.packed_slots(w, h, world), 65 * 65, 7)
        out = parallel.unpack_tiles(np.full_like(full, -1), world, packed)
        assert np.array_equal(out, full)
