#!/usr/bin/env python3
"""What one rank of an N-GPU render does, measured on a single GPU: the tracer renders only the tiles of rank 0 of `--world`
ranks (the other ranks do the same amount of work on their own GPUs), so wall time here ~ per-rank time of the N-GPU job
without the final framebuffer reduce.  Also times the host-side sampler-table generation (threads vs. 1 thread)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudatracerlib_amd as ctl
from cudatracerlib_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--instances", type=int, default=2000)
args = ap.parse_args()

g = ctl.SequenceGenerator()
n = 16
t1 = np.ones((n, api.SAMPLER_N1), np.float32); t2 = np.ones((n, 2 * api.SAMPLER_N1), np.float32)
for th in (1, 1, 4, 16, 16):
    t = time.perf_counter()
    api.lib.ctl_sequence_generator_compute_many(g._h, C.c_uint32(n), api._fp(t1), api._fp(t2), C.c_uint32(th))
    print("tables: 16 passes, %2d threads: %6.2f ms" % (th, (time.perf_counter() - t) * 1e3), flush=True)

sc = scenes.synthetic_sm(1920, 1080, n_instances=args.instances)
scene = ctl.Scene(sc.desc, flatten=True)
base = None
for world in args.world:
    tr = ctl.WavefrontPathTracer()
    p = tr.getParameters(); p.setValue("MaxPathLength", 8)
    tr.setTileShard(0, world); tr.Resize(1920, 1080); tr.InitializeScene(scene)
    img = ctl.Image(1920, 1080)
    tr.DoPasses(img, 2, new_trace=True)
    t = time.perf_counter()
    tr.DoPasses(img, args.steps, new_trace=False)
    dt = time.perf_counter() - t
    st = tr.stats()
    if base is None:
        base = dt
    print("world %d: rank-0 share of %d passes in %7.2f ms (%.1f Mrays/s on this rank, x%.2f vs world 1; intersect %.1f shade %.1f ms)"
          % (world, args.steps, dt * 1e3, st.rays_last_pass / dt / 1e6, base / dt, st.ms_intersect + st.ms_intersect_any, st.ms_shade), flush=True)

# load balance: every rank of the largest world in turn (the N-GPU job ends when its slowest rank does)
world = max(args.world)
times, rays = [], []
for rank in range(world):
    tr = ctl.WavefrontPathTracer()
    tr.getParameters().setValue("MaxPathLength", 8)
    tr.setTileShard(rank, world); tr.Resize(1920, 1080); tr.InitializeScene(scene)
    img = ctl.Image(1920, 1080)
    tr.DoPasses(img, 2, new_trace=True)
    t = time.perf_counter()
    tr.DoPasses(img, args.steps, new_trace=False)
    times.append((time.perf_counter() - t) * 1e3); rays.append(tr.stats().rays_last_pass)
print("world %d per-rank ms: %s" % (world, " ".join("%.1f" % x for x in times)))
print("world %d per-rank Mrays: %s" % (world, " ".join("%.1f" % (x / 1e6) for x in rays)))
print("slowest / mean = %.3f  -> strong-scaling ceiling from tile imbalance %.1f %%" % (max(times) / np.mean(times), 100 * np.mean(times) / max(times)))
