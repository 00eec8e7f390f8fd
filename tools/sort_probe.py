#!/usr/bin/env python3
"""What re-sorting a bounce's rays could buy the traversal kernel, measured before building a device-side sort:
second- and third-bounce rays of the bench scene (diffuse-like directions from the previous hit points, several passes worth) are timed through
ctl_intersect_device in (a) the order the wavefront produces (parent's queue order), (b) a random permutation, (c) sorted on the HOST by
(Morton code of the origin cell in a G^3 grid, direction octant) and (d) by (octant, Morton), for several G.  The sort itself is free here: the
numbers are the upper bound of what a device-side sort can return.  GPU box only."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudatracerlib_amd as ctl  # noqa: E402
from cudatracerlib_amd import scenes  # noqa: E402
from tools.intersect_bench import camera_rays, bounce_rays, run  # noqa: E402


def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def keys(rays, lo, hi, G, octant_major):
    c = np.clip(((rays[:, :3] - lo) / (hi - lo) * G).astype(np.int64), 0, G - 1)
    m = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    o = ((rays[:, 4] < 0).astype(np.uint64)) | ((rays[:, 5] < 0).astype(np.uint64) << 1) | ((rays[:, 6] < 0).astype(np.uint64) << 2)
    return (o << 32) | m if octant_major else (m << 3) | o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--instances", type=int, default=2000); ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--passes", type=int, default=4); ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    sc = scenes.synthetic_sm(a.width, a.height, n_instances=a.instances, subdiv=a.subdiv)
    t0 = time.time(); scene = ctl.Scene(sc.desc, flatten=True); print("scene upload + flatten %.1f s" % (time.time() - t0), flush=True)
    prim = camera_rays(sc.desc, a.width, a.height)
    hits = ctl.intersect(scene, prim)
    b1 = np.concatenate([bounce_rays(sc.desc, prim, hits, seed=s) for s in range(a.passes)])     # rays of bounce 2, `passes` passes in queue order
    h1 = ctl.intersect(scene, b1)
    b2 = bounce_rays(sc.desc, b1, h1, seed=77)                                                   # rays of bounce 3
    for name, rays in (("bounce2", b1), ("bounce3", b2)):
        lo = rays[:, :3].min(0); hi = rays[:, :3].max(0) + 1e-3
        n = len(rays)
        base = None
        rs = np.random.RandomState(5)
        variants = [("queue order", None), ("random permutation", rs.permutation(n))]
        for G in (8, 32, 128, 512):
            variants.append(("sorted (cell %d^3, octant)" % G, np.argsort(keys(rays, lo, hi, G, False), kind="stable")))
        variants.append(("sorted (octant, cell 32^3)", np.argsort(keys(rays, lo, hi, 32, True), kind="stable")))
        variants.append(("sorted (octant only)", np.argsort(keys(rays, lo, hi, 1, True), kind="stable")))
        for any_hit in (False, True):
            for vname, perm in variants:
                r = rays if perm is None else np.ascontiguousarray(rays[perm])
                ms = run(scene, r, any_hit, a.reps)
                c = ctl.intersect_count(scene, r, any_hit=any_hit)
                if base is None or vname == "queue order":
                    base = ms
                print("%-8s %-7s %-30s n=%d  %.3f ms  %.1f Mrays/s  x%.3f vs queue order | util inner %.3f tri %.3f" % (
                    name, "any" if any_hit else "closest", vname, n, ms, n / ms / 1e3, base / ms,
                    c["n_inner"] / max(1, 64 * c["wave_inner_iters"]), c["n_tri"] / max(1, 64 * c["wave_tri_iters"])), flush=True)


if __name__ == "__main__":
    main()
