#!/usr/bin/env python3
"""Kernel-tuning harness for row a7 (k_intersect): times ctl_intersect_device on (a) coherent camera rays and
(b) incoherent bounce rays (cosine-distributed directions from the primary hit points) of the bench scene and prints
ms, Mrays/s, per-ray N_inner/N_tri/N_inst and lane utilisation.  GPU box only."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudatracerlib_amd as ctl  # noqa: E402
from cudatracerlib_amd import scenes  # noqa: E402


def camera_rays(desc, w, h):
    cam = desc.camera
    tw = np.array(cam.to_world[:], np.float64).reshape(4, 4)
    aspect = w / h
    t = np.tan(cam.fov / 2)
    ys, xs = np.mgrid[0:h, 0:w]
    # 8x8 micro-tile order like the tracer's ray-gen (a wave = 8x8 pixels)
    order = ((ys // 8) * (w // 8) + (xs // 8)) * 64 + (ys % 8) * 8 + (xs % 8)
    px = (xs + 0.5) / w; py = (ys + 0.5) / h
    d = np.stack([(1 - 2 * px) * t, (1 - 2 * py) * t / aspect, np.ones_like(px)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    dw = d @ tw[:3, :3].T
    rays = np.zeros((h * w, 8), np.float32)
    idx = np.argsort(order.ravel(), kind="stable")
    rays[:, :3] = tw[:3, 3]; rays[:, 3] = desc.ray_trace_eps; rays[:, 4:7] = dw.reshape(-1, 3)[idx]; rays[:, 7] = 3.4e38
    return rays


def bounce_rays(desc, rays, hits, seed=1):
    rs = np.random.RandomState(seed)
    ok = hits["tri_idx"] >= 0
    P = rays[ok, :3] + hits["dist"][ok, None] * rays[ok, 4:7]
    n = len(P)
    # uniform sphere directions flipped away from the incoming ray (stand-in for cosine lobes around unknown normals)
    d = rs.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = (d * rays[ok, 4:7]).sum(1) > 0
    d[flip] *= -1
    out = np.zeros((n, 8), np.float32)
    out[:, :3] = P; out[:, 3] = desc.ray_trace_eps; out[:, 4:7] = d; out[:, 7] = 3.4e38
    return out


def run(scene, rays, any_hit, reps):
    n = len(rays)
    ro = np.ascontiguousarray(rays[:, :4]); rd = np.ascontiguousarray(rays[:, 4:])
    ptr = [C.c_void_p() for _ in range(4)]
    for p, nbytes in zip(ptr, (n * 16, n * 16, n * 16, n * 4)):
        ctl.api._check(ctl.lib.ctl_device_malloc(nbytes, C.byref(p)))
    ctl.api._check(ctl.lib.ctl_memcpy_h2d(ptr[0], ro.ctypes.data, n * 16)); ctl.api._check(ctl.lib.ctl_memcpy_h2d(ptr[1], rd.ctypes.data, n * 16))
    best = 1e9
    for _ in range(reps):
        ms = C.c_float()
        ctl.api._check(ctl.lib.ctl_intersect_device(scene._h, ptr[0], ptr[1], n, ptr[2], ptr[3], 1 if any_hit else 0, C.byref(ms)))
        best = min(best, ms.value)
    for p in ptr:
        ctl.lib.ctl_device_free(p)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--instances", type=int, default=2000); ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5); ap.add_argument("--flatten", type=int, default=0)
    a = ap.parse_args()
    sc = scenes.synthetic_sm(a.width, a.height, n_instances=a.instances, subdiv=a.subdiv)
    import time
    t0 = time.time(); scene = ctl.Scene(sc.desc, flatten=bool(a.flatten)); print("scene upload%s %.1f s" % (" + flatten" if a.flatten else "", time.time() - t0), flush=True)
    prim = camera_rays(sc.desc, a.width, a.height)
    hits = ctl.intersect(scene, prim)
    sec = bounce_rays(sc.desc, prim, hits)
    for name, rays in (("primary", prim), ("bounce", sec)):
        for any_hit in (False, True):
            ms = run(scene, rays, any_hit, a.reps)
            c = ctl.intersect_count(scene, rays, any_hit=any_hit)
            n = len(rays)
            print("%-8s %-7s n=%d  %.3f ms  %.1f Mrays/s  per-ray inner %.1f tri %.1f inst %.2f  util inner %.3f tri %.3f" % (
                name, "any" if any_hit else "closest", n, ms, n / ms / 1e3, c["n_inner"] / n, c["n_tri"] / n, c["n_inst"] / n,
                c["n_inner"] / max(1, 64 * c["wave_inner_iters"]), c["n_tri"] / max(1, 64 * c["wave_tri_iters"])), flush=True)


if __name__ == "__main__":
    main()
