#!/usr/bin/env python3
"""Quality of the flattened world-space BVH4, measured WITHOUT a GPU: the oracle renders a small frame of the bench workload in counting mode over the
product's own flattened arrays (the roofline's N_inner / N_tri, bench.py `per_ray`) and prints visits per path ray plus the traversal-cost proxy
`277 N_inner + 365 N_tri` (lane-slots per node step / leaf-entry step of k_intersect at its measured lane utilisation).  Builder knobs come
from the environment ($CTL_FLAT_MAX_LEAF, $CTL_FLAT_NODE_COST, $CTL_FLAT_BINS, $CTL_FLAT_SWEEP, $CTL_FLAT_COLLAPSE ...), one process per setting."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudatracerlib_amd as ctl  # noqa: E402
from cudatracerlib_amd import scenes, api  # noqa: E402
import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=480); ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--instances", type=int, default=2000); ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--workload", default="synthetic-sm")
    ap.add_argument("--format", default="q8", choices=sorted(api.FLAT_FORMATS))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    if a.workload == "synthetic-sm":
        sc = scenes.synthetic_sm(a.width, a.height, n_instances=a.instances, subdiv=a.subdiv)
    elif a.workload == "synthetic-bathroom":
        sc = scenes.synthetic_bathroom(a.width, a.height)
    elif a.workload == "synthetic-sm-hard":   # through the loader, as bench.py builds it (full size: 8.4 M terrain triangles)
        d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_scene_sm_hard_probe_%dx%d" % (a.width, a.height))
        if not os.path.exists(os.path.join(d, "scene.xml")):
            scenes.write_sm_hard_mitsuba(d, a.width, a.height)
        sc = scenes.load_mitsuba(os.path.join(d, "scene.xml"), a.width, a.height)
    else:
        sc = scenes.cornell_box(a.width, a.height, glass_sphere=True)
    t0 = time.time(); fb = api.FlatBvh(sc.desc, api.FLAT_FORMATS[a.format]); t_build = time.time() - t0
    orc = oracle.Oracle()
    import ctypes as C
    orc.lib.orc_slab_probe.argtypes = [C.c_int, C.c_void_p]
    orc.lib.orc_slab_probe(1, None)   # count what an oriented slab in front of the entry fetch would reject (oracle/ocore.h traceRayFlat)
    counts = {}
    t0 = time.time()
    _, rays = orc.render(sc.desc, a.width, a.height, n_passes=1, threads=a.threads, flat=fb.desc, counts=counts, direct=True, max_path_length=8, rr_start=5)
    t_r = time.time() - t0
    pr = counts["path_rays"]
    ni, nt = counts["path_inner"] / pr, counts["path_tri"] / pr
    sl = (C.c_uint64 * 2)(); orc.lib.orc_slab_probe(0, sl)
    print("oriented slabs: %d of %d leaf children that the box test of a slab node lets in (path + occlusion rays) are kept out by the slab = %.1f %%" % (sl[1], sl[0], 100.0 * sl[1] / max(1, sl[0])), flush=True)
    tp = (C.c_uint64 * 8)(); orc.lib.orc_top_probe_read(tp)
    print("node visits (path + occlusion rays) by array position (breadth-first top): " + ", ".join("< %s: %.1f %%" % (n, 100.0 * tp[i] / max(1, tp[7])) for i, n in enumerate(("85", "256", "341", "512", "1365", "5461", "65536"))), flush=True)
    knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CTL_FLAT"))
    print("%-60s nodes %9d leaves %9d depth %2d | per path ray: inner %.2f tri %.2f | cost proxy %.0f | build %.1f s, count %.1f s" % (
        knobs or "(defaults)", fb.desc.n_nodes, fb.desc.n_leaves, fb.desc.max_depth, ni, nt, 277 * ni + 365 * nt, t_build, t_r), flush=True)


if __name__ == "__main__":
    main()
