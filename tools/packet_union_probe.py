#!/usr/bin/env python3
"""Feasibility of a PACKET traversal for the coherent first bounce, measured without building it (CPU, the oracle's counting traversal of the product's flattened tree):
the 64 primary rays of an 8 x 8 pixel block (= one wave of k_raygen's order) — node steps and entry tests ray by ray against the UNION of nodes / entries the block looks at
(what a wave-uniform traversal would step through, every lane testing everything).  python tools/packet_union_probe.py [workload: sm | bathroom]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cudatracerlib_amd import scenes, api
import oracle
which = sys.argv[1] if len(sys.argv) > 1 else "sm"
W, H = 1920, 1080
api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
sc = scenes.synthetic_sm(W, H, n_instances=2000) if which == "sm" else scenes.synthetic_bathroom(W, H)
d = sc.desc
fb = api.FlatBvh(d, api.FLAT_Q4)
orc = oracle.Oracle()
orc.lib.orc_packet_union_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
out = (C.c_uint64 * 6)()
orc.lib.orc_set_flat_bvh(C.addressof(fb.desc))
orc.lib.orc_packet_union_probe(C.addressof(d), W, H, 1500, out)
orc.lib.orc_set_flat_bvh(None)
rays, n_in, n_en, u_in, u_en, blocks = [int(v) for v in out]
print("workload %s: %d blocks of 8 x 8 primary rays" % (which, blocks))
print("  per ray, traced alone:        %.1f node steps + %.1f entry tests" % (n_in / rays, n_en / rays))
print("  per block, union of the 64:   %.1f nodes + %.1f entries  (= %.2f x / %.2f x one ray's)" % (u_in / blocks, u_en / blocks, u_in / blocks / (n_in / rays), u_en / blocks / (n_en / rays)))
# VALU wave-instructions per 64 rays: one ray per lane (253 per node step at 0.78 of the lanes busy, 149 per entry test at 0.34: DESIGN §3) against a wave-uniform packet
# (~80 per node step — scalar node fetch, 6 fma + 2 min / max per child and lane, no per-lane stack / links / ordering — and the same 149 per entry, every lane testing every entry)
alone = (n_in / rays) * 253 / 0.78 + (n_en / rays) * 149 / 0.34
packet = (u_in / blocks) * 80 + (u_en / blocks) * 149
print("  VALU wave-instructions per wave of 64 rays: one ray per lane %.0f, packet %.0f  (%.2f x)" % (alone, packet, packet / alone))
