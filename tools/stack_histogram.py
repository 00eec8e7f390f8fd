#!/usr/bin/env python3
"""Traversal-stack depth of the bench workload's rays: one counting pass of bench.py's frame (synthetic-SM 1080p depth 8 by default), then
ctl_traversal_stack_histogram — how many rays used which deepest stack entry, and how many went past the LDS rows (csrc/traverse_flat.h kFlatLdsRows = 19)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cudatracerlib_amd as ctl
from cudatracerlib_amd import scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "synthetic-sm"
w, h = 1920, 1080
ctl.api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
sc = scenes.synthetic_sm(w, h) if wl == "synthetic-sm" else (scenes.synthetic_bathroom(w, h) if wl == "synthetic-bathroom" else scenes.cornell_box(1024, 1024, glass_sphere=True))
if wl == "cornell-glass": w = h = 1024
scene = ctl.Scene(sc.desc, flatten=True)
tr = ctl.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", 8); tr.Resize(w, h); tr.InitializeScene(scene)
img = ctl.Image(w, h)
hist = (C.c_uint64 * 96)()
ctl.lib.ctl_traversal_stack_histogram.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
ctl.api._check(ctl.lib.ctl_traversal_stack_histogram(hist, 96, 1))
tr.setCounting(True); tr.DoPasses(img, 1, new_trace=True)
ctl.api._check(ctl.lib.ctl_traversal_stack_histogram(hist, 96, 1))
a = np.array(hist[:], np.float64); n = a.sum()
print("%s: %d rays (path + shadow) of one pass; deepest stack entry used -> share of rays" % (wl, int(n)))
for d in range(96):
    if a[d]: print("  %2d  %9d  %7.4f %%" % (d, int(a[d]), 100 * a[d] / n))
print("rays past the 19 LDS rows (entries in scratch): %d = %.5f %%; deepest: %d" % (int(a[19:].sum()), 100 * a[19:].sum() / n, int(np.nonzero(a)[0].max())))
