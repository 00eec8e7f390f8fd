#!/usr/bin/env python3
"""GPU frame against both oracle builds (glibc / shared math), fraction of pixels within the render tolerance by path depth — where GPU and checker part ways."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cudatracerlib_amd as ctl
from cudatracerlib_amd import scenes
import oracle

w, h = 256, 144
sc = scenes.synthetic_sm(w, h, n_instances=int(sys.argv[1]) if len(sys.argv) > 1 else 400, subdiv=3); d = sc.desc
libm, sm = oracle.Oracle(), oracle.Oracle(shared_math=True)
tables = libm.sequence_tables(2)
scene = ctl.Scene(d)
for depth in (1, 2, 3, 4, 8):
    tr = ctl.WavefrontPathTracer(); tr.getParameters().setValue("MaxPathLength", depth); tr.Resize(w, h); tr.InitializeScene(scene)
    img = ctl.Image(w, h)
    for k in range(2):
        tr.setSamplerTables(*tables[k]); tr.DoPass(img, new_trace=(k == 0))
    got = img.getPixelData()
    out = []
    for o in (libm, sm):
        want, _ = o.render(d, w, h, n_passes=2, tables=tables, max_path_length=depth, threads=os.cpu_count() or 8)
        ok = (np.abs(got[..., :3] - want[..., :3]) <= 2e-3 * (1 + np.abs(want[..., :3]))).all(axis=2)
        exact = (got[..., :3] == want[..., :3]).all(axis=2)
        out.append("%.4f (bit-equal %.4f)" % (ok.mean(), exact.mean()))
    print("depth %d: vs glibc oracle %s | vs shared-math oracle %s" % (depth, out[0], out[1]), flush=True)
