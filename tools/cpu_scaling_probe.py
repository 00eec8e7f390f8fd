#!/usr/bin/env python3
"""Oracle (CPU baseline) rays/s by thread count on the bench workload: where does the host stop scaling?  GPU box or any host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudatracerlib_amd import scenes, api
import oracle
orc = oracle.Oracle()
api.set_cache_dir(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ctl_amd_cache"))
sc = scenes.synthetic_sm(1920, 1080, n_instances=2000, subdiv=4)
d = sc.desc
print("cpus", os.cpu_count(), flush=True)
os.system("lscpu | grep -i 'model name\\|socket\\|numa node' | head -8")
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    rows = (400, min(1080, 400 + max(8, 4 * th)))
    t = time.time(); _, rays = orc.render(d, 1920, 1080, n_passes=1, max_path_length=8, threads=th, rows=rows); dt = time.time() - t
    print("%4d threads  rows %s  %8.3f Mrays/s  %.2f s" % (th, rows, rays / dt / 1e6, dt), flush=True)
