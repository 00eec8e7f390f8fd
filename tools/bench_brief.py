#!/usr/bin/env python3
"""one-line digest of a bench.py JSON line on stdin"""
import json, sys
for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        continue
    b = json.loads(l); r = b.get("roofline", {})
    print("  %.1f Mrays/s  %.3f ms/step  traversal %.2f shade %.2f ms | %s  avg launch %.3f ms  per ray %s  util %s  frac %.3f" % (
        b["value"], b["ms_per_step"], r.get("ms_intersect", 0) / b["steps"], r.get("ms_shade", 0) / b["steps"], r.get("kernel", "")[:16], r.get("avg_launch_ms", 0),
        r.get("per_ray"), r.get("lane_utilisation"), r.get("frac") or 0.0))
