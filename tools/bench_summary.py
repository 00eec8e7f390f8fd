import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["value"], "Mrays/s", d["ms_per_step"], "ms/pass | traversal", round(r["ms_intersect"]/d["steps"],2), "shade", round(r["ms_shade"]/d["steps"],2), "frac", r["frac"])
