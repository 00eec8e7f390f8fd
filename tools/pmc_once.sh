#!/bin/bash
# one extra PMC pass over a bench run: tools/pmc_once.sh <tag> "<counters>" [bench args]; prints per-kernel sums of each counter
TAG=$1; CTRS=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc -o p -- python bench.py "$@" --no-cpu-baseline > $OUT/pmc.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("ctl::", "").replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
with open(sys.argv[1] + "/pmc_sums.txt", "w") as o:
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        line = "%-28s " % k[:28] + "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()))
        print(line); o.write(line + "\n")
PY
find $OUT/pmc -name '*.csv' -size +4M -delete
