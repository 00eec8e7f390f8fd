#!/bin/bash
# per-dispatch durations of the traversal / shade launches of one rank of N (tools/shard_time_probe.py under rocprofv3 --kernel-trace): tools/launch_trace.sh <tag> <world>
TAG=$1; WORLD=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
FUSE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/shard_time_probe.py $WORLD > $OUT/probe.txt 2>$OUT/err.log
python - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ctl::", "").replace("void ", "")[:34]))
rows.sort()
# the last DoPasses call: from the last k_raygen on
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_raygen"))
t0 = rows[last][0]; prev_end = None
with open(sys.argv[1] + "/launches.txt", "w") as o:
    for s, e, k in rows[last:]:
        line = "%-36s start %9.3f ms  dur %8.3f ms  gap before %7.3f ms" % (k, (s - t0) * 1e-6, (e - s) * 1e-6, 0.0 if prev_end is None else (s - prev_end) * 1e-6)
        print(line); o.write(line + "\n"); prev_end = e
PY
rm -rf $OUT/trace
