#!/bin/bash
# same-box A/B of library variants on the bench workload: tools/r05_ab.sh <tag> <variant> [rounds]  (cudatracerlib_amd/libctl_<variant>.so against libctl_amd.so)
TAG=${1:-r05ab}; V=${2:-anyu}; N=${3:-3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; L=$PWD/cudatracerlib_amd
for i in $(seq $N); do for v in amd $V; do CTL_AMD_LIB=$L/libctl_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_EXTRA} > $OUT/b.json 2>$OUT/err; python - $OUT/b.json $v <<'PY' | tee -a $OUT/ab.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("%-6s %8.1f Mrays/s  ms_intersect %7.2f  ms_shade %6.2f  per path ray %s  per shadow ray %s" % (sys.argv[2], j["value"], r["ms_intersect"], r["ms_shade"], r["per_ray"], r["per_shadow_ray"]))
PY
done; done
