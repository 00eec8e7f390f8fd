#!/bin/bash
# round 4: the wave-queue variant on the workload where entry tests dominate (synthetic-sm-hard: 30 entries per path ray)
TAG=${1:-r04k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
run() { env CTL_AMD_LIB=$L/$1 $2 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/b.json 2> $OUT/b.err; echo "$1 $2 $(python tools/bench_brief.py < $OUT/b.json | cut -c1-230)"; tail -1 $OUT/b.err | cut -c1-200; }
run libctl_amd.so ""
run libctl_wq.so ""
run libctl_wq.so "CTL_WQ_FLUSH=32"
run libctl_wq.so "CTL_WQ_FLUSH=64"
run libctl_knobs.so "CTL_LEAF_BATCH=24"
run libctl_knobs.so "CTL_LEAF_BATCH=32"
run libctl_knobs.so "CTL_LEAF_BATCH=12"
