#!/bin/bash
TAG=${1:-r04g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
run() { env CTL_AMD_LIB=$L/$1 $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; echo "$1 $2 $(python tools/bench_brief.py < $OUT/b.json | cut -c1-210)"; tail -1 $OUT/b.err | cut -c1-200; }
run libctl_amd.so ""
for s in "" "CTL_WQ_FLUSH=32" "CTL_WQ_FLUSH=24" "CTL_WQ_FLUSH=16" "CTL_WQ_FLUSH=64"; do run libctl_wqp.so "$s"; done
run libctl_wq.so "CTL_WQ_FLUSH=24"
run libctl_wq.so "CTL_WQ_FLUSH=16"
