#!/bin/bash
# round 4: entry tests from a wave-wide queue (traverse_flat_wq.h) against the parked-leaf kernel, one box
TAG=${1:-r04c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
CTL_AMD_LIB=$L/libctl_wq.so timeout 900 python -m pytest tests/test_gpu_intersect.py -x -q -m gpu -k "not explicit" > $OUT/pytest_wq.log 2>&1; tail -3 $OUT/pytest_wq.log
run() { env CTL_AMD_LIB=$L/$1 $2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; echo "$1 $2 $(python tools/bench_brief.py < $OUT/b.json | cut -c1-210)"; tail -1 $OUT/b.err | cut -c1-200; }
run libctl_amd.so ""
run libctl_wq.so ""
run libctl_amd.so ""
for s in ${SWEEP:-"CTL_WQ_FLUSH=32" "CTL_WQ_FLUSH=64" "CTL_WQ_MIN_INNER=32"}; do run libctl_wq.so "$s"; done
