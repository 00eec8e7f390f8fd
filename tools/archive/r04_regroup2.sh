#!/bin/bash
TAG=${1:-r04n}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
run() { env CTL_AMD_LIB=$L/$1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 > $OUT/b.json 2> $OUT/b.err; echo "$1 $2 $(python tools/bench_brief.py < $OUT/b.json | cut -c1-90)"; }
for i in 1 2; do for v in amd bs128 bs256 bs512; do run libctl_$v.so ""; done; done
