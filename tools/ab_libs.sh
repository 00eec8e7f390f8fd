#!/bin/bash
# same-box comparison of library variants: tools/ab_libs.sh <tag> <variant> [<variant> ...]   (cudatracerlib_amd/libctl_<variant>.so), two rounds; extra bench args in $BENCH_ARGS
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for i in 1 2; do
for v in "$@"; do
CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS > gpurun_out/$TAG/$v$i.json 2>gpurun_out/$TAG/$v.err
echo "$v $i $(python tools/bench_brief.py < gpurun_out/$TAG/$v$i.json | cut -c1-120)"
done
done
