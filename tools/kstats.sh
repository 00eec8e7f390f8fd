#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py run: tools/kstats.sh <tag> [bench args]
TAG=$1; shift; out=gpurun_out/$TAG; mkdir -p $out; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; head -14 $out/kernel_stats.csv | cut -c1-150; rm -rf $out/prof
