#!/bin/bash
# per-kernel times of the model-class shade launches (rocprofv3 --kernel-trace --stats on synthetic-bathroom)
out=gpurun_out/${1:-r04clsp}; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o cls -- python bench.py --workload synthetic-bathroom --steps 20 --warmup 5 --no-cpu-baseline ${2:+--tracer-param $2} > $out/bench.json 2> $out/bench.err
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); cp $f $out/kernel_stats.csv; head -12 $out/kernel_stats.csv | cut -c1-160
rm -rf $out/prof
