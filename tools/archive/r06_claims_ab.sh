#!/bin/bash
# Round 6: static first ray claim + exact-resident traversal grids + LDS stack rows — same-box A/B of library variants (libctl_<v>.so), whole frame and one rank of eight.
TAG=${1:-r06f}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; L=$PWD/cudatracerlib_amd
VARS=${*:-head knobs nostatic r23 r23w6}
for rep in 1 2; do for v in $VARS; do
  echo "== bench $v" | tee -a $OUT/summary.txt
  CTL_AMD_LIB=$L/libctl_$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/last.json 2> $OUT/last.err; python tools/bench_brief.py < $OUT/last.json | tee -a $OUT/summary.txt; tail -2 $OUT/last.err
done; done
for v in $VARS; do
  echo "== shard probe $v" | tee -a $OUT/summary.txt
  CTL_AMD_LIB=$L/libctl_$v.so FUSE=1 timeout 600 python tools/shard_time_probe.py 1 8 2>&1 | tail -2 | tee -a $OUT/summary.txt
done
