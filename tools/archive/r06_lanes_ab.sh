#!/bin/bash
# Round 6: pipeline lanes A/B on one box — bench.py (driver style) under the settings of the heredoc / file: "<library variant> <lanes> [ENV=value ...]"
# tools/archive/r06_lanes_ab.sh <tag> [workload] < settings
TAG=${1:-r06c}; WL=${2:-synthetic-sm}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; L=$PWD/cudatracerlib_amd
while read -r lib lanes envs; do
  [ -z "$lib" ] && continue
  echo "== $WL lib $lib lanes $lanes $envs" | tee -a $OUT/summary.txt
  env CTL_AMD_LIB=$L/libctl_$lib.so $envs timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $WL  > $OUT/last.json 2> $OUT/last.err
  python tools/bench_brief.py < $OUT/last.json | tee -a $OUT/summary.txt; tail -2 $OUT/last.err; cat $OUT/last.json >> $OUT/runs.jsonl
done
