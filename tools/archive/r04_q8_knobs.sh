#!/bin/bash
# round 4: scheduling thresholds of the 8-wide kernel (knobs build), one box
TAG=${1:-r04b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_knobs.so
for s in "CTL_LEAF_BATCH=16" "CTL_LEAF_BATCH=8" "CTL_LEAF_BATCH=24" "CTL_LEAF_BATCH=32" "CTL_LEAF_BATCH=48" "CTL_LEAF_BATCH=24 CTL_REFILL_IDLE=8" "CTL_LEAF_BATCH=24 CTL_REFILL_IDLE=20"; do
  env $s timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flat-format q8 > $OUT/b.json 2> $OUT/b.err
  echo "$s $(python tools/bench_brief.py < $OUT/b.json | cut -c1-200)"
done
