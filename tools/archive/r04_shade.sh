#!/bin/bash
# round 4: where k_shade_basic spends its time
TAG=${1:-r04l}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
python tools/shade_basic_probe.py 2>$OUT/err.log | tee $OUT/probe.jsonl
for v in p1 p2; do CTL_AMD_LIB=$L/libctl_$v.so PROBE_SCENE_VARIANTS=0 python tools/shade_basic_probe.py 2>>$OUT/err.log | tee -a $OUT/probe.jsonl; done
tail -3 $OUT/err.log
