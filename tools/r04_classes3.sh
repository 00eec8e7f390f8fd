#!/bin/bash
out=gpurun_out/${1:-r04cls4}; mkdir -p $out; L=$PWD/cudatracerlib_amd
for v in ${VARIANTS}; do CTL_AMD_LIB=$L/libctl_$v.so PROBE_CLASS_ONLY=1 python tools/shade_class_probe.py ${SETS:-all} 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl; done
tools/r04_classes_prof.sh $1
