#!/bin/bash
# class-probe over variant libraries (VARIANTS="a b", SETS=all,basic,single); BASIC=1 adds the synthetic-SM probe of k_shade_basic
out=gpurun_out/${1:-r04cls4}; mkdir -p $out; L=$PWD/cudatracerlib_amd
for v in amd ${VARIANTS}; do
  CTL_AMD_LIB=$L/libctl_$v.so PROBE_CLASS_ONLY=1 python tools/shade_class_probe.py ${SETS:-all} 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl
  [ "${BASIC:-0}" = 1 ] && CTL_AMD_LIB=$L/libctl_$v.so PROBE_SCENE_VARIANTS=0 python tools/shade_basic_probe.py 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl
done
