#!/bin/bash
# k_shade_basic build variants (libctl_<v>.so) on synthetic-SM: shade ms per pass
out=gpurun_out/${1:-r04bv}; mkdir -p $out; L=$PWD/cudatracerlib_amd; export PROBE_SCENE_VARIANTS=0
for v in amd ${VARIANTS}; do CTL_AMD_LIB=$L/libctl_$v.so timeout 300 python tools/shade_basic_probe.py 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl; done
