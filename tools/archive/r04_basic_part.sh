#!/bin/bash
# k_shade_basic over the model-ordered list of k_class_partition (variant libraries) against the shipped in-kernel regrouping; synthetic-SM 1080p depth 8
out=gpurun_out/${1:-r04bp}; mkdir -p $out; L=$PWD/cudatracerlib_amd; export PROBE_SCENE_VARIANTS=0
python tools/shade_basic_probe.py 2>$out/err.log | grep "^{" | tee $out/probe.jsonl
for v in ${VARIANTS:-bp bp512 bp5}; do CTL_AMD_LIB=$L/libctl_$v.so python tools/shade_basic_probe.py 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl; done
PROBE_CLASS_ONLY=1 python tools/shade_class_probe.py all 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl
python -m pytest tests/test_gpu_render.py -m gpu -x -q > $out/pytest_render.log 2>&1; grep -a "passed\|failed" $out/pytest_render.log
