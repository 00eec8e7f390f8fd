#!/bin/bash
# model-class shade launches behind the partition pre-pass: render tests, then the probe with the default library and its variants
out=gpurun_out/${1:-r04cls2}; mkdir -p $out; L=$PWD/cudatracerlib_amd
python -m pytest tests/test_gpu_render.py -m gpu -x -q > $out/pytest_render.log 2>&1; grep -a "passed\|failed" $out/pytest_render.log
python tools/shade_class_probe.py 2>$out/err.log | grep "^{" | tee $out/probe.jsonl
for v in ${VARIANTS:-bout w3 a5 b512}; do CTL_AMD_LIB=$L/libctl_$v.so PROBE_CLASS_ONLY=1 python tools/shade_class_probe.py 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl; done
