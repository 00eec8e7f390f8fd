#!/bin/bash
# round 6: after the zero-estimate poison fix — the fuzz tests, then seed ranges of the sweep that had findings, then the any-hit leaf batch sweep
OUT=gpurun_out/r06i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_render.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for m in default plugin; do echo "== mode $m seeds 380 400" >> $OUT/fuzz.log; timeout 600 python tools/fuzz_sweep.py 380 400 $m >> $OUT/fuzz.log 2>&1; done; cat $OUT/fuzz.log | cut -c1-200
bash tools/archive/r06_lanes_ab.sh r06i synthetic-sm < tools/archive/r06h_settings.txt
