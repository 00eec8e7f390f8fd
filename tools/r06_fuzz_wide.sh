#!/bin/bash
# round 6: the strict parity fuzz over seeds no sweep has seen before
OUT=gpurun_out/${1:-r06fuzz3}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== mode default seeds 400 1400" >> $OUT/fuzz.log; timeout 2400 python tools/fuzz_sweep.py 400 1400 default >> $OUT/fuzz.log 2>&1
for m in wavefront plugin; do echo "== mode $m seeds 400 900" >> $OUT/fuzz.log; timeout 1800 python tools/fuzz_sweep.py 400 900 $m >> $OUT/fuzz.log 2>&1; done
for m in sensors alpha nodirect wild; do echo "== mode $m seeds 250 500" >> $OUT/fuzz.log; timeout 1500 python tools/fuzz_sweep.py 250 500 $m >> $OUT/fuzz.log 2>&1; done
grep "mode\|zero_stop_samples_total" $OUT/fuzz.log | cut -c1-150
