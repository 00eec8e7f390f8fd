#!/bin/bash
# round 6: the parity fuzz with the strict every-pixel check (oracle frame + zero-stop side image), seeds 100..400 in the three path modes + the sweeps of round 5's other modes
OUT=gpurun_out/${1:-r06fuzz}; mkdir -p $OUT; export TMPDIR=/tmp
for m in default wavefront plugin; do echo "== mode $m seeds 100 400" >> $OUT/fuzz.log; timeout 1500 python tools/fuzz_sweep.py 100 400 $m >> $OUT/fuzz.log 2>&1; done
for m in sensors alpha nodirect wild; do echo "== mode $m seeds 100 250" >> $OUT/fuzz.log; timeout 1200 python tools/fuzz_sweep.py 100 250 $m >> $OUT/fuzz.log 2>&1; done
grep -c seed $OUT/fuzz.log; grep "mode\|zero_stop_samples_total" $OUT/fuzz.log
