#!/bin/bash
# round 6, last check of the committed tree on a fresh box: what the driver runs at round end (GPU suite, smoke, the default bench line), then one more wide fuzz sweep
OUT=gpurun_out/${1:-r06z}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python tools/bench_brief.py < $OUT/bench_driver.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python tools/bench_brief.py < $OUT/bench_default.json
echo "== mode default seeds 1400 3400" >> $OUT/fuzz.log; timeout 3000 python tools/fuzz_sweep.py 1400 3400 default >> $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log | cut -c1-300
