#!/bin/bash
# round 5: GPU test suite + the bench lines of the three big workloads (driver style: --steps 20 --warmup 5)
TAG=${1:-r05chk}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED|Error" $OUT/pytest.log | head -5; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sm.json 2>$OUT/e0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom.json 2>$OUT/e2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/bench_sm_hard.json 2>$OUT/e6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload cornell-glass --width 1024 --height 1024 > $OUT/bench_cornell.json 2>$OUT/e3
for f in sm bathroom sm_hard cornell; do echo "$f $(python tools/bench_brief.py < $OUT/bench_$f.json | cut -c1-220)"; done
