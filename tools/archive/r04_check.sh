#!/bin/bash
# GPU suite + the two headline bench lines (one gpurun call)
out=gpurun_out/${1:-r04chk}; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; grep -a "passed\|failed" $out/gpu_tests.log
python bench.py --workload synthetic-bathroom --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_bathroom.json 2> $out/bench_bathroom.err; python tools/bench_brief.py < $out/bench_bathroom.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_sm.json 2> $out/bench_sm.err; python tools/bench_brief.py < $out/bench_sm.json
