#!/bin/bash
# early split clipping (flatten.cpp): GPU suite + the bench lines it moves
out=gpurun_out/${1:-r04split}; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; grep -a "passed\|failed" $out/gpu_tests.log | tail -2
for w in synthetic-sm synthetic-sm-hard synthetic-bathroom; do
  timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; echo "$w $(python tools/bench_brief.py < $out/bench_$w.json | cut -c1-200) build $(python -c "import json,sys; print(json.loads(open('$out/bench_$w.json').read().strip().splitlines()[-1])['config'].get('scene_build_s'))")"
done
