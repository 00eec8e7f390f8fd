#!/bin/bash
# round 4: the 8-wide node format against the 4-wide one on one box: GPU parity tests of both, then bench.py alternating
TAG=${1:-r04a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_intersect.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for i in 1 2; do for f in q4 q8; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flat-format $f > $OUT/bench_$f$i.json 2> $OUT/bench_$f$i.err
  echo "$f $i $(python tools/bench_brief.py < $OUT/bench_$f$i.json | cut -c1-200)"; tail -2 $OUT/bench_$f$i.err
done; done
