#!/bin/bash
# round 4: synthetic-sm-hard next to synthetic-SM (bench.py, no CPU baseline), then the full-size tests that changed
TAG=${1:-r04i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() { n=$1; shift; timeout 1500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_$n.json 2> $OUT/bench_$n.err; echo "$n $(python tools/bench_brief.py < $OUT/bench_$n.json | cut -c1-230)"; tail -2 $OUT/bench_$n.err | cut -c1-200; }
run hard --workload synthetic-sm-hard
run hard_q8 --workload synthetic-sm-hard --flat-format q8
run hard_alpha --workload synthetic-sm-hard --tracer-param AlphaTest=true
run sm
cat $OUT/bench_hard.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b[\"config\"])"
