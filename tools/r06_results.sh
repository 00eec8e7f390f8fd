#!/bin/bash
# round 6: the bench lines of the shipped build WITH its committed counter profile (profiles/roofline_traffic.json carries this tree's source hash), and RESULTS.md's records
TAG=${1:-r06n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench_64spp.json 2>$OUT/e1
python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2>$OUT/e0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom.json 2>$OUT/e2
python bench.py --steps 128 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom_128spp.json 2>$OUT/e2b
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom --reduced-rough-transmittance > $OUT/bench_bathroom_reduced.json 2>$OUT/e2c
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/bench_sm_hard.json 2>$OUT/e6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --via-loader > $OUT/bench_loader.json 2>$OUT/e4
for f in final 64spp bathroom bathroom_128spp bathroom_reduced sm_hard loader; do echo "$f $(python tools/bench_brief.py < $OUT/bench_$f.json | cut -c1-260)"; done
timeout 2400 python tools/results_round.py --tag r06res > $OUT/results.log 2>&1; tail -c 600 $OUT/results.log
