#!/bin/bash
# round 4, the build the round ends with: GPU test suite, profile (tools/profile_round.sh), the other workloads' bench lines, plugin comparison, one-rank-of-N timing
TAG=${1:-r04z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; fi
if [ "${SKIP_PROFILE:-0}" != "1" ]; then bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; fi
python bench.py > $OUT/bench_64spp.json 2>$OUT/e1
python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2>$OUT/e0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom.json 2>$OUT/e2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload cornell-glass --width 1024 --height 1024 > $OUT/bench_cornell.json 2>$OUT/e3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --via-loader > $OUT/bench_loader.json 2>$OUT/e4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tracer-param PathSemantics=1 > $OUT/bench_wavefront_rules.json 2>$OUT/e5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/bench_sm_hard.json 2>$OUT/e6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flat-format q8 > $OUT/bench_q8.json 2>$OUT/e7
python tools/plugin_compare.py > $OUT/plugin_compare.txt 2>&1
FUSE=1 python tools/shard_time_probe.py 1 2 4 8 > $OUT/shard_time_probe.txt 2>&1
CTL_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2ranks_shared_gpu.json 2>$OUT/e8
for f in final 64spp bathroom cornell loader wavefront_rules sm_hard q8 2ranks_shared_gpu; do echo "$f $(python tools/bench_brief.py < $OUT/bench_$f.json | cut -c1-150)"; done
tail -3 $OUT/plugin_compare.txt; cat $OUT/shard_time_probe.txt | tail -4
