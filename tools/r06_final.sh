#!/bin/bash
# round 6, the build the round ends with: GPU test suite, counter profiles of the three big workloads (tools/profile_round.sh), the other bench lines, the 8-rank rehearsal on one
# device, one rank of 1 / 2 / 4 / 8 at 20 / 64 / 256 passes, plugin comparison
TAG=${1:-r06x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; fi
if [ "${SKIP_PROFILE:-0}" != "1" ]; then
  bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
  bash tools/profile_round.sh ${TAG}bath --steps 20 --warmup 5 --workload synthetic-bathroom > $OUT/profile_bath.log 2>&1
  bash tools/profile_round.sh ${TAG}hard --steps 20 --warmup 5 --workload synthetic-sm-hard > $OUT/profile_hard.log 2>&1
fi
python bench.py > $OUT/bench_64spp.json 2>$OUT/e1
python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2>$OUT/e0
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom.json 2>$OUT/e2
python bench.py --steps 128 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > $OUT/bench_bathroom_128spp.json 2>$OUT/e2b
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom --reduced-rough-transmittance > $OUT/bench_bathroom_reduced.json 2>$OUT/e2c
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload cornell-glass --width 1024 --height 1024 > $OUT/bench_cornell.json 2>$OUT/e3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --via-loader > $OUT/bench_loader.json 2>$OUT/e4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tracer-param PathSemantics=1 > $OUT/bench_wavefront_rules.json 2>$OUT/e5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/bench_sm_hard.json 2>$OUT/e6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --flat-format q8 > $OUT/bench_q8.json 2>$OUT/e7
python tools/plugin_compare.py > $OUT/plugin_compare.txt 2>&1
for P in 20 64 256; do PASSES=$P FUSE=1 python tools/shard_time_probe.py 1 2 4 8 >> $OUT/shard_time_probe.txt 2>&1; done
CTL_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2ranks_shared_gpu.json 2>$OUT/e8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-frame $OUT/frame1.npy > /dev/null 2>&1
CTL_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --dump-frame $OUT/frame8.npy > $OUT/bench_8ranks_shared_gpu.json 2>$OUT/e9
python - <<PY > $OUT/frames_8_vs_1.txt 2>&1
import numpy as np
a, b = np.load("$OUT/frame1.npy"), np.load("$OUT/frame8.npy")
print("8-rank frame (gather of packed tiles, 8 processes on one device) vs 1-rank frame, 1920x1080, 25 passes: weights equal", np.array_equal(a[..., 6], b[..., 6]), "| rgb bit-equal fraction of pixels", float((a[..., :3] == b[..., :3]).all(-1).mean()), "| max abs diff", float(np.abs(a - b).max()), "| max rel diff", float((np.abs(a - b) / (1 + np.abs(a))).max()))
PY
rm -f $OUT/frame1.npy $OUT/frame8.npy; cat $OUT/frames_8_vs_1.txt
for f in final 64spp bathroom bathroom_128spp bathroom_reduced cornell loader wavefront_rules sm_hard q8 2ranks_shared_gpu 8ranks_shared_gpu; do echo "$f $(python tools/bench_brief.py < $OUT/bench_$f.json | cut -c1-150)"; done
tail -3 $OUT/plugin_compare.txt; cat $OUT/shard_time_probe.txt
