#!/bin/bash
# round 5: the framebuffer gather on the GPU — the comm tests, the 8-rank dress rehearsal on one device (frame == the 1-rank frame), one rank of 1/2/4/8 at 20 / 64 / 256 passes
TAG=${1:-r05a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -m gpu -x -q -k "framebuffer or gather or self_launch or batching_and_sharding" > $OUT/pytest_comm.log 2>&1; tail -3 $OUT/pytest_comm.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-frame $OUT/frame1.npy > $OUT/bench_1rank.json 2>$OUT/e1
CTL_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --dump-frame $OUT/frame8.npy > $OUT/bench_8ranks_shared_gpu.json 2>$OUT/e8
python - <<PY
import numpy as np, json
a, b = np.load("$OUT/frame1.npy"), np.load("$OUT/frame8.npy")
print("8-rank frame vs 1-rank frame: weights equal", np.array_equal(a[..., 6], b[..., 6]), " rgb bit-equal fraction", float((a[..., :3] == b[..., :3]).all(-1).mean()), " max abs diff", float(np.abs(a - b).max()), " max rel diff", float((np.abs(a - b) / (1 + np.abs(a))).max()))
j = json.loads(open("$OUT/bench_8ranks_shared_gpu.json").read().strip().splitlines()[-1])
print(j["n_gpus"], j["value"], j["config"].get("framebuffer_reduce"), j.get("rank_ms"), j.get("reduce_ms"))
PY
rm -f $OUT/frame1.npy $OUT/frame8.npy
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; fi
if [ "${SKIP_PROBE:-0}" != "1" ]; then for P in 20 64 256; do PASSES=$P FUSE=1 python tools/shard_time_probe.py 1 2 4 8 >> $OUT/shard_time_probe.txt 2>&1; done; cat $OUT/shard_time_probe.txt; fi
