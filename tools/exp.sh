#!/bin/bash
# GPU-box experiment runner: tools/exp.sh <tag> — runs the GPU tests, then bench.py under the env-variable settings listed in the
# heredoc below (one JSON line each into gpurun_out/<tag>/runs.jsonl with the setting prepended).
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"; fi
ARGS=${BENCH_ARGS:---steps 32 --warmup 2 --no-cpu-baseline}
while read -r line; do
  [ -z "$line" ] && continue
  echo "== $line" | tee -a "$OUT/runs.jsonl"
  env $line timeout 600 python bench.py $ARGS $(echo "$line" | tr " " "\n" | sed -n "s/^BENCH_EXTRA=//p" | tr "," " ") 2> "$OUT/last.err" | tee -a "$OUT/runs.jsonl" | python tools/bench_brief.py
  tail -2 "$OUT/last.err"
done < "${2:-/dev/stdin}"
