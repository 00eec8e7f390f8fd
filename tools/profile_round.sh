#!/bin/bash
# Collect the evidence behind bench.py's numbers on the MI355X box (run through gpurun from the repo root):
#   tools/profile_round.sh r01b
# writes gpurun_out/<tag>/{bench.json, stats/, pmc_FETCH_SIZE/, pmc_WRITE_SIZE/, pmc_TCC/}; tools/summarize_profile.py then
# condenses them into profiles/<tag>_*.  PMC passes are separate runs with --kernel-trace only (never with sys/hip traces).
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--no-cpu-baseline"   # bench.py defaults (64 steps, 2 warm-up): the profiled launches are the ones the JSON line is about
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python bench.py $ARGS > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o p -- python bench.py $ARGS > "$OUT/pmc_$C.log" 2>&1
done
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_TCC" -o p -- python bench.py $ARGS > "$OUT/pmc_TCC.log" 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
ls -R "$OUT" | head -40
cat "$OUT/bench.json"
