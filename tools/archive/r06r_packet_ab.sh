#!/bin/bash
# round 6: packet traversal of the first bounce — GPU parity tests, then bench.py A/B (PacketFirstBounce false / true), alternating, three workloads
OUT=gpurun_out/${1:-r06r}; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -B2 -A12 "^E  " $OUT/pytest.log | head -40; fi
for w in ${WORKLOADS:-synthetic-sm synthetic-bathroom synthetic-sm-hard cornell-glass}; do for rep in 1 2; do for v in false true; do
  echo "== $w PacketFirstBounce=$v" | tee -a $OUT/summary.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w --tracer-param PacketFirstBounce=$v > $OUT/last.json 2> $OUT/last.err; python tools/bench_brief.py < $OUT/last.json | cut -c1-120 | tee -a $OUT/summary.txt; python -c "
import json; j=json.loads(open('$OUT/last.json').read().strip().splitlines()[-1]); s=j['roofline'].get('separate_launches',{}).get('closest_hit',{}); print('   first-bounce closest-hit launch:', s)" | tee -a $OUT/summary.txt; tail -1 $OUT/last.err
done; done; done
