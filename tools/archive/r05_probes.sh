#!/bin/bash
# round 5: the product with BVH reinsertion (tests + the four bench lines), the any-hit-unordered variant, the corrected VALU probe
TAG=${1:-r05e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SKIP_TESTS=${SKIP_TESTS:-0} bash tools/archive/r05_check.sh $TAG
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-sm-hard > $OUT/bench_sm_hard_again.json 2>$OUT/e6b; echo "sm_hard again $(python tools/bench_brief.py < $OUT/bench_sm_hard_again.json | cut -c1-120)"
L=$PWD/cudatracerlib_amd
for v in amd anyu amd anyu; do CTL_AMD_LIB=$L/libctl_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$v.json 2>$OUT/e_$v; echo "$v $(python tools/bench_brief.py < $OUT/bench_$v.json | cut -c1-220)"; python - "$OUT/bench_$v.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("   visits per shadow ray", r.get("per_shadow_ray"), " per path ray", r.get("per_ray"), " ms_intersect", r.get("ms_intersect"))
PY
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_probe3.hip -o $OUT/valu_probe3 2>/dev/null && timeout 300 $OUT/valu_probe3 > $OUT/valu_probe3.log 2>&1; rm -f $OUT/valu_probe3; cat $OUT/valu_probe3.log
