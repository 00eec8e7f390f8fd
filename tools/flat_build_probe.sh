#!/bin/bash
# Sweep of the world-space BVH builder settings (leaf size, SAH node cost) on the synthetic-SM workload: run on the GPU box,
#   gpurun -- 'tools/flat_build_probe.sh > gpurun_out/flat_build.log 2>&1'
cd "$(dirname "$0")/.." || exit 1
for cfg in ${*:-4:1 4:0.5 2:0.5 2:1 8:1 8:0.5 4:0.25}; do
  ml=${cfg%%:*}; nc=${cfg##*:}
  echo "max_leaf $ml node_cost $nc"
  CTL_FLAT_MAX_LEAF=$ml CTL_FLAT_NODE_COST=$nc python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null |
    python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("  %.0f Mrays/s  closest %.2f ms/launch  per ray: %s  B_ray %.0f  build %.1f s" % (d["value"], r["avg_launch_ms"], r["per_ray"], r["bytes_per_ray"], d["config"]["scene_build_s"]))'
done
