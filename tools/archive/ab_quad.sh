#!/bin/bash
# A/B of the quad-cooperative node fetch against the own-node fetch (libctl_ownfetch.so = -DCTL_NODE_FETCH_QUAD=0) on one GPU box
mkdir -p gpurun_out/r03c
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for i in 1 2; do
for v in amd ownfetch prev; do
CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03c/$v$i.json 2>gpurun_out/r03c/$v.err
echo $v $i; python tools/bench_brief.py < gpurun_out/r03c/$v$i.json
done
done
CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_ownfetch.so timeout 900 python -m pytest tests/test_gpu_intersect.py -x -q -m gpu 2>&1 | tail -2
