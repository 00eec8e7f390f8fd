#!/bin/bash
# Round 6 feasibility probe for co-resident shading: (a) what the traversal kernels lose when LDS padding holds them to 6 / 5 / 4 workgroups per CU (room for a shade wave per SIMD next to
# them), (b) what the basic shade kernel takes with ONE workgroup per CU (libctl_sg1.so: -DCTL_SHADE_GRID_WAVES=1), the residency it would have beside five traversal waves.
TAG=${1:-r06a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; L=$PWD/cudatracerlib_amd
run() { name=$1; shift; echo "== $name $*" | tee -a $OUT/summary.txt; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; python tools/bench_brief.py < $OUT/$name.json | tee -a $OUT/summary.txt; tail -2 $OUT/$name.err; }
run base CTL_AMD_LIB=$L/libctl_knobs.so
run pad6 CTL_AMD_LIB=$L/libctl_knobs.so CTL_LDS_PAD=4096
run pad5 CTL_AMD_LIB=$L/libctl_knobs.so CTL_LDS_PAD=7680
run pad4 CTL_AMD_LIB=$L/libctl_knobs.so CTL_LDS_PAD=16384
run sg1 CTL_AMD_LIB=$L/libctl_sg1.so
run sg2 CTL_AMD_LIB=$L/libctl_sg2.so
run base2 CTL_AMD_LIB=$L/libctl_knobs.so
