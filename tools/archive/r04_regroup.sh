#!/bin/bash
# round 4: the shade kernels' workgroup-local regrouping keyed by the BSDF model the TRAVERSAL leaves per ray
TAG=${1:-r04m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/cudatracerlib_amd
timeout 1200 python -m pytest tests/test_gpu_render.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
CTL_AMD_LIB=$L/libctl_bs512.so timeout 1200 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "not knob" > $OUT/pytest_bs.log 2>&1; tail -3 $OUT/pytest_bs.log
run() { env CTL_AMD_LIB=$L/$1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 > $OUT/b.json 2> $OUT/b.err; echo "$1 $2 $(python tools/bench_brief.py < $OUT/b.json | cut -c1-110)"; tail -1 $OUT/b.err | cut -c1-200; }
run libctl_amd.so ""
run libctl_bs512.so ""
run libctl_bs128.so ""
run libctl_bs512.so "--tracer-param BlockSort=false"
run libctl_amd.so ""
run libctl_amd.so "--workload synthetic-bathroom"
run libctl_amd.so "--workload synthetic-bathroom --tracer-param BlockSort=false"
run libctl_amd.so "--workload synthetic-sm-hard"
run libctl_amd.so "--workload synthetic-sm-hard --tracer-param BlockSort=false"
