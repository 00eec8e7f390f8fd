#!/bin/bash
# SQ counter passes for library variants on one box: tools/pmc_ab.sh <tag> <variant> [<variant> ...]  (cudatracerlib_amd/libctl_<variant>.so) -> gpurun_out/<tag>/<variant>_{sqA,sqB}/
# Each pass is its own rocprofv3 run with --kernel-trace only.
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-}"
for v in "$@"; do
  export CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_$v.so
  python bench.py $ARGS > "$OUT/$v.json" 2> "$OUT/$v.err"
  run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/${v}_$name" -o p -- python bench.py $ARGS > "$OUT/${v}_$name.log" 2>&1; }
  run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run sqB SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE
done
find "$OUT" -name '*_agent_info.csv' -delete
find "$OUT" -name '*.csv' -size +8M -delete
for v in "$@"; do python tools/pmc_table.py "$OUT/${v}_sqA" "$OUT/${v}_sqB" 2>/dev/null | grep -E "^==|intersect_pair"; done
