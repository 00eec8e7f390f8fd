#!/bin/bash
# SQ / TCC counter passes over a short bench run + FETCH_SIZE calibration on the gather probe (known byte counts).
#   tools/pmc_probe.sh <tag> [bench args]      -> gpurun_out/<tag>/{sqA,sqB,tccC,tccD,cal_*}/
# Every pass is its own rocprofv3 run with --kernel-trace only (gpurun refuses --pmc together with sys/hip traces).
set -u
TAG=${1:-pmc}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 2 --no-cpu-baseline $*"
python bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"     # also warms the geometry cache
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python bench.py $ARGS > "$OUT/$name.log" 2>&1; }
run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE
run tccC TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run tccD TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tccE FETCH_SIZE
run tccF TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS
# calibration: dependent random gathers over a 2 GiB table (beyond L2 and the Infinity Cache): bytes = records x record size
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o "$OUT/gather_probe" 2> "$OUT/gather_build.log"
"$OUT/gather_probe" 2048 > "$OUT/gather_plain.log" 2>&1
cal() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- "$OUT/gather_probe" 2048 > "$OUT/$name.log" 2>&1; }
cal cal_fetch FETCH_SIZE
cal cal_req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
cal cal_hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
rm -f "$OUT/gather_probe"
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*_agent_info.csv' -delete
cat "$OUT/bench.json"; cat "$OUT/gather_plain.log"
