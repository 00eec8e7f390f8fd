#!/bin/bash
# counters of the shade kernels under bench.py (default: synthetic-SM, the driver's 20 passes); tools/shade_pmc.sh <tag> [bench args]
set -u
TAG=${1:-r03s}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${*:---steps 20 --warmup 5} --no-cpu-baseline"
python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python bench.py $ARGS > "$OUT/$name.log" 2>&1; }
run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU
run sqC SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_GDS SQ_INSTS_BRANCH
run tccE FETCH_SIZE WRITE_SIZE
run tccF TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*_agent_info.csv' -delete
for p in sqA sqB sqC tccE tccF; do python tools/pmc_table.py $OUT/$p | grep -i "==\|shade\|finalize" ; done > $OUT/shade_pmc.txt
cat $OUT/shade_pmc.txt
