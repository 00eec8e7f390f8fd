set -u
OUT=gpurun_out/r02n; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--workload synthetic-bathroom --steps 8 --warmup 2 --no-cpu-baseline --tracer-param BlockSort=true"
python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
run() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python bench.py $ARGS > "$OUT/$name.log" 2>&1; }
run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE
run tccE FETCH_SIZE WRITE_SIZE
run tccF TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*_agent_info.csv' -delete
for p in sqA sqB tccE tccF; do python tools/pmc_table.py $OUT/$p | grep -i "==\|shade" ; done > $OUT/shade_pmc.txt
