#!/bin/bash
# Collect the evidence behind bench.py's numbers on the MI355X box (run through gpurun from the repo root):
#   tools/profile_round.sh r02a [bench args, default: the driver's --steps 20 --warmup 5]
# writes gpurun_out/<tag>/{bench.json, stats/, pmc_*/}; tools/summarize_profile.py then condenses them into profiles/<tag>_*.
# PMC passes are separate runs with --kernel-trace only (never with sys/hip traces).
set -u
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
RUN=${*:---steps 20 --warmup 5}
timeout 900 python bench.py $RUN > "$OUT/bench.json" 2> "$OUT/bench.err"
ARGS="$RUN --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python bench.py $ARGS > "$OUT/stats.log" 2>&1
pmc() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc TCC TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
pmc SQ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pmc SQ2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pmc TCP TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*_agent_info.csv' -delete
cat "$OUT/bench.json"
