#!/bin/bash
# PC sampling of the bench (rocprofv3 beta feature; never combined with --pmc).  Each attempt under its own timeout; only the aggregated summaries are kept.
# usage: [PCS_BENCH_ARGS="--workload ..."] tools/archive/r05_pcsample.sh TAG
tag=$1
export TMPDIR=/tmp ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-cpu-baseline $PCS_BENCH_ARGS > gpurun_out/${tag}_warm.json 2> gpurun_out/${tag}_warm.err     # fills the geometry cache
while read m unit iv; do
  d=/tmp/pcs_${m}
  rm -rf $d
  timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $m --pc-sampling-interval $iv --kernel-trace --output-format csv -d $d -- \
      python bench.py --steps 6 --warmup 1 --no-cpu-baseline $PCS_BENCH_ARGS > gpurun_out/${tag}_${m}.log 2>&1 < /dev/null
  echo "rc=$?" >> gpurun_out/${tag}_${m}.log
  du -sh $d >> gpurun_out/${tag}_${m}.log 2>&1
  python tools/archive/pcsample_aggregate.py $d gpurun_out/${tag}_${m}_summary.txt >> gpurun_out/${tag}_${m}.log 2>&1
done <<SPECS
stochastic cycles 1048576
host_trap time 100
SPECS
