#!/bin/bash
# rough-plastic transmittance, three ways on one box: the default (the reference's sum over the material's sixteen rows), the generic 3-D lookup on every call (CTL_RT_ROWS=0) and
# the opt-in 1-D reduction (--reduced-rough-transmittance): bit-equal pixels of the bathroom miniature and of fuzz seeds, and what the bathroom workload pays
out=gpurun_out/${1:-r05rt}; mkdir -p $out; export TMPDIR=/tmp
for mode in rows generic reduced; do
  echo "== $mode" >> $out/log.txt
  case $mode in rows) env=""; arg="";; generic) env="CTL_RT_ROWS=0"; arg="";; reduced) env=""; arg="--reduced-rough-transmittance";; esac
  env $env RT_MODE=$mode timeout 600 python tools/archive/rt_exact_frames.py >> $out/log.txt 2>&1
  for k in 1 2; do env $env timeout 900 python bench.py --workload synthetic-bathroom --steps 20 --warmup 5 --no-cpu-baseline $arg 2>>$out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bathroom', d['value'], d['ms_per_step'])" >> $out/log.txt; done
done
cat $out/log.txt
