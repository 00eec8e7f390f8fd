#!/bin/bash
# rough-plastic transmittance: the per-material 1-D reduction against the 3-D lookup on every call (CTL_RT_REDUCTION=0): bit-equal pixels of the bathroom miniature
# and the fuzz seeds, and what the bathroom workload pays
out=gpurun_out/${1:-r05rt}; mkdir -p $out; export TMPDIR=/tmp
for r in 1 0; do
  echo "== CTL_RT_REDUCTION=$r" >> $out/log.txt
  CTL_RT_REDUCTION=$r timeout 600 python tools/rt_exact_frames.py >> $out/log.txt 2>&1
  for k in 1 2; do CTL_RT_REDUCTION=$r timeout 900 python bench.py --workload synthetic-bathroom --steps 20 --warmup 5 --no-cpu-baseline 2>>$out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bathroom', d['value'], d['ms_per_step'], d.get('ms_shade'), d.get('ms_intersect'))" >> $out/log.txt; done
done
cat $out/log.txt
