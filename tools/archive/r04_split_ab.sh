#!/bin/bash
# early split clipping A/B on one box: knobs library, SPLITS = "ratio:gain ..." ($CTL_FLAT_SPLIT: longest reference side in medians of the scene's triangle boxes, 0 = off; $CTL_FLAT_SPLIT_GAIN)
out=gpurun_out/${1:-r04splitab}; mkdir -p $out; export CTL_AMD_LIB=$PWD/cudatracerlib_amd/libctl_knobs.so
for w in ${WORKLOADS:-synthetic-sm synthetic-bathroom}; do for s in ${SPLITS:-0:1 8:1}; do
  CTL_FLAT_SPLIT=${s%%:*} CTL_FLAT_SPLIT_GAIN=${s##*:} timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $out/b_${w}_$s.json 2> $out/b_${w}_$s.err; echo "$w split $s $(python tools/bench_brief.py < $out/b_${w}_$s.json | cut -c1-190)"
done; done
