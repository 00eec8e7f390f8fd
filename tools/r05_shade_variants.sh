#!/bin/bash
# round 5, VERDICT r4 item 5: the shade kernels' dependent chain.  k_shade_basic build variants (cudatracerlib_amd/libctl_<v>.so) on synthetic-SM: shade ms per pass
# (tools/shade_basic_probe.py, same box), then SQ_WAIT_ANY / SQ_WAVE_CYCLES of k_shade_basic for each (one PMC pass per library).
out=gpurun_out/${1:-r05s}; mkdir -p $out; L=$PWD/cudatracerlib_amd; export PROBE_SCENE_VARIANTS=0 TMPDIR=/tmp
for rep in 1 2; do for v in amd ${VARIANTS}; do CTL_AMD_LIB=$L/libctl_$v.so timeout 300 python tools/shade_basic_probe.py 2>>$out/err.log | grep "^{" | tee -a $out/probe.jsonl; done; done
if [ "${SKIP_PMC:-0}" != "1" ]; then
for v in amd ${VARIANTS}; do
  CTL_AMD_LIB=$L/libctl_$v.so timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_$v -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/pmc_$v.log 2>&1
  python - "$out/pmc_$v" "$v" <<'PY' | tee -a $out/pmc_sums.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("ctl::", "").replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen and "Start_Timestamp" in r: seen.add(r["Dispatch_Id"]); dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
for k, v in acc.items():
    if k.startswith("k_shade"):
        print("%-8s %-16s wait_any %.3f  wait_inst %.3f  VALU %.4g  VMEM_RD %.4g  LDS %.4g  ms %.2f" % (sys.argv[2], k, v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"], v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_LDS"], dur[k]))
PY
  rm -rf $out/pmc_$v
done
fi
