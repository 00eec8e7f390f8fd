#!/bin/bash
# Per-kernel register / spill / occupancy report of the HIP translation units (compiler remarks, no GPU needed).
# Usage: tools/kernel_resources.sh [kernels shade_full shade_basic megakernel]
cd "$(dirname "$0")/.." || exit 1
units=${*:-kernels shade_full shade_basic megakernel}
for f in $units; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -munsafe-fp-atomics $KRES_FLAGS -c cudatracerlib_amd/csrc/$f.hip -o /tmp/kres_$$.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v unit="$f" '
      /Function Name:/ { name=$0; sub(/.*Function Name: /, "", name); sub(/ \[-Rpass.*/, "", name) }
      /VGPRs:/ && !/Agpr|AGPRs/ { v=$0; sub(/.*VGPRs: /, "", v); sub(/ \[.*/, "", v) }
      /ScratchSize/ { s=$0; sub(/.*: /, "", s); sub(/ \[.*/, "", s) }
      /VGPRs Spill/ { sp=$0; sub(/.*: /, "", sp); sub(/ \[.*/, "", sp) }
      /Occupancy/ { o=$0; sub(/.*: /, "", o); sub(/ \[.*/, "", o) }
      /LDS Size/ { l=$0; sub(/.*: /, "", l); sub(/ \[.*/, "", l); printf "%-12s vgpr %-4s spill %-4s scratch %-6s occ %-3s lds %-6s %s\n", unit, v, sp, s, o, l, name }'
done
rm -f /tmp/kres_$$.o
