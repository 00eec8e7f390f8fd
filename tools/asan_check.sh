#!/bin/bash
# Run the CPU test suite against an AddressSanitizer build of libctl_amd.so (host code instrumented, device code left alone).
# The tests load it through $CTL_AMD_LIB; the regular library is not touched.  Usage: tools/asan_check.sh [pytest args]
set -u
cd "$(dirname "$0")/.." || exit 1
OUT=/tmp/ctl_asan; mkdir -p $OUT
FL="-O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -munsafe-fp-atomics -pthread -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
SRCS=$(python3 -c "import runpy; print(' '.join(runpy.run_path('cudatracerlib_amd/build.py')['SRCS']))")
for f in $SRCS; do
  x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc $FL $x -c cudatracerlib_amd/csrc/$f -o $OUT/$f.o || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -shared-libasan -o $OUT/libctl_amd.so $OUT/*.o -pthread -lz -ldl || exit 1
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
rm -f $OUT/log.*
CTL_AMD_LIB=$OUT/libctl_amd.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$OUT/log python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider \
  --deselect tests/test_distributed_cpu.py::test_two_rank_gloo_reduce_reproduces_single_rank_frame "$@" 2>&1 | grep -v "SplitBVHBuilder: progress" | tail -5
ls $OUT/log.* 2>/dev/null && head -12 $OUT/log.* | cut -c1-200
