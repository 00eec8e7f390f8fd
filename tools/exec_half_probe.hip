// exec_half_probe.hip — does gfx950 skip the half of a wave64 VALU instruction whose 32 lanes are all inactive?  (A wave64 instruction issues over two passes of 32 lanes on the
// SIMD-32 units.)  The same loop of independent v_fma_f32 / v_cvt_f32_ubyte / v_max3_f32 with (a) all 64 lanes active, (b) lanes 0..31, (c) lanes 32..63, (d) the even lanes,
// (e) lanes 0..15, at 8 waves per SIMD.  build: hipcc -O3 --offload-arch=gfx950 tools/exec_half_probe.hip -o gpurun_out/exec_half_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (MODE == 1) on = lane < 32; if (MODE == 2) on = lane >= 32; if (MODE == 3) on = (lane & 1) == 0; if (MODE == 4) on = lane < 16; if (MODE == 5) on = (lane & 3) == 0;
    float r[16]; const float a = out[threadIdx.x & 7], b = out[8 + (threadIdx.x & 7)]; const unsigned u = __float_as_uint(a) | 0x01020304u;
    for (int i = 0; i < 16; i++) r[i] = a + i;
    if (on) {
        for (int it = 0; it < iters; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(u));
#define MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(FMA) REP16(CVT) REP16(MAX3) REP16(FMA)
        }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += r[i];
    out[1024 + blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; CHECK(hipMalloc(&d, (1024 + 2048 * 256) * 4)); CHECK(hipMemset(d, 0, (1024 + 2048 * 256) * 4));
    const int iters = 20000; const char* names[6] = { "all 64 lanes", "lanes 0..31", "lanes 32..63", "even lanes", "lanes 0..15", "every 4th lane" };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) for (int m = 0; m < 6; m++) {
        CHECK(hipEventRecord(e0));
        switch (m) { case 0: k<0><<<2048, 256>>>(d, iters); break; case 1: k<1><<<2048, 256>>>(d, iters); break; case 2: k<2><<<2048, 256>>>(d, iters); break;
                     case 3: k<3><<<2048, 256>>>(d, iters); break; case 4: k<4><<<2048, 256>>>(d, iters); break; default: k<5><<<2048, 256>>>(d, iters); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // 8 waves per SIMD x 64 instructions per iteration
        if (rep) std::printf("%-16s %8.3f ms   %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms, ms * 1e-3 * 2.4e9 / (8.0 * 64 * iters));
    }
    return 0;
}
