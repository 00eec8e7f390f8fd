// valu_probe.hip — issue rate of the VALU / SALU instructions the traversal kernel is made of, per SIMD, on gfx950.
// Each wave runs `iters` rounds of 16 independent instructions of one kind (inline asm, so the compiler cannot fold or pack them);
// 8 waves per SIMD (2048 blocks of 256 threads on 256 CUs).  Prints cycles per wave-instruction per SIMD at the 2.4 GHz nominal clock
// and from s_memtime (shader clock).   build: hipcc -O3 --offload-arch=gfx950 tools/archive/valu_probe.hip -o gpurun_out/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int KIND> __global__ __launch_bounds__(256) void k_rate(float* out, int iters, unsigned long long* clk) {
    float r[16]; float a = out[threadIdx.x & 7], b = out[8 + (threadIdx.x & 7)];
    unsigned u = __float_as_uint(a) | 0x01020304u;
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = a + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define CND(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(a));
#define CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(u));
#define AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define SHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
#define MINU(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define SAND(i) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21");
#define BFE(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(r[i]) : "v"(u));
#define PERM(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(u));
        if (KIND == 0) { REP16(FMA) } if (KIND == 1) { REP16(MUL) } if (KIND == 2) { REP16(MAX) } if (KIND == 3) { REP16(MAX3) }
        if (KIND == 4) { REP16(CND) } if (KIND == 5) { REP16(CVT) } if (KIND == 6) { REP16(AND) } if (KIND == 7) { REP16(SHL) }
        if (KIND == 8) { REP16(ADDU) } if (KIND == 9) { REP16(CMP) } if (KIND == 10) { REP16(MOV) } if (KIND == 11) { REP16(MINU) }
        if (KIND == 12) { REP16(RCP) } if (KIND == 13) { REP16(SAND) } if (KIND == 14) { REP16(BFE) } if (KIND == 15) { REP16(PERM) }
        if (KIND == 16) {   // packed fp32 fma on register pairs
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[0]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[2]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[4]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[6]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[8]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[10]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[0]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&r[2]) : "v"(*(double*)&r[14]), "v"(*(double*)&r[12]));
        }
        if (KIND == 17) {   // VALU and SALU interleaved: do they co-issue?
#define MIX(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(r[i]) : "v"(a), "v"(b) : "s20", "s21");
            REP16(MIX)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x + 16] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[KIND] = t1 - t0;
}

int main() {
    const int blocks = 256 * 8, iters = 4096;
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, (blocks * 256 + 16) * 4)); CHECK(hipMemset(out, 0, (blocks * 256 + 16) * 4)); CHECK(hipMalloc(&clk, 32 * 8)); CHECK(hipMemset(clk, 0, 32 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[18] = { "v_fma_f32", "v_mul_f32", "v_max_f32", "v_max3_f32", "v_cndmask_b32", "v_cvt_f32_ubyte1", "v_and_b32", "v_lshlrev_b32", "v_add_u32", "v_cmp_lt_f32", "v_mov_b32", "v_min_u32", "v_rcp_f32", "s_and_b64", "v_bfe_u32", "v_perm_b32", "v_pk_fma_f32 (8/round)", "v_fma + s_and pairs" };
#define LAUNCH(K) hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, out, iters, clk)
    for (int k = 0; k < 18; k++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            switch (k) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; case 4: LAUNCH(4); break; case 5: LAUNCH(5); break;
                case 6: LAUNCH(6); break; case 7: LAUNCH(7); break; case 8: LAUNCH(8); break; case 9: LAUNCH(9); break; case 10: LAUNCH(10); break; case 11: LAUNCH(11); break;
                case 12: LAUNCH(12); break; case 13: LAUNCH(13); break; case 14: LAUNCH(14); break; case 15: LAUNCH(15); break; case 16: LAUNCH(16); break; case 17: LAUNCH(17); break; }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned long long c; CHECK(hipMemcpy(&c, clk + k, 8, hipMemcpyDeviceToHost));
        const double per_round = k == 16 ? 8.0 : (k == 17 ? 32.0 : 16.0);
        const double wave_instr_per_simd = 8.0 * iters * per_round;   // 8 waves per SIMD
        std::printf("%-26s %8.3f ms   %6.2f cycles per wave-instruction per SIMD @2.4GHz   (one wave: %.2f clk per instr by the cycle counter)\n", names[k], best,
                    best * 1e-3 * 2.4e9 / wave_instr_per_simd, (double)c / (iters * per_round));
    }
    return 0;
}
