// valu_probe2.hip — follow-up to valu_probe.hip: why does v_cndmask_b32 look slow, and what do SALU / exec-mask instructions cost?
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
    unsigned long long m = __ballot(threadIdx.x & 1), m2 = __ballot(threadIdx.x & 2);
    asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define CNDVCC(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(a));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "s"(m2));
#define CNDNEW(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r[i]) : "v"(a), "v"(b), "s"(m2));
#define CMPS(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(r[i]), "v"(a));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define SAND(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m) : "s"(m2));
#define SAVEEXEC(i) asm volatile("s_and_saveexec_b64 %0, %1\n s_mov_b64 exec, %0" : "=&s"(m) : "s"(m2));
#define FMAS(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n s_and_b64 %1, %1, %4" : "+v"(r[i]), "+s"(m) : "v"(a), "v"(b), "s"(m2));
#define MAXMIN(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define FMAMAX(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n v_max_f32 %1, %2, %1" : "+v"(r[i]), "+v"(r[(i + 8) & 15]) : "v"(a), "v"(b));
#define MINI(i) asm volatile("v_min_i32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define MED3(i) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(u), "v"(a));
#define XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(u));
#define ADD3(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(u), "v"(a));
#define SUBF(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define CVTU(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r[i]) : "v"(u));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r[i]) : "v"(a));
#define RDL(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(u) : "v"(r[i]));
        if (KIND == 0) { REP16(CNDVCC) } if (KIND == 1) { REP16(CNDS) } if (KIND == 2) { REP16(CNDNEW) } if (KIND == 3) { REP16(CMPS) }
        if (KIND == 4) { REP16(CMPCND) } if (KIND == 5) { REP16(SAND) } if (KIND == 6) { REP16(SAVEEXEC) } if (KIND == 7) { REP16(FMAS) }
        if (KIND == 8) { REP16(MAXMIN) } if (KIND == 9) { REP16(FMAMAX) } if (KIND == 10) { REP16(MINI) } if (KIND == 11) { REP16(MED3) }
        if (KIND == 12) { REP16(BFI) } if (KIND == 13) { REP16(XOR) } if (KIND == 14) { REP16(LSHLOR) } if (KIND == 15) { REP16(ADD3) }
        if (KIND == 16) { REP16(SUBF) } if (KIND == 17) { REP16(CVTU) } if (KIND == 18) { REP16(DPP) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x + 16] = s + (float)(m & 1) + (float)(m2 & 1);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[KIND] = t1 - t0;
}

template <int K> void run(const char* name, double per_round, float* out, unsigned long long* clk) {
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long c; CHECK(hipMemcpy(&c, clk + K, 8, hipMemcpyDeviceToHost));
    std::printf("%-44s %8.3f ms   %6.2f cycles per wave-instr per SIMD @2.4GHz   (wave 0: %.2f clk/instr)\n", name, best, best * 1e-3 * 2.4e9 / (8.0 * iters * per_round), (double)c / (iters * per_round));
}

int main() {
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, (256 * 8 * 256 + 16) * 4)); CHECK(hipMemset(out, 0, (256 * 8 * 256 + 16) * 4)); CHECK(hipMalloc(&clk, 32 * 8)); CHECK(hipMemset(clk, 0, 32 * 8));
    run<0>("v_cndmask_b32 vcc (vcc set once)", 16, out, clk);
    run<1>("v_cndmask_b32_e64 sgpr pair, dst=src1", 16, out, clk);
    run<2>("v_cndmask_b32_e64 sgpr pair, fresh dst", 16, out, clk);
    run<3>("v_cmp_lt_f32_e64 -> sgpr pair", 16, out, clk);
    run<4>("v_cmp vcc + v_cndmask vcc pairs", 32, out, clk);
    run<5>("s_and_b64", 16, out, clk);
    run<6>("s_and_saveexec_b64 + s_mov exec pairs", 32, out, clk);
    run<7>("v_fma + s_and_b64 pairs", 32, out, clk);
    run<8>("v_max_f32", 16, out, clk);
    run<9>("v_fma + v_max pairs", 32, out, clk);
    run<10>("v_min_i32", 16, out, clk);
    run<11>("v_med3_f32", 16, out, clk);
    run<12>("v_bfi_b32", 16, out, clk);
    run<13>("v_xor_b32", 16, out, clk);
    run<14>("v_lshl_or_b32", 16, out, clk);
    run<15>("v_add3_u32", 16, out, clk);
    run<16>("v_sub_f32", 16, out, clk);
    run<17>("v_cvt_f32_u32", 16, out, clk);
    run<18>("v_mov_b32_dpp quad_perm", 16, out, clk);
    return 0;
}
