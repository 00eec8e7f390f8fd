// valu_probe3.hip — issue cost of the instructions the traversal kernel's node step is made of, per SIMD, on gfx950 — the round-2 probe (tools/archive/valu_probe.hip) redone without its
// two artefacts: (1) its v_cndmask_b32 read VCC that nothing in the kernel had written (22.7 "cycles"), (2) its SALU kernels
// ended their loop after a round (0.01): s_and_b64 writes SCC, which the loop's own compare-and-branch uses.  Here every mask is written inside the measured asm, SALU operands are compiler-allocated, every launch is checked, and each kind is measured with
// 1, 2, 4 and 8 waves per SIMD so that latency (1 wave) and issue rate (8 waves) can be told apart.
//   build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe3.hip -o gpurun_out/valu_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

enum { K_FMA, K_MUL, K_ADD, K_MAX, K_MAX3, K_MIN3, K_CVT_UB, K_AND, K_LSHL, K_ADDU, K_BFE, K_ALIGNBIT, K_PERM, K_MOV, K_RCP,
       K_CMP_VCC, K_CMP_SGPR, K_CND_VCC, K_CND_SGPR, K_CMP_CND_PAIR, K_SALU_AND, K_FMA_SALU_PAIR, K_PK_FMA_F16, K_PK_MAX_F16, K_CVT_F16, K_DS_WRITE, K_DS_READ, K_COUNT };
static const char* kNames[K_COUNT] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_cvt_f32_ubyte1", "v_and_b32", "v_lshlrev_b32", "v_add_u32", "v_bfe_u32",
    "v_alignbit_b32", "v_perm_b32", "v_mov_b32", "v_rcp_f32", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)",
    "v_cmp + v_cndmask pairs (per instr)", "s_and_b64", "v_fma_f32 + s_and_b64 pairs (per pair)", "v_pk_fma_f16", "v_pk_max_f16", "v_cvt_f16_f32", "ds_write_b32", "ds_read_b32" };

template <int KIND> __global__ __launch_bounds__(256) void k_rate(float* out, int iters, unsigned long long* clk) {
    __shared__ float lds[256 * 16];
    float r[16]; const float a = out[threadIdx.x & 7], b = out[8 + (threadIdx.x & 7)];
    const unsigned u = __float_as_uint(a) | 0x01020304u;
    unsigned long long m = __ballot(a < b + (float)(threadIdx.x & 1));          // a lane mask in an SGPR pair
    unsigned long long sacc = m;
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = a + i;
    float* my = lds + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define MAX3(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define MIN3(i) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[i]) : "v"(u));
#define AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define SHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
#define BFE(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(r[i]) : "v"(u));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %1, %0, 6" : "+v"(r[i]) : "v"(u));
#define PERM(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(u));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define CMPV(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define CMPS(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(r[i]), "v"(a));
#define CNDV(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "s"(m));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define SAND(i) asm volatile("s_and_b64 %0, %0, exec" : "+s"(sacc) : : "scc");   // (s_and writes SCC: undeclared, it broke the loop's own compare-and-branch — the 0.01 of the round-2 probe)
#define FMASAND(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n s_and_b64 %1, %1, exec" : "+v"(r[i]), "+s"(sacc) : "v"(a), "v"(b) : "scc");
#define PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define PKMAX16(i) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define CVT16(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r[i]) : "v"(a));
#define DSW(i) my[256 * i] = r[i];
#define DSR(i) r[i] = my[256 * i];
        if (KIND == K_FMA) { REP16(FMA) } if (KIND == K_MUL) { REP16(MUL) } if (KIND == K_ADD) { REP16(ADD) } if (KIND == K_MAX) { REP16(MAX) } if (KIND == K_MAX3) { REP16(MAX3) }
        if (KIND == K_MIN3) { REP16(MIN3) } if (KIND == K_CVT_UB) { REP16(CVT) } if (KIND == K_AND) { REP16(AND) } if (KIND == K_LSHL) { REP16(SHL) } if (KIND == K_ADDU) { REP16(ADDU) }
        if (KIND == K_BFE) { REP16(BFE) } if (KIND == K_ALIGNBIT) { REP16(ALIGN) } if (KIND == K_PERM) { REP16(PERM) } if (KIND == K_MOV) { REP16(MOV) } if (KIND == K_RCP) { REP16(RCP) }
        if (KIND == K_CMP_VCC) { REP16(CMPV) } if (KIND == K_CMP_SGPR) { REP16(CMPS) }
        if (KIND == K_CND_VCC) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[0]), "v"(b) : "vcc"); REP16(CNDV) }       // (one more instruction per round: VCC written right here)
        if (KIND == K_CND_SGPR) { REP16(CNDS) } if (KIND == K_CMP_CND_PAIR) { REP16(CMPCND) } if (KIND == K_SALU_AND) { REP16(SAND) } if (KIND == K_FMA_SALU_PAIR) { REP16(FMASAND) }
        if (KIND == K_PK_FMA_F16) { REP16(PKFMA16) } if (KIND == K_PK_MAX_F16) { REP16(PKMAX16) } if (KIND == K_CVT_F16) { REP16(CVT16) }
        if (KIND == K_DS_WRITE) { REP16(DSW) asm volatile("" ::: "memory"); } if (KIND == K_DS_READ) { asm volatile("" ::: "memory"); REP16(DSR) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)(unsigned)(sacc ^ m);
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x + 16] = s + my[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[KIND] = t1 - t0;
}

template <int K> struct runner {
    static void go(float* out, unsigned long long* clk, hipEvent_t e0, hipEvent_t e1, const hipDeviceProp_t& prop) {
        const int iters = 2048, cus = prop.multiProcessorCount;
        double cyc[4]; unsigned long long one = 0;
        for (int w = 0; w < 4; w++) {
            const int waves_per_simd = 1 << w, blocks = cus * waves_per_simd;      // a 256-lane block = one wave on each of a CU's 4 SIMDs
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double per_round = (K == K_CMP_CND_PAIR) ? 32.0 : (K == K_CND_VCC ? 17.0 : 16.0);
            cyc[w] = best * 1e-3 * (prop.clockRate * 1e3) / ((double)iters * per_round * waves_per_simd);       // cycles per wave-instruction per SIMD at the reported shader clock
            if (w == 0) CHECK(hipMemcpy(&one, clk + K, 8, hipMemcpyDeviceToHost));
        }
        std::printf("%-40s 1 wave %6.2f   2 waves %6.2f   4 waves %6.2f   8 waves %6.2f   cycles per wave-instruction per SIMD\n", kNames[K], cyc[0], cyc[1], cyc[2], cyc[3]);
        runner<K + 1>::go(out, clk, e0, e1, prop);
    }
};
template <> struct runner<K_COUNT> { static void go(float*, unsigned long long*, hipEvent_t, hipEvent_t, const hipDeviceProp_t&) {} };

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    float* out; unsigned long long* clk;
    const size_t n = (size_t)prop.multiProcessorCount * 8 * 256 + 16;
    CHECK(hipMalloc(&out, n * 4)); CHECK(hipMemset(out, 0, n * 4)); CHECK(hipMalloc(&clk, 64 * 8)); CHECK(hipMemset(clk, 0, 64 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::printf("%s, %d CUs, shader clock %.0f MHz; 2048 rounds of 16 independent instructions per wave; time of the launch / (rounds x 16 x waves per SIMD)\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3);
    runner<0>::go(out, clk, e0, e1, prop);
    return 0;
}
