// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// orng.h — XORWOW generator + the per-pass sampling-sequence tables.
//
// PARITY UNPINNED at one boundary: the reference calls cuRAND's curand_init(1234, 7539414, 0), whose sub-sequence
// skip-ahead multiplies the state by pre-computed GF(2) matrices shipped in NVIDIA's curand_precalc.h (CUDA toolkit
// 9.0, README.md:18) — absent from /root/reference and from this image.  The published algorithm is restated here:
// the matrices are T^(2^67 * 4^i) for the XORWOW transition T (XORWOW_SEQUENCE_SPACING = 67), computed from the
// in-tree recurrence (Base/CudaRandom.h:112-123).  tests/ cross-check T^(2^67 * 4^i) against rocRAND's independent
// pre-computed tables of the same recurrence (/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

struct Xorwow {
    uint32_t d, v[5];
    // Base/CudaRandom.h:112-123
    uint32_t next() {
        uint32_t t = (v[0] ^ (v[0] >> 2));
        v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
        v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
        d += 362437;
        return v[4] + d;
    }
};

// 160x160 GF(2) map stored as the images of the 160 basis vectors — the layout of Base/CudaRandom.h:128-142:
// row (i*32+j) = image (5 words) of bit j of word i.
struct XMat { uint32_t r[160][5]; };
inline void xmatApply(const XMat& M, const uint32_t in[5], uint32_t out[5]) {   // __curand_matvec
    uint32_t res[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) if (in[i] & (1u << j)) for (int k = 0; k < 5; k++) res[k] ^= M.r[i * 32 + j][k];
    std::memcpy(out, res, sizeof(res));
}
inline void xmatMul(XMat& A, const XMat& B) {   // __curand_matmat: A <- (apply A, then B)
    for (int i = 0; i < 160; i++) { uint32_t t[5]; xmatApply(B, A.r[i], t); std::memcpy(A.r[i], t, sizeof(t)); }
}
inline XMat xmatTransition() {
    XMat T;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) {
        Xorwow s; s.d = 0; for (int k = 0; k < 5; k++) s.v[k] = 0; s.v[i] = 1u << j;
        s.next();
        for (int k = 0; k < 5; k++) T.r[i * 32 + j][k] = s.v[k];
    }
    return T;
}
// T^(2^log2pow)
inline XMat xmatPow2(int log2pow) { XMat M = xmatTransition(); for (int i = 0; i < log2pow; i++) { XMat C = M; xmatMul(M, C); } return M; }

// Base/CudaRandom.h:253-272 (_curand_init_scratch) with offset = 0; skipahead_sequence = T^(2^67 * subsequence)
inline Xorwow xorwowInit(uint64_t seed, uint64_t subsequence) {
    Xorwow st;
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49UL, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddUL;
    uint32_t t0 = 1099087573UL * s0, t1 = 2591861531UL * s1;
    st.d = 6615241 + t1 + t0;
    st.v[0] = 123456789UL + t0; st.v[1] = 362436069UL ^ t0; st.v[2] = 521288629UL + t1; st.v[3] = 88675123UL ^ t1; st.v[4] = 5783321UL + t0;
    // Base/CudaRandom.h:166-208: base-4 digits of `subsequence`, digit k applies (T^(2^67 * 4^k)) digit times
    XMat M = xmatPow2(67);
    uint64_t p = subsequence;
    while (p) {
        for (unsigned t = 0; t < (p & 3); t++) { uint32_t o[5]; xmatApply(M, st.v, o); std::memcpy(st.v, o, sizeof(o)); }
        p >>= 2;
        if (p) { XMat C = M; xmatMul(M, C); C = M; xmatMul(M, C); }   // M <- M^4
    }
    return st;
}

// Base/CudaRandom.cu:7-16 + CudaRandom.h:124-127
inline float xorwowFloat(Xorwow& s) {
    const float CURAND_2POW32_INV = 2.3283064e-10f;
    float f = s.next() * CURAND_2POW32_INV + (CURAND_2POW32_INV / 2.0f);
    return f * (1 - 1e-5f);
}

// SamplingSequenceGeneratorHost<IndependantSamplingSequenceGenerator>::Compute (Kernel/Sampler.h:36-85).
// One generator persists across passes; per call it emits, for s = 0..4095, 30 floats (1-D row) then 30 Vec2f.
// `Vec2f(rng.randomFloat(), rng.randomFloat())` (Sampler.h:83) has unspecified argument evaluation order; gcc and MSVC —
// the reference's host compilers — evaluate right to left, so the FIRST draw lands in .y.
struct SequenceGenerator {
    Xorwow rng;
    SequenceGenerator() : rng(xorwowInit(1234, 7539414)) {}
    void compute(float* t1, float* t2, unsigned numSeq = 4096, unsigned seqLen = 30) {
        for (unsigned s = 0; s < numSeq; s++) {
            for (unsigned i = 0; i < seqLen; i++) t1[i * numSeq + s] = xorwowFloat(rng);
            for (unsigned i = 0; i < seqLen; i++) { float y = xorwowFloat(rng), x = xorwowFloat(rng); t2[2 * (i * numSeq + s)] = x; t2[2 * (i * numSeq + s) + 1] = y; }
        }
    }
};

} // namespace orc
