// sequence_generator.h — host generator of the per-pass sampling-sequence tables.
// Behaviour: SamplingSequenceGeneratorHost<IndependantSamplingSequenceGenerator>::Compute (Kernel/Sampler.h:36-85)
// on a CudaRNG(7539414) = cuRAND XORWOW curand_init(1234, 7539414, 0) (Base/CudaRandom.cu:26-33, CudaRandom.h:112-272).
// The sub-sequence jump (2^67 steps per sub-sequence) is computed here by GF(2) matrix powers of the XORWOW
// transition instead of cuRAND's pre-computed tables.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace ctl {

class xorwow {
public:
    uint32_t d, v[5];
    inline uint32_t next() {   // Base/CudaRandom.h:112-123
        const uint32_t t = (v[0] ^ (v[0] >> 2));
        v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
        v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
        d += 362437;
        return v[4] + d;
    }
    inline float uniform() {   // curand_uniform2 * (1 - 1e-5) (Base/CudaRandom.h:124-127, CudaRandom.cu:7-16)
        const float inv = 2.3283064e-10f;
        const float f = next() * inv + (inv / 2.0f);
        return f * (1 - 1e-5f);
    }
    // curand_init(seed, subsequence, 0)
    void init(uint64_t seed, uint64_t subsequence) {
        const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
        const uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
        d = 6615241u + t1 + t0;
        v[0] = 123456789u + t0; v[1] = 362436069u ^ t0; v[2] = 521288629u + t1; v[3] = 88675123u ^ t1; v[4] = 5783321u + t0;
        // bit-matrix of one step: rows = images of the 160 state bits
        std::vector<uint32_t> M(160 * 5), T(160 * 5);
        for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) {
            xorwow s; s.d = 0; std::memset(s.v, 0, sizeof(s.v)); s.v[i] = 1u << j; s.next();
            std::memcpy(&M[(i * 32 + j) * 5], s.v, 20);
        }
        auto apply = [](const std::vector<uint32_t>& A, const uint32_t* in, uint32_t* out) {
            uint32_t r[5] = { 0, 0, 0, 0, 0 };
            for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) if (in[i] & (1u << j)) for (int k = 0; k < 5; k++) r[k] ^= A[(i * 32 + j) * 5 + k];
            std::memcpy(out, r, 20);
        };
        auto square = [&](std::vector<uint32_t>& A) { T = A; for (int r = 0; r < 160; r++) apply(T, &T[r * 5], &A[r * 5]); };
        for (int i = 0; i < 67; i++) square(M);   // one sub-sequence = 2^67 steps
        for (uint64_t p = subsequence; p; p >>= 1) { if (p & 1) apply(M, v, v); if (p >> 1) square(M); }
    }
};

class sequence_generator {
    xorwow rng_;
public:
    sequence_generator() { rng_.init(1234, 7539414); }
    // tables for the next pass: t1[e*4096 + s], t2[2*(e*4096 + s) + {0,1}]
    void compute(float* t1, float* t2) {
        const unsigned N = 4096, L = 30;
        for (unsigned s = 0; s < N; s++) {
            for (unsigned i = 0; i < L; i++) t1[i * N + s] = rng_.uniform();
            // Vec2f(rng.randomFloat(), rng.randomFloat()) (Sampler.h:83): the reference's host compilers evaluate
            // constructor arguments right to left, so the first draw is .y
            for (unsigned i = 0; i < L; i++) { const float y = rng_.uniform(), x = rng_.uniform(); t2[2 * (i * N + s)] = x; t2[2 * (i * N + s) + 1] = y; }
        }
    }
};

} // namespace ctl
