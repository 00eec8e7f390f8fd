// sequence_generator.h — host generator of the per-pass sampling-sequence tables.
// Behaviour: SamplingSequenceGeneratorHost<IndependantSamplingSequenceGenerator>::Compute (Kernel/Sampler.h:36-85)
// on a CudaRNG(7539414) = cuRAND XORWOW curand_init(1234, 7539414, 0) (Base/CudaRandom.cu:26-33, CudaRandom.h:112-272).
// The sub-sequence jump (2^67 steps per sub-sequence) is computed here by GF(2) matrix powers of the XORWOW
// transition instead of cuRAND's pre-computed tables.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <thread>
#include <algorithm>

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

// 160x160 bit matrices over GF(2) acting on the xorwow v-state (rows = images of the state bits)
struct xorwow_matrix {
    std::vector<uint32_t> r;   // 160 rows x 5 words
    xorwow_matrix() : r(160 * 5, 0) {}
    static xorwow_matrix one_step() {
        xorwow_matrix M;
        for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) {
            xorwow s; s.d = 0; std::memset(s.v, 0, sizeof(s.v)); s.v[i] = 1u << j; s.next();
            std::memcpy(&M.r[(i * 32 + j) * 5], s.v, 20);
        }
        return M;
    }
    void apply(const uint32_t* in, uint32_t* out) const {
        uint32_t o[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < 5; i++) for (int j = 0; j < 32; j++) if (in[i] & (1u << j)) for (int k = 0; k < 5; k++) o[k] ^= r[(i * 32 + j) * 5 + k];
        std::memcpy(out, o, 20);
    }
    xorwow_matrix then(const xorwow_matrix& B) const {   // first *this, then B
        xorwow_matrix C; for (int row = 0; row < 160; row++) B.apply(&r[row * 5], &C.r[row * 5]); return C;
    }
    static xorwow_matrix power(uint64_t n) {   // (one step)^n
        xorwow_matrix result; for (int i = 0; i < 160; i++) result.r[i * 5 + i / 32] = 1u << (i % 32);   // identity
        xorwow_matrix sq = one_step();
        for (; n; n >>= 1) { if (n & 1) result = result.then(sq); if (n >> 1) sq = sq.then(sq); }
        return result;
    }
};

class sequence_generator {
    xorwow rng_;
    xorwow_matrix pass_jump_; bool have_jump_ = false;
    static constexpr unsigned N = 4096, L = 30;
    static constexpr uint64_t kDrawsPerPass = (uint64_t)N * L * 3;   // L 1-D values + L 2-D values per sequence
    static void fill(xorwow& rng, float* t1, float* t2) {
        // the draws come in sequence-major order (Sampler.h:76-84) while the tables are element-major: go through a 16-sequence
        // tile so that the table is written one whole cache line at a time
        constexpr unsigned TS = 16;
        float a[TS][L], b[TS][2 * L];
        for (unsigned s0 = 0; s0 < N; s0 += TS) {
            for (unsigned s = 0; s < TS; s++) {
                for (unsigned i = 0; i < L; i++) a[s][i] = rng.uniform();
                // Vec2f(rng.randomFloat(), rng.randomFloat()) (Sampler.h:83): the reference's host compilers evaluate
                // constructor arguments right to left, so the first draw is .y
                for (unsigned i = 0; i < L; i++) { const float y = rng.uniform(), x = rng.uniform(); b[s][2 * i] = x; b[s][2 * i + 1] = y; }
            }
            for (unsigned i = 0; i < L; i++) for (unsigned s = 0; s < TS; s++) {
                t1[i * N + s0 + s] = a[s][i];
                t2[2 * (i * N + s0 + s)] = b[s][2 * i]; t2[2 * (i * N + s0 + s) + 1] = b[s][2 * i + 1];
            }
        }
    }
public:
    sequence_generator() { rng_.init(1234, 7539414); }
    // tables for the next pass: t1[e*4096 + s], t2[2*(e*4096 + s) + {0,1}]
    void compute(float* t1, float* t2) { fill(rng_, t1, t2); }
    // ---- generation on the GPU (k_sequence_fill, kernels.hip): the host only advances the stream.
    // A pass is cut into kChunks chunks of kSeqPerChunk sequences; chunk c = 16 a + b starts (90 * kSeqPerChunk * c) draws into the pass, reached
    // from the pass's start state with two jump matrices, hi[a] = step^(draws(16 a)) and lo[b] = step^(draws(b)).
    static constexpr unsigned kSeqPerChunk = 16, kChunks = N / kSeqPerChunk;   // 256 chunks of 1440 draws
    static constexpr uint32_t kDrawsPerChunk = kSeqPerChunk * L * 3;
    struct pass_start { uint32_t v[5]; uint32_t d; };
    // start states of the next n passes; the generator moves on by n passes
    void take_pass_starts(unsigned n, pass_start* out) {
        if (!have_jump_) { pass_jump_ = xorwow_matrix::power(kDrawsPerPass); have_jump_ = true; }
        for (unsigned k = 0; k < n; k++) {
            std::memcpy(out[k].v, rng_.v, 20); out[k].d = rng_.d;
            xorwow nx = rng_; pass_jump_.apply(rng_.v, nx.v); nx.d = rng_.d + (uint32_t)(362437u * (uint32_t)kDrawsPerPass); rng_ = nx;
        }
    }
    // 32 matrices x 160 rows x 5 words: lo[0..15] then hi[0..15]
    static const std::vector<uint32_t>& chunk_jump_matrices() {
        static const std::vector<uint32_t> M = [] {
            std::vector<uint32_t> m; m.reserve(32 * 800);
            const xorwow_matrix lo1 = xorwow_matrix::power(kDrawsPerChunk), hi1 = xorwow_matrix::power((uint64_t)kDrawsPerChunk * 16);
            for (const xorwow_matrix* step : { &lo1, &hi1 }) {
                xorwow_matrix cur = xorwow_matrix::power(0);
                for (int k = 0; k < 16; k++) { m.insert(m.end(), cur.r.begin(), cur.r.end()); cur = cur.then(*step); }
            }
            return m;
        }();
        return M;
    }
    // tables of the next n passes, generated by up to `threads` host threads.  The stream is the same single XORWOW stream the
    // reference draws from: pass k starts kDrawsPerPass * k draws further on, reached with the GF(2) jump matrix of one pass
    // (the Weyl counter d advances linearly).
    void compute_many(float* t1, float* t2, unsigned n, size_t stride1, size_t stride2, unsigned threads);
};

inline void sequence_generator::compute_many(float* t1, float* t2, unsigned n, size_t stride1, size_t stride2, unsigned threads) {
    if (n == 0) return;
    if (n == 1 || threads <= 1) { for (unsigned k = 0; k < n; k++) compute(t1 + k * stride1, t2 + k * stride2); return; }
    if (!have_jump_) { pass_jump_ = xorwow_matrix::power(kDrawsPerPass); have_jump_ = true; }
    std::vector<xorwow> start(n + 1);
    start[0] = rng_;
    for (unsigned k = 0; k < n; k++) { start[k + 1] = start[k]; pass_jump_.apply(start[k].v, start[k + 1].v); start[k + 1].d = start[k].d + (uint32_t)(362437u * (uint32_t)kDrawsPerPass); }
    const unsigned T = std::min(threads, n);
    std::vector<std::thread> pool;
    auto work = [&](unsigned tid) { for (unsigned k = tid; k < n; k += T) { xorwow r = start[k]; fill(r, t1 + k * stride1, t2 + k * stride2); } };
    for (unsigned t = 1; t < T; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    rng_ = start[n];
}

} // namespace ctl
