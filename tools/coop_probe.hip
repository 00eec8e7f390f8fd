// coop_probe.hip — is the L1 (TCP) cost of a random 64-byte record fetch paid per LANE-LOAD or per LINE?
//   A: every lane reads its own 64-B record with 4 x global_load_dwordx4 (what k_intersect did in round 1)
//   T: quad-cooperative: in round r the four lanes of a quad read the record of the quad's lane r, 16 B each (one contiguous 64-B access per quad),
//      the chunks are transposed through LDS (4 x ds_write_b128, 4 x ds_read_b128) so that every lane ends up with its own record
// Tables from L1-resident to HBM-resident; each lane follows a dependent chain like a traversal does.  Prints G records/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

constexpr int kRow = 1024 + 16;   // one round's 64 x 16 B + 16 B of padding: the transposed reads are bank-conflict free

template <int MODE> __global__ __launch_bounds__(256) void k_chase(const float4* __restrict__ data, uint32_t n_rec, int steps, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 4 * kRow];
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    float acc = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* wbase = stage + wave * 4 * kRow;
    float4* wr = (float4*)(wbase + lane * 16);                                    // + r * kRow
    const float4* rd = (const float4*)(wbase + (lane & 3) * kRow + (lane >> 2) * 64);   // + k * 16
    for (int s = 0; s < steps; s++) {
        float4 a, b, c, d;
        if (MODE == 0) {
            const float4* p = data + (size_t)idx * 4;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else if (MODE == 2 || MODE == 3) {
            // quad-cooperative, transposed in registers: two butterfly stages of v_cndmask with a DPP source (32 VALU for 64 B per lane, no LDS).  MODE 3: 48 B (lane 3 of the quad idles)
            uint32_t t[4][4];
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xf, 0xf, false), i1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xf, 0xf, false);
            const uint32_t i2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xaa, 0xf, 0xf, false), i3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xff, 0xf, 0xf, false);
            const uint32_t ii[4] = { i0, i1, i2, i3 };
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (MODE == 2 || (lane & 3) != 3) v = *(const uint4*)&data[(size_t)ii[r] * 4 + (lane & 3)];
                t[r][0] = v.x; t[r][1] = v.y; t[r][2] = v.z; t[r][3] = v.w;
            }
            const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                // stage 1: lanes q and q^1 exchange (reg r <-> reg r^1) where bit 0 of q and r differ
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t[1][w], 0xb1, 0xf, 0xf, false), p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t[0][w], 0xb1, 0xf, 0xf, false);
                const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t[3][w], 0xb1, 0xf, 0xf, false), p3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t[2][w], 0xb1, 0xf, 0xf, false);
                const uint32_t u0 = b0 ? p0 : t[0][w], u1 = b0 ? t[1][w] : p1, u2 = b0 ? p2 : t[2][w], u3 = b0 ? t[3][w] : p3;
                // stage 2: lanes q and q^2, reg r <-> r^2
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)u2, 0x4e, 0xf, 0xf, false), q2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)u0, 0x4e, 0xf, 0xf, false);
                const uint32_t q1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)u3, 0x4e, 0xf, 0xf, false), q3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)u1, 0x4e, 0xf, 0xf, false);
                t[0][w] = b1 ? q0 : u0; t[1][w] = b1 ? q1 : u1; t[2][w] = b1 ? u2 : q2; t[3][w] = b1 ? u3 : q3;
            }
            a = make_float4(__uint_as_float(t[0][0]), __uint_as_float(t[0][1]), __uint_as_float(t[0][2]), __uint_as_float(t[0][3]));
            b = make_float4(__uint_as_float(t[1][0]), __uint_as_float(t[1][1]), __uint_as_float(t[1][2]), __uint_as_float(t[1][3]));
            c = make_float4(__uint_as_float(t[2][0]), __uint_as_float(t[2][1]), __uint_as_float(t[2][2]), __uint_as_float(t[2][3]));
            d = make_float4(__uint_as_float(t[3][0]), __uint_as_float(t[3][1]), __uint_as_float(t[3][2]), __uint_as_float(t[3][3]));
            if (MODE == 3) d = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
        } else if (MODE == 5 || MODE == 6) {   // A3 with half / a quarter of the lanes taking part: is the L1's cost per wave instruction or per active lane?
            a = b = c = d = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
            if ((lane & (MODE == 5 ? 1 : 3)) == 0) { const float4* p = data + (size_t)idx * 4; a = p[0]; b = p[1]; c = p[2]; }
        } else if (MODE == 4) {   // own record, 3 x dwordx4 (48 B): what the node step does today
            const float4* p = data + (size_t)idx * 4;
            a = p[0]; b = p[1]; c = p[2]; d = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
        } else {
            float4 t[4];
            // quad_perm [r,r,r,r]: every lane of a quad gets the index held by the quad's lane r
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xf, 0xf, false), i1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xf, 0xf, false);
            const uint32_t i2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xaa, 0xf, 0xf, false), i3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xff, 0xf, 0xf, false);
            t[0] = data[(size_t)i0 * 4 + (lane & 3)]; t[1] = data[(size_t)i1 * 4 + (lane & 3)]; t[2] = data[(size_t)i2 * 4 + (lane & 3)]; t[3] = data[(size_t)i3 * 4 + (lane & 3)];
#pragma unroll
            for (int r = 0; r < 4; r++) *(float4*)((unsigned char*)wr + r * kRow) = t[r];
            a = rd[0]; b = rd[1]; c = rd[2]; d = rd[3];
        }
        acc += a.x + b.y + c.z + d.w;
        idx = (__float_as_uint(a.w) + s) % n_rec;
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(acc) ^ idx;
}

int main() {
    const size_t max_mb = 1024;
    const uint32_t max_rec = (uint32_t)(max_mb * 1024 * 1024 / 64);
    std::vector<float4> h((size_t)max_rec * 4);
    float4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * sizeof(float4)));
    const int blocks = 256 * 8, steps = 200;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[7] = { "A  own record, 4 x dwordx4 per lane", "T  quad-cooperative + LDS transpose", "D  quad-cooperative + DPP transpose", "D3 quad-cooperative 48 B + DPP transpose", "A3 own record, 3 x dwordx4 per lane", "H2 like A3, every 2nd lane only (records/s counts all lanes: x 1/2)", "H4 like A3, every 4th lane only (x 1/4)" };
    const double kb_list[] = { 8, 256, 4096, 65536, 184320, 1048576 };   // per-table KiB: L1, L2-of-one-XCD, L2, Infinity Cache, HBM
    for (double kb : kb_list) {
        const uint32_t n_rec = (uint32_t)(kb * 1024 / 64);
        uint32_t x = 123456789u;
        for (uint32_t i = 0; i < n_rec; i++) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const uint32_t nxt = x % n_rec; float nf; std::memcpy(&nf, &nxt, 4);
            h[(size_t)i * 4] = make_float4(1.0f, 2.0f, 3.0f, nf);
            h[(size_t)i * 4 + 1] = h[(size_t)i * 4 + 2] = h[(size_t)i * 4 + 3] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
        }
        CHECK(hipMemcpy(d, h.data(), (size_t)n_rec * 64, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 7; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else if (mode == 3) hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else if (mode == 4) hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else if (mode == 5) hipLaunchKernelGGL(k_chase<5>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                else hipLaunchKernelGGL(k_chase<6>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double recs = (double)blocks * 256 * steps;
            std::printf("table %8.0f KiB  %-44s %8.3f ms  %7.2f G records/s\n", kb, names[mode], best, recs / best / 1e6);
        }
    }
    // check that T returns what A returns
    uint32_t* oa = (uint32_t*)std::malloc(blocks * 256 * 4); uint32_t* ob = (uint32_t*)std::malloc(blocks * 256 * 4);
    hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), 0, 0, d, max_rec, 50, out); CHECK(hipMemcpy(oa, out, blocks * 256 * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), 0, 0, d, max_rec, 50, out); CHECK(hipMemcpy(ob, out, blocks * 256 * 4, hipMemcpyDeviceToHost));
    std::printf("T == A: %s\n", std::memcmp(oa, ob, blocks * 256 * 4) == 0 ? "yes" : "NO");
    hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), 0, 0, d, max_rec, 50, out); CHECK(hipMemcpy(ob, out, blocks * 256 * 4, hipMemcpyDeviceToHost));
    std::printf("D == A: %s\n", std::memcmp(oa, ob, blocks * 256 * 4) == 0 ? "yes" : "NO");
    hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(256), 0, 0, d, max_rec, 50, out); CHECK(hipMemcpy(ob, out, blocks * 256 * 4, hipMemcpyDeviceToHost));
    std::printf("D3 == A: %s\n", std::memcmp(oa, ob, blocks * 256 * 4) == 0 ? "yes" : "NO");
    return 0;
}
