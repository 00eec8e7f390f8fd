// gather_probe.hip — how fast can an MI355X gather random 64-byte records?  (the access pattern of the BVH traversal kernel)
//   A: every lane reads its own record with 4 x global_load_dwordx4 (what k_intersect does)
//   B: 4 consecutive lanes read one record cooperatively (one dwordx4 each), 4 rounds -> the wave still gets 64 records
//   C: like A but only 3 of the 4 dwordx4 (48-byte records)
//   D: like A with 1 dwordx4 (16-byte records)
// Each lane follows a dependent chain (the next index comes from the data), like traversal does.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o gpurun_out/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_chase(const float4* __restrict__ data, uint32_t n_rec, int steps, uint32_t* __restrict__ out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    float acc = 0.0f;
    const int lane = threadIdx.x & 63;
    for (int s = 0; s < steps; s++) {
        if (MODE == 1) {
            // round r: lanes 16r..16r+15's records are fetched by the whole wave: lane l reads quarter (l & 3) of record owner (16r + l/4)
            float4 mine[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int owner = 16 * r + (lane >> 2);
                const uint32_t oidx = __shfl(idx, owner, 64);
                const float4 part = data[(size_t)oidx * 4 + (lane & 3)];
                // hand the four quarters to the owner: owner lane o gets quarter q from lane 4*(o & 15) + q of round o / 16
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int src = 4 * (lane & 15) + q;
                    const float x = __shfl(part.x, src, 64), y = __shfl(part.y, src, 64), z = __shfl(part.z, src, 64), w = __shfl(part.w, src, 64);
                    if ((lane >> 4) == r) mine[q] = make_float4(x, y, z, w);
                }
            }
            acc += mine[0].x + mine[1].y + mine[2].z + mine[3].w;
            idx = (__float_as_uint(mine[0].w) + s) % n_rec;
        } else if (MODE >= 4) {   // 128-byte records (two per 256 B; idx counts 128-B records): E reads all 8 quarters, F reads 6
            const float4* p = data + (size_t)(idx & ~1u) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5];
            float4 g = make_float4(0, 0, 0, 0), h = g;
            if (MODE == 4) { g = p[6]; h = p[7]; }
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
            idx = (__float_as_uint(a.w) + s) % n_rec;
        } else {
            const float4* p = data + (size_t)idx * 4;
            const float4 a = p[0];
            float4 b = make_float4(0, 0, 0, 0), c = b, d = b;
            if (MODE == 0 || MODE == 2) { b = p[1]; c = p[2]; }
            if (MODE == 0) d = p[3];
            acc += a.x + b.y + c.z + d.w;
            idx = (__float_as_uint(a.w) + s) % n_rec;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(acc) ^ idx;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? std::atol(argv[1]) : 550;
    const uint32_t n_rec = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<float4> h((size_t)n_rec * 4);
    uint32_t x = 123456789u;
    for (uint32_t i = 0; i < n_rec; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t nxt = x % n_rec; float nf; std::memcpy(&nf, &nxt, 4);
        h[(size_t)i * 4] = make_float4(1.0f, 2.0f, 3.0f, nf);
        h[(size_t)i * 4 + 1] = h[(size_t)i * 4 + 2] = h[(size_t)i * 4 + 3] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
    }
    float4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * sizeof(float4))); CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    const int blocks = 256 * 8, steps = 200;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[6] = { "A  4 x dwordx4 per lane (64 B records)", "B  cooperative: 4 lanes per record + shuffles", "C  3 x dwordx4 per lane (48 B)", "D  1 x dwordx4 per lane (16 B)", "E  8 x dwordx4 per lane (128 B aligned records)", "F  6 x dwordx4 per lane (96 B of a 128 B record)" };
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            if (mode == 3) hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            if (mode == 4) hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            if (mode == 5) hipLaunchKernelGGL(k_chase<5>, dim3(blocks), dim3(256), 0, 0, d, n_rec, steps, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) {
                const double recs = (double)blocks * 256 * steps;
                std::printf("%-52s %8.3f ms  %7.2f G records/s  %7.2f TB/s of 64-B records\n", names[mode], ms, recs / ms / 1e6, recs * 64 / ms / 1e9);
            }
        }
    }
    return 0;
}
