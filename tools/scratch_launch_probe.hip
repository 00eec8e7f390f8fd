// What a launch of 2048 x 256 lanes costs when every wave has to be given scratch and / or LDS, and nothing else happens (the waves return at once).
// hipcc --offload-arch=gfx950 -O3 tools/scratch_launch_probe.hip -o /tmp/slp && /tmp/slp      (profiles/r04_pipeline_lanes.log)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCRATCH_INTS, int LDS_INTS>
__global__ __launch_bounds__(256) void k(const int* __restrict__ n, int* __restrict__ out) {
    __shared__ int lds[LDS_INTS > 0 ? LDS_INTS : 1];
    int priv[SCRATCH_INTS > 0 ? SCRATCH_INTS : 1];
    const int m = *n;
    if (m == 0) return;                       // every launch of the probe: nothing to do
    for (int i = 0; i < (SCRATCH_INTS > 0 ? SCRATCH_INTS : 1); i++) priv[i] = i * m;
    if (LDS_INTS > 0) lds[threadIdx.x % LDS_INTS] = m;
    __syncthreads();
    int s = 0;
    for (int i = 0; i < m; i++) s += priv[(i * 7 + threadIdx.x) % (SCRATCH_INTS > 0 ? SCRATCH_INTS : 1)] + (LDS_INTS > 0 ? lds[(i + threadIdx.x) % LDS_INTS] : 0);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int S, int L> float run(const int* n, int* out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<S, L>), dim3(grid), dim3(256), 0, 0, n, out);
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((k<S, L>), dim3(grid), dim3(256), 0, 0, n, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / 200 * 1000;
}
int main() {
    int *n, *out; hipMalloc(&n, 4); hipMemset(n, 0, 4); hipMalloc(&out, 2048 * 256 * 4);
    for (int grid : { 256, 1024, 2048 }) {
        printf("grid %4d x 256:  no scratch, no LDS %7.1f us | 79 ints of scratch %7.1f us | 20 KB of LDS %7.1f us | both %7.1f us | 8 ints of scratch + LDS %7.1f us\n", grid,
               run<0, 0>(n, out, grid), run<79, 0>(n, out, grid), run<0, 5120>(n, out, grid), run<79, 5120>(n, out, grid), run<8, 5120>(n, out, grid));
    }
    return 0;
}
