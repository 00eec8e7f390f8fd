// spline.h — the cubic spline lookup of the rough-transmittance tables: Spline::evalCubicInterp2D / 3D (Math/Spline.cu:223-296, 376-453) in their node-weight form,
// knots on [0,1]^n, extrapolate = false — the only way RoughTransmittance::Evaluate / EvaluateDiffuse call them (Engine/RoughTransmittance.cu:55-119).
// ONE statement of the arithmetic for its three users: the shade kernels (bsdf_rough.h), the host-side reduction of a constant-roughness material's table to 1-D
// (tracer.hip), and — compiled by g++ — tests/test_oracle_golden.py, which holds it bit for bit against the reference's own Math/Spline.cu (tests/golden/spline.npz).
// Expression order is the reference's; build with -ffp-contract=off.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __HIPCC__
#define CTL_SPLINE_FN __host__ __device__ __forceinline__
#else
#define CTL_SPLINE_FN inline
#endif

namespace ctl {

// one dimension (Spline.cu:232-277 / 385-430): the four node weights of the cubic through knot-1 .. knot+2 and the left knot; false = p outside [0,1] or NaN
CTL_SPLINE_FN bool spline_weights(float p, uint32_t size, float* w, uint32_t& knot) {
    if (!(p >= 0.0f && p <= 1.0f)) return false;
    float t = ((p - 0.0f) * (size - 1)) / (1.0f - 0.0f);
    knot = (uint32_t)t < size - 2 ? (uint32_t)t : size - 2;
    t = t - (float)knot;
    const float t2 = t * t, t3 = t2 * t;
    w[0] = 0.0f; w[1] = 2 * t3 - 3 * t2 + 1; w[2] = -2 * t3 + 3 * t2; w[3] = 0.0f;
    const float d0 = t3 - 2 * t2 + t, d1 = t3 - t2;
    if (knot > 0) { w[2] += 0.5f * d0; w[0] -= 0.5f * d0; } else { w[2] += d0; w[1] -= d0; }
    if (knot + 2 < size) { w[3] += 0.5f * d1; w[1] -= 0.5f * d1; } else { w[2] += d1; w[1] -= d1; }
    return true;
}
// Spline.cu:279-295 — x fastest
CTL_SPLINE_FN float spline_eval_2d(float px, float py, const float* values, uint32_t sx, uint32_t sy) {
    float wx[4], wy[4]; uint32_t kx, ky;
    if (!spline_weights(px, sx, wx, kx) || !spline_weights(py, sy, wy, ky)) return 0.0f;
    float result = 0.0f;
    for (int y = -1; y <= 2; ++y)
        for (int x = -1; x <= 2; ++x) {
            const float wxy = wx[x + 1] * wy[y + 1];
            if (wxy == 0) continue;
            result += values[(size_t)(ky + y) * sx + kx + x] * wxy;
        }
    return result;
}
// Spline.cu:432-452
CTL_SPLINE_FN float spline_eval_3d(float px, float py, float pz, const float* values, uint32_t sx, uint32_t sy, uint32_t sz) {
    float wx[4], wy[4], wz[4]; uint32_t kx, ky, kz;
    if (!spline_weights(px, sx, wx, kx) || !spline_weights(py, sy, wy, ky) || !spline_weights(pz, sz, wz, kz)) return 0.0f;
    float result = 0.0f;
    for (int z = -1; z <= 2; ++z)
        for (int y = -1; y <= 2; ++y) {
            const float wyz = wy[y + 1] * wz[z + 1];
            for (int x = -1; x <= 2; ++x) {
                const float wxyz = wx[x + 1] * wyz;
                if (wxyz == 0) continue;
                result += values[((size_t)(kz + z) * sy + (ky + y)) * sx + kx + x] * wxyz;
            }
        }
    return result;
}
// The build's own 1-D form: the same node weights over one axis.  (The reference's evalCubicInterp1D, Spline.cu:6-44, is the derivative form — equal up to rounding; it is
// not on the path: RoughTransmittance only calls the 2-D and 3-D functions.)  Used on the 1-D table the 3-D one is reduced to for a constant-roughness material.
CTL_SPLINE_FN float spline_eval_1d(float p, const float* values, uint32_t size) {
    float w[4]; uint32_t knot;
    if (!spline_weights(p, size, w, knot)) return 0.0f;
    float result = 0.0f;
    for (int x = -1; x <= 2; ++x) { if (w[x + 1] == 0) continue; result += values[knot + x] * w[x + 1]; }
    return result;
}
// the 3-D table at fixed (py, pz) as a 1-D table over x: out[x] = sum_yz values[z][y][x] * wy * wz (tracer.hip; summed y / z first instead of last: equal up to fp32 rounding)
inline bool spline_reduce_3d_to_1d(float py, float pz, const float* values, uint32_t sx, uint32_t sy, uint32_t sz, float* out) {
    float wy[4], wz[4]; uint32_t ky, kz;
    const bool ok = spline_weights(py, sy, wy, ky) && spline_weights(pz, sz, wz, kz);
    for (uint32_t x = 0; x < sx; x++) {
        float v = 0.0f;
        if (ok) for (int z = -1; z <= 2; ++z) for (int y = -1; y <= 2; ++y) {
            const float wyz = wy[y + 1] * wz[z + 1];
            if (wyz == 0) continue;
            v += values[((size_t)(kz + z) * sy + (ky + y)) * sx + x] * wyz;
        }
        out[x] = v;
    }
    return ok;
}

}  // namespace ctl
