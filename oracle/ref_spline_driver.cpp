// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_spline_driver.cpp — extern "C" driver around the reference's own Math/Spline.cu: Spline::evalCubicInterp1D / 2D / 3D (Math/Spline.cu:6-44, 223-296, 376-453), the
// 64-tap arithmetic behind RoughTransmittance::Evaluate / EvaluateDiffuse (Engine/RoughTransmittance.cu:55-119), i.e. behind roughplastic and roughcoating.
// `make ref` compiles the reference file where it lies, behind ONE preface line — `using std::min; using std::max;` — because the file calls CUDA's global ::min / ::max
// (Spline.cu:17,244), which a host compiler does not have; nothing else is added or removed.  This file contains no reference source.
#include <Math/Spline.h>
#include <cstdint>

using namespace CudaTracerLib;

extern "C" {

float ref_spline_eval_1d(float x, const float* values, uint32_t size, float lo, float hi, int extrapolate) {
    return Spline::evalCubicInterp1D(x, values, (size_t)size, lo, hi, extrapolate != 0);
}
float ref_spline_eval_2d(const float p[2], const float* values, const uint32_t size[2], const float lo[2], const float hi[2], int extrapolate) {
    return Spline::evalCubicInterp2D(Vec2f(p[0], p[1]), values, make_uint2(size[0], size[1]), Vec2f(lo[0], lo[1]), Vec2f(hi[0], hi[1]), extrapolate != 0);
}
float ref_spline_eval_3d(const float p[3], const float* values, const uint32_t size[3], const float lo[3], const float hi[3], int extrapolate) {
    return Spline::evalCubicInterp3D(Vec3f(p[0], p[1], p[2]), values, make_uint3(size[0], size[1], size[2]), Vec3f(lo[0], lo[1], lo[2]), Vec3f(hi[0], hi[1], hi[2]), extrapolate != 0);
}

}  // extern "C"
