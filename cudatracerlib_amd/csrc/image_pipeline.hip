// image_pipeline.hip — applyImagePipeline with a reconstruction filter and / or the tone-mapping post-process
// (Kernel/ImagePipeline/ImagePipeline.cu:54-84, Filter/CanonicalFilter.cu:6-44, PostProcess/ToneMapPostProcess.cu:6-42,
// Engine/Image.cu:88-168).  Pure bandwidth kernels over the PixelData frame; one pixel per lane, rows contiguous.
#include "kernels.h"
#include "tracer.h"
#include <cfloat>
#include <climits>
#include <cmath>

namespace ctl {

namespace {

__device__ __forceinline__ f3 to_spectrum(const ctl_pixel_data& p, float splat_scale) {   // PixelData::toSpectrum (Engine/Image.h:21-28)
    const float w = p.weight_sum != 0 ? p.weight_sum : 1;
    return f3(p.rgb[0] / w + p.rgb_splat[0] * splat_scale, p.rgb[1] / w + p.rgb_splat[1] * splat_scale, p.rgb[2] / w + p.rgb_splat[2] * splat_scale);
}
// SpectrumConverter::Float3ToRGBE / RGBEToFloat3 (Math/Spectrum.h:534-565)
__device__ __forceinline__ uint32_t to_rgbe(f3 c) {
    float m = max2(max2(c.x, c.y), c.z);
    if (m < 1e-32f) return 0u;
    int e; m = (float)frexp((double)m, &e) * 256.0f / m;
    // float -> unsigned char saturates on the reference's device (negative lobes of the Mitchell / Lanczos filters reach here): say so explicitly
    auto u8 = [](float v) { return (uint32_t)min2(max2(v, 0.0f), 255.0f); };
    return u8(c.x * m) | (u8(c.y * m) << 8) | (u8(c.z * m) << 16) | ((uint32_t)(unsigned char)(e + 128) << 24);
}
__device__ __forceinline__ f3 from_rgbe(uint32_t v) {
    const uint32_t w = v >> 24;
    if (!w) return f3(0.0f);
    const float e = ldexpf(1.0f, (int)w - (128 + 8));
    return f3((v & 0xff) * e, ((v >> 8) & 0xff) * e, ((v >> 16) & 0xff) * e);
}
// Float3ToCOLORREF / COLORREFToFloat3 (Math/Spectrum.h:521-532)
__device__ __forceinline__ uint32_t to_rgbcol(f3 c) {
    return (uint32_t)(unsigned char)(clampf(c.x, 0.0f, 1.0f) * 255.0f) | ((uint32_t)(unsigned char)(clampf(c.y, 0.0f, 1.0f) * 255.0f) << 8) |
           ((uint32_t)(unsigned char)(clampf(c.z, 0.0f, 1.0f) * 255.0f) << 16) | (255u << 24);
}
__device__ __forceinline__ f3 from_rgbcol(uint32_t v) { return f3(float(v & 0xff) / 255.0f, float((v >> 8) & 0xff) / 255.0f, float((v >> 16) & 0xff) / 255.0f); }
__device__ __forceinline__ float srgb(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, (float)(1.0 / 2.4)) - 0.055f; }   // Spectrum.cu:229-235
__device__ __forceinline__ uint32_t gamma_correct(f3 c) { return to_rgbcol(f3(srgb(c.x), srgb(c.y), srgb(c.z))); }                           // gammaCorrecture (ImagePipeline.cu:7-12)
__device__ __forceinline__ float lum(f3 s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }

// SceneTypes/Filter.h: Evaluate(|dx|, |dy|) of the five reconstruction filters
struct dev_filter { uint32_t type; float xw, yw, ixw, iyw, p0, p1, ex, ey; };
__device__ __forceinline__ float mitchell1d(float x, float B, float C) {
    x = fabsf(2.f * x);
    if (x > 1.f) return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) * (1.f / 6.f);
    return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) * (1.f / 6.f);
}
__device__ __forceinline__ float sinc1d(float x, float tau) {
    x = fabsf(x);
    if (x < 1e-5f) return 1.f;
    if (x > 1.f) return 0.f;
    x *= 3.14159265358979323846f;
    return (sinf(x) / x) * (sinf(x * tau) / (x * tau));
}
__device__ __forceinline__ float filter_eval(const dev_filter& F, float x, float y) {
    switch (F.type) {
    case CTL_RFILTER_BOX: return 1.0f;
    case CTL_RFILTER_GAUSSIAN: return max2(0.f, expf(-F.p0 * x * x) - F.ex) * max2(0.f, expf(-F.p0 * y * y) - F.ey);
    case CTL_RFILTER_MITCHELL: return mitchell1d(x * F.ixw, F.p0, F.p1) * mitchell1d(y * F.iyw, F.p0, F.p1);
    case CTL_RFILTER_LANCZOS: return sinc1d(x * F.ixw, F.p0) * sinc1d(y * F.iyw, F.p0);
    default: return max2(0.f, F.xw - fabsf(x)) * max2(0.f, F.yw - fabsf(y));
    }
}

// rtm_Copy + evalFilter (CanonicalFilter.cu:6-36): weighted mean of the pixel values under the filter footprint -> RGBE
__global__ __launch_bounds__(256) void k_filter(const ctl_pixel_data* __restrict__ px, int w, int h, float splat_scale, dev_filter F, uint32_t* __restrict__ filtered) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int x0 = max(0, (int)ceilf(x - F.xw)), x1 = min(w - 1, (int)floorf(x + F.xw)), y0 = max(0, (int)ceilf(y - F.yw)), y1 = min(h - 1, (int)floorf(y + F.yw));
    f3 acc(0.0f); float accw = 0;
    if (x1 - x0 >= 0 && y1 - y0 >= 0) {
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
                const float wt = filter_eval(F, (float)abs(xx - x), (float)abs(yy - y));
                acc = acc + to_spectrum(px[(size_t)yy * w + xx], splat_scale) * wt;
                accw += wt;
            }
        acc = acc / accw;
    }
    filtered[(size_t)y * w + x] = to_rgbe(acc);
}
// copySamplesToFiltered (ImagePipeline.cu:24-31)
__global__ __launch_bounds__(256) void k_to_filtered(const ctl_pixel_data* __restrict__ px, uint32_t n, float splat_scale, uint32_t* __restrict__ filtered) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) filtered[i] = to_rgbe(to_spectrum(px[i], splat_scale));
}
// copyFilteredToOutput (ImagePipeline.cu:33-41)
__global__ __launch_bounds__(256) void k_filtered_to_output(const uint32_t* __restrict__ filtered, uint32_t n, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = gamma_correct(from_rgbe(filtered[i]));
}
// applyGammaCorrectureToOutput (ImagePipeline.cu:43-52)
__global__ __launch_bounds__(256) void k_gamma_in_place(uint32_t* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = gamma_correct(from_rgbcol(out[i]));
}
// computeLuminanceInfo (Engine/Image.cu:93-130): min / max as ordered ints, sums of Y and log(2.3e-5 + Y); wave shuffles, then one atomic per wave
struct lum_info { int min_i, max_i; float sum, sum_log; };
__device__ __forceinline__ int ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__global__ __launch_bounds__(256) void k_luminance_info(const uint32_t* __restrict__ filtered, uint32_t n, lum_info* __restrict__ out) {
    int mn = INT_MAX, mx = INT_MIN; float s = 0, sl = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float Y = lum(from_rgbe(filtered[i])); const int iy = ordered(Y);
        mn = min(mn, iy); mx = max(mx, iy); s += Y; sl += logf(2.3e-5f + Y);
    }
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_down(mn, off, 64)); mx = max(mx, __shfl_down(mx, off, 64)); s += __shfl_down(s, off, 64); sl += __shfl_down(sl, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out->min_i, mn); atomicMax(&out->max_i, mx); atomicAdd(&out->sum, s); atomicAdd(&out->sum_log, sl); }
}
// Reinhard05Kernel (ToneMapPostProcess.cu:6-26) with Spectrum::toYxy / fromYxy (Spectrum.cu:160-172,286-302)
__global__ __launch_bounds__(256) void k_reinhard(const uint32_t* __restrict__ filtered, uint32_t n, float scale, float invWp2, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const f3 c = from_rgbe(filtered[i]);
        const float X = c.x * 0.412453f + c.y * 0.357580f + c.z * 0.180423f, Y0 = c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f, Z = c.x * 0.019334f + c.y * 0.119193f + c.z * 0.950227f;
        const float s = clampf(X + Y0 + Z, 0.001f, 100000.0f);
        const float x = X / s, y = Y0 / s;
        const float Lp = scale * Y0;
        const float Y = Lp * (1.0f + Lp * invWp2) / (1.0f + Lp);
        const float yc = clampf(y, 0.001f, 100000.0f);
        const float X2 = Y / yc * x, Z2 = Y / yc * (1 - x - y);
        out[i] = to_rgbcol(f3(3.240479f * X2 + -1.537150f * Y + -0.498535f * Z2, -0.969256f * X2 + 1.875991f * Y + 0.041556f * Z2, 0.055648f * X2 + -0.204043f * Y + 1.057311f * Z2));
    }
}

} // namespace

void Image::apply_pipeline_ex(float splat_scale, const ctl_reconstruction_filter* filter, const ctl_tonemap* process, uint32_t* host_rgbcol) {
    if (!filter && !process) { apply_pipeline(splat_scale, host_rgbcol); return; }
    if (filter) {
        if (filter->type < CTL_RFILTER_BOX || filter->type > CTL_RFILTER_TRIANGLE) throw std::runtime_error("applyImagePipeline: unknown reconstruction filter type");
        if (!(filter->x_width > 0) || !(filter->y_width > 0) || filter->x_width > 64 || filter->y_width > 64) throw std::runtime_error("applyImagePipeline: filter width must be in (0, 64]");
    }
    const uint32_t n = (uint32_t)px_.n;
    if (!out_.p) out_.alloc(n);
    if (!filtered_.p) filtered_.alloc(n);
    CTL_HIP(hipDeviceSynchronize());
    const int grid = (int)std::min<uint32_t>(4096, (n + 255) / 256);
    if (filter) {
        dev_filter F{ filter->type, filter->x_width, filter->y_width, 1.0f / filter->x_width, 1.0f / filter->y_width, filter->p0, filter->p1, 0, 0 };
        if (F.type == CTL_RFILTER_GAUSSIAN) { F.ex = expf(-F.p0 * F.xw * F.xw); F.ey = expf(-F.p0 * F.yw * F.yw); }   // GaussianFilter::Update
        hipLaunchKernelGGL(k_filter, dim3((w_ + 63) / 64, (h_ + 3) / 4), dim3(256), 0, nullptr, px_.p, (int)w_, (int)h_, splat_scale, F, filtered_.p);
    } else {
        hipLaunchKernelGGL(k_to_filtered, dim3(grid), dim3(256), 0, nullptr, px_.p, n, splat_scale, filtered_.p);
    }
    if (process) {
        if (!lum_.p) lum_.alloc(4);
        const lum_info init{ INT_MAX, 0, 0.0f, 0.0f };   // g_minLum = INT_MAX, the other symbols zeroed (Image.cu:157-158)
        CTL_HIP(hipMemcpy(lum_.p, &init, sizeof(init), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_luminance_info, dim3(grid), dim3(256), 0, nullptr, filtered_.p, n, (lum_info*)lum_.p);
        lum_info li; CTL_HIP(hipMemcpy(&li, lum_.p, sizeof(li), hipMemcpyDeviceToHost));
        auto unordered = [](int i) { const int v = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; std::memcpy(&f, &v, 4); return f; };
        const float maxLum = unordered(li.max_i), logAvg = expf(li.sum_log / float(w_ * h_));
        const float scale = process->key / logAvg, Lwhite = maxLum * scale;
        const float burn = std::min(1.0f, std::max(1e-8f, 1.0f - process->burn));
        const float invWp2 = 1 / (Lwhite * Lwhite * std::pow(burn, 4.0f));
        hipLaunchKernelGGL(k_reinhard, dim3(grid), dim3(256), 0, nullptr, filtered_.p, n, scale, invWp2, out_.p);
        hipLaunchKernelGGL(k_gamma_in_place, dim3(grid), dim3(256), 0, nullptr, out_.p, n);
    } else {
        hipLaunchKernelGGL(k_filtered_to_output, dim3(grid), dim3(256), 0, nullptr, filtered_.p, n, out_.p);
    }
    CTL_HIP(hipGetLastError());
    CTL_HIP(hipDeviceSynchronize());
    CTL_HIP(hipMemcpy(host_rgbcol, out_.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
}

} // namespace ctl
