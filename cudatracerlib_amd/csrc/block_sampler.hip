// block_sampler.hip — see block_sampler.h
#include "block_sampler.h"
#include "tracer.h"
#include <algorithm>
#include <cfloat>
#include <cmath>

namespace ctl {

namespace {

__device__ __forceinline__ float lum3(float r, float g, float b) { return r * 0.212671f + g * 0.715160f + b * 0.072169f; }

// updateVarianceBuffer + PixelVarianceInfo::updateMoments (PixelVarianceBuffer.cu:10-19, PixelVarianceBuffer.h:22-45): only the pixels of
// blocks that were sampled in this pass, with the number of samples the block got
__global__ __launch_bounds__(256) void k_update_variance(pixel_variance* __restrict__ var, const ctl_pixel_data* __restrict__ image, uint32_t w, uint32_t h, uint32_t bx,
                                                         const unsigned char* __restrict__ counts, float splat_scale) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t x = i % w, y = i / w;
    const unsigned char c = counts[(y / kSamplerBlock) * bx + x / kSamplerBlock];
    if (!c) return;
    const ctl_pixel_data p = image[i]; pixel_variance v = var[i];
    const float performed = (float)c;
    const float n0 = p.rgb[0] + p.rgb_splat[0] * splat_scale, n1 = p.rgb[1] + p.rgb_splat[1] * splat_scale, n2 = p.rgb[2] + p.rgb_splat[2] * splat_scale;   // value of the pixel sum after the pass
    const float e0 = (n0 - v.prev_I[0]) / performed, e1 = (n1 - v.prev_I[1]) / performed, e2 = (n2 - v.prev_I[2]) / performed;
    v.prev_I[0] = n0; v.prev_I[1] = n1; v.prev_I[2] = n2;
    v.weight = p.weight_sum;
    if (v.iterations_done++ % 2 == 1) { v.half_buffer[0] += e0; v.half_buffer[1] += e1; v.half_buffer[2] += e2; }
    const float l = lum3(e0, e1, e2);
    v.sum_x += l; v.sum_x2 += l * l; v.num_samples_var++;
    var[i] = v;
}

// VarianceBlockSampler's and DifferenceBlockSampler's updateInfo (VarianceBlockSampler.cu:7-32, DifferenceBlockSampler.cu:7-24) for one
// sampler block per workgroup: sums over the block's pixels by wave shuffles and LDS instead of one float atomic per pixel
__global__ __launch_bounds__(256) void k_block_stats(const pixel_variance* __restrict__ var, block_stats* __restrict__ out, uint32_t w, uint32_t h, uint32_t bx) {
    __shared__ float s_f[4][4]; __shared__ uint32_t s_u[3][4];
    const uint32_t b = blockIdx.x, x0 = (b % bx) * kSamplerBlock, y0 = (b / bx) * kSamplerBlock;
    float var_i = 0, e_i = 0, e_i2 = 0, err = 0; uint32_t n_var = 0, n_e = 0, n_err = 0;
    for (uint32_t k = threadIdx.x; k < kSamplerBlock * kSamplerBlock; k += 256) {
        const uint32_t x = x0 + (k & 63u), y = y0 + (k >> 6);
        if (x >= w || y >= h) continue;
        const pixel_variance v = var[(size_t)y * w + x];
        const float N = (float)v.num_samples_var, invN = 1.0f / N;
        const float vr = (v.sum_x2 - (v.sum_x * v.sum_x) * invN) * invN, e = v.sum_x / N;   // VarAccumulator::Var / E
        if (vr >= 0 && !isnan(vr)) { var_i += vr; n_var++; }
        e_i += e; e_i2 += e * e; n_e++;
        // PixelVarianceInfo::computeError (PixelVarianceBuffer.h:57-63)
        const float I0 = v.prev_I[0] / v.weight, I1 = v.prev_I[1] / v.weight, I2 = v.prev_I[2] / v.weight;
        const float hd = (float)(v.iterations_done / 2);
        const float A0 = v.half_buffer[0] / hd, A1 = v.half_buffer[1] / hd, A2 = v.half_buffer[2] / hd;
        const float e_p = (fabsf(I0 - A0) + fabsf(I1 - A1) + fabsf(I2 - A2)) / sqrtf(I0 + I1 + I2);
        const bool skip = (I0 == 0 && I1 == 0 && I2 == 0) || isnan(I0) || isnan(I1) || isnan(I2) || isnan(A0) || isnan(A1) || isnan(A2);
        err += skip ? 0.0f : fmaxf(e_p, 1e-2f); n_err++;
    }
    for (int off = 32; off > 0; off >>= 1) {
        var_i += __shfl_down(var_i, off, 64); e_i += __shfl_down(e_i, off, 64); e_i2 += __shfl_down(e_i2, off, 64); err += __shfl_down(err, off, 64);
        n_var += __shfl_down(n_var, off, 64); n_e += __shfl_down(n_e, off, 64); n_err += __shfl_down(n_err, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_f[0][wave] = var_i; s_f[1][wave] = e_i; s_f[2][wave] = e_i2; s_f[3][wave] = err; s_u[0][wave] = n_var; s_u[1][wave] = n_e; s_u[2][wave] = n_err; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_stats r{};
        for (int k = 0; k < 4; k++) { r.var_i += s_f[0][k]; r.e_i += s_f[1][k]; r.e_i2 += s_f[2][k]; r.sum_err += s_f[3][k]; r.n_var += s_u[0][k]; r.n_e += s_u[1][k]; r.n_err += s_u[2][k]; }
        out[b] = r;
    }
}

}  // namespace

BlockSampler::BlockSampler(Type t, uint32_t w, uint32_t h) : type_(t), w_(w), h_(h), bx_((w + kSamplerBlock - 1) / kSamplerBlock), by_((h + kSamplerBlock - 1) / kSamplerBlock) {
    user_w_.assign(n_blocks(), t == Select ? 0.0f : 1.0f);
    indices_.resize(n_blocks()); for (uint32_t i = 0; i < n_blocks(); i++) indices_[i] = (int)i;
    CTL_HIP(hipMalloc((void**)&d_var_, (size_t)w * h * sizeof(pixel_variance)));
    CTL_HIP(hipMalloc((void**)&d_stats_, (size_t)n_blocks() * sizeof(block_stats)));
    CTL_HIP(hipMalloc((void**)&d_counts_, n_blocks()));
    CTL_HIP(hipMemset(d_var_, 0, (size_t)w * h * sizeof(pixel_variance)));
}
BlockSampler::~BlockSampler() { (void)hipFree(d_var_); (void)hipFree(d_stats_); (void)hipFree(d_counts_); }

void BlockSampler::set_weight(uint32_t x, uint32_t y, float w) { if (x >= bx_ || y >= by_) throw std::runtime_error("block sampler: block index out of range"); user_w_[y * bx_ + x] = w; }
float BlockSampler::get_weight(uint32_t x, uint32_t y) const { if (x >= bx_ || y >= by_) throw std::runtime_error("block sampler: block index out of range"); return user_w_[y * bx_ + x]; }
bool BlockSampler::every_block_once() const {
    if (type_ != Uniform) return false;
    for (float w : user_w_) if (w != 1.0f) return false;
    return !non_zero_;
}

void BlockSampler::start_new_rendering(hipStream_t s) {
    passes_done_ = 0;
    CTL_HIP(hipMemsetAsync(d_var_, 0, (size_t)w_ * h_ * sizeof(pixel_variance), s));   // PixelVarianceBuffer::Clear
}

void BlockSampler::mixed(std::vector<unsigned char>& c) const {   // IBlockSampler::MixedBlockIterate (IBlockSampler.h:131-153)
    const int N = (int)n_blocks();
    for (int i = 0; i < N / fraction_weighted; i++) c[indices_[i]]++;
    for (int i = (int)(passes_done_ % (unsigned)fraction_deterministic); i < N; i += fraction_deterministic) c[i]++;
}

void BlockSampler::counts(std::vector<unsigned char>& c) const {
    const uint32_t N = n_blocks();
    c.assign(N, 0);
    switch (type_) {
    case Uniform:
        if (non_zero_) { for (uint32_t k = 0; k < N; k++) { const int b = indices_[k]; if (user_w_[b] <= 0) break; c[b]++; } }
        else c.assign(N, 1);
        break;
    case Variance: case Difference:
        if (passes_done_ < 10) c.assign(N, 1); else mixed(c);
        break;
    case Select:
        for (uint32_t b = 0; b < N; b++) if (user_w_[b] != 0.0f) c[b]++;
        break;
    }
}

const unsigned char* BlockSampler::upload_counts(const std::vector<unsigned char>& c, hipStream_t s) {
    CTL_HIP(hipMemcpyAsync(d_counts_, c.data(), n_blocks(), hipMemcpyHostToDevice, s));
    CTL_HIP(hipStreamSynchronize(s));   // `c` is pageable host memory of the caller
    return d_counts_;
}

void BlockSampler::add_pass(const ctl_pixel_data* image, float splat_scale, const std::vector<unsigned char>& c, hipStream_t s) {
    last_counts_ = c;
    const uint32_t n = w_ * h_;
    hipLaunchKernelGGL(k_update_variance, dim3((n + 255) / 256), dim3(256), 0, s, d_var_, image, w_, h_, bx_, (const unsigned char*)d_counts_, splat_scale);
    auto fetch_stats = [&]() {
        hipLaunchKernelGGL(k_block_stats, dim3(n_blocks()), dim3(256), 0, s, (const pixel_variance*)d_var_, d_stats_, w_, h_, bx_);
        stats_host_.resize(n_blocks());
        CTL_HIP(hipMemcpyAsync(stats_host_.data(), d_stats_, n_blocks() * sizeof(block_stats), hipMemcpyDeviceToHost, s));
        CTL_HIP(hipStreamSynchronize(s));
    };
    auto sqr = [](float v) { return v * v; };
    std::vector<float> key(n_blocks(), 0.0f);
    if (type_ == Variance) {   // VarianceBlockSampler::AddPass (VarianceBlockSampler.cu:57-89)
        passes_done_++;
        fetch_stats();
        auto w1 = [](const block_stats& b) { return b.n_var == 0 ? 0.0f : std::sqrt(b.var_i / b.n_var); };                                        // mean std-dev of the pixel estimators
        auto w2 = [&](const block_stats& b) { if (b.n_e == 0) return 0.0f; const float E = b.e_i / b.n_e; return std::sqrt(b.e_i2 / b.n_e - sqr(E)); };   // std-dev of the pixel values in the block
        float min_block = FLT_MAX, max_block = -FLT_MAX, min_est = FLT_MAX, max_est = -FLT_MAX;
        for (auto& b : stats_host_) { const float est = w1(b), blk = w2(b); min_block = std::min(min_block, blk); max_block = std::max(max_block, blk); min_est = std::min(min_est, est); max_est = std::max(max_est, est); }
        const float lambda = 0.85f;
        for (uint32_t i = 0; i < n_blocks(); i++) {
            // (x - min) / (max - min); a frame whose blocks all agree makes the reference divide 0 by 0 — those weights are taken as 0 here
            const float a = max_est > min_est ? (w1(stats_host_[i]) - min_est) / (max_est - min_est) : 0.0f, b = max_block > min_block ? (w2(stats_host_[i]) - min_block) / (max_block - min_block) : 0.0f;
            const float wgt = lambda * a + (1 - lambda) * b;
            key[i] = (std::isnan(wgt) ? 0.0f : wgt) * sqr(user_w_[i]);
        }
        std::stable_sort(indices_.begin(), indices_.end(), [&](int i1, int i2) { return key[i1] > key[i2]; });
    } else if (type_ == Difference) {   // DifferenceBlockSampler::AddPass (DifferenceBlockSampler.cu:32-52)
        if (passes_done_++ == 0) { CTL_HIP(hipStreamSynchronize(s)); return; }
        fetch_stats();
        for (uint32_t i = 0; i < n_blocks(); i++) { const block_stats& b = stats_host_[i]; const float e = 1.0f / b.n_err * b.sum_err; key[i] = (std::isnan(e) ? 0.0f : e) * sqr(user_w_[i]); }
        std::stable_sort(indices_.begin(), indices_.end(), [&](int i1, int i2) { return key[i1] > key[i2]; });
    } else if (type_ == Uniform) {      // UniformBlockSampler::AddPass (UniformBlockSampler.h:36-40)
        if (n_blocks() >= 2) for (float w : user_w_) non_zero_ |= w != 1.0f;
        std::stable_sort(indices_.begin(), indices_.end(), [&](int i1, int i2) { return user_w_[i1] > user_w_[i2]; });
        CTL_HIP(hipStreamSynchronize(s));
    } else CTL_HIP(hipStreamSynchronize(s));
}

} // namespace ctl
