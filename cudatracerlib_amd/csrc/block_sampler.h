// block_sampler.h — the reference's block samplers and pixel-variance buffer behind `Tracer<true>` (Kernel/BlockSampler/*.h,
// Kernel/PixelVarianceBuffer.{h,cu}, Kernel/Tracer.h:209-248): every pass, a sampler decides how many samples each 64x64-pixel block of the
// film gets (0, 1 or 2) from statistics of the frame so far.  Blocks are this build's image tiles (BLOCK_SAMPLER_BlockSize = 32 *
// BLOCK_FACTOR = 64 in the reference's default build, IBlockSampler_device.h:7-22), flattened as block_y * blocks_x + block_x.
//   Uniform     every block once per pass; with user weights: the heaviest first, weight <= 0 never (UniformBlockSampler.h)
//   Variance    after 10 uniform passes: the quarter of the blocks with the largest weight 0.85 sigma_estimator + 0.15 sigma_colour
//               (both min-max normalised over the blocks) plus every second block in turn (VarianceBlockSampler.{h,cu}, MixedBlockIterate)
//   Difference  the same scheme on the mean per-pixel error |I - A| / sqrt(I) between the frame and its half buffer (DifferenceBlockSampler)
//   Select      only the blocks the user gave a weight (SelectBlockSampler.h)
// Statistics are reduced on the device (one workgroup per block), the ordering is done on the host as in the reference.
#pragma once
#include "../../include/ctl_amd.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace ctl {

constexpr uint32_t kSamplerBlock = 64;

// PixelVarianceInfo (Kernel/PixelVarianceBuffer.h:10-63)
struct pixel_variance { float prev_I[3]; float half_buffer[3]; int iterations_done; float weight; float sum_x, sum_x2; int num_samples_var; int pad_; };
// per block: VarianceBlockSampler::TmpBlockInfo + DifferenceBlockSampler::blockInfo
struct block_stats { float var_i; uint32_t n_var; float e_i, e_i2; uint32_t n_e; float sum_err; uint32_t n_err; uint32_t pad_; };

class BlockSampler {
public:
    enum Type { Uniform = 0, Variance = 1, Difference = 2, Select = 3 };
    BlockSampler(Type t, uint32_t w, uint32_t h);
    ~BlockSampler();
    BlockSampler(const BlockSampler&) = delete; BlockSampler& operator=(const BlockSampler&) = delete;
    Type type() const { return type_; }
    uint32_t blocks_x() const { return bx_; }
    uint32_t width() const { return w_; }
    uint32_t height() const { return h_; }
    uint32_t n_blocks() const { return bx_ * by_; }
    bool every_block_once() const;                            // the plain uniform case: nothing to do per pass
    void set_weight(uint32_t block_x, uint32_t block_y, float w);   // IUserPreferenceSampler::setWeight
    float get_weight(uint32_t block_x, uint32_t block_y) const;
    int fraction_deterministic = 2, fraction_weighted = 4;   // KEY_FractionDeterministic / KEY_FractionWeighted (IBlockSampler.h:157-163)

    void start_new_rendering(hipStream_t s);                  // StartNewRendering + PixelVarianceBuffer::Clear
    // IterateBlocks -> BlockSamplerBuffer::Update (BlockSamplerBuffer.h:32-49): samples per block for the NEXT pass
    void counts(std::vector<unsigned char>& per_block) const;
    const unsigned char* upload_counts(const std::vector<unsigned char>& per_block, hipStream_t s);
    // after a pass: PixelVarianceBuffer::AddPass(img, splatScale, sampler) then sampler->AddPass(img, tracer, varBuffer) (Tracer.h:233-237).
    // Synchronises the stream (the reference reads the block statistics back on the host here too).
    void add_pass(const ctl_pixel_data* image, float splat_scale, const std::vector<unsigned char>& per_block, hipStream_t s);
    const std::vector<unsigned char>& last_counts() const { return last_counts_; }
    const std::vector<block_stats>& last_stats() const { return stats_host_; }
private:
    Type type_; uint32_t w_, h_, bx_, by_;
    std::vector<float> user_w_; std::vector<int> indices_; bool non_zero_ = false;
    unsigned int passes_done_ = 0;                            // m_uPassesDone of the Variance / Difference samplers
    pixel_variance* d_var_ = nullptr; block_stats* d_stats_ = nullptr; unsigned char* d_counts_ = nullptr;
    std::vector<block_stats> stats_host_; std::vector<unsigned char> last_counts_;
    void mixed(std::vector<unsigned char>& c) const;
};

} // namespace ctl
