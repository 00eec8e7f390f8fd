// tracer.h — host-side plugin API: the reference's TracerBase / Tracer<PROGRESSIVE> / WavefrontPathTracer surface
// (Kernel/Tracer.h:67-294, Integrators/PseudoRealtime/WavefrontPathTracer.h:24-67) over HIP.
// Same virtuals, same parameter keys, same counters; errors are std::runtime_error as in the reference.
#pragma once
#include "../../include/ctl_amd.h"
#include "device_scene.h"
#include "kernels.h"
#include "sequence_generator.h"
#include "block_sampler.h"
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>
#include <memory>
#include <stdexcept>
#include <climits>

namespace ctl {

struct hip_error : std::runtime_error { using std::runtime_error::runtime_error; };
void throw_hip(hipError_t e, const char* file, int line);
#define CTL_HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) ::ctl::throw_hip(_e, __FILE__, __LINE__); } while (0)   // ThrowCudaErrors (Defines.h:98-99)

template <typename T> struct dbuf {   // tracked device allocation (Base/CudaMemoryManager.h)
    T* p = nullptr; size_t n = 0;
    dbuf() {}
    dbuf(const dbuf&) = delete; dbuf& operator=(const dbuf&) = delete;
    ~dbuf() { free(); }
    void alloc(size_t count) { free(); n = count; if (count) CTL_HIP(hipMalloc((void**)&p, count * sizeof(T))); }
    void free() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
    void upload(const T* h, size_t count, hipStream_t s = 0) { if (count > n) alloc(count); if (count) CTL_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s)); }
};

// KernelDynamicScene in HBM (UpdateKernel, Kernel/TraceHelper.cu:182-217)
class Scene {
public:
    // flatten: also build the single-level world-space BVH (flatten.cpp) and make the intersect kernels use it
    // flat_format: flat_format of flatten.h, or -1 for the default (Q4 / $CTL_FLAT_FORMAT)
    explicit Scene(const ctl_scene_desc& d, bool flatten = false, int flat_format = -1, bool reduced_rough_transmittance = false);
    bool flattened() const { return S.flat_nodes != nullptr; }
    dev_scene S{};
    uint32_t n_nodes = 0;
    float box_min[3] = { 0, 0, 0 }, box_max[3] = { 0, 0, 0 };   // KernelDynamicScene::m_sBox
    float near_depth = 0, far_depth = 0;                        // SensorBase::m_fNearFarDepths of the scene's camera (DeviceDepthImage::NormalizeDepthD3D)
private:
    dbuf<float4> top_nodes_, bot_nodes_, leaf_tris_, inst_, inst_fwd_, flat_nodes_, flat_leaves_; dbuf<float2> normal_lut_;
    dbuf<uint4> tri_data_, node_info_;
    dbuf<ctl_material> mats_; dbuf<ctl_light> lights_; dbuf<unsigned char> anim_; dbuf<uint32_t> texels_; dbuf<ctl_mipmap> images_; dbuf<dev_mip_levels> mip_levels_; dbuf<float> mip_lut_; dbuf<float> rt_data_, rt_reduced_; dbuf<ctl_rough_transmittance> rt_;
};

// Engine/Image.h:31-91 (accumulator part)
class Image {
public:
    Image(uint32_t w, uint32_t h);
    void Clear();
    uint32_t getWidth() const { return w_; }
    uint32_t getHeight() const { return h_; }
    ctl_pixel_data* device() { return px_.p; }
    bool holds_reduced_frame() const { return reduced_; }   // set on every rank by an in-place ctl_image_reduce / ctl_image_gather of more than one rank (comm.cpp), cleared by Clear() / write()
    void mark_reduced(bool v) { reduced_ = v; }
    void read(ctl_pixel_data* host);
    void write(const ctl_pixel_data* host);
    void add_samples(uint32_t n, const float* host_samples5);   // Image::AddSample (Engine/Image.cu:22-44) for n host samples {sx, sy, r, g, b}
    void resolve_rgb(float splat_scale, float* host_rgb);
    void apply_pipeline(float splat_scale, uint32_t* host_rgbcol);                 // applyImagePipeline without filter / post-process
    void apply_pipeline_ex(float splat_scale, const ctl_reconstruction_filter* filter, const ctl_tonemap* process, uint32_t* host_rgbcol);   // image_pipeline.hip
    void write_file(float splat_scale, const char* path);                          // Image::WriteDisplayImage (Engine/Image.cpp:67-75)
private:
    bool reduced_ = false;
    uint32_t w_, h_; dbuf<ctl_pixel_data> px_; dbuf<float> rgb_; dbuf<uint32_t> out_, filtered_; dbuf<int> lum_;   // filtered_: m_filteredColorsDevice (RGBE)
};

// Kernel/TracerSettings.h:14-350 — typed parameters with interval / set constraints: bool, int and float intervals (IntervalParameterConstraint), and
// enumerations (SetParameterConstraint over the enum's values, addressed by value or by name as TracerParameter<enum> does through its string table)
struct TracerParameter {
    enum kind_t { Bool, Int, Float, Enum } kind; int value, lo, hi;
    float fvalue = 0, flo = 0, fhi = 0;
    std::vector<std::string> names;   // Enum: name of value i
};
class TracerParameterCollection {
public:
    void addBool(const std::string& key, bool v) { p_[key] = { TracerParameter::Bool, v ? 1 : 0, 0, 1 }; }
    void addInterval(const std::string& key, int v, int lo, int hi) { p_[key] = { TracerParameter::Int, v, lo, hi }; }
    void addFloatInterval(const std::string& key, float v, float lo, float hi) { TracerParameter p{ TracerParameter::Float, 0, 0, 0 }; p.fvalue = v; p.flo = lo; p.fhi = hi; p_[key] = p; }
    void addEnum(const std::string& key, int v, const std::vector<std::string>& names) { TracerParameter p{ TracerParameter::Enum, v, 0, (int)names.size() - 1 }; p.names = names; p_[key] = p; }
    int getValue(const std::string& key) const { const TracerParameter& p = find(key); if (p.kind == TracerParameter::Float) throw std::runtime_error("Parameter type mismatch for key: " + key); return p.value; }
    float getFloat(const std::string& key) const { const TracerParameter& p = find(key); if (p.kind != TracerParameter::Float) throw std::runtime_error("Parameter type mismatch for key: " + key); return p.fvalue; }
    const std::string& getEnumName(const std::string& key) const { const TracerParameter& p = find(key); if (p.kind != TracerParameter::Enum) throw std::runtime_error("Parameter type mismatch for key: " + key); return p.names[p.value]; }
    void setValue(const std::string& key, int v, TracerParameter::kind_t kind) {
        TracerParameter& p = find(key);
        if (p.kind != kind) throw std::runtime_error("Parameter type mismatch for key: " + key);
        if (v < p.lo || v > p.hi) throw std::runtime_error("Parameter value outside of its interval: " + key);
        p.value = v;
    }
    void setFloat(const std::string& key, float v) {
        TracerParameter& p = find(key);
        if (p.kind != TracerParameter::Float) throw std::runtime_error("Parameter type mismatch for key: " + key);
        if (!(v >= p.flo && v <= p.fhi)) throw std::runtime_error("Parameter value outside of its interval: " + key);
        p.fvalue = v;
    }
    void setEnumByName(const std::string& key, const std::string& name) {
        TracerParameter& p = find(key);
        if (p.kind != TracerParameter::Enum) throw std::runtime_error("Parameter type mismatch for key: " + key);
        for (size_t i = 0; i < p.names.size(); i++) if (p.names[i] == name) { p.value = (int)i; return; }
        throw std::runtime_error("Parameter value outside of its set: " + key + " = " + name);
    }
    TracerParameter::kind_t kindOf(const std::string& key) const { return find(key).kind; }
    bool has(const std::string& key) const { return p_.count(key) != 0; }
private:
    std::map<std::string, TracerParameter> p_;
    const TracerParameter& find(const std::string& key) const { auto it = p_.find(key); if (it == p_.end()) throw std::runtime_error("Unknown parameter key: " + key); return it->second; }
    TracerParameter& find(const std::string& key) { auto it = p_.find(key); if (it == p_.end()) throw std::runtime_error("Unknown parameter key: " + key); return it->second; }
};

class event_timer {   // hipEvent pairs on the tracer's stream, summed per kernel class after synchronisation
public:
    ~event_timer();
    void begin(hipStream_t s, int cls);
    void end(hipStream_t s);
    void collect(double ms_out[5]);   // adds elapsed ms per class, recycles the events
private:
    struct rec { hipEvent_t a, b; int cls; };
    std::vector<rec> used_; std::vector<hipEvent_t> free_;
    hipEvent_t get();
};

class TracerBase {
public:
    TracerBase();
    virtual ~TracerBase();
    virtual void InitializeScene(Scene* s) { m_pScene = s; }
    virtual void Resize(unsigned int _w, unsigned int _h) { w = _w; h = _h; }
    virtual void DoPass(Image* I, bool a_NewTrace) = 0;
    virtual bool isMultiPass() const = 0;
    virtual float getSplatScale() const = 0;
    unsigned int getNumPassesDone() const { return m_uPassesDone; }
    uint64_t getRaysInLastPass() const { return m_uLastNumRaysTraced; }
    float getLastTimeSpentRenderingSec() const { return m_fLastRuntime; }
    uint64_t getAccRays() const { return m_uAccNumRaysTraced; }
    float getAccTimeSpentRenderingSec() const { return m_fAccRuntime; }
    TracerParameterCollection& getParameters() { return m_sParameters; }
    // build-specific additions
    virtual void DoPasses(Image* I, bool a_NewTrace, unsigned int n) { for (unsigned int i = 0; i < n; i++) DoPass(I, a_NewTrace && i == 0); }
    // TracerBase::Debug(Image*, pixel) (Kernel/Tracer.h:119-123): UpdateKernel draws the NEXT set of sampling tables from the tracer's generator (so the following
    // DoPass uses the set after it), then DebugInternal follows one path for that pixel.  rgb (may be null): the radiance of that path (the reference discards it)
    virtual void Debug(Image* I, unsigned int x, unsigned int y, float rgb[3]) = 0;
    // IDepthTracer::setDepthBuffer (Kernel/Tracer.h:34-57): a device buffer of w x h floats that receives the normalised depth of every primary hit
    virtual void setDepthBuffer(float* device_data, unsigned int dw, unsigned int dh);   // default: refuses (the tracer is not an IDepthTracer)
    virtual void reservePasses(unsigned int n) { (void)n; }   // size the queues now for a DoPasses(n) to come (otherwise they grow inside that call)
    void setTileShard(uint32_t rank, uint32_t world) { if (world == 0 || rank >= world) throw std::runtime_error("bad tile shard"); shard_rank = rank; shard_world = world; if (w != 0xffffffffu) Resize(w, h); }
    void setSamplerTables(const float* t1, const float* t2);
    // IBlockSampler of Tracer<true> (Kernel/Tracer.h:151-152,181-190): chosen by the parameter BlockSamplerType (0 Uniform, 1 Variance,
    // 2 Difference, 3 Select), re-created by Resize; user weights as IUserPreferenceSampler::setWeight
    BlockSampler* getBlockSampler();
    void setBlockWeight(uint32_t block_x, uint32_t block_y, float w) { getBlockSampler()->set_weight(block_x, block_y, w); }
    void getKernelStats(ctl_tracer_stats& s) const;
protected:
    Scene* m_pScene = nullptr;
    TracerParameterCollection m_sParameters;
    unsigned int w = 0xffffffffu, h = 0xffffffffu;
    unsigned int m_uPassesDone = 0;
    uint64_t m_uLastNumRaysTraced = 0, m_uAccNumRaysTraced = 0;
    float m_fLastRuntime = 0, m_fAccRuntime = 0;
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream = nullptr;
    sequence_generator m_SamplingSequenceGenerator;   // IndependantSamplingSequenceGenerator
    std::vector<float> user_t1, user_t2; bool have_user_tables = false;
    uint32_t shard_rank = 0, shard_world = 1;
    std::unique_ptr<BlockSampler> block_sampler_; const unsigned char* pass_block_counts_ = nullptr; uint32_t pass_max_block_count_ = 1; uint64_t pass_paths_ = 0;
    event_timer timer; double kernel_ms[5] = { 0, 0, 0, 0, 0 };   // 0 raygen, 1 closest-hit intersect, 2 shade/finalize, 3 any-hit intersect, 4 fused closest + any-hit launches
    uint64_t intersect_rays = 0, intersect_launches = 0, shadow_rays = 0, shadow_launches = 0, fused_launches = 0, fused_shadow_rays = 0, fused_closest_rays = 0;
    bool counting = false; ctl_traversal_counts closest_counts{}, any_counts{};
public:
    void setCounting(bool on) { counting = on; }
};

template <bool PROGRESSIVE> class Tracer : public TracerBase {
public:
    void DoPass(Image* I, bool a_NewTrace) override { DoPasses(I, a_NewTrace, 1); }
    void DoPasses(Image* I, bool a_NewTrace, unsigned int n) override;
    void Debug(Image* I, unsigned int x, unsigned int y, float rgb[3]) override;
    bool isMultiPass() const override { return PROGRESSIVE; }
    float getSplatScale() const override { return PROGRESSIVE ? 1.0f / float(m_uPassesDone) : 0.0f; }
protected:
    // render `n_batch` passes together; their sampler tables are consecutive at (d_t1, d_t2) (device); asynchronous on `stream`
    virtual void DoRender(Image* I, const float* d_t1, const float* d_t2, unsigned int n_batch) = 0;
    virtual void takeRayCounts(uint64_t& path_rays, uint64_t& shadow_rays_) = 0;
    virtual void DebugInternal(Image* I, unsigned int x, unsigned int y, const float* d_t1, const float* d_t2, float rgb[3]) { (void)I; (void)x; (void)y; (void)d_t1; (void)d_t2; rgb[0] = rgb[1] = rgb[2] = 0.0f; }   // TracerBase::DebugInternal: nothing unless the integrator overrides it
    dbuf<float> d_t1, d_t2;                       // ring of sampler-table sets in HBM
    float *h_t1 = nullptr, *h_t2 = nullptr; size_t h_cap = 0;   // their pinned host staging (tables handed in by setSamplerTables, host-generated tables)
    // tables generated in HBM (k_sequence_fill): the chunk jump matrices, and a ring of the batches' pass start states
    dbuf<uint32_t> d_jumps, d_starts; sequence_generator::pass_start* h_starts = nullptr; size_t starts_cap = 0;
    std::vector<hipEvent_t> slot_done;
    virtual unsigned int passBatch() const { return 1; }
    static constexpr unsigned int kTableRing = 2;
    void ensureTableRing(unsigned int B);   // device + pinned host staging for `kTableRing` batches of B passes
public:
    ~Tracer() override { if (h_t1) (void)hipHostFree(h_t1); if (h_t2) (void)hipHostFree(h_t2); if (h_starts) (void)hipHostFree(h_starts); for (auto e : slot_done) (void)hipEventDestroy(e); }
};

// Integrators/PseudoRealtime/WavefrontPathTracer.h:24-67
class WavefrontPathTracer : public Tracer<true> {
public:
    WavefrontPathTracer();
    void Resize(unsigned int w, unsigned int h) override;
    void setDepthBuffer(float* device_data, unsigned int dw, unsigned int dh) override { depth_buffer_ = device_data; depth_w_ = dw; depth_h_ = dh; }   // WavefrontPathTracer : IDepthTracer (WavefrontPathTracer.h:24)
    void reservePasses(unsigned int n) override { const unsigned int b = std::min(passBatch(), std::max(1u, n)); growBatch(b, "reservePasses"); ensureTableRing(b); if (w != 0xffffffffu) (void)ensureStage(b); }
protected:
    void DoRender(Image* I, const float* d_t1, const float* d_t2, unsigned int n_batch) override;
    void takeRayCounts(uint64_t& path_rays, uint64_t& shadow_rays_) override;
    unsigned int passBatch() const override;
private:
    void growBatch(unsigned int b, const char* who);
    float4* ensureStage(unsigned int b);   // queues for b passes per wavefront; validated before anything changes
    wave_queues Q{};
    uint32_t capacity = 0, n_local_pixels = 0, alloc_batch_ = 1;
    float* depth_buffer_ = nullptr; unsigned int depth_w_ = 0, depth_h_ = 0;
    std::vector<std::unique_ptr<dbuf<float4>>> f4_; dbuf<float2> px_[2]; dbuf<float4> stage_; dbuf<int> hit_node_; dbuf<uint32_t> occ_[2], counts_, work_, order_, class_order_, mat_counts_; dbuf<unsigned char> mat_key_; dbuf<unsigned long long> stats_;
    int grid_blocks = 0;
    float4* new_f4(size_t n);
};

// Integrators/PathTracer.h:7-31 — the megakernel integrator behind the same plugin API (megakernel.hip)
class PathTracer : public Tracer<true> {
public:
    PathTracer();
    void Resize(unsigned int w, unsigned int h) override;
    void InitializeScene(Scene* s) override;
protected:
    void DoRender(Image* I, const float* d_t1, const float* d_t2, unsigned int n_batch) override;
    void takeRayCounts(uint64_t& path_rays, uint64_t& shadow_rays_) override;
    void DebugInternal(Image* I, unsigned int x, unsigned int y, const float* d_t1, const float* d_t2, float rgb[3]) override;
private:
    dbuf<float> debug_;
    dbuf<unsigned long long> count_; unsigned long long host_count_ = 0; uint64_t total_rays_ = 0; dbuf<float> mollifier_;
    uint32_t n_local_pixels = 0; int grid_blocks = 0;
};

int device_count();
void require_device();

// comm.cpp: the framebuffer reduce of a multi-GPU render over RCCL (ctl_comm_* in include/ctl_amd.h)
struct Comm;
void comm_unique_id(unsigned char out[128]);
Comm* comm_create(const unsigned char id[128], int rank, int world, int timeout_ms = 0);
void comm_destroy(Comm* c);
void comm_reduce_image(Comm* c, Image* src, Image* dst, int root);
void comm_gather_image(Comm* c, Image* src, Image* dst, int root);
size_t comm_gather_bytes(Comm* c, uint32_t W, uint32_t H);
size_t image_packed_tile_bytes(uint32_t W, uint32_t H, uint32_t world);
void image_pack_tiles(Image* img, uint32_t rank, uint32_t world, void* host_out);
void image_unpack_tiles(Image* img, uint32_t world, const void* host_in_all_ranks);

} // namespace ctl
