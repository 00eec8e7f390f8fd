// adapter_calls.cpp — the C-ABI half of INTEGRATION.md §1's adapter class (WavefrontPathTracerAMD : Tracer<true>) and of §4's multi-GPU flow, compiled: every call the
// adapter makes into libctl_amd.so, with the argument types the reference side has in its hands, minus the inheritance from Kernel/Tracer.h (whose include chain needs
// curand_kernel.h — not in this image).  Kept next to the prose so that the prose cannot rot: tests/test_host_examples.py compiles this file with -Wall -Werror on every
// CPU run and, on a GPU box, runs it.
//     g++ -std=c++11 -Iinclude examples/adapter_calls.cpp -Lcudatracerlib_amd -lctl_amd -Wl,-rpath,$PWD/cudatracerlib_amd -o adapter_calls
#include "ctl_amd.h"

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace amd_adapter {

static void check(int rc) { if (rc != CTL_OK) throw std::runtime_error(ctl_last_error()); }   // ThrowCudaErrors convention (Defines.cpp:15-29)

// INTEGRATION.md §1, member for member
class WavefrontPathTracerAMDCalls {
    ctl_tracer* m_t = nullptr; ctl_scene* m_s = nullptr; ctl_image* m_img = nullptr;
    unsigned m_uPassesDone = 0;
public:
    bool Direct = true; int MaxPathLength = 50, RRStartDepth = 5;                        // the plugin's parameter keys (WavefrontPathTracer.h:29-39)
    WavefrontPathTracerAMDCalls() { check(ctl_tracer_create("WavefrontPathTracer", &m_t)); }
    ~WavefrontPathTracerAMDCalls() { ctl_tracer_destroy(m_t); ctl_scene_destroy(m_s); ctl_image_destroy(m_img); }
    ctl_image* image() { return m_img; }
    ctl_scene* scene() { return m_s; }
    void setTileShard(uint32_t rank, uint32_t world) { check(ctl_tracer_set_tile_shard(m_t, rank, world)); }   // §4: before Resize
    void Resize(unsigned w, unsigned h) {
        check(ctl_tracer_resize(m_t, w, h));
        ctl_image_destroy(m_img); m_img = nullptr; check(ctl_image_create(w, h, &m_img));
    }
    void InitializeScene(const ctl_scene_desc& d) {                                      // d = toDesc(s->getKernelSceneData(false)), §2
        ctl_scene_destroy(m_s); m_s = nullptr; check(ctl_scene_create_ex(&d, CTL_SCENE_FLATTEN, &m_s));
        check(ctl_tracer_initialize_scene(m_t, m_s));
    }
    void DoRender(ctl_pixel_data* reference_image_pixels) {                              // I->getPixelDataBuffer(): PixelData is 28 B on both sides
        m_uPassesDone++;                                                                 // (Tracer<true>::DoPass bumps the counter before DoRender)
        check(ctl_tracer_set_param_bool(m_t, "Direct", Direct ? 1 : 0));
        check(ctl_tracer_set_param_int(m_t, "MaxPathLength", MaxPathLength));
        check(ctl_tracer_set_param_int(m_t, "RRStartDepth", RRStartDepth));
        check(ctl_tracer_do_pass(m_t, m_img, m_uPassesDone == 1));
        if (reference_image_pixels) check(ctl_image_read_pixels(m_img, reference_image_pixels));
    }
    void Debug(int x, int y, float rgb[3]) { check(ctl_tracer_debug_pixel(m_t, m_img, (uint32_t)x, (uint32_t)y, rgb)); }
    void setDepthBuffer(float* device_data, unsigned w, unsigned h) { check(ctl_tracer_set_depth_buffer(m_t, device_data, w, h)); }   // DeviceDepthImage::m_pData
    uint64_t getRaysInLastPass() { ctl_tracer_stats st; check(ctl_tracer_get_stats(m_t, &st)); return st.rays_last_pass; }
    double getLastTimeSpentRenderingSec() { ctl_tracer_stats st; check(ctl_tracer_get_stats(m_t, &st)); return st.seconds_last_pass; }
};

// TracerBase::TraceSingleRay (Tracer.cu:74-78) for a scene the library holds: traversalRay a = (origin, tmin), b = (direction, tmax)
inline bool TraceSingleRayAMD(ctl_scene* s, const float o[3], const float d[3], float eps, ctl_hit& h) {
    ctl_ray cr = { { o[0], o[1], o[2], eps }, { d[0], d[1], d[2], FLT_MAX } };
    check(ctl_trace_single_ray(s, &cr, &h));
    return h.tri_idx >= 0;
}

// one process per GPU.  `bcast128` hands rank 0's 128 id bytes to everybody (MPI_Bcast, a file, a socket).
template <class Bcast>
void render_sharded(WavefrontPathTracerAMDCalls& tracer, const ctl_scene_desc& scene, unsigned w, unsigned h, unsigned spp, int rank, int world, int local_rank, Bcast bcast128, ctl_image* display_on_root) {
    check(ctl_set_device(local_rank));
    uint8_t id[128] = { 0 };
    if (rank == 0) check(ctl_comm_get_unique_id(id));
    bcast128(id);
    ctl_comm* comm = nullptr; check(ctl_comm_create_timeout(id, rank, world, 120000, &comm));   // collective: ncclCommInitRank, with a deadline
    tracer.setTileShard((uint32_t)rank, (uint32_t)world);
    tracer.Resize(w, h); tracer.InitializeScene(scene);
    for (unsigned i = 0; i < spp; i++) {
        tracer.DoRender(nullptr);
        if (display_on_root || rank != 0) { int rc = ctl_image_gather_to(tracer.image(), rank == 0 ? display_on_root : nullptr, comm, 0); if (rc != CTL_OK && rank == 0) check(rc); }   // progressive: after every pass
    }
    int rc = ctl_image_gather(tracer.image(), comm, 0);                                  // ONE ncclGather of the ranks' own tiles, in place on the root — once per frame
    if (rc != CTL_OK) {                                                                  // (a communicator whose gather timed out is aborted: the fallback needs a new one)
        ctl_comm_destroy(comm); comm = nullptr;
        if (rank == 0) check(ctl_comm_get_unique_id(id));
        bcast128(id);
        check(ctl_comm_create(id, rank, world, &comm));
        check(ctl_image_reduce(tracer.image(), comm, 0));                                // the whole-frame sum: same frame, 8x the bytes
    }
    ctl_comm_destroy(comm);
}

}  // namespace amd_adapter

// a host that exercises every call above on one GPU: one quad under a point light
int main() {
    using namespace amd_adapter;
    if (ctl_device_count() < 1) { std::printf("{\"skipped\": \"no HIP device\"}\n"); return 0; }
    try {
        ctl_builder* b = nullptr; check(ctl_builder_create(&b));
        const float P[12] = { -1, 0, -1, 1, 0, -1, 1, 0, 1, -1, 0, 1 }; const uint32_t I[6] = { 0, 2, 1, 0, 3, 2 };
        ctl_material m; std::memset(&m, 0, sizeof m); m.bsdf_type = CTL_BSDF_DIFFUSE; m.combined_type = CTL_EDiffuseReflection; m.node_light_index = 0xffffffffu; m.two_sided = 1;
        for (int i = 0; i < 4; i++) { m.tex[i].type = CTL_TEX_CONSTANT; m.tex[i].uv_scale[0] = m.tex[i].uv_scale[1] = 1.0f; }
        m.tex[0].value[0] = m.tex[0].value[1] = m.tex[0].value[2] = 0.5f;
        uint32_t mesh = 0, node = 0; check(ctl_builder_add_mesh(b, P, 4, I, 2, nullptr, nullptr, nullptr, &m, 1, &mesh)); check(ctl_builder_add_node(b, mesh, nullptr, &node));
        const float lp[3] = { 0, 2, 0 }, li[3] = { 10, 10, 10 }; check(ctl_builder_add_point_light(b, lp, li));
        const float eye[3] = { 0, 3, -3 }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 }; check(ctl_builder_set_camera_lookat(b, eye, at, up, 45.0f, 128, 96));
        ctl_scene_desc d; check(ctl_builder_finalize(b, &d));

        WavefrontPathTracerAMDCalls tracer; tracer.MaxPathLength = 4;
        std::vector<ctl_pixel_data> pixels(128 * 96);
        ctl_image* display = nullptr; check(ctl_image_create(128, 96, &display));
        render_sharded(tracer, d, 128, 96, 2, 0, 1, 0, [](uint8_t*) {}, display);           // world = 1: the same calls, the collective over one rank
        check(ctl_image_read_pixels(tracer.image(), pixels.data()));
        std::vector<ctl_pixel_data> shown(128 * 96); check(ctl_image_read_pixels(display, shown.data()));
        double weight = 0, lum = 0; bool same = std::memcmp(pixels.data(), shown.data(), pixels.size() * sizeof(ctl_pixel_data)) == 0;
        for (const ctl_pixel_data& p : pixels) { weight += p.weight_sum; lum += p.rgb[0] + p.rgb[1] + p.rgb[2]; }
        float rgb[3]; tracer.Debug(64, 60, rgb);
        void* depth = nullptr; check(ctl_device_malloc(128 * 96 * sizeof(float), &depth)); tracer.setDepthBuffer((float*)depth, 128, 96); tracer.DoRender(pixels.data());
        tracer.setDepthBuffer(nullptr, 0, 0); check(ctl_device_free(depth));
        ctl_hit hit; const float o[3] = { 0, 1, 0 }, dir[3] = { 0, -1, 0 }; const bool found = TraceSingleRayAMD(tracer.scene(), o, dir, 1e-4f, hit);
        // Image::AddSample from the host side (a plugin that deposits its own radiance): three samples into the display image — one lands, one is NaN and one lies outside the film (both dropped)
        const float smp[15] = { 5.5f, 7.25f, 1.0f, 2.0f, 3.0f,   6.5f, 7.5f, 1.0f, NAN, 0.0f,   -0.5f, 3.0f, 1.0f, 1.0f, 1.0f };
        check(ctl_image_clear(display)); check(ctl_image_add_samples(display, 3, smp)); check(ctl_image_read_pixels(display, shown.data()));
        double added = 0; for (const ctl_pixel_data& p : shown) added += p.weight_sum;
        const bool landed = shown[7 * 128 + 5].weight_sum == 1.0f && shown[7 * 128 + 5].rgb[1] == 2.0f;
        std::printf("{\"weight_sum_after_2_passes\": %.0f, \"luminance_sum\": %.4f, \"display_equals_frame\": %s, \"rays_last_pass\": %llu, \"hit\": %s, \"hit_dist\": %.6f, \"add_samples_kept\": %.0f, \"add_sample_landed\": %s}\n",
                    weight, lum, same ? "true" : "false", (unsigned long long)tracer.getRaysInLastPass(), found ? "true" : "false", found ? hit.dist : -1.0f, added, landed ? "true" : "false");
        ctl_image_destroy(display); ctl_builder_destroy(b);
    } catch (const std::exception& e) { std::fprintf(stderr, "adapter_calls: %s\n", e.what()); return 1; }
    return 0;
}
