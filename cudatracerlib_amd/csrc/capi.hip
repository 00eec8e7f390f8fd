// capi.hip — the extern "C" surface declared in include/ctl_amd.h.  Exceptions of the C++ layer (the reference's
// error convention, Defines.cpp:15-29) are mapped to ctl_status codes + ctl_last_error().
#include "../../include/ctl_amd.h"
#include "tracer.h"
#include "kernels.h"
#include "scene_builder.h"
#include "mitsuba_loader.h"
#include "flatten.h"
#include "material_factory.h"
#include "material_textures.h"
#include "image_io.h"
#include <memory>
#include "scene_cache.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>

using namespace ctl;

struct ctl_builder { scene_builder b; };
struct ctl_scene { Scene s; ctl_scene(const ctl_scene_desc& d, bool flatten, int fmt = -1, bool reduced_rt = false) : s(d, flatten, fmt, reduced_rt) {} };
struct ctl_flat_bvh { flat_scene f; };
struct ctl_image { Image img; ctl_image(uint32_t w, uint32_t h) : img(w, h) {} };
struct ctl_tracer { std::unique_ptr<TracerBase> t; };
struct ctl_sequence_generator { sequence_generator g; };

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define CTL_TRY try {
#define CTL_CATCH                                                                    \
    } catch (const hip_error& e) { return fail(device_count() > 0 ? CTL_ERR_HIP : CTL_ERR_NO_DEVICE, e.what()); } \
    catch (const io_error& e) { return fail(CTL_ERR_IO, e.what()); }             \
    catch (const unsupported_error& e) { return fail(CTL_ERR_UNSUPPORTED, e.what()); } \
    catch (const std::bad_alloc&) { return fail(CTL_ERR_INVALID, "out of host memory"); } \
    catch (const std::exception& e) { return fail(CTL_ERR_INVALID, e.what()); }      \
    return CTL_OK;
#define CTL_REQUIRE(c, msg) if (!(c)) return fail(CTL_ERR_INVALID, msg)

extern "C" {

const char* ctl_last_error(void) { return g_err.c_str(); }
const char* ctl_version(void) { return "cudatracerlib_amd 0.1 (gfx950)"; }
int ctl_device_count(void) { return device_count(); }

// ---- builder
int ctl_builder_create(ctl_builder** out) { CTL_REQUIRE(out, "null out"); CTL_TRY *out = new ctl_builder(); CTL_CATCH }
void ctl_builder_destroy(ctl_builder* b) { delete b; }
int ctl_builder_add_mesh(ctl_builder* b, const float* positions, uint32_t n_vert, const uint32_t* indices, uint32_t n_tri, const float* normals, const float* uvs,
                         const uint8_t* tri_material, const ctl_material* materials, uint32_t n_mat, uint32_t* mesh_index_out) {
    CTL_REQUIRE(b, "null builder");
    CTL_TRY uint32_t i = b->b.add_mesh(positions, n_vert, indices, n_tri, normals, uvs, tri_material, materials, n_mat); if (mesh_index_out) *mesh_index_out = i; CTL_CATCH
}
int ctl_builder_set_bvh_mode(ctl_builder* b, uint32_t mode) { CTL_REQUIRE(b && mode <= CTL_BVH_BINNED, "null builder or unknown mode"); b->b.bvh_mode = mode; return CTL_OK; }
int ctl_builder_add_node(ctl_builder* b, uint32_t mesh_index, const ctl_float4x4* to_world, uint32_t* node_index_out) {
    CTL_REQUIRE(b, "null builder");
    CTL_TRY uint32_t i = b->b.add_node(mesh_index, to_world); if (node_index_out) *node_index_out = i; CTL_CATCH
}
int ctl_builder_add_area_light(ctl_builder* b, uint32_t node_index, uint32_t local_material, const float radiance[3]) {
    CTL_REQUIRE(b && radiance, "null argument");
    CTL_TRY b->b.add_area_light(node_index, local_material, radiance); CTL_CATCH
}
int ctl_builder_add_area_light_ex(ctl_builder* b, uint32_t node_index, uint32_t local_material, const float radiance[3], const ctl_texture* rad_texture, int32_t orthogonal) {
    CTL_REQUIRE(b && radiance, "null argument");
    CTL_TRY b->b.add_area_light(node_index, local_material, radiance, rad_texture, orthogonal != 0); CTL_CATCH
}
int ctl_builder_add_point_light(ctl_builder* b, const float position[3], const float intensity[3]) {
    CTL_REQUIRE(b && position && intensity, "null argument");
    CTL_TRY b->b.add_point_light(position, intensity); CTL_CATCH
}
int ctl_builder_add_spot_light(ctl_builder* b, const float position[3], const float target[3], const float intensity[3], float cutoff_angle_degrees, float beam_width_degrees) {
    CTL_REQUIRE(b && position && target && intensity, "null argument");
    CTL_TRY b->b.add_spot_light(position, target, intensity, cutoff_angle_degrees, beam_width_degrees); CTL_CATCH
}
int ctl_builder_add_distant_light(ctl_builder* b, const float direction[3], const float irradiance[3], float scene_radius) {
    CTL_REQUIRE(b && direction && irradiance, "null argument");
    CTL_TRY b->b.add_distant_light(direction, irradiance, scene_radius); CTL_CATCH
}
int ctl_builder_add_image(ctl_builder* b, const uint32_t* texels, uint32_t width, uint32_t height, uint32_t texel_type, uint32_t wrap_mode, uint32_t filter_mode, uint32_t* image_index_out) {
    CTL_REQUIRE(b && texels, "null argument");
    CTL_TRY uint32_t i = b->b.add_image(texels, width, height, texel_type, wrap_mode, filter_mode); if (image_index_out) *image_index_out = i; CTL_CATCH
}
int ctl_builder_set_environment_map(ctl_builder* b, uint32_t image_index, const float scale[3], const ctl_float4x4* to_world) {
    CTL_REQUIRE(b && scale, "null argument");
    CTL_TRY b->b.set_environment_map(image_index, scale, to_world); CTL_CATCH
}
int ctl_builder_add_material(ctl_builder* b, const ctl_material* material, uint32_t* index_out) {
    CTL_REQUIRE(b && material, "null argument");
    CTL_TRY uint32_t i = b->b.add_aux_material(*material); if (index_out) *index_out = i; CTL_CATCH
}
int ctl_builder_set_rough_transmittance(ctl_builder* b, uint32_t slot, const ctl_rough_transmittance* table) {
    CTL_REQUIRE(b && table, "null argument");
    CTL_TRY b->b.set_rough_transmittance(slot, *table); CTL_CATCH
}
int ctl_builder_load_rough_transmittance(ctl_builder* b, uint32_t slot, const char* dat_path) {
    CTL_REQUIRE(b && dat_path, "null argument");
    CTL_TRY b->b.load_rough_transmittance(slot, dat_path); CTL_CATCH
}
int ctl_builder_set_camera_lookat(ctl_builder* b, const float pos[3], const float target[3], const float up[3], float fov_degrees, uint32_t width, uint32_t height) {
    CTL_REQUIRE(b && pos && target && up && width && height, "bad argument");
    CTL_TRY b->b.set_camera_lookat(pos, target, up, fov_degrees, width, height); CTL_CATCH
}
int ctl_builder_set_camera(ctl_builder* b, const ctl_sensor* sensor) { CTL_REQUIRE(b && sensor, "null argument"); CTL_TRY b->b.set_camera(*sensor); CTL_CATCH }
int ctl_builder_finalize(ctl_builder* b, ctl_scene_desc* out) { CTL_REQUIRE(b && out, "null argument"); CTL_TRY b->b.finalize(*out); CTL_CATCH }

int ctl_material_update(ctl_material* m) {   // (a weight from an IMAGE texture's average is final after ctl_builder_finalize, material_textures.h)
    CTL_REQUIRE(m, "null argument"); CTL_REQUIRE(material_update(*m), "unknown bsdf_type"); material_update_textures(*m, image_set()); return CTL_OK;
}
float ctl_fresnel_diffuse_reflectance(float eta) { return fresnel_diffuse_reflectance(eta); }
// ---- the shared transcendental functions (ctl_fmath.h), evaluated on the host or by a kernel: the parity tests hold the two bit-identical
__global__ void k_shared_math(int which, int n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    switch (which) {
    case 0: out[i] = fm::sin(x[i]); break; case 1: out[i] = fm::cos(x[i]); break; case 2: out[i] = fm::tan(x[i]); break; case 3: out[i] = fm::acos(x[i]); break;
    case 4: out[i] = fm::atan(x[i]); break; case 5: out[i] = fm::atan2(x[i], y[i]); break; case 6: out[i] = fm::exp(x[i]); break; case 7: out[i] = fm::log(x[i]); break;
    case 8: out[i] = fm::log2(x[i]); break; default: out[i] = fm::pow(x[i], y[i]); break;
    }
}
int ctl_shared_math_eval(int32_t which, uint32_t n, const float* x, const float* y, float* out, int32_t on_device) {
    CTL_REQUIRE(x && y && out && which >= 0 && which <= 9, "null argument or unknown function");
    CTL_TRY
        if (!on_device) {
            for (uint32_t i = 0; i < n; i++) {
                switch (which) {
                case 0: out[i] = fm::sin(x[i]); break; case 1: out[i] = fm::cos(x[i]); break; case 2: out[i] = fm::tan(x[i]); break; case 3: out[i] = fm::acos(x[i]); break;
                case 4: out[i] = fm::atan(x[i]); break; case 5: out[i] = fm::atan2(x[i], y[i]); break; case 6: out[i] = fm::exp(x[i]); break; case 7: out[i] = fm::log(x[i]); break;
                case 8: out[i] = fm::log2(x[i]); break; default: out[i] = fm::pow(x[i], y[i]); break;
                }
            }
        } else {
            require_device();
            dbuf<float> dx, dy, dout; dx.alloc(n); dy.alloc(n); dout.alloc(n);
            CTL_HIP(hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice)); CTL_HIP(hipMemcpy(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_shared_math, dim3((n + 255) / 256), dim3(256), 0, 0, (int)which, (int)n, (const float*)dx.p, (const float*)dy.p, dout.p);
            CTL_HIP(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
    CTL_CATCH
}
int ctl_traversal_stack_histogram(uint64_t* out, uint32_t n_bins, int reset) {
    CTL_REQUIRE(out && n_bins >= 1, "null argument");
    CTL_TRY
        require_device();
        unsigned long long h[kStackSize];
        read_stack_histogram(h, reset != 0);
        for (uint32_t i = 0; i < n_bins; i++) out[i] = 0;
        for (int i = 0; i < kStackSize; i++) out[(uint32_t)i < n_bins ? (uint32_t)i : n_bins - 1] += h[i];
    CTL_CATCH
}
// ---- scene
int ctl_scene_create(const ctl_scene_desc* desc, ctl_scene** out) { CTL_REQUIRE(desc && out, "null argument"); CTL_TRY *out = new ctl_scene(*desc, false); CTL_CATCH }
int ctl_scene_create_ex(const ctl_scene_desc* desc, uint32_t flags, ctl_scene** out) { CTL_REQUIRE(desc && out, "null argument"); CTL_TRY
    const uint32_t fmt = (flags >> 8) & 7u;   // 0: default, else CTL_FLAT_* + 1
    *out = new ctl_scene(*desc, (flags & CTL_SCENE_FLATTEN) != 0, (int)fmt - 1, (flags & CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE) != 0);
CTL_CATCH }
void ctl_scene_destroy(ctl_scene* s) { delete s; }
int ctl_set_cache_dir(const char* dir) { CTL_TRY set_cache_dir(dir); CTL_CATCH }
int ctl_flatten_probe(const ctl_scene_desc* desc, uint32_t format, uint64_t* out4) {
    CTL_REQUIRE(desc && out4 && format <= CTL_FLAT_Q8, "null argument or bad format");
    CTL_TRY
        flat_scene F;
        if (!flatten_scene(*desc, F, (size_t)1 << 30, (int)format)) throw std::runtime_error("ctl_flatten_probe: nothing to flatten");
        content_hash H; H.add_vector(F.nodes); H.add_vector(F.child_links); H.add_vector(F.nodes_f4); H.add_vector(F.nodes_f2); H.add_vector(F.leaves); if (!F.nodes_q8.empty()) H.add_vector(F.nodes_q8);
        out4[0] = F.nodes.size() + F.nodes_f4.size() + F.nodes_f2.size() + F.nodes_q8.size(); out4[1] = F.leaves.size(); out4[2] = (uint64_t)F.max_depth;
        out4[3] = std::strtoull(H.hex().substr(16).c_str(), nullptr, 16);
    CTL_CATCH
}
int ctl_flat_bvh_build(const ctl_scene_desc* desc, uint32_t format, ctl_flat_bvh** out) {
    CTL_REQUIRE(desc && out && format <= CTL_FLAT_Q8, "null argument or bad format");
    CTL_TRY
        std::unique_ptr<ctl_flat_bvh> h(new ctl_flat_bvh());
        if (!flatten_scene(*desc, h->f, (size_t)1 << 30, (int)format)) throw std::runtime_error("ctl_flat_bvh_build: nothing to flatten");
        *out = h.release();
    CTL_CATCH
}
int ctl_flat_bvh_arrays(const ctl_flat_bvh* h, ctl_flat_bvh_desc* out) {
    CTL_REQUIRE(h && out, "null argument");
    const flat_scene& F = h->f;
    out->format = (uint32_t)F.format; out->max_depth = (uint32_t)F.max_depth;
    if (F.format == kFlatQ4) { out->nodes = F.nodes.data(); out->n_nodes = F.nodes.size(); out->node_bytes = 64; }
    else if (F.format == kFlatQ8) { out->nodes = F.nodes_q8.data(); out->n_nodes = F.nodes_q8.size(); out->node_bytes = 128; }
    else if (F.format == kFlatF4) { out->nodes = F.nodes_f4.data(); out->n_nodes = F.nodes_f4.size(); out->node_bytes = 128; }
    else { out->nodes = F.nodes_f2.data(); out->n_nodes = F.nodes_f2.size(); out->node_bytes = 64; }
    out->leaves = F.leaves.data(); out->n_leaves = F.leaves.size();
    out->child_links = (F.format == kFlatQ4 || F.format == kFlatQ8) ? F.child_links.data() : nullptr; out->compact = ((F.format == kFlatQ4 && F.compact_links) || F.format == kFlatQ8) ? 1u : 0u;
    out->root_slab = F.root_slab ? 1u : 0u; out->n_slab_nodes = F.slab_nodes;
    return CTL_OK;
}
void ctl_flat_bvh_destroy(ctl_flat_bvh* h) { delete h; }
int ctl_parse_mitsuba_scene(ctl_builder* b, const char* xml_path, int32_t* width_inout, int32_t* height_inout) {
    CTL_REQUIRE(b && xml_path, "null argument");
    CTL_TRY parse_mitsuba_scene(b->b, xml_path, width_inout, height_inout); CTL_CATCH
}
int ctl_decode_image_file(const char* path, uint32_t* width, uint32_t* height, int32_t* is_float, float* rgb, uint8_t* rgba8) {
    CTL_REQUIRE(path && width && height && is_float, "null argument");
    CTL_TRY
        const decoded_image img = load_image_file(path);
        *width = img.width; *height = img.height; *is_float = img.is_float ? 1 : 0;
        if (img.is_float && rgb) std::memcpy(rgb, img.rgb.data(), img.rgb.size() * sizeof(float));
        if (!img.is_float && rgba8) std::memcpy(rgba8, img.rgba8.data(), img.rgba8.size());
    CTL_CATCH
}

// ---- sampler
int ctl_sequence_generator_create(ctl_sequence_generator** out) { CTL_REQUIRE(out, "null out"); CTL_TRY *out = new ctl_sequence_generator(); CTL_CATCH }
void ctl_sequence_generator_destroy(ctl_sequence_generator* g) { delete g; }
int ctl_sequence_generator_compute(ctl_sequence_generator* g, float* tables_1d, float* tables_2d) { CTL_REQUIRE(g && tables_1d && tables_2d, "null argument"); CTL_TRY g->g.compute(tables_1d, tables_2d); CTL_CATCH }

int ctl_sequence_generator_compute_many(ctl_sequence_generator* g, uint32_t n_passes, float* tables_1d, float* tables_2d, uint32_t threads) {
    CTL_REQUIRE(g && tables_1d && tables_2d, "null argument");
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    CTL_TRY g->g.compute_many(tables_1d, tables_2d, n_passes, n1, 2 * n1, threads); CTL_CATCH
}
int ctl_sequence_generator_compute_many_device(ctl_sequence_generator* g, uint32_t n_passes, float* tables_1d, float* tables_2d) {
    CTL_REQUIRE(g && tables_1d && tables_2d, "null argument");
    const size_t n1 = (size_t)CTL_SAMPLER_NUM_SEQUENCES * CTL_SAMPLER_SEQUENCE_LENGTH;
    CTL_TRY
        require_device();
        if (n_passes == 0) return CTL_OK;
        std::vector<sequence_generator::pass_start> starts(n_passes);
        g->g.take_pass_starts(n_passes, starts.data());
        const std::vector<uint32_t>& J = sequence_generator::chunk_jump_matrices();
        dbuf<uint32_t> dj, ds; dbuf<float> d1, d2;
        dj.alloc(J.size()); ds.alloc(starts.size() * sizeof(sequence_generator::pass_start) / 4); d1.alloc(n1 * n_passes); d2.alloc(2 * n1 * n_passes);
        CTL_HIP(hipMemcpy(dj.p, J.data(), J.size() * 4, hipMemcpyHostToDevice));
        CTL_HIP(hipMemcpy(ds.p, starts.data(), starts.size() * sizeof(sequence_generator::pass_start), hipMemcpyHostToDevice));
        launch_sequence_fill(nullptr, dj.p, ds.p, n_passes, d1.p, d2.p);
        CTL_HIP(hipDeviceSynchronize());
        CTL_HIP(hipMemcpy(tables_1d, d1.p, n1 * n_passes * 4, hipMemcpyDeviceToHost));
        CTL_HIP(hipMemcpy(tables_2d, d2.p, 2 * n1 * n_passes * 4, hipMemcpyDeviceToHost));
    CTL_CATCH
}

// ---- image
int ctl_image_create(uint32_t width, uint32_t height, ctl_image** out) { CTL_REQUIRE(out && width && height, "bad argument"); CTL_TRY *out = new ctl_image(width, height); CTL_CATCH }
void ctl_image_destroy(ctl_image* img) { delete img; }
int ctl_image_clear(ctl_image* img) { CTL_REQUIRE(img, "null image"); CTL_TRY img->img.Clear(); CTL_CATCH }
int ctl_image_read_pixels(ctl_image* img, ctl_pixel_data* host_out) { CTL_REQUIRE(img && host_out, "null argument"); CTL_TRY img->img.read(host_out); CTL_CATCH }
int ctl_image_write_pixels(ctl_image* img, const ctl_pixel_data* host_in) { CTL_REQUIRE(img && host_in, "null argument"); CTL_TRY img->img.write(host_in); CTL_CATCH }
int ctl_image_add_samples(ctl_image* img, uint32_t n, const float* host_samples5) { CTL_REQUIRE(img && (host_samples5 || !n), "null argument"); CTL_TRY img->img.add_samples(n, host_samples5); CTL_CATCH }
// ---- multi-GPU: the one collective (comm.cpp)
struct ctl_comm { Comm* c; };
int ctl_comm_get_unique_id(uint8_t out128[128]) { CTL_REQUIRE(out128, "null argument"); CTL_TRY comm_unique_id(out128); CTL_CATCH }
int ctl_comm_create(const uint8_t id128[128], int32_t rank, int32_t world, ctl_comm** out) {
    CTL_REQUIRE(id128 && out, "null argument");
    CTL_TRY *out = new ctl_comm{ comm_create(id128, rank, world) }; CTL_CATCH
}
void ctl_comm_destroy(ctl_comm* c) { if (c) { comm_destroy(c->c); delete c; } }
int ctl_comm_create_timeout(const uint8_t id128[128], int32_t rank, int32_t world, int32_t timeout_ms, ctl_comm** out) {
    CTL_REQUIRE(id128 && out, "null argument");
    CTL_TRY *out = new ctl_comm{ comm_create(id128, rank, world, timeout_ms) }; CTL_CATCH
}
int ctl_image_reduce(ctl_image* img, ctl_comm* comm, int32_t root) { CTL_REQUIRE(img && comm, "null argument"); CTL_TRY comm_reduce_image(comm->c, &img->img, &img->img, root); CTL_CATCH }
int ctl_image_reduce_to(ctl_image* src, ctl_image* dst, ctl_comm* comm, int32_t root) {
    CTL_REQUIRE(src && comm, "null argument");
    CTL_REQUIRE(dst != src, "ctl_image_reduce_to: source and destination must differ (ctl_image_reduce is the in-place form)");
    CTL_TRY comm_reduce_image(comm->c, &src->img, dst ? &dst->img : nullptr, root); CTL_CATCH
}
int ctl_image_gather(ctl_image* img, ctl_comm* comm, int32_t root) { CTL_REQUIRE(img && comm, "null argument"); CTL_TRY comm_gather_image(comm->c, &img->img, &img->img, root); CTL_CATCH }
int ctl_image_gather_to(ctl_image* src, ctl_image* dst, ctl_comm* comm, int32_t root) {
    CTL_REQUIRE(src && comm, "null argument");
    CTL_REQUIRE(dst != src, "ctl_image_gather_to: source and destination must differ (ctl_image_gather is the in-place form)");
    CTL_TRY comm_gather_image(comm->c, &src->img, dst ? &dst->img : nullptr, root); CTL_CATCH
}
int ctl_image_packed_tile_bytes(uint32_t width, uint32_t height, uint32_t world, uint64_t* out_bytes) {
    CTL_REQUIRE(out_bytes && width && height, "bad argument"); CTL_TRY *out_bytes = image_packed_tile_bytes(width, height, world); CTL_CATCH
}
int ctl_image_pack_tiles(ctl_image* img, uint32_t rank, uint32_t world, void* host_out) { CTL_REQUIRE(img && host_out, "null argument"); CTL_TRY image_pack_tiles(&img->img, rank, world, host_out); CTL_CATCH }
int ctl_image_unpack_tiles(ctl_image* img, uint32_t world, const void* host_in_all_ranks) { CTL_REQUIRE(img && host_in_all_ranks, "null argument"); CTL_TRY image_unpack_tiles(&img->img, world, host_in_all_ranks); CTL_CATCH }
void* ctl_image_device_ptr(ctl_image* img) { return img ? (void*)img->img.device() : nullptr; }
int ctl_image_resolve_rgb(ctl_image* img, float splat_scale, float* host_rgb_out) { CTL_REQUIRE(img && host_rgb_out, "null argument"); CTL_TRY img->img.resolve_rgb(splat_scale, host_rgb_out); CTL_CATCH }

int ctl_image_apply_pipeline_ex(ctl_image* img, float splat_scale, const ctl_reconstruction_filter* filter, const ctl_tonemap* process, uint32_t* host_rgbcol_out) {
    CTL_REQUIRE(img && host_rgbcol_out, "null argument"); CTL_TRY img->img.apply_pipeline_ex(splat_scale, filter, process, host_rgbcol_out); CTL_CATCH
}
int ctl_image_apply_pipeline(ctl_image* img, float splat_scale, uint32_t* host_rgbcol_out) { CTL_REQUIRE(img && host_rgbcol_out, "null argument"); CTL_TRY img->img.apply_pipeline(splat_scale, host_rgbcol_out); CTL_CATCH }
int ctl_image_write_file(ctl_image* img, float splat_scale, const char* path) { CTL_REQUIRE(img && path, "null argument"); CTL_TRY img->img.write_file(splat_scale, path); CTL_CATCH }

// ---- tracer
int ctl_tracer_create(const char* plugin, ctl_tracer** out) {
    CTL_REQUIRE(plugin && out, "null argument");
    const bool wave = !std::strcmp(plugin, "WavefrontPathTracer") || !std::strcmp(plugin, "PT_Wave"), mega = !std::strcmp(plugin, "PathTracer") || !std::strcmp(plugin, "PT");   // main.cpp:91-96
    if (!wave && !mega) return fail(CTL_ERR_UNSUPPORTED, std::string("unknown tracer plugin: ") + plugin);
    CTL_TRY ctl_tracer* t = new ctl_tracer(); try { if (wave) t->t.reset(new WavefrontPathTracer()); else t->t.reset(new PathTracer()); } catch (...) { delete t; throw; } *out = t; CTL_CATCH
}
void ctl_tracer_destroy(ctl_tracer* t) { delete t; }
int ctl_tracer_set_param_bool(ctl_tracer* t, const char* key, int value) { CTL_REQUIRE(t && key, "null argument"); CTL_TRY t->t->getParameters().setValue(key, value ? 1 : 0, TracerParameter::Bool); CTL_CATCH }
int ctl_tracer_set_param_int(ctl_tracer* t, const char* key, int value) { CTL_REQUIRE(t && key, "null argument"); CTL_TRY
    auto& P = t->t->getParameters();
    P.setValue(key, value, P.kindOf(key) == TracerParameter::Enum ? TracerParameter::Enum : TracerParameter::Int);   // an enum may be set by its index
CTL_CATCH }
int ctl_tracer_set_param_float(ctl_tracer* t, const char* key, float value) { CTL_REQUIRE(t && key, "null argument"); CTL_TRY t->t->getParameters().setFloat(key, value); CTL_CATCH }
int ctl_tracer_get_param_float(ctl_tracer* t, const char* key, float* value_out) { CTL_REQUIRE(t && key && value_out, "null argument"); CTL_TRY *value_out = t->t->getParameters().getFloat(key); CTL_CATCH }
int ctl_tracer_set_param_enum(ctl_tracer* t, const char* key, const char* value_name) { CTL_REQUIRE(t && key && value_name, "null argument"); CTL_TRY t->t->getParameters().setEnumByName(key, value_name); CTL_CATCH }
int ctl_tracer_get_param_int(ctl_tracer* t, const char* key, int* value_out) { CTL_REQUIRE(t && key && value_out, "null argument"); CTL_TRY *value_out = t->t->getParameters().getValue(key); CTL_CATCH }
int ctl_tracer_resize(ctl_tracer* t, uint32_t width, uint32_t height) { CTL_REQUIRE(t && width && height, "bad argument"); CTL_TRY t->t->Resize(width, height); CTL_CATCH }
int ctl_tracer_initialize_scene(ctl_tracer* t, ctl_scene* s) { CTL_REQUIRE(t && s, "null argument"); CTL_TRY t->t->InitializeScene(&s->s); CTL_CATCH }
int ctl_tracer_set_tile_shard(ctl_tracer* t, uint32_t rank, uint32_t world) { CTL_REQUIRE(t, "null tracer"); CTL_TRY t->t->setTileShard(rank, world); CTL_CATCH }
int ctl_tracer_set_sampler_tables(ctl_tracer* t, const float* tables_1d, const float* tables_2d) { CTL_REQUIRE(t && tables_1d && tables_2d, "null argument"); CTL_TRY t->t->setSamplerTables(tables_1d, tables_2d); CTL_CATCH }
int ctl_tracer_do_pass(ctl_tracer* t, ctl_image* img, int new_trace) { CTL_REQUIRE(t && img, "null argument"); CTL_TRY t->t->DoPass(&img->img, new_trace != 0); CTL_CATCH }
int ctl_tracer_do_passes(ctl_tracer* t, ctl_image* img, int new_trace, uint32_t n_passes) { CTL_REQUIRE(t && img, "null argument"); CTL_TRY t->t->DoPasses(&img->img, new_trace != 0, n_passes); CTL_CATCH }
int ctl_tracer_reserve_passes(ctl_tracer* t, uint32_t n) { CTL_REQUIRE(t, "null tracer"); CTL_TRY t->t->reservePasses(n); CTL_CATCH }
int ctl_tracer_set_block_weight(ctl_tracer* t, uint32_t bx, uint32_t by, float w) { CTL_REQUIRE(t, "null tracer"); CTL_TRY t->t->setBlockWeight(bx, by, w); CTL_CATCH }
int ctl_tracer_get_block_counts(ctl_tracer* t, uint8_t* out, uint32_t n) {
    CTL_REQUIRE(t && out, "null argument");
    CTL_TRY
        BlockSampler* bs = t->t->getBlockSampler();
        if (n != bs->n_blocks()) throw std::runtime_error("ctl_tracer_get_block_counts: the film has " + std::to_string(bs->n_blocks()) + " blocks");
        const auto& c = bs->last_counts();
        for (uint32_t i = 0; i < n; i++) out[i] = c.size() == n ? c[i] : 1;
    CTL_CATCH
}
int ctl_tracer_get_stats(ctl_tracer* t, ctl_tracer_stats* out) { CTL_REQUIRE(t && out, "null argument"); CTL_TRY t->t->getKernelStats(*out); CTL_CATCH }
int ctl_tracer_debug_pixel(ctl_tracer* t, ctl_image* img, uint32_t x, uint32_t y, float* rgb_out) { CTL_REQUIRE(t && img, "null argument"); CTL_TRY t->t->Debug(&img->img, x, y, rgb_out); CTL_CATCH }
int ctl_tracer_set_depth_buffer(ctl_tracer* t, float* device_depth, uint32_t width, uint32_t height) { CTL_REQUIRE(t, "null tracer"); CTL_REQUIRE(device_depth || (width == 0 && height == 0), "null buffer"); CTL_TRY t->t->setDepthBuffer(device_depth, width, height); CTL_CATCH }
int ctl_tracer_set_counting(ctl_tracer* t, int on) { CTL_REQUIRE(t, "null tracer"); CTL_TRY t->t->setCounting(on != 0); CTL_CATCH }

// ---- intersect (row a7 on its own)
static void run_intersect(Scene& sc, const float4* d_ro, const float4* d_rd, uint32_t n, float4* d_hit, int* d_node, uint32_t* d_occ, int any_hit, unsigned long long* d_counts, float* ms_out) {
    dbuf<uint32_t> ctl; ctl.alloc(2);
    uint32_t h[2] = { n, 0 };
    CTL_HIP(hipMemcpy(ctl.p, h, sizeof(h), hipMemcpyHostToDevice));
    int dev = 0; hipDeviceProp_t prop; CTL_HIP(hipGetDevice(&dev)); CTL_HIP(hipGetDeviceProperties(&prop, dev));
    launch_ctx lc{ nullptr, prop.multiProcessorCount * 8 };
    hipEvent_t a, b; CTL_HIP(hipEventCreate(&a)); CTL_HIP(hipEventCreate(&b));
    CTL_HIP(hipEventRecord(a, nullptr));
    if (d_counts) launch_intersect_count(lc, sc.S, d_ro, d_rd, ctl.p, ctl.p + 1, d_hit, d_node, nullptr, any_hit, d_counts);
    else if (any_hit) launch_intersect_any(lc, sc.S, d_ro, d_rd, ctl.p, ctl.p + 1, d_occ, d_hit, d_node);
    else launch_intersect_closest(lc, sc.S, d_ro, d_rd, ctl.p, ctl.p + 1, d_hit, d_node);
    CTL_HIP(hipEventRecord(b, nullptr));
    CTL_HIP(hipEventSynchronize(b));
    CTL_HIP(hipGetLastError());
    float ms = 0; CTL_HIP(hipEventElapsedTime(&ms, a, b));
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}
static void intersect_host(Scene& sc, const ctl_ray* rays, uint32_t n, ctl_hit* hits, int any_hit, ctl_traversal_counts* counts) {
    std::vector<float4> ro(n), rd(n);
    for (uint32_t i = 0; i < n; i++) { ro[i] = make_float4(rays[i].a[0], rays[i].a[1], rays[i].a[2], rays[i].a[3]); rd[i] = make_float4(rays[i].b[0], rays[i].b[1], rays[i].b[2], rays[i].b[3]); }
    dbuf<float4> d_ro, d_rd, d_hit; dbuf<int> d_node; dbuf<unsigned long long> d_cnt;
    d_ro.upload(ro.data(), n); d_rd.upload(rd.data(), n); d_hit.alloc(n); d_node.alloc(n);
    if (counts) { d_cnt.alloc(5); CTL_HIP(hipMemset(d_cnt.p, 0, 40)); }
    CTL_HIP(hipDeviceSynchronize());
    run_intersect(sc, d_ro.p, d_rd.p, n, d_hit.p, d_node.p, nullptr, any_hit, counts ? d_cnt.p : nullptr, nullptr);
    if (hits) {
        std::vector<float4> hh(n); std::vector<int> hn(n);
        CTL_HIP(hipMemcpy(hh.data(), d_hit.p, (size_t)n * 16, hipMemcpyDeviceToHost)); CTL_HIP(hipMemcpy(hn.data(), d_node.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) { hits[i].dist = hh[i].x; hits[i].u = hh[i].y; hits[i].v = hh[i].z; hits[i].tri_idx = __builtin_bit_cast(int, hh[i].w); hits[i].node_idx = hn[i]; }
    }
    if (counts) { unsigned long long c[5]; CTL_HIP(hipMemcpy(c, d_cnt.p, 40, hipMemcpyDeviceToHost)); *counts = ctl_traversal_counts{ c[0], c[1], c[2], c[3], c[4] }; }
}
int ctl_trace_single_ray(ctl_scene* s, const ctl_ray* ray, ctl_hit* hit_out) { CTL_REQUIRE(s && ray && hit_out, "null argument"); return ctl_intersect(s, ray, 1, hit_out, 0); }
int ctl_intersect(ctl_scene* s, const ctl_ray* rays, uint32_t n, ctl_hit* hits, int any_hit) {
    CTL_REQUIRE(s && (rays || !n) && (hits || !n), "null argument");
    if (!n) return CTL_OK;
    CTL_TRY intersect_host(s->s, rays, n, hits, any_hit, nullptr); CTL_CATCH
}
int ctl_intersect_count(ctl_scene* s, const ctl_ray* rays, uint32_t n, int any_hit, ctl_traversal_counts* out) {
    CTL_REQUIRE(s && rays && out, "null argument");
    CTL_TRY intersect_host(s->s, rays, n, nullptr, any_hit, out); CTL_CATCH
}
int ctl_intersect_device(ctl_scene* s, const void* d_ray_o, const void* d_ray_d, uint32_t n, void* d_hit4, void* d_hit_node, int any_hit, float* ms_out) {
    CTL_REQUIRE(s && d_ray_o && d_ray_d && d_hit4 && d_hit_node, "null argument");
    CTL_TRY run_intersect(s->s, (const float4*)d_ray_o, (const float4*)d_ray_d, n, (float4*)d_hit4, (int*)d_hit_node, nullptr, any_hit, nullptr, ms_out); CTL_CATCH
}

// ---- device memory helpers
int ctl_device_malloc(size_t bytes, void** out) { CTL_REQUIRE(out, "null out"); CTL_TRY require_device(); CTL_HIP(hipMalloc(out, bytes)); CTL_CATCH }
int ctl_device_free(void* p) { CTL_TRY CTL_HIP(hipFree(p)); CTL_CATCH }
int ctl_memcpy_h2d(void* dst, const void* src, size_t bytes) { CTL_TRY CTL_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); CTL_CATCH }
int ctl_memcpy_d2h(void* dst, const void* src, size_t bytes) { CTL_TRY CTL_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); CTL_CATCH }
int ctl_memcpy_d2d(void* dst, const void* src, size_t bytes) { CTL_TRY CTL_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)); CTL_CATCH }
int ctl_device_synchronize(void) { CTL_TRY require_device(); CTL_HIP(hipDeviceSynchronize()); CTL_CATCH }
int ctl_set_device(int ordinal) { CTL_TRY require_device(); CTL_HIP(hipSetDevice(ordinal)); CTL_CATCH }

} // extern "C"
