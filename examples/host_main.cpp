// host_main.cpp — a compiled C++ host on the reference's side of the boundary: the flow of the reference's main.cpp:160-172
//     Resize -> InitializeScene -> UpdateScene -> Debug(132, 472) -> DoPass x N -> applyImagePipeline(BoxFilter(0.5, 0.5)) -> WriteDisplayImage("result.png")
// through nothing but include/ctl_amd.h and libctl_amd.so.  C++11, no HIP, no torch.
//     g++ -std=c++11 -Iinclude examples/host_main.cpp -Lcudatracerlib_amd -lctl_amd -Wl,-rpath,$PWD/cudatracerlib_amd -o host_main
//     host_main <scene.xml | --cornell> [passes = 16] [out.png = result.png] [--frame frame.bin] [--size W H]
// <scene.xml>: ParseMitsubaScene, as main.cpp:151 does.  --cornell: the Cornell box put together with the ctl_builder_* calls a DynamicScene host makes
// (CreateNode / CreateLight / setCamera).  --frame: the raw PixelData array (W * H * 28 B) for tests/test_gpu_host_example.py, which holds it against the
// frame the Python ctypes path renders from the same scene, bit for bit.  Prints one line: size, passes, rays, Mrays/s, debug pixel, FNV-1a of the frame.
#include "ctl_amd.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(call) do { const int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctl_last_error()); std::exit(1); } } while (0)

static ctl_texture constant_texture(float r, float g, float b) {
    ctl_texture t; std::memset(&t, 0, sizeof t);
    t.type = CTL_TEX_CONSTANT; t.value[0] = r; t.value[1] = g; t.value[2] = b; t.uv_scale[0] = t.uv_scale[1] = 1.0f;
    return t;
}
static ctl_material diffuse(float r, float g, float b) {   // diffuse(reflectance), SceneTypes/BSDF_Simple.h:6-24
    ctl_material m; std::memset(&m, 0, sizeof m);
    m.bsdf_type = CTL_BSDF_DIFFUSE; m.combined_type = CTL_EDiffuseReflection; m.node_light_index = 0xffffffffu;
    for (int i = 0; i < 4; i++) m.tex[i] = constant_texture(0, 0, 0);
    m.tex[0] = constant_texture(r, g, b);
    return m;
}

struct mesh_acc {
    std::vector<float> P, N; std::vector<uint32_t> I; std::vector<uint8_t> M;
    // a quad as two triangles with one normal, facing `inward` (the room's centre) or away from it
    void quad(const float q[4][3], const double inward[3], bool towards, uint8_t mat) {
        double e1[3], e2[3], c[3] = { 0, 0, 0 }, n[3];
        for (int k = 0; k < 3; k++) { e1[k] = (double)q[1][k] - q[0][k]; e2[k] = (double)q[3][k] - q[0][k]; for (int v = 0; v < 4; v++) c[k] += q[v][k] / 4.0; }
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        double d = 0; for (int k = 0; k < 3; k++) { n[k] /= len; d += (inward[k] - c[k]) * n[k]; }
        if ((d < 0) == towards) for (int k = 0; k < 3; k++) n[k] = -n[k];
        const uint32_t base = (uint32_t)(P.size() / 3);
        for (int v = 0; v < 4; v++) for (int k = 0; k < 3; k++) { P.push_back(q[v][k]); N.push_back((float)n[k]); }
        const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
        for (int k = 0; k < 6; k++) I.push_back(base + idx[k]);
        M.push_back(mat); M.push_back(mat);
    }
};

// the Cornell box (cornell-box.com data): local materials 0 white, 1 red, 2 green, 3 light
static void build_cornell(ctl_builder* b, uint32_t W, uint32_t H) {
    const double centre[3] = { 278, 274, 280 };
    mesh_acc m;
    const float room[5][4][3] = {
        { { 552.8f, 0, 0 }, { 0, 0, 0 }, { 0, 0, 559.2f }, { 549.6f, 0, 559.2f } },                    // floor
        { { 556, 548.8f, 0 }, { 556, 548.8f, 559.2f }, { 0, 548.8f, 559.2f }, { 0, 548.8f, 0 } },       // ceiling
        { { 549.6f, 0, 559.2f }, { 0, 0, 559.2f }, { 0, 548.8f, 559.2f }, { 556, 548.8f, 559.2f } },    // back wall
        { { 0, 0, 559.2f }, { 0, 0, 0 }, { 0, 548.8f, 0 }, { 0, 548.8f, 559.2f } },                     // right wall (green)
        { { 552.8f, 0, 0 }, { 549.6f, 0, 559.2f }, { 556, 548.8f, 559.2f }, { 556, 548.8f, 0 } } };     // left wall (red)
    const uint8_t room_mat[5] = { 0, 0, 0, 2, 1 };
    for (int i = 0; i < 5; i++) m.quad(room[i], centre, true, room_mat[i]);
    const float light[4][3] = { { 343, 548.3f, 227 }, { 343, 548.3f, 332 }, { 213, 548.3f, 332 }, { 213, 548.3f, 227 } };
    m.quad(light, centre, true, 3);
    const float blocks[2][5][4][3] = {
        { { { 130, 165, 65 }, { 82, 165, 225 }, { 240, 165, 272 }, { 290, 165, 114 } }, { { 290, 0, 114 }, { 290, 165, 114 }, { 240, 165, 272 }, { 240, 0, 272 } },
          { { 130, 0, 65 }, { 130, 165, 65 }, { 290, 165, 114 }, { 290, 0, 114 } }, { { 82, 0, 225 }, { 82, 165, 225 }, { 130, 165, 65 }, { 130, 0, 65 } },
          { { 240, 0, 272 }, { 240, 165, 272 }, { 82, 165, 225 }, { 82, 0, 225 } } },
        { { { 423, 330, 247 }, { 265, 330, 296 }, { 314, 330, 456 }, { 472, 330, 406 } }, { { 423, 0, 247 }, { 423, 330, 247 }, { 472, 330, 406 }, { 472, 0, 406 } },
          { { 472, 0, 406 }, { 472, 330, 406 }, { 314, 330, 456 }, { 314, 0, 456 } }, { { 314, 0, 456 }, { 314, 330, 456 }, { 265, 330, 296 }, { 265, 0, 296 } },
          { { 265, 0, 296 }, { 265, 330, 296 }, { 423, 330, 247 }, { 423, 0, 247 } } } };
    for (int blk = 0; blk < 2; blk++) {
        double c[3] = { 0, 80.0, 0 };
        for (int f = 0; f < 5; f++) for (int v = 0; v < 4; v++) { c[0] += blocks[blk][f][v][0] / 20.0; c[2] += blocks[blk][f][v][2] / 20.0; }
        for (int f = 0; f < 5; f++) m.quad(blocks[blk][f], c, false, 0);
    }
    const ctl_material mats[4] = { diffuse(0.725f, 0.71f, 0.68f), diffuse(0.63f, 0.065f, 0.05f), diffuse(0.14f, 0.45f, 0.091f), diffuse(0.78f, 0.78f, 0.78f) };
    uint32_t mesh = 0, node = 0;
    CHECK(ctl_builder_add_mesh(b, m.P.data(), (uint32_t)(m.P.size() / 3), m.I.data(), (uint32_t)(m.I.size() / 3), m.N.data(), nullptr, m.M.data(), mats, 4, &mesh));
    CHECK(ctl_builder_add_node(b, mesh, nullptr, &node));                           // DynamicScene::CreateNode
    const float radiance[3] = { 17.0f, 12.0f, 4.0f };
    CHECK(ctl_builder_add_area_light(b, node, 3, radiance));                       // DynamicScene::CreateLight(node, "light", L)
    const float pos[3] = { 278, 273, -800 }, target[3] = { 278, 273, -799 }, up[3] = { 0, 1, 0 };
    CHECK(ctl_builder_set_camera_lookat(b, pos, target, up, 39.3077f, W, H));      // DynamicScene::setCamera
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <scene.xml | --cornell> [passes] [out.png] [--frame frame.bin] [--size W H]\n", argv[0]); return 2; }
    const std::string scene_arg = argv[1];
    int n_passes = 16; std::string out_png = "result.png", frame_path; int32_t width = 1024, height = 1024;   // main.cpp:143
    int pos = 0;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--frame" && i + 1 < argc) frame_path = argv[++i];
        else if (a == "--size" && i + 2 < argc) { width = std::atoi(argv[++i]); height = std::atoi(argv[++i]); }
        else if (pos == 0) { n_passes = std::atoi(argv[i]); pos++; }
        else if (pos == 1) { out_png = a; pos++; }
    }
    if (ctl_device_count() < 1) { std::fprintf(stderr, "no HIP device: %s\n", ctl_version()); return 3; }   // InitializeCuda4Tracer
    CHECK(ctl_set_device(0));

    ctl_builder* builder = nullptr; CHECK(ctl_builder_create(&builder));            // DynamicScene scene(&camera, ...)
    if (scene_arg == "--cornell") build_cornell(builder, (uint32_t)width, (uint32_t)height);
    else { int32_t w = -1, h = -1; CHECK(ctl_parse_mitsuba_scene(builder, scene_arg.c_str(), &w, &h)); if (w > 0 && h > 0) { width = w; height = h; } }   // main.cpp:151-156

    ctl_image* image = nullptr; CHECK(ctl_image_create((uint32_t)width, (uint32_t)height, &image));   // Image outImage(width, height)
    ctl_tracer* tracer = nullptr; CHECK(ctl_tracer_create("WavefrontPathTracer", &tracer));          // options.tracer
    CHECK(ctl_tracer_resize(tracer, (uint32_t)width, (uint32_t)height));            // tracer->Resize(width, height)
    ctl_scene_desc desc; CHECK(ctl_builder_finalize(builder, &desc));               // scene.UpdateScene() + getKernelSceneData
    ctl_scene* scene = nullptr; CHECK(ctl_scene_create_ex(&desc, 1u, &scene));      // UpdateKernel: the scene in HBM (flag 1: + the flattened BVH)
    CHECK(ctl_tracer_initialize_scene(tracer, scene));                              // tracer->InitializeScene(&scene)

    float dbg[3] = { 0, 0, 0 };                                                     // tracer->Debug(&outImage, Vec2i(132, 472)) — scaled from main.cpp's 1024 x 1024
    CHECK(ctl_tracer_debug_pixel(tracer, image, (uint32_t)(132 * width / 1024), (uint32_t)(472 * height / 1024), dbg));

    for (int i = 0; i < n_passes; i++) CHECK(ctl_tracer_do_pass(tracer, image, i == 0));   // tracer->DoPass(&outImage, !i)
    ctl_tracer_stats st; CHECK(ctl_tracer_get_stats(tracer, &st));

    std::vector<uint32_t> display((size_t)width * height);                          // applyImagePipeline(*tracer, outImage, BoxFilter(0.5f, 0.5f))
    ctl_reconstruction_filter box; box.type = CTL_RFILTER_BOX; box.x_width = box.y_width = 0.5f; box.p0 = box.p1 = 0.0f;
    CHECK(ctl_image_apply_pipeline_ex(image, 0.0f, &box, nullptr, display.data()));
    CHECK(ctl_image_write_file(image, 0.0f, out_png.c_str()));                      // outImage.WriteDisplayImage("result.png")

    std::vector<ctl_pixel_data> frame((size_t)width * height);
    CHECK(ctl_image_read_pixels(image, frame.data()));
    uint64_t fnv = 1469598103934665603ull; double weight = 0;
    const unsigned char* bytes = (const unsigned char*)frame.data();
    for (size_t i = 0; i < frame.size() * sizeof(ctl_pixel_data); i++) { fnv ^= bytes[i]; fnv *= 1099511628211ull; }
    for (size_t i = 0; i < frame.size(); i++) weight += frame[i].weight_sum;
    if (!frame_path.empty()) {
        FILE* f = std::fopen(frame_path.c_str(), "wb");
        if (!f || std::fwrite(frame.data(), sizeof(ctl_pixel_data), frame.size(), f) != frame.size()) { std::fprintf(stderr, "cannot write %s\n", frame_path.c_str()); return 4; }
        std::fclose(f);
    }
    std::printf("{\"width\": %d, \"height\": %d, \"passes\": %u, \"rays\": %llu, \"mrays_per_s\": %.1f, \"weight_sum\": %.0f, \"debug_rgb\": [%.9g, %.9g, %.9g], \"display_pixel0\": %u, \"frame_fnv1a\": \"%016llx\"}\n",
                width, height, st.passes_done, (unsigned long long)st.rays_total, st.seconds_total > 0 ? st.rays_total / st.seconds_total / 1e6 : 0.0, weight, dbg[0], dbg[1], dbg[2],
                display[0], (unsigned long long)fnv);

    ctl_tracer_destroy(tracer); ctl_scene_destroy(scene); ctl_image_destroy(image); ctl_builder_destroy(builder);   // outImage.Free(); DeInitializeCuda4Tracer()
    return 0;
}
