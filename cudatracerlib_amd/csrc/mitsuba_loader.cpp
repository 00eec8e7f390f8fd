// mitsuba_loader.cpp — Mitsuba-0.5 XML scene loader feeding the scene builder: the counterpart of ParseMitsubaScene
// (Engine/SceneLoader/Mitsuba/MitsubaLoader.cpp:11-73) with the element handlers of ObjectParser.h / PropertyParser.cpp.
// Behaviour follows the reference handler by handler (cited at each function), including its defaults and quirks; what the
// HIP path cannot render yet (coatings, blends, bump / opacity maps, media, sun / sky) is rejected with unsupported_error, or
// skipped with a warning on stderr when CTL_LOADER_LENIENT=1 is set.
#include "mitsuba_loader.h"
#include "xml_lite.h"
#include "image_io.h"
#include "mesh_io.h"
#include "material_factory.h"
#include "sequence_generator.h"   // xorwow = CudaRNG
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace ctl {
namespace {

struct mat4 { float m[16]; };
mat4 identity() { mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
mat4 mul(const mat4& l, const mat4& r) { mat4 o; mat_mul(l.m, r.m, o.m); return o; }   // float4x4 operator% (float4x4.h:365-372)
mat4 translate(float x, float y, float z) { mat4 r = identity(); r.m[3] = x; r.m[7] = y; r.m[11] = z; return r; }
mat4 scale(float x, float y, float z) { mat4 r = identity(); r.m[0] = x; r.m[5] = y; r.m[10] = z; return r; }
mat4 rotate_x(float a) { mat4 r = identity(); const float c = cosf(a), s = sinf(a); r.m[5] = c; r.m[6] = -s; r.m[9] = s; r.m[10] = c; return r; }   // float4x4.h:523-532
mat4 rotation_axis(f3 n, float angle) {   // float4x4::RotationAxis (float4x4.h:556-582); the loader hands the XML angle over unconverted
    const float s = sinf(angle), c = cosf(angle); mat4 r = identity();
    r.m[0] = n.x * n.x + (1.0f - n.x * n.x) * c; r.m[1] = n.x * n.y * (1.0f - c) - n.z * s; r.m[2] = n.x * n.z * (1.0f - c) + n.y * s;
    r.m[4] = n.x * n.y * (1.0f - c) + n.z * s; r.m[5] = n.y * n.y + (1.0f - n.y * n.y) * c; r.m[6] = n.y * n.z * (1.0f - c) - n.x * s;
    r.m[8] = n.x * n.z * (1.0f - c) - n.y * s; r.m[9] = n.y * n.z * (1.0f - c) + n.x * s; r.m[10] = n.z * n.z + (1.0f - n.z * n.z) * c;
    return r;
}
mat4 look_at(f3 p, f3 t, f3 up) {   // float4x4::lookAt (float4x4.h:622-633)
    const f3 dir = normalize(t - p), left = normalize(cross(up, dir)), nu = cross(dir, left);
    mat4 r = identity();
    r.m[0] = left.x; r.m[4] = left.y; r.m[8] = left.z; r.m[1] = nu.x; r.m[5] = nu.y; r.m[9] = nu.z; r.m[2] = dir.x; r.m[6] = dir.y; r.m[10] = dir.z;
    r.m[3] = p.x; r.m[7] = p.y; r.m[11] = p.z;
    return r;
}
f3 xf_dir(const mat4& m, f3 d) { return f3(m.m[0] * d.x + m.m[1] * d.y + m.m[2] * d.z, m.m[4] * d.x + m.m[5] * d.y + m.m[6] * d.z, m.m[8] * d.x + m.m[9] * d.y + m.m[10] * d.z); }
f3 translation(const mat4& m) { return f3(m.m[3], m.m[7], m.m[11]); }
f3 forward(const mat4& m) { return xf_dir(m, f3(0, 0, 1)); }

struct rgb { float r, g, b; rgb(float v = 0) : r(v), g(v), b(v) {} rgb(float R, float G, float B) : r(R), g(G), b(B) {} };

// BsdfData (ObjectParser.h:600-610); the height / alpha map travel inside ctl_material (map_tex, alpha_tex)
struct bsdf_data { ctl_material mat; bool two_sided = false; };
struct shape_result;
struct group_data { std::vector<std::pair<mat4, uint32_t>> nodes; int instanciations = 0; };
struct shape_result { int type = 3; uint32_t node = 0; group_data group; };   // 1 node, 2 group, 3 nothing (ObjectParser.h:1147-1166)

struct loader {
    scene_builder& B;
    std::string dir;
    bool lenient;
    std::map<std::string, std::string> defaults;
    std::map<std::string, bsdf_data> ref_bsdf; std::map<std::string, ctl_texture> ref_tex; std::map<std::string, mat4> ref_mat; std::map<std::string, rgb> ref_rgb;
    std::map<std::string, f3> ref_vec; std::map<std::string, std::shared_ptr<shape_result>> ref_shape;
    std::map<std::string, uint32_t> mesh_cache;      // file path (+ options) -> mesh index, CachedBuffer<Mesh> of the reference
    std::map<std::string, uint32_t> image_cache;     // file path -> image index
    std::map<uint32_t, std::vector<float>> mesh_emission;   // OBJ "Ke" per mesh material
    std::map<uint32_t, uint32_t> node_mesh;
    int film_w = 768, film_h = 576; bool have_film = false, have_sensor = false;
    mat4 id_matrix = identity();   // assume_rotated_coords = false (main.cpp passes false)

    loader(scene_builder& b, const std::string& d) : B(b), dir(d) { const char* e = std::getenv("CTL_LOADER_LENIENT"); lenient = e && std::atoi(e) != 0; }

    [[noreturn]] static void bad(const std::string& msg) { throw std::runtime_error(msg); }
    void unsupported(const std::string& what) {
        if (!lenient) throw unsupported_error("ParseMitsubaScene: " + what + " is not supported by this build (set CTL_LOADER_LENIENT=1 to skip it)");
        std::fprintf(stderr, "[ctl loader] skipping unsupported %s\n", what.c_str());
    }

    // ---- DefaultValueStorage (Utils.h:148-263)
    std::string map_default(const std::string& data, int depth = 0) const {
        const size_t s = data.find('$');
        if (s == std::string::npos) return data;
        if (depth > 64) bad("invalid default value (a <default> that refers to itself)");
        size_t e = data.find(' ', s); if (e == std::string::npos) e = data.size();
        const std::string key = data.substr(s + 1, e - s - 1);
        auto it = defaults.find(key);
        if (it == defaults.end()) bad("invalid default value");
        // (the reference keeps the '$' when the key is not at the start of the string, Utils.h:175; kept)
        return map_default((s != 0 ? data.substr(0, s + 1) : "") + it->second + (e != data.size() ? data.substr(e) : ""), depth + 1);
    }
    float as_float(const std::string& v) const { return std::stof(map_default(v)); }
    std::string attr_s(const xml_node& n, const char* a) const { return map_default(n.attr(a)); }
    float attr_f(const xml_node& n, const char* a) const { return as_float(n.attr(a)); }
    float attr_f(const xml_node& n, const char* a, float def) const { return n.has_attr(a) ? as_float(n.attr(a)) : def; }
    float prop_f(const xml_node& n, const char* p, float def) const { const xml_node* c = n.property(p); return c ? attr_f(*c, "value", def) : def; }
    float prop_f(const xml_node& n, const char* p) const { const xml_node* c = n.property(p); if (!c) bad("no default value passed but property doesn't exist!"); return attr_f(*c, "value"); }
    int prop_i(const xml_node& n, const char* p, int def) const { const xml_node* c = n.property(p); return (c && c->has_attr("value")) ? std::stoi(map_default(c->attr("value"))) : def; }
    int prop_i(const xml_node& n, const char* p) const { const xml_node* c = n.property(p); if (!c) bad("no default value passed but property doesn't exist!"); return std::stoi(map_default(c->attr("value"))); }
    std::string prop_s(const xml_node& n, const char* p) const { const xml_node* c = n.property(p); if (!c) bad("no default value passed but property doesn't exist!"); return attr_s(*c, "value"); }
    std::string prop_s(const xml_node& n, const char* p, const std::string& def) const { const xml_node* c = n.property(p); return (c && c->has_attr("value")) ? attr_s(*c, "value") : def; }
    bool prop_b(const xml_node& n, const char* p, bool def) const { const xml_node* c = n.property(p); return (c && c->has_attr("value")) ? map_default(c->attr("value")) == "True" : def; }   // as_bool compares with "True" (Utils.h:203-207)
    std::string asset(const std::string& f) const { return dir + "/" + f; }
    static bool is_ref(const xml_node& n) { return n.lname() == "ref"; }

    // split_string_array (Utils.h:62-65): separators ',' and ' ', empty entries removed
    static std::vector<std::string> split_array(const std::string& s) {
        std::vector<std::string> out; std::string cur;
        for (char c : s) { if (c == ',' || c == ' ' || c == '\t' || c == '\n' || c == '\r') { if (!cur.empty()) out.push_back(cur); cur.clear(); } else cur += c; }
        if (!cur.empty()) out.push_back(cur);
        return out;
    }

    // ---- PropertyParser.cpp
    f3 parse_vector(const xml_node& n) {   // :6-14
        if (is_ref(n)) { auto it = ref_vec.find(n.attr("id")); if (it == ref_vec.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        f3 v(attr_f(n, "x"), attr_f(n, "y"), attr_f(n, "z"));
        v = xf_dir(id_matrix, v);
        if (n.has_attr("id")) ref_vec[n.attr("id")] = v;
        return v;
    }
    mat4 parse_matrix(const xml_node& n, bool apply_id = true) {   // :16-82
        if (is_ref(n)) { auto it = ref_mat.find(n.attr("id")); if (it == ref_mat.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        mat4 T = identity();
        for (const xml_node& t : n.children) {
            mat4 I = identity(); const std::string N = t.lname();
            if (N == "translate") I = translate(attr_f(t, "x", 0), attr_f(t, "y", 0), attr_f(t, "z", 0));
            else if (N == "rotate") I = rotation_axis(f3(attr_f(t, "x", 0), attr_f(t, "y", 0), attr_f(t, "z", 0)), attr_f(t, "angle"));
            else if (N == "scale") { if (t.has_attr("value")) { const float v = attr_f(t, "value"); I = scale(v, v, v); } else I = scale(attr_f(t, "x", 1), attr_f(t, "y", 1), attr_f(t, "z", 1)); }
            else if (N == "matrix") { auto s = split_array(attr_s(t, "value")); if (s.size() < 16) bad("invalid matrix"); for (int i = 0; i < 16; i++) I.m[i] = std::stof(s[i]); }
            else if (N == "lookat") {
                auto conv = [&](const char* a) { auto s = split_array(attr_s(t, a)); if (s.size() != 3) bad("invalid vector"); return f3(std::stof(s[0]), std::stof(s[1]), std::stof(s[2])); };
                I = look_at(conv("origin"), conv("target"), conv("up"));
            }
            else bad("invalid matrix operation : " + N);
            T = mul(I, T);
        }
        if (apply_id) T = mul(id_matrix, T);
        if (n.has_attr("id")) ref_mat[n.attr("id")] = T;
        return T;
    }
    static float srgb_to_linear(float v) { return v <= 0.04045f ? v * (1.0f / 12.92f) : powf((v + 0.055f) * (1.0f / 1.055f), 2.4f); }   // Spectrum::fromSRGB (Spectrum.cu)
    rgb parse_rgb(const xml_node& n, bool srgb) {   // :84-112
        if (is_ref(n)) { auto it = ref_rgb.find(n.attr("id")); if (it == ref_rgb.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        const std::string s = attr_s(n, "value"); rgb C;
        if (srgb && !s.empty() && s[0] == '#') { const int hex = std::stoi(s.substr(1)); C = rgb(((hex >> 16) & 0xFF) / 255.0f, ((hex >> 8) & 0xFF) / 255.0f, (hex & 0xFF) / 255.0f); }   // (decimal stoi, as the reference)
        else {
            auto v = split_array(s); if (v.size() == 1) v = { v[0], v[0], v[0] }; if (v.size() < 3) bad("invalid colour");
            C = rgb(std::stof(v[0]), std::stof(v[1]), std::stof(v[2]));
            if (srgb) C = rgb(srgb_to_linear(C.r), srgb_to_linear(C.g), srgb_to_linear(C.b));
        }
        if (n.has_attr("id")) ref_rgb[n.attr("id")] = C;
        return C;
    }
    rgb parse_spectrum(const xml_node& n) {   // :114-176 — RGB triples and single values; wavelength tables need the reference's CIE data
        if (is_ref(n)) { auto it = ref_rgb.find(n.attr("id")); if (it == ref_rgb.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        if (n.has_attr("filename")) { unsupported("<spectrum filename=...> (.spd tables)"); return rgb(0.0f); }
        const std::string data = attr_s(n, "value");
        if (data.find(':') != std::string::npos) { unsupported("<spectrum> given as wavelength:value pairs"); return rgb(0.0f); }
        auto v = split_array(data); rgb C;
        if (v.size() == 3) C = rgb(std::stof(v[0]), std::stof(v[1]), std::stof(v[2]));
        else { try { C = rgb(std::stof(data)); } catch (...) { bad("invalid spectrum data"); } }
        if (n.has_attr("id")) ref_rgb[n.attr("id")] = C;
        return C;
    }
    rgb parse_color(const xml_node& n) {   // ObjectParser.h:54-66
        const std::string t = n.lname();
        if (t == "srgb") return parse_rgb(n, true);
        if (t == "rgb") return parse_rgb(n, false);
        if (t == "spectrum") return parse_spectrum(n);
        if (t == "float") return rgb(attr_f(n, "value"));
        if (t == "ref") { auto it = ref_rgb.find(n.attr("id")); if (it != ref_rgb.end()) return it->second; }
        bad("invalid spectrum type : " + t);
    }
    rgb try_color(const xml_node& n, const char* p, rgb def) { const xml_node* c = n.property(p); return c ? parse_color(*c) : def; }

    // ---- textures (ObjectParser.h:74-133)
    uint32_t load_image(const std::string& file, uint32_t wrap, uint32_t filter) {
        auto it = image_cache.find(file); if (it != image_cache.end()) return it->second;
        const decoded_image img = load_image_file(file);
        std::vector<uint32_t> texels; const uint32_t type = image_to_texels(img, texels);
        const uint32_t idx = B.add_image(texels.data(), img.width, img.height, type, wrap, filter);
        image_cache[file] = idx; return idx;
    }
    ctl_texture parse_texture(const xml_node& n) {
        if (is_ref(n)) { auto it = ref_tex.find(n.attr("id")); if (it == ref_tex.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        const std::string T = n.attr("type"); ctl_texture t;
        const float uoff = prop_f(n, "uoff", 0.0f), voff = prop_f(n, "voff", 0.0f), uscale = prop_f(n, "uscale", 1.0f), vscale = prop_f(n, "vscale", 1.0f);
        if (T == "bitmap") {   // ImageTexture(TextureMapping2D(su, sv, du, dv), file, 1.0f); MIPMap defaults: repeat wrap, trilinear filter (-> level-0 triangle filter)
            t = tex_const(1.0f); t.type = CTL_TEX_IMAGE;
            t.image = load_image(asset(prop_s(n, "filename")), CTL_WRAP_REPEAT, CTL_FILTER_TRILINEAR);
        } else if (T == "checkerboard") {
            const rgb c0 = try_color(n, "color0", rgb(0.4f)), c1 = try_color(n, "color1", rgb(0.2f));
            t = tex_const(c0.r, c0.g, c0.b); t.type = CTL_TEX_CHECKER; t.value1[0] = c1.r; t.value1[1] = c1.g; t.value1[2] = c1.b;
        } else if (T == "wireframe") { unsupported("texture type wireframe"); const rgb c = try_color(n, "interiorColor", rgb(0.5f)); t = tex_const(c.r, c.g, c.b); }
        else bad("invalid Texture type : " + T);
        t.uv_scale[0] = uscale; t.uv_scale[1] = vscale; t.uv_offset[0] = uoff; t.uv_offset[1] = voff;
        if (n.has_attr("id")) ref_tex[n.attr("id")] = t;
        return t;
    }
    ctl_texture tex_or_color(const xml_node& n) {   // :118-125
        if (n.lname() == "texture") return parse_texture(n);
        if (n.lname() == "ref") { auto it = ref_tex.find(n.attr("id")); if (it != ref_tex.end()) return it->second; auto ic = ref_rgb.find(n.attr("id")); if (ic != ref_rgb.end()) return tex_const(ic->second.r, ic->second.g, ic->second.b); bad("unknown reference : " + n.attr("id")); }
        const rgb c = parse_color(n); return tex_const(c.r, c.g, c.b);
    }
    ctl_texture try_tex(const xml_node& n, const char* p, rgb def) { const xml_node* c = n.property(p); return c ? tex_or_color(*c) : tex_const(def.r, def.g, def.b); }

    // ---- IoRLibrary (Utils.h:265-318)
    static bool ior_name(const std::string& name, float& v) {
        static const std::map<std::string, float> lib = { { "vacuum", 1.0f }, { "helium", 1.00004f }, { "hydrogen", 1.00013f }, { "air", 1.00028f }, { "carbon dioxide", 1.00045f }, { "water", 1.3330f },
            { "acetone", 1.36f }, { "ethanol", 1.361f }, { "carbon tetrachloride", 1.451f }, { "glycerol", 1.4729f }, { "benzene", 1.501f }, { "silicone oil", 1.52045f }, { "bromine", 1.661f },
            { "water ice", 1.31f }, { "fused quartz", 1.458f }, { "pyrex", 1.470f }, { "acrylic glass ", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f }, { "sodium chloride", 1.544f },
            { "amber", 1.55f }, { "pet", 1.575f }, { "diamond", 2.419f } };
        auto it = lib.find(name); if (it == lib.end()) return false; v = it->second; return true;
    }
    float ior_of(const xml_node& n) const {
        const std::string v = map_default(n.attr("value"));
        try { return std::stof(v); } catch (...) { float r; if (!ior_name(v, r)) bad("Invalid IOR material name"); return r; }
    }

    // ---- BSDFs (ObjectParser.h:596-1010)
    void generic_rough(const xml_node& n, ctl_texture& aU, ctl_texture& aV, uint32_t& dist) {   // :631-645
        aU = aV = tex_const(0.1f);
        if (const xml_node* c = n.property("alpha")) aU = aV = tex_or_color(*c);
        if (const xml_node* c = n.property("alphaU")) aU = tex_or_color(*c);
        if (const xml_node* c = n.property("alphaV")) aV = tex_or_color(*c);
        dist = prop_s(n, "distribution", "beckmann") == "beckmann" ? CTL_MF_BECKMANN : CTL_MF_GGX;   // every other name selects GGX (the ternary at :644)
    }
    void generic_ior(const xml_node& n, ctl_texture& refl, ctl_texture& trans, float& ior) {   // :647-662
        float intIOR = 1.5046f, extIOR = 1.00028f;
        refl = try_tex(n, "specularReflectance", rgb(1.0f)); trans = try_tex(n, "specularTransmittance", rgb(1.0f));
        if (const xml_node* c = n.property("intIOR")) intIOR = ior_of(*c);
        if (const xml_node* c = n.property("extIOR")) extIOR = ior_of(*c);
        ior = intIOR / extIOR;
    }
    void generic_eta(const xml_node& n, float eta[3], float k[3], ctl_texture& refl) {   // :664-711
        rgb e(0.0f), kk(0.0f); refl = tex_const(1.0f); float extEta = 1.00028f;
        if (n.property("material") && prop_s(n, "material") != "none") unsupported("conductor material presets (.spd tables)");
        e = try_color(n, "eta", e); kk = try_color(n, "k", kk);
        extEta = prop_f(n, "extEta", extEta);
        eta[0] = e.r / extEta; eta[1] = e.g / extEta; eta[2] = e.b / extEta; k[0] = kk.r / extEta; k[1] = kk.g / extEta; k[2] = kk.b / extEta;
    }
    std::vector<bsdf_data> all_nested(const xml_node& n, int depth) {
        std::vector<bsdf_data> out;
        for (const xml_node& c : n.children) if (c.lname() == "ref" || c.lname() == "bsdf") out.push_back(parse_bsdf(c, depth + 1));
        return out;
    }
    static bsdf_data make(const ctl_material& m, const std::vector<bsdf_data>* others = nullptr) {   // BsdfParser::create (:605-626)
        bsdf_data d; d.mat = m;
        if (others) for (auto& c : *others) {   // the first nested BSDF that has a map hands it up (:621-624)
            d.two_sided |= c.two_sided;
            if (d.mat.map_kind == CTL_MAP_NONE && c.mat.map_kind != CTL_MAP_NONE) { d.mat.map_kind = c.mat.map_kind; d.mat.map_tex = c.mat.map_tex; }
            if (d.mat.alpha_state == CTL_ALPHA_DISABLED && c.mat.alpha_state != CTL_ALPHA_DISABLED) {
                d.mat.alpha_state = c.mat.alpha_state; d.mat.alpha_tex = c.mat.alpha_tex; d.mat.alpha_test_scalar = c.mat.alpha_test_scalar;
                std::memcpy(d.mat.alpha_test_color, c.mat.alpha_test_color, sizeof(d.mat.alpha_test_color));
            }
        }
        return d;
    }
    bsdf_data parse_bsdf(const xml_node& n, int depth) {   // BsdfParser::parse (:968-992)
        if (is_ref(n)) { auto it = ref_bsdf.find(n.attr("id")); if (it == ref_bsdf.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        const std::string T = n.attr("type"); bsdf_data d;
        ctl_texture refl, trans, aU, aV, diff, spec; float ior = 1.0f, eta[3], k[3]; uint32_t dist = 0;
        if (T == "diffuse") d = make(make_diffuse(try_tex(n, "reflectance", rgb(0.5f))));
        else if (T == "roughdiffuse") d = make(make_roughdiffuse(try_tex(n, "reflectance", rgb(0.5f)), try_tex(n, "reflectance", rgb(0.2f))));   // alpha read from "reflectance" (:759)
        else if (T == "dielectric") { generic_ior(n, refl, trans, ior); d = make(make_dielectric(ior, refl, trans)); }
        else if (T == "thindielectric") { generic_ior(n, refl, trans, ior); d = make(make_thindielectric(ior, refl, trans)); }
        else if (T == "roughdielectric") { generic_ior(n, refl, trans, ior); generic_rough(n, aU, aV, dist); d = make(make_roughdielectric(dist, ior, aU, aV, refl, trans)); }
        else if (T == "conductor") { generic_eta(n, eta, k, refl); d = make(make_conductor(eta, k, refl)); }
        else if (T == "roughconductor") { generic_eta(n, eta, k, refl); generic_rough(n, aU, aV, dist); d = make(make_roughconductor(dist, eta, k, aU, aV, refl)); }
        else if (T == "plastic" || T == "roughplastic") {
            generic_ior(n, refl, trans, ior);
            const bool nonlinear = prop_b(n, "nonlinear", false);
            spec = try_tex(n, "specularReflectance", rgb(1.0f)); diff = try_tex(n, "diffuseReflectance", rgb(0.5f));
            if (T == "plastic") d = make(make_plastic(ior, diff, spec, nonlinear));
            else { generic_rough(n, aU, aV, dist); d = make(make_roughplastic(dist, ior, aU, diff, spec, nonlinear)); }
        }
        else if (T == "phong") d = make(make_phong(try_tex(n, "diffuseReflectance", rgb(0.5f)), try_tex(n, "specularReflectance", rgb(0.2f)), try_tex(n, "exponent", rgb(30.0f))));
        else if (T == "ward") {
            const std::string v = prop_s(n, "variant", "balanced");
            d = make(make_ward(v == "ward" ? 0u : (v == "ward-duer" ? 1u : 2u), try_tex(n, "diffuseReflectance", rgb(0.5f)), try_tex(n, "specularReflectance", rgb(0.2f)), try_tex(n, "alphaU", rgb(0.1f)), try_tex(n, "alphaV", rgb(0.1f))));
        }
        else if (T == "difftrans") { ctl_material m = make_diffuse(try_tex(n, "reflectance", rgb(0.5f))); m.combined_type = CTL_EDiffuseTransmission; d = make(m); }
        else if (T == "twosided") { auto nested = all_nested(n, depth - 1); if (nested.size() != 1) bad("expected 1 nested bsdf in twosided!"); d = nested[0]; d.two_sided = true; }
        else if (T == "bumpmap") { auto nested = all_nested(n, depth - 1); if (nested.size() != 1) bad("expected 1 nested bsdf in bumpmap!"); d = nested[0]; d.mat.map_kind = CTL_MAP_HEIGHT; d.mat.map_tex = try_tex(n, "texture", rgb(0.0f)); }   // :867-877
        else if (T == "mask") { auto nested = all_nested(n, depth - 1); if (nested.size() != 1) bad("expected 1 nested bsdf in mask!");   // :930-947
            const ctl_texture opacity = try_tex(n, "opacity", rgb(0.0f));
            if (nested[0].mat.bsdf_type == CTL_BSDF_DIFFUSE) {   // the reference turns a masked diffuse into a diffuse TRANSMITTER coloured by the opacity (:940-944)
                ctl_material m = make_diffuse(opacity); m.combined_type = CTL_EDiffuseTransmission; d = make(m, &nested);
            } else {
                // SetAlphaMap(opacity, AlphaMap_Luminance) (:1006-1007).  The reference leaves AlphaBlendData::test_val_scalar uninitialised on this
                // path; the build uses 1.0, the value the reference's OBJ loader sets for its alpha maps (ObjParser.cpp:849)
                d = nested[0]; d.mat.alpha_state = CTL_ALPHA_MAP_LUMINANCE; d.mat.alpha_tex = opacity; d.mat.alpha_test_scalar = 1.0f;
            }
        }
        else if (T == "coating" || T == "roughcoating") {
            auto nested = all_nested(n, depth); if (nested.size() != 1) bad("expected 1 nested bsdf in coating!");
            if (depth == 1 || nested[0].mat.bsdf_type >= CTL_BSDF_HK) d = nested[0];   // is_max_depth(): the reference returns the nested BSDF (:873-874); BSDFFirst cannot hold a nesting model
            else {
                generic_ior(n, refl, trans, ior);
                const float thickness = prop_f(n, "thickness", 1.0f); const ctl_texture sigmaA = try_tex(n, "sigmaA", rgb(0.0f));
                const uint32_t ni = B.add_aux_material(nested[0].mat);
                if (T == "coating") d = make(make_coating(ni, nested[0].mat.combined_type, ior, thickness, sigmaA, refl), &nested);
                else { generic_rough(n, aU, aV, dist); d = make(make_roughcoating(dist, ni, nested[0].mat.combined_type, ior, thickness, sigmaA, aU, refl), &nested); }
            }
        }
        else if (T == "mixturebsdf" || T == "blendbsdf") {
            auto nested = all_nested(n, depth); if (nested.size() != 2) bad("expected 2 nested bsdf in " + T + "!");
            if (depth == 1 || nested[0].mat.bsdf_type >= CTL_BSDF_HK || nested[1].mat.bsdf_type >= CTL_BSDF_HK) d = nested[0];
            else {
                ctl_texture weight;
                if (T == "mixturebsdf") {
                    auto w = split_array(prop_s(n, "weights")); if (w.size() != 2) bad("not able to get 2 weights from weights property");
                    const float w1 = std::stof(w[0]), w2 = std::stof(w[1]); weight = tex_const(w1 / (w1 + w2));
                } else weight = try_tex(n, "weight", rgb(0.5f));
                const uint32_t n0 = B.add_aux_material(nested[0].mat), n1 = B.add_aux_material(nested[1].mat);
                d = make(make_blend(n0, nested[0].mat.combined_type, n1, nested[1].mat.combined_type, weight), &nested);
            }
        }
        else bad("invalid BsdfData type : " + T);
        if (n.has_attr("id")) ref_bsdf[n.attr("id")] = d;
        return d;
    }
    void apply_bsdf(const xml_node& n, uint32_t node) {   // :996-1010 — material 0 of the node
        const bsdf_data d = parse_bsdf(n, 0);
        ctl_material m = d.mat; m.two_sided = d.two_sided ? 1u : 0u;
        B.set_node_bsdf(node, 0, m);
    }

    // ---- shapes (ObjectParser.h:1012-1300)
    uint32_t create_node(const std::string& key, const mesh_data* data) {
        uint32_t mesh;
        auto it = mesh_cache.find(key);
        if (it != mesh_cache.end()) mesh = it->second;
        else {
            const mesh_data& M = *data;
            std::vector<ctl_material> mats = M.materials;
            if (!M.image_files.empty()) {   // bitmaps of an OBJ's .mtl: load (MIPMap defaults: repeat wrap, trilinear filter) and point the textures at the scene's image table
                std::vector<uint32_t> idx(M.image_files.size());
                for (size_t i = 0; i < idx.size(); i++) idx[i] = load_image(M.image_files[i], CTL_WRAP_REPEAT, CTL_FILTER_TRILINEAR);
                auto fix = [&](ctl_texture& t) { if (t.type == CTL_TEX_IMAGE && t.image < idx.size()) t.image = idx[t.image]; };
                for (auto& m : mats) { for (auto& t : m.tex) fix(t); if (m.map_kind != CTL_MAP_NONE) fix(m.map_tex); if (m.alpha_state != CTL_ALPHA_DISABLED) fix(m.alpha_tex); }
            }
            mesh = B.add_mesh(M.positions.data(), M.n_vertices(), M.indices.data(), M.n_triangles(), M.normals.empty() ? nullptr : M.normals.data(), M.uvs.empty() ? nullptr : M.uvs.data(),
                              M.tri_material.data(), mats.data(), (uint32_t)mats.size(), M.flip_normals, M.face_normals, M.max_smooth_angle);
            mesh_cache[key] = mesh; mesh_emission[mesh] = M.emission;
        }
        const uint32_t node = B.add_node(mesh, nullptr);
        node_mesh[node] = mesh;
        return node;
    }
    uint32_t node_from_file(const std::string& file) {   // parseFiles (:1014-1029) -> DynamicScene::CreateNode
        const std::string path = asset(file);
        if (mesh_cache.count(path)) return create_node(path, nullptr);
        const mesh_data M = load_mesh_file(path);
        return create_node(path, &M);
    }
    uint32_t node_virtual(const char* name, mesh_data (*gen)()) {   // load_virtual (:44-52)
        if (mesh_cache.count(name)) return create_node(name, nullptr);
        const mesh_data M = gen();
        return create_node(name, &M);
    }
    void mesh_lights(uint32_t node) {   // MeshPartLight entries of a compiled OBJ become area lights when the node is created (DynamicScene::CreateNode)
        const std::vector<float>& E = mesh_emission[node_mesh[node]];
        for (size_t m = 0; m * 3 + 2 < E.size(); m++) if (E[m * 3] != 0 || E[m * 3 + 1] != 0 || E[m * 3 + 2] != 0) B.add_area_light(node, (uint32_t)m, &E[m * 3]);
    }
    void parse_generic(uint32_t node, const xml_node& n, const mat4& local = identity(), bool in_coord = false) {   // :1031-1102
        const mat4 T = n.property("toWorld") ? parse_matrix(*n.property("toWorld")) : (in_coord ? identity() : id_matrix);
        ctl_float4x4 m; std::memcpy(m.m, mul(T, local).m, 64);
        B.set_node_transform(node, m);
        mesh_lights(node);
        if (const xml_node* em = n.child("emitter")) {
            if (em->attr("type") != "area") bad("only supports area light sources on meshes, not : " + em->attr("type"));
            const xml_node* r = em->property("radiance"); if (!r) bad("no property of that name : radiance");
            const rgb e = parse_color(*r); const float rad[3] = { e.r, e.g, e.b };
            B.add_area_light(node, 0, rad);
        }
        if (const xml_node* b = n.child("bsdf")) apply_bsdf(*b, node);
        for (const xml_node& c : n.children) if (c.lname() == "ref" && (!c.has_attr("name") || c.attr("name") == "bsdf")) apply_bsdf(c, node);
        if (n.property("interior") || n.property("exterior")) { /* media are only created when the caller asks for BSSRDFs (main.cpp passes false) */ }
    }
    mat4 node_transform(uint32_t node) const { mat4 r; std::memcpy(r.m, B.xf[node].m, 64); return r; }
    void assign_instance(uint32_t src, const mat4& m) {   // ShapegroupData::assign (:1127-1145): new node of the same mesh with the source's BSDF
        const uint32_t mesh = node_mesh[src];
        const uint32_t node = B.add_node(mesh, nullptr); node_mesh[node] = mesh;
        ctl_float4x4 x; std::memcpy(x.m, m.m, 64); B.set_node_transform(node, x);
        mesh_lights(node);
        B.set_node_bsdf(node, 0, B.node_material(src, 0));
    }
    std::shared_ptr<shape_result> parse_shape(const xml_node& n) {
        if (is_ref(n)) { auto it = ref_shape.find(n.attr("id")); if (it == ref_shape.end()) bad("unknown reference : " + n.attr("id")); return it->second; }
        const std::string T = n.attr("type"); auto R = std::make_shared<shape_result>();
        const float pi = 3.14159265358979f;
        auto flip = [&]() { return prop_b(n, "flipNormals", false); };
        if (T == "obj" || T == "ply") { R->node = node_from_file(prop_s(n, "filename")); parse_generic(R->node, n); R->type = 1; }
        else if (T == "serialized") {
            const std::string path = asset(prop_s(n, "filename")); const int idx = prop_i(n, "shapeIndex");
            const bool fl = flip(), fn = prop_b(n, "faceNormals", false); const float msa = prop_f(n, "maxSmoothAngle", 0.0f);
            const std::string key = path + "#" + std::to_string(idx) + (fl ? "f" : "") + (fn ? "n" : "") + (msa != 0 ? "a" + std::to_string(msa) : "");
            if (mesh_cache.count(key)) R->node = create_node(key, nullptr);
            else { mesh_data M = load_serialized(path, idx); M.flip_normals = fl; M.face_normals = fn; M.max_smooth_angle = msa; R->node = create_node(key, &M); }
            parse_generic(R->node, n); R->type = 1;
        }
        else if (T == "rectangle") { R->node = node_virtual("rect", make_plane); parse_generic(R->node, n, mul(rotate_x(pi / 2), scale(1, 1, flip() ? -1.0f : 1.0f))); R->type = 1; }
        else if (T == "sphere") {
            const float radius = prop_f(n, "radius", 1.0f); const f3 pos = n.property("center") ? parse_vector(*n.property("center")) : f3(0.0f);
            R->node = node_virtual("sphere", make_sphere); const float r = (flip() ? -1.0f : 1.0f) * radius;
            parse_generic(R->node, n, mul(translate(pos.x, pos.y, pos.z), scale(r, r, r)), true); R->type = 1;
        }
        else if (T == "cube") { R->node = node_virtual("cube", make_cube); const float s = flip() ? -2.0f : 2.0f; parse_generic(R->node, n, mul(scale(s, s, s), translate(-0.5f, -0.5f, -0.5f))); R->type = 1; }
        else if (T == "cylinder") {
            const float radius = prop_f(n, "radius", 1.0f);
            const f3 p0 = n.property("p0") ? parse_vector(*n.property("p0")) : f3(0, 0, 0), p1 = n.property("p1") ? parse_vector(*n.property("p1")) : f3(0, 0, 1);
            R->node = node_virtual("cylinder", make_cylinder); const float r = (flip() ? -1.0f : 1.0f) * radius;
            const f3 ax = normalize(p1 - p0); f3 s, t; coordinate_system(ax, s, t);
            mat4 fr = identity(); fr.m[0] = s.x; fr.m[4] = s.y; fr.m[8] = s.z; fr.m[1] = t.x; fr.m[5] = t.y; fr.m[9] = t.z; fr.m[2] = ax.x; fr.m[6] = ax.y; fr.m[10] = ax.z;   // Frame::ToWorldMatrix
            parse_generic(R->node, n, mul(mul(translate(p0.x, p0.y, p0.z), fr), scale(r, r, length(p1 - p0) / 2)), true); R->type = 1;
        }
        else if (T == "disk") { R->node = node_virtual("disk", make_disk); const float s = flip() ? -1.0f : 1.0f; parse_generic(R->node, n, scale(s, s, s)); R->type = 1; }
        else if (T == "shapegroup") {
            for (const xml_node& c : n.children) {
                if (c.lname() != "shape") bad("invalid node in shapegroup : " + c.lname());
                auto o = parse_shape(c); if (o->type != 1) bad("invalid xml parsed, expected node");
                R->group.nodes.emplace_back(node_transform(o->node), o->node);
            }
            R->type = 2;
        }
        else if (T == "instance") {
            const mat4 Tm = n.property("toWorld") ? parse_matrix(*n.property("toWorld")) : id_matrix;
            const xml_node* rf = n.child("ref"); if (!rf) bad("instance without <ref>");
            auto it = ref_shape.find(attr_s(*rf, "id")); if (it == ref_shape.end()) bad("unknown reference : " + rf->attr("id"));
            shape_result& g = *it->second;
            if (g.type == 2) {   // ShapegroupData::instanciate (:1114-1125)
                mat4 id_inv; mat_inverse(id_matrix.m, id_inv.m);
                for (auto& el : g.group.nodes) {
                    const mat4 m = mul(mul(Tm, id_inv), el.first);
                    if (g.group.instanciations == 0) { ctl_float4x4 x; std::memcpy(x.m, m.m, 64); B.set_node_transform(el.second, x); } else assign_instance(el.second, m);
                }
                g.group.instanciations++;
            } else if (g.type == 1) assign_instance(g.node, Tm);
            else bad("invalid ref type : " + std::to_string(g.type));
            R->type = 3;
        }
        else if (T == "hair") { std::fprintf(stderr, "hair model is not implemented\n"); R->type = 3; }
        else bad("invalid ShapeParseResult type : " + T);
        if (n.has_attr("id")) ref_shape[n.attr("id")] = R;
        return R;
    }

    // ---- sensor (ObjectParser.h:226-345): perspective, thinlens, orthographic.  `telecentric` is refused: the reference default-constructs it and never sets its
    //      aperture, focus distance or screen scale (ObjectParser.h:329-334, Sensor.h:453-457), i.e. it renders from uninitialised memory; the sensor itself is
    //      available through ctl_builder_set_camera
    void parse_sensor(const xml_node& n) {
        const std::string T = n.attr("type");
        if (T == "telecentric") throw unsupported_error("ParseMitsubaScene: the reference leaves a telecentric sensor's aperture, focus distance and screen scale uninitialised; set the camera through ctl_builder_set_camera");
        if (T != "perspective" && T != "thinlens" && T != "orthographic") bad("invalid Sensor type : " + T);
        int width = 768, height = 576;
        if (const xml_node* film = n.child("film")) { width = prop_i(*film, "width", width); height = prop_i(*film, "height", height); }
        film_w = width; film_h = height; have_film = true;
        float fov_deg = 50.0f;   // PerspectiveSensor() leaves fov uninitialised (Sensor.h:196-200); Mitsuba's documented default is used
        auto set_diagonal = [&](float diag_fov) {   // the reference feeds this value to tan() and the result to SetFov() unconverted (:246-252)
            const float aspect = width / float(height), diagonal = 2 * std::tan(0.5f * diag_fov), w = diagonal / std::sqrt(1.0f + 1.0f / (aspect * aspect));
            fov_deg = 2 * std::atan(w * 0.5f);
        };
        if (n.property("focalLength")) { const float fl = prop_f(n, "focalLength"); const float c = std::sqrt((float)(36 * 36 + 24 * 24)); set_diagonal(2 * std::atan(c / (2 * fl))); }
        if (n.property("fov")) {
            const float fov = prop_f(n, "fov"); const std::string axis = prop_s(n, "fovAxis", "x");
            auto vertical = [&](float f) { fov_deg = f * height / (float)width; };
            if (axis == "x") fov_deg = fov; else if (axis == "y") vertical(fov); else if (axis == "diagonal") set_diagonal(fov);
            else if (axis == "smaller") { if (width < height) fov_deg = fov; else vertical(fov); }
            else if (axis == "larger") { if (width < height) vertical(fov); else fov_deg = fov; }
            else bad("invalid fov axis type : " + axis);
        }
        ctl_sensor S; std::memset(&S, 0, sizeof(S));
        f3 pos(0.0f), fwd(0, 0, 1);
        if (const xml_node* tw = n.property("toWorld")) { const mat4 T2 = parse_matrix(*tw); pos = translation(T2); fwd = forward(T2); }
        // Sensor::SetToWorld(pos, f) (SceneTypes/Sensor.cu:680-686): r = normalize(f x (0,1,0)), u = normalize(r x f), then (pos, pos + f, u)
        const f3 f = normalize(fwd), r = normalize(cross(f, f3(0, 1, 0))), u = normalize(cross(r, f)), tar = pos + f;
        const float p3[3] = { pos.x, pos.y, pos.z }, t3[3] = { tar.x, tar.y, tar.z }, u3[3] = { u.x, u.y, u.z };
        B.set_camera_lookat(p3, t3, u3, fov_deg, (uint32_t)width, (uint32_t)height);
        B.camera.near_depth = prop_f(n, "nearClip", 1e-2f); B.camera.far_depth = prop_f(n, "farClip", 10000.0f);
        if (T == "thinlens") {   // ObjectParser.h:309-320: SetApperture(0) in parseGeneric, then focusDistance (default 0) and apertureRadius when present
            B.camera.type = CTL_SENSOR_THINLENS;
            B.camera.focus_distance = prop_f(n, "focusDistance", 0.0f);
            B.camera.aperture_radius = n.property("apertureRadius") ? prop_f(n, "apertureRadius") : 0.0f;
        } else if (T == "orthographic") B.camera.type = CTL_SENSOR_ORTHOGRAPHIC;
        have_sensor = true;
    }

    // ---- emitters (ObjectParser.h:347-594)
    void parse_emitter(const xml_node& n) {
        const std::string T = n.attr("type");
        if (T == "point") {
            const rgb e = try_color(n, "intensity", rgb(1.0f)); f3 pos;
            if (const xml_node* p = n.property("position")) pos = parse_vector(*p); else if (const xml_node* tw = n.property("toWorld")) pos = translation(parse_matrix(*tw)); else bad("no position specified for point light");
            const float p3[3] = { pos.x, pos.y, pos.z }, i3[3] = { e.r, e.g, e.b }; B.add_point_light(p3, i3);
        } else if (T == "spot") {
            const rgb e = try_color(n, "intensity", rgb(1.0f)); const float cutoff = prop_f(n, "cutoffAngle", 20.0f), beam = prop_f(n, "beamWidth", 20 * 0.75f);
            const xml_node* tw = n.property("toWorld"); if (!tw) bad("no property of that name : toWorld");
            const mat4 Tm = parse_matrix(*tw); const f3 p = translation(Tm), t = p + forward(Tm);
            const float p3[3] = { p.x, p.y, p.z }, t3[3] = { t.x, t.y, t.z }, i3[3] = { e.r, e.g, e.b }; B.add_spot_light(p3, t3, i3, cutoff, beam);
        } else if (T == "directional") {
            const rgb e = try_color(n, "irradiance", rgb(1.0f)); f3 d;
            if (const xml_node* p = n.property("direction")) d = parse_vector(*p); else if (const xml_node* tw = n.property("toWorld")) d = forward(parse_matrix(*tw)); else bad("no direction specified for directional light");
            d = normalize(d); const float d3[3] = { d.x, d.y, d.z }, i3[3] = { e.r, e.g, e.b }; B.add_distant_light(d3, i3, 1.0f);
        } else if (T == "envmap") {
            const float sc = prop_f(n, "scale", 1.0f); const mat4 Tm = n.property("toWorld") ? parse_matrix(*n.property("toWorld")) : id_matrix;
            const uint32_t img = load_image(asset(prop_s(n, "filename")), CTL_WRAP_REPEAT, CTL_FILTER_TRILINEAR);
            const float s3[3] = { sc, sc, sc }; ctl_float4x4 x; std::memcpy(x.m, Tm.m, 64); B.set_environment_map(img, s3, &x);
        } else if (T == "constant") {   // :559-583: an inward-facing sphere of the scene's size carrying an area light
            rgb e = try_color(n, "radiance", rgb(1.0f));
            const uint32_t node = node_virtual("sphere", make_sphere);
            const aabb box = B.scene_box();
            const float rad = length(f3(box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]));
            ctl_float4x4 x; std::memcpy(x.m, scale(-rad, -rad, -rad).m, 64); B.set_node_transform(node, x);
            const float k = 4 * 3.14159265358979f * rad * rad; const float r3[3] = { e.r / k, e.g / k, e.b / k };
            B.add_area_light(node, 0, r3);
        } else if (T == "sunsky" || T == "sun") parse_sun(n);
        else if (T == "sky") bad("NOT YET IMPLEMENTED");   // LightParser::sky (:554-557)
        else bad("invalid StreamReference<Light> type : " + T);
    }
    // LightParser::parseSun (ObjectParser.h:354-493): the sun (and "sunsky": the sky part is dropped there too) becomes EIGHT wide spot lights on a
    // small disk half a scene diameter up-sun of the scene centre.  The sun position is Mitsuba's (Blanco-Muriel et al., "Computing the Solar
    // Vector", 2001).  The reference jitters the eight positions with CudaRNG(time(0)), i.e. differently in every run; this loader seeds the
    // same generator with 0 ($CTL_SUN_SEED overrides), as SURVEY §8d fixes it for the San Miguel configs.
    void parse_sun(const xml_node& n) {
        f3 dir;
        if (const xml_node* sd = n.property("sunDirection")) dir = parse_vector(*sd);
        else {
            const int year = prop_i(n, "year", 2010), month = prop_i(n, "month", 7), day = prop_i(n, "day", 10);
            const float hour = prop_f(n, "hour", 15.0f), minute = prop_f(n, "minute", 0.0f), second = prop_f(n, "second", 0.0f);
            const float latitude = prop_f(n, "latitude", 35.6894f), longitude = prop_f(n, "longitude", 139.6917f), timezone = (float)prop_i(n, "timezone", 9);
            const double PI_ = 3.14159265358979323846f;   // the reference's PI is a float constant
            const double decHours = hour - timezone + (minute + second / 60.0) / 60.0;
            const int liAux1 = (month - 14) / 12;
            const int liAux2 = (1461 * (year + 4800 + liAux1)) / 4 + (367 * (month - 2 - 12 * liAux1)) / 12 - (3 * ((year + 4900 + liAux1) / 100)) / 4 + day - 32075;
            const double elapsedJulianDays = ((double)liAux2 - 0.5 + decHours / 24.0) - 2451545.0;
            const double omega = 2.1429 - 0.0010394594 * elapsedJulianDays, meanLongitude = 4.8950630 + 0.017202791698 * elapsedJulianDays, anomaly = 6.2400600 + 0.0172019699 * elapsedJulianDays;
            const double eclipticLongitude = meanLongitude + 0.03341607 * std::sin(anomaly) + 0.00034894 * std::sin(2 * anomaly) - 0.0001134 - 0.0000203 * std::sin(omega);
            const double eclipticObliquity = 0.4090928 - 6.2140e-9 * elapsedJulianDays + 0.0000396 * std::cos(omega);
            const double sinEclipticLongitude = std::sin(eclipticLongitude);
            double dY = std::cos(eclipticObliquity) * sinEclipticLongitude, dX = std::cos(eclipticLongitude);
            double rightAscension = std::atan2(dY, dX);
            if (rightAscension < 0.0) rightAscension += 2 * PI_;
            const double declination = std::asin(std::sin(eclipticObliquity) * sinEclipticLongitude);
            const double greenwichMeanSiderealTime = 6.6974243242 + 0.0657098283 * elapsedJulianDays + decHours;
            auto degToRad = [&](double f) { return f * PI_ / 180.0f; };
            const double localMeanSiderealTime = degToRad(greenwichMeanSiderealTime * 15 + longitude), latitudeInRadians = degToRad(latitude);
            const double cosLatitude = std::cos(latitudeInRadians), sinLatitude = std::sin(latitudeInRadians);
            const double hourAngle = localMeanSiderealTime - rightAscension, cosHourAngle = std::cos(hourAngle);
            double elevation = std::acos(cosLatitude * cosHourAngle * std::cos(declination) + std::sin(declination) * sinLatitude);
            dY = -std::sin(hourAngle); dX = std::tan(declination) * cosLatitude - sinLatitude * cosHourAngle;
            double azimuth = std::atan2(dY, dX);
            if (azimuth < 0.0) azimuth += 2 * PI_;
            const float EARTH_MEAN_RADIUS = 6371.01f, ASTRONOMICAL_UNIT = 149597890.0f;
            elevation += (EARTH_MEAN_RADIUS / ASTRONOMICAL_UNIT) * std::sin(elevation);
            const float sinTheta = sinf((float)elevation), cosTheta = cosf((float)elevation), sinPhi = sinf((float)azimuth), cosPhi = cosf((float)azimuth);
            dir = f3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
        }
        const float scale_ = prop_f(n, "scale", 1.0f), sunRadiusScale = prop_f(n, "sunRadiusScale", 1.0f);
        if (const xml_node* tw = n.property("toWorld")) dir = xf_dir(parse_matrix(*tw), dir);
        const aabb box = B.scene_box();   // of what has been loaded so far, as in the reference
        const f3 size(box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]), center((box.hi[0] + box.lo[0]) * 0.5f, (box.hi[1] + box.lo[1]) * 0.5f, (box.hi[2] + box.lo[2]) * 0.5f);
        const float scene_rad = length(size);
        const char* seed_env = std::getenv("CTL_SUN_SEED");
        xorwow rng; rng.init(1234, seed_env ? (uint64_t)std::strtoull(seed_env, nullptr, 10) : 0);   // CudaRNG(seed) = curand_init(1234, seed, 0)
        const int N_lights = 8;
        const float EARTH_MEAN_RADIUS = 6371.01f;
        const float rel_radius = scene_rad / EARTH_MEAN_RADIUS * sunRadiusScale * 10;
        const f3 nd = normalize(dir);
        f3 s, t; coordinate_system(nd, s, t);
        for (int i = 0; i < N_lights; i++) {
            f3 p = center - nd * scene_rad / 2;
            const float jx = 2 * rng.uniform() - 1, jy = 2 * rng.uniform() - 1;
            p = p + (s * jx + t * jy) * rel_radius;
            const float e = scale_ * scene_rad * scene_rad / N_lights;
            const float p3[3] = { p.x, p.y, p.z }, c3[3] = { center.x, center.y, center.z }, i3[3] = { e, e, e };
            B.add_spot_light(p3, c3, i3, 90.0f, 90.0f);
        }
    }

    void parse_file(const std::string& file);
    int include_depth = 0;   // <include> chains: a file that (directly or through others) includes itself is refused instead of recursing until the stack is gone
};

void loader::parse_file(const std::string& file) {
    struct level { int& d; explicit level(int& x) : d(x) { d++; } ~level() { d--; } } guard(include_depth);
    if (include_depth > 32) throw io_error("couldn't loader scene xml! (<include> nested more than 32 deep: " + file + ")");
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) throw io_error("couldn't loader scene xml! (" + file + ")");
    std::string text; char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    std::fclose(f);
    const xml_node doc = xml_parser(text).parse();
    const xml_node* scene = doc.child("scene");
    if (!scene) bad("couldn't loader scene xml! (no <scene> element)");
    for (const xml_node& n : scene->children) {   // MitsubaLoader.cpp:22-63
        const std::string t = n.lname();
        if (t == "include") parse_file(asset(attr_s(n, "filename")));
        else if (t == "default") defaults[n.attr("name")] = n.attr("value");
        else if (t == "alias") {
            const std::string id = n.attr("id"), as = n.attr("as");
            if (ref_bsdf.count(id)) ref_bsdf[as] = ref_bsdf[id]; if (ref_tex.count(id)) ref_tex[as] = ref_tex[id]; if (ref_mat.count(id)) ref_mat[as] = ref_mat[id];
            if (ref_rgb.count(id)) ref_rgb[as] = ref_rgb[id]; if (ref_vec.count(id)) ref_vec[as] = ref_vec[id]; if (ref_shape.count(id)) ref_shape[as] = ref_shape[id];
        }
        else if (t == "sensor") parse_sensor(n);
        else if (t == "emitter") parse_emitter(n);
        else if (t == "bsdf") parse_bsdf(n, 0);
        else if (t == "shape") parse_shape(n);
        else if (t == "texture") parse_texture(n);
        else if (t == "medium") unsupported("<medium>");
        // other elements (integrator, ...) are ignored, as in the reference
    }
}

} // namespace

void parse_mitsuba_scene(scene_builder& b, const char* xml_path, int32_t* width_inout, int32_t* height_inout) {
    const std::string path(xml_path);
    const size_t s = path.find_last_of("/\\");
    loader L(b, s == std::string::npos ? "." : path.substr(0, s));
    L.parse_file(path);
    if (!L.have_sensor) throw std::runtime_error("ParseMitsubaScene: the scene has no <sensor>");
    int w = L.film_w, h = L.film_h;
    if (width_inout && height_inout && *width_inout > 0 && *height_inout > 0) {   // caller override of the film size (main.cpp resizes the tracer, not the sensor)
        w = *width_inout; h = *height_inout;
        b.camera.resolution[0] = (float)w; b.camera.resolution[1] = (float)h;
    }
    if (width_inout) *width_inout = w;
    if (height_inout) *height_inout = h;
}

} // namespace ctl
