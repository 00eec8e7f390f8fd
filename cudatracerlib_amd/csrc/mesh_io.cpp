// mesh_io.cpp — see mesh_io.h.
#include "mesh_io.h"
#include "mitsuba_loader.h"     // io_error / unsupported_error
#include "material_factory.h"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <array>
#include <sstream>
#include <algorithm>

namespace ctl {

static std::vector<uint8_t> slurp(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw io_error("Could not open file : " + path);
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); throw io_error("short read : " + path); }
    std::fclose(f);
    return buf;
}
static std::string dir_of(const std::string& path) { size_t s = path.find_last_of("/\\"); return s == std::string::npos ? "." : path.substr(0, s); }
static std::string ext_of(const std::string& path) {
    size_t d = path.find_last_of('.'); std::string e = d == std::string::npos ? "" : path.substr(d + 1);
    for (auto& c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}

// ------------------------------------------------------------------------------------------------ OBJ (+ MTL)
// Semantics of Engine/MeshLoader/ObjParser.cpp: vertices are unique (p, t, n) index triples; vt is stored as (u, 1 - v) (:627);
// polygons are fanned (0, i-1, i) (:699-700); faces before any known `usemtl` go to a default sub-mesh with diffuse 0.75
// (:235,242); every triangle is emitted with reversed index order (:861-866); normals are used only if the file has `vn`.
struct mtl_rec {
    std::string name; float kd[3] = { 0.75f, 0.75f, 0.75f }, ks[3] = { 0, 0, 0 }, ke[3] = { 0, 0, 0 }, tf[3] = { 0, 0, 0 }; float ns = 0, ni = 1; int illum = 2; int submesh = -1;
    std::string map_kd, map_ks, map_alpha, map_disp;   // ObjMaterial::textures[TextureType_Diffuse / Specular / Alpha / Displacement]
};
// CreateTexture(path, colour) (ObjParser.cpp:583-588): a bitmap when the .mtl names one, else the constant
static ctl_texture mtl_texture(const std::string& file, const float* col, std::vector<std::string>& files) {
    if (file.empty()) return tex_const(col[0], col[1], col[2]);
    ctl_texture t = tex_const(1.0f); t.type = CTL_TEX_IMAGE;   // ImageTexture(TextureMapping2D(), file, Spectrum(1))
    size_t i = 0; while (i < files.size() && files[i] != file) i++;
    if (i == files.size()) files.push_back(file);
    t.image = (uint32_t)i;
    return t;
}
static ctl_material mtl_to_material(const mtl_rec& M, std::vector<std::string>& files) {   // ObjParser.cpp:808-850
    const float zero[3] = { 0, 0, 0 };
    ctl_material m;
    if (M.illum == 5) { const float e[3] = { 0, 0, 0 }, k[3] = { 1, 1, 1 }; m = make_conductor(e, k, tex_const(1.0f)); }
    else if (M.illum == 7) m = make_dielectric(M.ni, tex_const(M.ks[0], M.ks[1], M.ks[2]), tex_const(M.tf[0], M.tf[1], M.tf[2]));
    else if (M.illum == 9) m = make_dielectric(M.ni, tex_const(0.0f), tex_const(M.tf[0], M.tf[1], M.tf[2]));
    else if (M.illum == 2 && (M.ks[0] != 0 || M.ks[1] != 0 || M.ks[2] != 0 || !M.map_ks.empty()))
        m = make_phong(mtl_texture(M.map_kd, M.kd, files), mtl_texture(M.map_ks, M.ks, files), tex_const(M.ns));
    else if (M.illum == 2) m = make_diffuse(mtl_texture(M.map_kd, M.kd, files));
    else m = make_diffuse(tex_const(M.kd[0], M.kd[1], M.kd[2]));   // other models leave the Material's default BSDF; the build shades them diffuse
    if (!M.map_disp.empty()) { m.map_kind = CTL_MAP_HEIGHT; m.map_tex = mtl_texture(M.map_disp, zero, files); }                      // SetHeightMap (:842-845)
    if (!M.map_alpha.empty()) { m.alpha_state = CTL_ALPHA_MAP_LUMINANCE; m.alpha_tex = mtl_texture(M.map_alpha, zero, files); m.alpha_test_scalar = 1.0f; }   // :846-850
    return m;
}
static void load_mtl(const std::string& path, std::vector<mtl_rec>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return;   // a missing material library leaves every face in the default sub-mesh
    char line[4096]; mtl_rec* cur = nullptr;
    while (std::fgets(line, sizeof(line), f)) {
        std::istringstream ss(line); std::string key; ss >> key;
        if (key == "newmtl") { out.emplace_back(); cur = &out.back(); ss >> cur->name; }
        else if (!cur) continue;
        else if (key == "Kd") ss >> cur->kd[0] >> cur->kd[1] >> cur->kd[2];
        else if (key == "Ks") ss >> cur->ks[0] >> cur->ks[1] >> cur->ks[2];
        else if (key == "Ke") ss >> cur->ke[0] >> cur->ke[1] >> cur->ke[2];
        else if (key == "Tf") ss >> cur->tf[0] >> cur->tf[1] >> cur->tf[2];
        else if (key == "Ns") { ss >> cur->ns; if (cur->ns <= 0.0f) { cur->ns = 1.0f; cur->ks[0] = cur->ks[1] = cur->ks[2] = 0.0f; } }   // :487-491
        else if (key == "map_Kd" || key == "map_Ks" || key == "map_d" || key == "map_D" || key == "map_opacity" || key == "disp" || key == "bump" || key == "map_bump" || key == "map_Bump") {
            // the file name is the last token (options such as "-bm 0.5" precede it); resolved against the .mtl's directory when it exists there (get_texture_path, :366-371)
            std::string tok, file; while (ss >> tok) file = tok;
            if (file.empty()) continue;
            const std::string local = dir_of(path) + "/" + file;
            if (FILE* t = std::fopen(local.c_str(), "rb")) { std::fclose(t); file = local; }
            (key == "map_Kd" ? cur->map_kd : key == "map_Ks" ? cur->map_ks : (key == "map_d" || key == "map_D" || key == "map_opacity") ? cur->map_alpha : cur->map_disp) = file;
        }
        else if (key == "Ni") ss >> cur->ni;
        else if (key == "illum") ss >> cur->illum;
    }
    std::fclose(f);
}
mesh_data load_obj(const std::string& path) {
    const std::vector<uint8_t> buf = slurp(path);
    std::vector<std::array<float, 3>> P, N; std::vector<std::array<float, 2>> T;
    std::map<std::array<int, 3>, uint32_t> vhash;
    struct sub { mtl_rec mat; std::vector<std::array<uint32_t, 3>> tris; };
    std::vector<sub> subs; std::vector<mtl_rec> mtls;
    int cur = -1, def = -1;
    mesh_data M;
    std::vector<float> vp, vn, vt;
    bool any_vn = false, any_vt = false;
    const char* p = (const char*)buf.data(); const char* end = p + buf.size();
    std::vector<uint32_t> poly;
    while (p < end) {
        const char* e = (const char*)std::memchr(p, '\n', end - p); if (!e) e = end;
        std::string line(p, e); p = e + 1;
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        size_t s = 0; while (s < line.size() && std::isspace((unsigned char)line[s])) s++;
        if (s >= line.size() || line[s] == '#') continue;
        const char* c = line.c_str() + s;
        if (c[0] == 'v' && c[1] == ' ') { std::array<float, 3> v{}; std::sscanf(c + 2, "%f %f %f", &v[0], &v[1], &v[2]); P.push_back(v); }
        else if (c[0] == 'v' && c[1] == 't' && c[2] == ' ') { std::array<float, 2> v{}; std::sscanf(c + 3, "%f %f", &v[0], &v[1]); v[1] = 1.0f - v[1]; T.push_back(v); any_vt = true; }
        else if (c[0] == 'v' && c[1] == 'n' && c[2] == ' ') { std::array<float, 3> v{}; std::sscanf(c + 3, "%f %f %f", &v[0], &v[1], &v[2]); N.push_back(v); any_vn = true; }
        else if (c[0] == 'f' && c[1] == ' ') {
            poly.clear();
            const char* q = c + 2;
            while (*q) {
                while (*q == ' ' || *q == '\t') q++;
                if (!*q) break;
                std::array<int, 3> ptn = { 0, 0, 0 };
                char* nx; ptn[0] = (int)std::strtol(q, &nx, 10); if (nx == q) break; q = nx;
                for (int i = 1; i < 3 && *q == '/'; i++) { q++; ptn[i] = (int)std::strtol(q, &nx, 10); q = nx; }
                const int size[3] = { (int)P.size(), (int)T.size(), (int)N.size() };
                for (int i = 0; i < 3; i++) { if (ptn[i] < 0) ptn[i] += size[i]; else ptn[i]--; if (ptn[i] < 0 || ptn[i] >= size[i]) ptn[i] = -1; }
                auto it = vhash.find(ptn);
                if (it != vhash.end()) poly.push_back(it->second);
                else {
                    const uint32_t idx = (uint32_t)(vp.size() / 3); vhash[ptn] = idx; poly.push_back(idx);
                    for (int k = 0; k < 3; k++) vp.push_back(ptn[0] == -1 ? 0.0f : P[ptn[0]][k]);
                    for (int k = 0; k < 2; k++) vt.push_back(ptn[1] == -1 ? 0.0f : T[ptn[1]][k]);
                    for (int k = 0; k < 3; k++) vn.push_back(ptn[2] == -1 ? 0.0f : N[ptn[2]][k]);
                }
            }
            if (cur == -1) { if (def == -1) { subs.emplace_back(); def = (int)subs.size() - 1; } cur = def; }
            for (size_t i = 2; i < poly.size(); i++) subs[cur].tris.push_back({ poly[0], poly[i - 1], poly[i] });
        }
        else if (!std::strncmp(c, "usemtl ", 7)) {
            std::string name(c + 7); while (!name.empty() && std::isspace((unsigned char)name.front())) name.erase(name.begin());
            cur = -1;
            for (auto& m : mtls) if (m.name == name) { if (m.submesh == -1) { subs.emplace_back(); subs.back().mat = m; m.submesh = (int)subs.size() - 1; } cur = m.submesh; }
        }
        else if (!std::strncmp(c, "mtllib ", 7)) { std::string name(c + 7); while (!name.empty() && std::isspace((unsigned char)name.front())) name.erase(name.begin()); load_mtl(dir_of(path) + "/" + name, mtls); }
    }
    M.positions = std::move(vp);
    if (any_vn) M.normals = std::move(vn);
    if (any_vt) M.uvs = std::move(vt);
    for (size_t si = 0; si < subs.size(); si++) {
        if (subs[si].tris.empty() && subs.size() > 1) continue;
        if (M.materials.size() >= 255) throw unsupported_error("OBJ with more than 255 materials : " + path);
        const uint8_t mi = (uint8_t)M.materials.size();
        M.materials.push_back(mtl_to_material(subs[si].mat, M.image_files));
        for (int k = 0; k < 3; k++) M.emission.push_back(subs[si].mat.ke[k]);
        for (auto& t : subs[si].tris) { M.indices.push_back(t[2]); M.indices.push_back(t[1]); M.indices.push_back(t[0]); M.tri_material.push_back(mi); }
    }
    if (M.indices.empty()) throw io_error("OBJ file has no faces : " + path);
    return M;
}

// ------------------------------------------------------------------------------------------------ PLY
// Engine/MeshLoader/PlyParser.cpp: positions (+ u,v / s,t), faces fanned, index order reversed (:314-316,347), no normals
// (CompileMesh computes them, :371), one default diffuse material.
mesh_data load_ply(const std::string& path) {
    const std::vector<uint8_t> buf = slurp(path);
    size_t p = 0; auto line = [&]() { std::string l; while (p < buf.size() && buf[p] != '\n') l += (char)buf[p++]; p++; if (!l.empty() && l.back() == '\r') l.pop_back(); return l; };
    if (line() != "ply") throw io_error("not a PLY file : " + path);
    enum { ASCII, BLE, BBE } fmt = ASCII;
    struct prop { std::string name, type, count_type; bool list = false; };
    struct elem { std::string name; size_t count = 0; std::vector<prop> props; };
    std::vector<elem> elems;
    for (;;) {
        if (p >= buf.size()) throw io_error("truncated PLY header : " + path);
        std::istringstream ss(line()); std::string k; ss >> k;
        if (k == "format") { std::string f; ss >> f; fmt = f == "ascii" ? ASCII : (f == "binary_little_endian" ? BLE : BBE); }
        else if (k == "element") { elems.emplace_back(); ss >> elems.back().name >> elems.back().count; }
        else if (k == "property" && !elems.empty()) { prop pr; std::string t; ss >> t; if (t == "list") { pr.list = true; ss >> pr.count_type >> pr.type >> pr.name; } else { pr.type = t; ss >> pr.name; } elems.back().props.push_back(pr); }
        else if (k == "end_header") break;
    }
    auto tsize = [](const std::string& t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1; if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4; if (t == "double" || t == "float64") return 8;
        return 0; };
    auto read_num = [&](const std::string& t) -> double {
        if (fmt == ASCII) {
            while (p < buf.size() && std::isspace(buf[p])) p++;
            if (p >= buf.size()) throw io_error("truncated PLY : " + path);   // (a header that promises more elements than the file holds ends here, not in an endless row of zeros)
            size_t b = p; while (p < buf.size() && !std::isspace(buf[p])) p++; return std::strtod(std::string((const char*)&buf[b], p - b).c_str(), nullptr); }
        const int n = tsize(t); if (!n || p + n > buf.size()) throw io_error("truncated PLY : " + path);
        uint8_t b[8]; for (int i = 0; i < n; i++) b[i] = buf[p + (fmt == BLE ? i : n - 1 - i)]; p += n;
        if (t == "float" || t == "float32") { float f; std::memcpy(&f, b, 4); return f; }
        if (t == "double" || t == "float64") { double d; std::memcpy(&d, b, 8); return d; }
        if (t == "char" || t == "int8") return (int8_t)b[0]; if (t == "uchar" || t == "uint8") return b[0];
        if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); return v; } if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); return v; }
        if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); return v; }
        uint32_t v; std::memcpy(&v, b, 4); return v; };
    // entries of a list property: never negative, never more than the bytes that are left could hold
    auto read_count = [&](const prop& pr) -> int {
        const double c = read_num(pr.count_type);
        const size_t left = buf.size() - std::min(p, buf.size()), each = fmt == ASCII ? 1 : (size_t)std::max(1, tsize(pr.type));
        if (!(c >= 0) || c > (double)(left / each)) throw io_error("corrupt PLY list length : " + path);
        return (int)c; };
    mesh_data M; bool has_uv = false;
    for (auto& e : elems) {
        if (e.count > buf.size() - std::min(p, buf.size())) throw io_error("truncated PLY (the header promises more elements than the file holds) : " + path);
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, iu = -1, iv = -1;
            for (size_t i = 0; i < e.props.size(); i++) { const std::string& n = e.props[i].name; if (n == "x") ix = (int)i; else if (n == "y") iy = (int)i; else if (n == "z") iz = (int)i; else if (n == "u" || n == "s") iu = (int)i; else if (n == "v" || n == "t") iv = (int)i; }
            if (ix < 0 || iy < 0 || iz < 0) throw io_error("PLY without vertex positions : " + path);
            has_uv = iu >= 0 && iv >= 0;
            std::vector<double> vals(e.props.size());
            for (size_t k = 0; k < e.count; k++) {
                for (size_t i = 0; i < e.props.size(); i++) { if (e.props[i].list) { const int n = read_count(e.props[i]); for (int j = 0; j < n; j++) read_num(e.props[i].type); vals[i] = 0; } else vals[i] = read_num(e.props[i].type); }
                M.positions.push_back((float)vals[ix]); M.positions.push_back((float)vals[iy]); M.positions.push_back((float)vals[iz]);
                if (has_uv) { M.uvs.push_back((float)vals[iu]); M.uvs.push_back((float)vals[iv]); }
            }
        } else if (e.name == "face") {
            std::vector<uint32_t> poly;
            for (size_t k = 0; k < e.count; k++)
                for (auto& pr : e.props) {
                    if (!pr.list) { read_num(pr.type); continue; }
                    const int n = read_count(pr); poly.resize(n);
                    for (int j = 0; j < n; j++) poly[j] = (uint32_t)read_num(pr.type);
                    if (pr.name != "vertex_indices" && pr.name != "vertex_index") continue;
                    for (int i = 2; i < n; i++) { M.indices.push_back(poly[i]); M.indices.push_back(poly[i - 1]); M.indices.push_back(poly[0]); }
                }
        } else {
            for (size_t k = 0; k < e.count; k++) for (auto& pr : e.props) { if (pr.list) { const int n = read_count(pr); for (int j = 0; j < n; j++) read_num(pr.type); } else read_num(pr.type); }
        }
    }
    const uint32_t nv = M.n_vertices();
    for (auto& i : M.indices) if (i >= nv) i = 0;   // PlyParser.cpp:358
    if (M.indices.empty() || !nv) throw io_error("PLY file has no faces : " + path);
    M.tri_material.assign(M.n_triangles(), 0);
    M.materials.push_back(make_diffuse(tex_const(0.75f)));
    M.emission.assign(3, 0.0f);
    return M;
}

// ------------------------------------------------------------------------------------------------ Mitsuba .serialized
// ObjectParser.cpp:9-204: file magic 0x041C, version 3 or 4, a table of sub-mesh offsets at the end, each sub-mesh a zlib stream:
// flags, [name], #vertices, #triangles (u64), positions, [normals], [uvs], [colours], u32 indices (winding reversed, :187-188).
mesh_data load_serialized(const std::string& path, int shape_index) {
    const std::vector<uint8_t> d = slurp(path);
    auto u16 = [&](size_t o) { return (uint16_t)(d[o] | (d[o + 1] << 8)); };
    if (d.size() < 8 || u16(0) != 1052) throw io_error("corrupt file : " + path);
    const uint16_t file_version = u16(2);
    uint32_t n_meshes; std::memcpy(&n_meshes, &d[d.size() - 4], 4);
    const size_t osz = file_version == 4 ? 8 : 4;
    if (shape_index < 0 || (uint32_t)shape_index >= n_meshes || d.size() < 4 + osz * n_meshes) throw io_error("serialized mesh: shapeIndex out of range : " + path);
    uint64_t off = 0;
    const size_t tab = d.size() - 4 - osz * n_meshes + osz * (size_t)shape_index;
    if (osz == 8) std::memcpy(&off, &d[tab], 8); else { uint32_t o32; std::memcpy(&o32, &d[tab], 4); off = o32; }
    if (d.size() < 8 || off >= d.size() - 4 || u16(off) != 1052) throw io_error("corrupt sub-mesh header : " + path);
    const uint16_t version = u16(off + 2);
    if (version != 3 && version != 4) throw io_error("invalid version in serialized mesh file");
    z_stream zs; std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15) != Z_OK) throw io_error("inflateInit failed");
    zs.next_in = (Bytef*)&d[off + 4]; zs.avail_in = (uInt)std::min<size_t>(d.size() - off - 4, 0xffffffffu);
    auto zread = [&](void* dst, size_t n) {
        zs.next_out = (Bytef*)dst; zs.avail_out = (uInt)n;
        while (zs.avail_out) { const int r = inflate(&zs, Z_NO_FLUSH); if (r == Z_STREAM_END && zs.avail_out) { inflateEnd(&zs); throw io_error("inflate(): attempting to read past the end of the stream!"); } if (r != Z_OK && r != Z_STREAM_END) { inflateEnd(&zs); throw io_error("inflate(): data error!"); } }
    };
    uint32_t flag; zread(&flag, 4);
    if (version == 4) { char c; do zread(&c, 1); while (c != 0); }
    uint64_t nv, nt; zread(&nv, 8); zread(&nt, 8);
    // deflate expands by at most ~1032 : 1: a header that promises more vertex / index bytes than the rest of the stream can inflate to is refused before the arrays are sized
    const double inflatable = 1032.0 * (double)(d.size() - off) + 65536.0;
    if (nv > (1ull << 31) || nt > (1ull << 31) || (double)nv * 12.0 + (double)nt * 12.0 > inflatable) { inflateEnd(&zs); throw io_error("serialized mesh: the header promises more data than the stream holds : " + path); }
    const bool dbl = (flag & 0x2000) != 0;
    mesh_data M;
    auto read_vec = [&](int dim, std::vector<float>& out) {
        out.resize((size_t)nv * dim);
        if (!dbl) zread(out.data(), out.size() * 4);
        else { std::vector<double> tmp(out.size()); zread(tmp.data(), tmp.size() * 8); for (size_t i = 0; i < tmp.size(); i++) out[i] = (float)tmp[i]; }
    };
    read_vec(3, M.positions);
    if (flag & 0x0001) read_vec(3, M.normals); else M.normals.assign((size_t)nv * 3, 0.0f);   // the reference always hands a normals array over (:183)
    if (flag & 0x0002) read_vec(2, M.uvs); else M.uvs.assign((size_t)nv * 2, 0.0f);
    if (flag & 0x0008) { std::vector<float> colours; read_vec(3, colours); }
    M.indices.resize((size_t)nt * 3); zread(M.indices.data(), M.indices.size() * 4);
    inflateEnd(&zs);
    for (size_t i = 0; i < M.indices.size(); i += 3) std::swap(M.indices[i], M.indices[i + 2]);
    M.tri_material.assign((size_t)nt, 0);
    M.materials.push_back(make_diffuse(tex_const(0.5f)));   // diffuse() default reflectance (BSDF_Simple.h:11-15)
    M.emission.assign(3, 0.0f);
    return M;
}

mesh_data load_mesh_file(const std::string& path) {
    const std::string e = ext_of(path);
    if (e == "obj") return load_obj(path);
    if (e == "ply") return load_ply(path);
    throw unsupported_error("mesh format not supported : " + path);
}

// ------------------------------------------------------------------------------------------------ primitives
static void push_v(mesh_data& M, float x, float y, float z, float nx, float ny, float nz, float u, float v) {
    M.positions.insert(M.positions.end(), { x, y, z }); M.normals.insert(M.normals.end(), { nx, ny, nz }); M.uvs.insert(M.uvs.end(), { u, v });
}
static void push_t(mesh_data& M, uint32_t a, uint32_t b, uint32_t c) {   // a,b,c counter-clockwise seen from outside; stored reversed like every importer
    M.indices.insert(M.indices.end(), { c, b, a }); M.tri_material.push_back(0);
}
static void finish(mesh_data& M) { M.materials.push_back(make_diffuse(tex_const(0.75f))); M.emission.assign(3, 0.0f); }

mesh_data make_plane() {
    mesh_data M;
    push_v(M, -1, 0, 1, 0, 1, 0, 0, 0); push_v(M, 1, 0, 1, 0, 1, 0, 1, 0); push_v(M, 1, 0, -1, 0, 1, 0, 1, 1); push_v(M, -1, 0, -1, 0, 1, 0, 0, 1);
    push_t(M, 0, 1, 2); push_t(M, 0, 2, 3);
    finish(M); return M;
}
mesh_data make_cube() {
    mesh_data M;
    const float n[6][3] = { { 0, 0, 1 }, { 0, 0, -1 }, { 0, 1, 0 }, { 0, -1, 0 }, { 1, 0, 0 }, { -1, 0, 0 } };
    for (int f = 0; f < 6; f++) {
        const float* nn = n[f]; float s[3], t[3];   // s x t = n
        if (nn[0] != 0) { s[0] = 0; s[1] = nn[0]; s[2] = 0; t[0] = 0; t[1] = 0; t[2] = 1; }
        else if (nn[1] != 0) { s[0] = 0; s[1] = 0; s[2] = nn[1]; t[0] = 1; t[1] = 0; t[2] = 0; }
        else { s[0] = nn[2]; s[1] = 0; s[2] = 0; t[0] = 0; t[1] = 1; t[2] = 0; }
        const uint32_t b = M.n_vertices();
        for (int k = 0; k < 4; k++) {
            const float a = (k == 1 || k == 2) ? 0.5f : -0.5f, c = (k >= 2) ? 0.5f : -0.5f;
            push_v(M, 0.5f + 0.5f * nn[0] + a * s[0] + c * t[0], 0.5f + 0.5f * nn[1] + a * s[1] + c * t[1], 0.5f + 0.5f * nn[2] + a * s[2] + c * t[2], nn[0], nn[1], nn[2], a + 0.5f, c + 0.5f);
        }
        push_t(M, b, b + 1, b + 2); push_t(M, b, b + 2, b + 3);
    }
    finish(M); return M;
}
mesh_data make_sphere() {   // 16 segments x 8 rings like the reference's sphere text (114 vertices, 224 triangles)
    mesh_data M; const int S = 16, R = 8; const float pi = 3.14159265358979f;
    push_v(M, 0, 1, 0, 0, 1, 0, 0.5f, 0);
    for (int r = 1; r < R; r++) for (int s = 0; s < S; s++) {
        const float th = pi * r / R, ph = 2 * pi * s / S; const float x = sinf(th) * cosf(ph), y = cosf(th), z = sinf(th) * sinf(ph);
        push_v(M, x, y, z, x, y, z, (float)s / S, (float)r / R);
    }
    push_v(M, 0, -1, 0, 0, -1, 0, 0.5f, 1);
    const uint32_t south = M.n_vertices() - 1;
    auto ring = [&](int r, int s) { return (uint32_t)(1 + (r - 1) * S + (s % S)); };
    for (int s = 0; s < S; s++) { push_t(M, 0, ring(1, s + 1), ring(1, s)); push_t(M, south, ring(R - 1, s), ring(R - 1, s + 1)); }
    for (int r = 1; r < R - 1; r++) for (int s = 0; s < S; s++) { push_t(M, ring(r, s), ring(r, s + 1), ring(r + 1, s + 1)); push_t(M, ring(r, s), ring(r + 1, s + 1), ring(r + 1, s)); }
    finish(M); return M;
}
mesh_data make_disk() {
    mesh_data M; const int S = 64; const float pi = 3.14159265358979f;
    push_v(M, 0, 0, 0, 0, 0, 1, 0.5f, 0.5f);
    for (int s = 0; s < S; s++) { const float a = 2 * pi * s / S; push_v(M, cosf(a), sinf(a), 0, 0, 0, 1, 0.5f + 0.5f * cosf(a), 0.5f + 0.5f * sinf(a)); }
    for (int s = 0; s < S; s++) push_t(M, 0, 1 + s, 1 + (s + 1) % S);
    finish(M); return M;
}
mesh_data make_cylinder() {
    mesh_data M; const int S = 64; const float pi = 3.14159265358979f;
    for (int s = 0; s < S; s++) { const float a = 2 * pi * s / S, x = cosf(a), y = sinf(a); push_v(M, x, y, 0, x, y, 0, (float)s / S, 0); push_v(M, x, y, 2, x, y, 0, (float)s / S, 1); }
    for (int s = 0; s < S; s++) { const uint32_t a = 2 * s, b = 2 * ((s + 1) % S); push_t(M, a, b, b + 1); push_t(M, a, b + 1, a + 1); }
    finish(M); return M;
}

} // namespace ctl
