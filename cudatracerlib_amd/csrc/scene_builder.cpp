// scene_builder.cpp — host scene assembly behind the ctl_builder_* C-ABI: the subset of the reference's
// DynamicScene that the Mitsuba loader drives (Engine/DynamicScene.h:70-187), emitting the reference's
// KernelDynamicScene arrays (Engine/KernelDynamicScene.h:28-109) as a ctl_scene_desc.
#include "scene_builder.h"
#include "scene_cache.h"
#include "material_textures.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include "ctl_math.h"
#include <cstring>
#include <algorithm>
#include <stdexcept>

namespace ctl {

// float4x4::inverse (Math/float4x4.h:132-193) — cofactor expansion in the reference's expression order
void mat_inverse(const float* Q, float* out) {
#define M(i, j) Q[(i) * 4 + (j)]
    float m00 = M(0, 0), m01 = M(0, 1), m02 = M(0, 2), m03 = M(0, 3), m10 = M(1, 0), m11 = M(1, 1), m12 = M(1, 2), m13 = M(1, 3);
    float m20 = M(2, 0), m21 = M(2, 1), m22 = M(2, 2), m23 = M(2, 3), m30 = M(3, 0), m31 = M(3, 1), m32 = M(3, 2), m33 = M(3, 3);
#undef M
    float v0 = m20 * m31 - m21 * m30, v1 = m20 * m32 - m22 * m30, v2 = m20 * m33 - m23 * m30;
    float v3 = m21 * m32 - m22 * m31, v4 = m21 * m33 - m23 * m31, v5 = m22 * m33 - m23 * m32;
    float t00 = +(v5 * m11 - v4 * m12 + v3 * m13), t10 = -(v5 * m10 - v2 * m12 + v1 * m13);
    float t20 = +(v4 * m10 - v2 * m11 + v0 * m13), t30 = -(v3 * m10 - v1 * m11 + v0 * m12);
    float invDet = 1 / (t00 * m00 + t10 * m01 + t20 * m02 + t30 * m03);
    float d00 = t00 * invDet, d10 = t10 * invDet, d20 = t20 * invDet, d30 = t30 * invDet;
    float d01 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d11 = +(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d21 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d31 = +(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m10 * m31 - m11 * m30; v1 = m10 * m32 - m12 * m30; v2 = m10 * m33 - m13 * m30;
    v3 = m11 * m32 - m12 * m31; v4 = m11 * m33 - m13 * m31; v5 = m12 * m33 - m13 * m32;
    float d02 = +(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d12 = -(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d22 = +(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d32 = -(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    v0 = m21 * m10 - m20 * m11; v1 = m22 * m10 - m20 * m12; v2 = m23 * m10 - m20 * m13;
    v3 = m22 * m11 - m21 * m12; v4 = m23 * m11 - m21 * m13; v5 = m23 * m12 - m22 * m13;
    float d03 = -(v5 * m01 - v4 * m02 + v3 * m03) * invDet, d13 = +(v5 * m00 - v2 * m02 + v1 * m03) * invDet;
    float d23 = -(v4 * m00 - v2 * m01 + v0 * m03) * invDet, d33 = +(v3 * m00 - v1 * m01 + v0 * m02) * invDet;
    float r[16] = { d00, d01, d02, d03, d10, d11, d12, d13, d20, d21, d22, d23, d30, d31, d32, d33 };
    std::memcpy(out, r, sizeof(r));
}
void mat_mul(const float* l, const float* r, float* o) {   // float4x4.h:373-381
    float t[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0.0f; for (int k = 0; k < 4; k++) s += l[i * 4 + k] * r[k * 4 + j]; t[i * 4 + j] = s; }
    std::memcpy(o, t, sizeof(t));
}
static m34 as_m34(const float* m) { m34 r; std::memcpy(r.r, m, 48); return r; }

// Woop rows from the three vertices (Engine/TriIntersectorData.cu:5-18)
void woop_set_data(ctl_woop_tri& w, f3 a, f3 b, f3 c) {
    f3 e0 = a - c, e1 = b - c, n = cross(a - c, b - c);
    float m[16] = { e0.x, e1.x, n.x, c.x, e0.y, e1.y, n.y, c.y, e0.z, e1.z, n.z, c.z, 0, 0, 0, 1 }, inv[16];
    mat_inverse(m, inv);
    w.a[0] = inv[8]; w.a[1] = inv[9]; w.a[2] = inv[10]; w.a[3] = -inv[11];
    for (int j = 0; j < 4; j++) { w.b[j] = inv[j]; w.c[j] = inv[4 + j]; }
}
// and back (Engine/TriIntersectorData.cu:20-32) — area lights take their vertices from this round trip
void woop_get_data(const ctl_woop_tri& w, f3& v0, f3& v1, f3& v2) {
    float m[16] = { w.b[0], w.b[1], w.b[2], w.b[3], w.c[0], w.c[1], w.c[2], w.c[3], w.a[0], w.a[1], w.a[2], w.a[3] * -1.0f, 0, 0, 0, 1 }, inv[16];
    mat_inverse(m, inv);
    f3 e02(inv[0], inv[4], inv[8]), e12(inv[1], inv[5], inv[9]);
    v2 = f3(inv[3], inv[7], inv[11]);
    v0 = v2 + e02; v1 = v2 + e12;
}

// TriangleData::setUvSetData + setData (Engine/TriangleData.cu:18-65)
static void tri_data_pack(ctl_triangle_data& T, const f3 p[3], const f3 n[3], const f2 t[3], uint32_t mat_index) {
    std::memset(&T, 0, sizeof(T));
    T.nor_mat_extra[1] = (mat_index & 0xff) << 16;
    for (int i = 0; i < 3; i++) T.uv[i] = (uint32_t)float_to_half(t[i].x) | ((uint32_t)float_to_half(t[i].y) << 16);
    auto h = [](uint32_t bits) { return half_to_float((uint16_t)bits); };
    f2 t0{ h(T.uv[0]), h(T.uv[0] >> 16) }, t1{ h(T.uv[1]), h(T.uv[1] >> 16) }, t2{ h(T.uv[2]), h(T.uv[2] >> 16) };
    f3 dP1 = p[1] - p[0], dP2 = p[2] - p[0];
    f2 dUV1{ t1.x - t0.x, t1.y - t0.y }, dUV2{ t2.x - t0.x, t2.y - t0.y };
    float determinant = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
    f3 dpdu, dpdv;
    if (determinant == 0) { f3 a, b, nn = normalize(cross(dP1, dP2)); coordinate_system(nn, a, b); dpdu = a; dpdv = b; }
    else {
        float invDet = 1.0f / determinant;
        dpdu = ((dUV2.y * dP1 - dUV1.y * dP2) * invDet);
        dpdv = ((-dUV2.x * dP1 + dUV1.x * dP2) * invDet);
    }
    uint32_t ax = float_to_half(dpdu.x), ay = float_to_half(dpdu.y), az = float_to_half(dpdu.z);
    uint32_t bx = float_to_half(dpdv.x), by = float_to_half(dpdv.y), bz = float_to_half(dpdv.z);
    T.nor_mat_extra[0] = (uint32_t)normal_to_uchar2(n[0]) | ((uint32_t)normal_to_uchar2(n[1]) << 16);
    T.nor_mat_extra[1] = (uint32_t)normal_to_uchar2(n[2]) | (T.nor_mat_extra[1] & 0xffff0000);
    T.dpdu_dpdv[0] = ax | (ay << 16); T.dpdu_dpdv[1] = az | (bx << 16); T.dpdu_dpdv[2] = by | (bz << 16);
}

// shading normal of a triangle at barycentrics (1/3,1/3) under `l2w` — the dg.sys.n of TriangleData::fillDG
// (Engine/TriangleData.cu:75-103), used for ShapeSet::triData::n (Engine/ShapeSet.cu:11-23)
static f3 tri_data_center_normal(const ctl_triangle_data& T, const m34& l2w) {
    f3 na = uchar2_to_normal(T.nor_mat_extra[0] & 0xffff), nb = uchar2_to_normal(T.nor_mat_extra[0] >> 16), nc = uchar2_to_normal(T.nor_mat_extra[1] & 0xffff);
    float u = 1.0f / 3.0f, v = 1.0f / 3.0f, w = 1.0f - u - v;
    f3 n = normalize(u * na + v * nb + w * nc);
    auto h = [](uint32_t bits) { return half_to_float((uint16_t)bits); };
    f3 dpdu(h(T.dpdu_dpdv[0]), h(T.dpdu_dpdv[0] >> 16), h(T.dpdu_dpdv[1]));
    f3 s = dpdu - n * dot(n, dpdu);
    f3 t = cross(s, n);
    s = xform_dir(l2w, s); t = xform_dir(l2w, t);
    return normalize(cross(t, s));
}

// Mesh::ComputeVertexNormals (Engine/Mesh.cpp:151-190), "sphere inscribed polytope" weights
static aabb box_transform(const aabb& b, const float* m);

static void compute_vertex_normals(const float* V, const uint32_t* I, uint32_t nv, uint32_t nt, std::vector<f3>& N, bool flip = false) {
    N.assign(nv, f3(0.0f));
    auto vtx = [&](uint32_t i) { return f3(V[3 * i], V[3 * i + 1], V[3 * i + 2]); };
    const float flip_coeff = flip ? -1.0f : 1.0f;
    for (uint32_t f = 0; f < nt; f++) {
        uint32_t i1 = I ? I[f * 3] : f * 3, i2 = I ? I[f * 3 + 1] : f * 3 + 1, i3 = I ? I[f * 3 + 2] : f * 3 + 2;
        f3 v1 = vtx(i1), v2 = vtx(i2), v3 = vtx(i3);
        auto nor = [&](f3 pb, f3 n1, f3 n2) { return flip_coeff * cross(n1 - pb, n2 - pb) / (len_sqr(n1 - pb) * len_sqr(n2 - pb)); };
        N[i1] = N[i1] + nor(v1, v3, v2); N[i2] = N[i2] + nor(v2, v1, v3); N[i3] = N[i3] + nor(v3, v2, v1);
    }
    for (uint32_t a = 0; a < nv; a++) N[a] = normalize(N[a]);
}

uint32_t scene_builder::add_mesh(const float* positions, uint32_t n_vert, const uint32_t* indices, uint32_t n_tri, const float* normals,
                                 const float* uvs, const uint8_t* tri_material, const ctl_material* materials, uint32_t n_mat,
                                 bool flip_normals, bool face_normals, float max_smooth_angle) {
    if (!positions || n_tri == 0 || n_mat == 0 || !materials) throw std::runtime_error("ctl_builder_add_mesh: empty mesh or no material");
    if (!indices && n_vert != n_tri * 3) throw std::runtime_error("ctl_builder_add_mesh: triangle soup needs n_vert == 3*n_tri");
    if (indices) for (size_t i = 0; i < (size_t)n_tri * 3; i++) if (indices[i] >= n_vert) throw std::runtime_error("ctl_builder_add_mesh: vertex index out of range");   // before anything dereferences them
    if (tri_material) for (uint32_t i = 0; i < n_tri; i++) if (tri_material[i] >= n_mat) throw std::runtime_error("ctl_builder_add_mesh: triangle material index out of range");
    mesh_rec mr{};
    mr.tri_offset = (uint32_t)tri.size(); mr.n_tris = n_tri;
    mr.mat_offset = (uint32_t)mesh_materials.size(); mr.n_mat = n_mat;
    mesh_materials.insert(mesh_materials.end(), materials, materials + n_mat);
    // compiled-mesh cache (scene_cache.h; the reference's .xmsh, Engine/Mesh.cpp:46-98): keyed by every input of the compile step
    const bool use_sbvh = bvh_mode == CTL_BVH_SBVH || (bvh_mode == CTL_BVH_AUTO && n_tri <= kSbvhAutoLimit);
    std::string key;
    if (!cache_dir().empty()) {
        content_hash H; const uint32_t version = 2, flags = (use_sbvh ? 64u : 0u) |(flip_normals ? 1u : 0u) | (face_normals ? 2u : 0u) | (indices ? 4u : 0u) | (normals ? 8u : 0u) | (uvs ? 16u : 0u) | (tri_material ? 32u : 0u);
        H.add_value(version); H.add_value(flags); H.add_value(n_vert); H.add_value(n_tri); H.add_value(n_mat); H.add_value(max_smooth_angle);
        H.add(positions, (size_t)n_vert * 12);
        if (indices) H.add(indices, (size_t)n_tri * 12);
        if (normals) H.add(normals, (size_t)n_vert * 12);
        if (uvs) H.add(uvs, (size_t)n_vert * 8);
        if (tri_material) H.add(tri_material, n_tri);
        key = H.hex();
        cache_reader rd("mesh", key);
        std::vector<ctl_triangle_data> c_tri; std::vector<ctl_bvh_node> c_nodes; std::vector<ctl_woop_tri> c_woop; std::vector<ctl_woop_index> c_widx; aabb c_box; int c_depth = 0;
        if (rd.found() && rd.vector(c_tri) && rd.vector(c_nodes) && rd.vector(c_woop) && rd.vector(c_widx) && rd.value(c_box) && rd.value(c_depth) && rd.verify() &&
            c_tri.size() == n_tri && c_woop.size() == c_widx.size() && c_woop.size() >= n_tri) {
            tri.insert(tri.end(), c_tri.begin(), c_tri.end());
            mr.box = c_box; mr.max_depth = c_depth;
            mr.node_offset = (uint32_t)bvh.size(); mr.n_nodes = (uint32_t)c_nodes.size(); bvh.insert(bvh.end(), c_nodes.begin(), c_nodes.end());
            mr.woop_offset = (uint32_t)woop.size(); mr.n_woop = (uint32_t)c_woop.size(); woop.insert(woop.end(), c_woop.begin(), c_woop.end()); widx.insert(widx.end(), c_widx.begin(), c_widx.end());
            return finish_mesh(mr);
        }
    }
    std::vector<f3> comp;
    if (!normals || flip_normals) compute_vertex_normals(positions, indices, n_vert, n_tri, comp, flip_normals);   // Mesh.cpp:216-217
    auto vidx = [&](uint32_t ti, int j) { return indices ? indices[ti * 3 + j] : ti * 3 + j; };
    auto vtx = [&](uint32_t i) { return f3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]); };
    mr.box.reset();
    std::vector<aabb> boxes(n_tri);
    tri.resize(mr.tri_offset + n_tri);
    for (uint32_t ti = 0; ti < n_tri; ti++) {   // Mesh::CompileMesh (Engine/Mesh.cpp:225-272)
        f3 p[3], n[3]; f2 t[3];
        boxes[ti].reset();
        for (int j = 0; j < 3; j++) {
            uint32_t l = vidx(ti, j);
            if (l >= n_vert) throw std::runtime_error("ctl_builder_add_mesh: vertex index out of range");
            p[j] = vtx(l);
            t[j] = uvs ? f2{ uvs[2 * l], uvs[2 * l + 1] } : f2{ 0.0f, 0.0f };
            n[j] = (normals && !flip_normals) ? normalize(f3(normals[3 * l], normals[3 * l + 1], normals[3 * l + 2])) : comp[l];
            float pp[3] = { p[j].x, p[j].y, p[j].z };
            boxes[ti].grow(pp); mr.box.grow(pp);
        }
        if (face_normals || max_smooth_angle != 0) {   // Mesh.cpp:247-267
            f3 n_face = normalize(cross(p[0] - p[1], p[2] - p[1]));
            if (flip_normals) n_face = -n_face;
            bool use_face = face_normals;
            if (!face_normals) for (int j = 0; j < 3; j++) if (acosf(dot(n_face, n[j])) > max_smooth_angle) use_face = true;
            if (use_face) n[0] = n[1] = n[2] = n_face;
        }
        uint32_t mi = tri_material ? tri_material[ti] : 0;
        if (mi >= n_mat) throw std::runtime_error("ctl_builder_add_mesh: triangle material index out of range");
        tri_data_pack(tri[mr.tri_offset + ti], p, n, t, mi);
    }
    // ConstructBVH (Engine/MeshLoader/BVHBuilderHelper.cpp:116-147): max leaf size 8
    bvh_result R;
    if (use_sbvh) build_sbvh(positions, indices, n_tri, 8, R);   // the reference's own tree: SplitBVHBuilder with spatial splits
    else build_bvh(boxes, 8, true, 44, R);                        // binned SAH, object splits only, threaded: for meshes the sweep builder takes minutes on
    mr.node_offset = (uint32_t)bvh.size(); mr.n_nodes = (uint32_t)R.nodes.size();
    bvh.insert(bvh.end(), R.nodes.begin(), R.nodes.end());
    mr.woop_offset = (uint32_t)woop.size(); mr.n_woop = (uint32_t)R.leaf_prims.size();
    woop.resize(mr.woop_offset + mr.n_woop); widx.resize(mr.woop_offset + mr.n_woop);
    for (uint32_t i = 0; i < mr.n_woop; i++) {   // createLeafNode (BVHBuilderHelper.cpp:51-62)
        uint32_t t = R.leaf_prims[i];
        woop_set_data(woop[mr.woop_offset + i], vtx(vidx(t, 0)), vtx(vidx(t, 1)), vtx(vidx(t, 2)));
        widx[mr.woop_offset + i].index = (t << 1) | (R.leaf_last[i] ? 1u : 0u);
    }
    mr.max_depth = R.max_depth;
    if (!key.empty()) {
        cache_writer wr("mesh", key);
        if (wr.active()) {
            wr.section(tri.data() + mr.tri_offset, (size_t)n_tri * sizeof(ctl_triangle_data)); wr.vector(R.nodes);
            wr.section(woop.data() + mr.woop_offset, (size_t)mr.n_woop * sizeof(ctl_woop_tri)); wr.section(widx.data() + mr.woop_offset, (size_t)mr.n_woop * sizeof(ctl_woop_index));
            wr.value(mr.box); wr.value(mr.max_depth); wr.commit();
        }
    }
    return finish_mesh(mr);
}

uint32_t scene_builder::default_bvh_mode() {
    const char* e = std::getenv("CTL_BVH_MODE");
    if (!e) return CTL_BVH_AUTO;
    const std::string s(e);
    return s == "sbvh" ? CTL_BVH_SBVH : (s == "binned" ? CTL_BVH_BINNED : CTL_BVH_AUTO);
}

uint32_t scene_builder::finish_mesh(const mesh_rec& mr) {
    mesh_info.push_back(mr);
    ctl_kernel_mesh km;   // Mesh::getKernelData (Engine/Mesh.cpp:100-109)
    km.tri_offset = mr.tri_offset; km.bvh_node_offset = mr.node_offset * 4; km.bvh_tri_offset = mr.woop_offset * 3;
    km.bvh_index_offset = mr.woop_offset; km.std_material_offset = mr.mat_offset;
    meshes.push_back(km);
    return (uint32_t)meshes.size() - 1;
}

uint32_t scene_builder::add_node(uint32_t mesh_index, const ctl_float4x4* to_world) {
    if (mesh_index >= meshes.size()) throw std::runtime_error("ctl_builder_add_node: bad mesh index");
    const mesh_rec& mr = mesh_info[mesh_index];
    // validate the transform before anything is appended: a rejected call must leave nodes / mats / xf / ixf untouched
    ctl_float4x4 m;
    if (to_world) m = *to_world; else { std::memset(&m, 0, sizeof(m)); m.m[0] = m.m[5] = m.m[10] = m.m[15] = 1.0f; }
    if (m.m[12] != 0.0f || m.m[13] != 0.0f || m.m[14] != 0.0f || m.m[15] != 1.0f)
        throw std::runtime_error("ctl_builder_add_node: node transform must be affine (last row 0 0 0 1)");
    ctl_float4x4 inv; mat_inverse(m.m, inv.m);   // kept as computed: the reference divides by the inverse's own w (float4x4.h:402-406)
    if (inv.m[12] != 0.0f || inv.m[13] != 0.0f || inv.m[14] != 0.0f || !(inv.m[15] > 0.0f))
        throw std::runtime_error("ctl_builder_add_node: node transform is singular");
    ctl_node n{};   // Node::Node (SceneTypes/Node.cpp:10-17): every node owns a copy of the mesh's materials
    n.mesh_index = mesh_index; n.material_offset = (uint32_t)mats.size(); n.instanciated_material = 0;
    n.lights[0] = n.lights[1] = 0xffffffffu; n.n_lights = 0;
    mats.insert(mats.end(), mesh_materials.begin() + mr.mat_offset, mesh_materials.begin() + mr.mat_offset + mr.n_mat);
    nodes.push_back(n);
    xf.push_back(m); ixf.push_back(inv);
    return (uint32_t)nodes.size() - 1;
}

// DynamicScene::SetNodeTransform (Engine/DynamicScene.cpp:338-346).  Area lights created from the node BEFORE the call keep the
// triangles of the old transform (the reference recalculates them; the loader always sets the transform first).
void scene_builder::set_node_transform(uint32_t node_index, const ctl_float4x4& m) {
    if (node_index >= nodes.size()) throw std::runtime_error("set_node_transform: bad node index");
    if (m.m[12] != 0.0f || m.m[13] != 0.0f || m.m[14] != 0.0f || m.m[15] != 1.0f) throw std::runtime_error("set_node_transform: node transform must be affine (last row 0 0 0 1)");
    ctl_float4x4 inv; mat_inverse(m.m, inv.m);
    if (inv.m[12] != 0.0f || inv.m[13] != 0.0f || inv.m[14] != 0.0f || !(inv.m[15] > 0.0f)) throw std::runtime_error("set_node_transform: node transform is singular");
    xf[node_index] = m; ixf[node_index] = inv;
}
// `mat->bsdf = ...; mat->bsdf.As()->m_enableTwoSided = ...` of BsdfParser::apply_bsdf (ObjectParser.h:996-1010): replaces the BSDF
// of one of the node's materials and keeps its NodeLightIndex
void scene_builder::set_node_bsdf(uint32_t node_index, uint32_t local_material, const ctl_material& m) {
    if (node_index >= nodes.size()) throw std::runtime_error("set_node_bsdf: bad node index");
    const ctl_node& N = nodes[node_index];
    if (local_material >= mesh_info[N.mesh_index].n_mat) throw std::runtime_error("set_node_bsdf: bad material index");
    ctl_material& dst = mats[N.material_offset + local_material];
    const uint32_t nli = dst.node_light_index;
    dst = m; dst.node_light_index = nli;
}
uint32_t scene_builder::add_aux_material(const ctl_material& m) {
    if (m.bsdf_type >= CTL_BSDF_HK) throw std::runtime_error("add_aux_material: a nested BSDF must be one of the simple models (BSDFFirst, SceneTypes/BSDF.h:102)");
    mats.push_back(m); mats.back().node_light_index = 0xffffffffu;
    return (uint32_t)mats.size() - 1;
}
const ctl_material& scene_builder::node_material(uint32_t node_index, uint32_t local_material) const {
    const ctl_node& N = nodes.at(node_index);
    if (local_material >= mesh_info[N.mesh_index].n_mat) throw std::runtime_error("node_material: bad material index");
    return mats[N.material_offset + local_material];
}
// AABB of all nodes (DynamicScene::getSceneBox)
aabb scene_builder::scene_box() const {
    aabb scene; scene.reset();
    for (size_t i = 0; i < nodes.size(); i++) { const aabb b = box_transform(mesh_info[nodes[i].mesh_index].box, xf[i].m); scene.grow(b); }
    return scene;
}

// DynamicScene::CreateLight(node, matName, L) + CreateShape (Engine/DynamicScene.cpp:689-767) + ShapeSet (Engine/ShapeSet.cpp:17-59)
uint32_t scene_builder::add_area_light(uint32_t node_index, uint32_t local_material, const float radiance[3], const ctl_texture* rad_texture, bool orthogonal) {
    if (node_index >= nodes.size()) throw std::runtime_error("ctl_builder_add_area_light: bad node index");
    ctl_node& N = nodes[node_index];
    const mesh_rec& mr = mesh_info[N.mesh_index];
    if (local_material >= mr.n_mat) throw std::runtime_error("Could not find material name in mesh!");
    ctl_material& mat = mats[N.material_offset + local_material];
    if (mat.node_light_index == 0xffffffffu && N.n_lights >= 2) throw std::runtime_error("Node already has maximum number of area lights!");
    std::vector<uint32_t> sel_woop, sel_tri;
    for (uint32_t i = 0; i < mr.n_woop; i++) {
        uint32_t i2 = widx[mr.woop_offset + i].index >> 1;
        const ctl_triangle_data& d = tri[mr.tri_offset + i2];
        if (((d.nor_mat_extra[1] >> 16) & 0xff) != local_material) continue;
        if (std::find(sel_tri.begin(), sel_tri.end(), mr.tri_offset + i2) != sel_tri.end()) continue;
        sel_woop.push_back(mr.woop_offset + i); sel_tri.push_back(mr.tri_offset + i2);
    }
    if (sel_woop.empty()) throw std::runtime_error("ctl_builder_add_area_light: material has no triangles");
    uint32_t count = (uint32_t)sel_woop.size();
    auto align_to = [&](size_t a) { while (anim.size() % a) anim.push_back(0); };
    align_to(4); uint32_t cdf_off = (uint32_t)anim.size(); anim.resize(anim.size() + (count + 1) * sizeof(float));
    align_to(16); uint32_t tri_off = (uint32_t)anim.size(); anim.resize(anim.size() + (size_t)count * sizeof(ctl_shape_tri));
    m34 l2w = as_m34(xf[node_index].m);
    std::vector<float> cdf(count + 1); std::vector<ctl_shape_tri> st(count);
    float sumArea = 0; cdf[0] = 0.0f;
    for (uint32_t i = 0; i < count; i++) {   // ShapeSet::triData::Recalculate (Engine/ShapeSet.cu:11-23)
        f3 p[3]; woop_get_data(woop[sel_woop[i]], p[0], p[1], p[2]);
        f3 n = tri_data_center_normal(tri[sel_tri[i]], l2w);
        for (int k = 0; k < 3; k++) p[k] = xform_point(l2w, p[k]);
        float area = 0.5f * length(cross(p[2] - p[0], p[1] - p[0]));
        ctl_shape_tri& s = st[i];
        for (int k = 0; k < 3; k++) { s.p[k][0] = p[k].x; s.p[k][1] = p[k].y; s.p[k][2] = p[k].z; }
        s.n[0] = n.x; s.n[1] = n.y; s.n[2] = n.z; s.area = area; s.i_dat = sel_woop[i]; s.t_dat = sel_tri[i];
        sumArea += area; cdf[i + 1] = cdf[i] + area;
    }
    for (uint32_t i = 0; i <= count; i++) cdf[i] = cdf[i] / sumArea;
    std::memcpy(anim.data() + cdf_off, cdf.data(), cdf.size() * sizeof(float));
    std::memcpy(anim.data() + tri_off, st.data(), st.size() * sizeof(ctl_shape_tri));
    ctl_light L{};
    L.type = CTL_LIGHT_DIFFUSE; L.radiance[0] = radiance[0]; L.radiance[1] = radiance[1]; L.radiance[2] = radiance[2];
    L.area_dist_index = cdf_off; L.triangles_index = tri_off; L.sum_area = sumArea; L.count = count; L.orthogonal = orthogonal ? 1u : 0u; L.node_idx = node_index;
    if (rad_texture) {
        if (rad_texture->type != CTL_TEX_CONSTANT && rad_texture->type != CTL_TEX_CHECKER && rad_texture->type != CTL_TEX_IMAGE) throw std::runtime_error("ctl_builder_add_area_light_ex: unknown texture type");
        L.rad_texture = *rad_texture;
        if (rad_texture->type == CTL_TEX_CONSTANT) { L.radiance[0] = rad_texture->value[0]; L.radiance[1] = rad_texture->value[1]; L.radiance[2] = rad_texture->value[2]; }
    }
    uint32_t li;
    if (mat.node_light_index != 0xffffffffu) { li = N.lights[mat.node_light_index]; lights[li] = L; }
    else { li = (uint32_t)lights.size(); lights.push_back(L); mat.node_light_index = N.n_lights; N.lights[N.n_lights++] = li; }
    return li;
}

uint32_t scene_builder::add_point_light(const float position[3], const float intensity[3]) {
    ctl_light L{};
    L.type = CTL_LIGHT_POINT;
    for (int i = 0; i < 3; i++) { L.position[i] = position[i]; L.radiance[i] = intensity[i]; }
    lights.push_back(L);
    return (uint32_t)lights.size() - 1;
}

// Frame(n) (Math/Frame.h:31-35) -> rows s, t, n of ctl_light::to_world
static void store_frame(ctl_light& L, f3 n) {
    f3 s, t; coordinate_system(n, s, t);
    std::memset(L.to_world, 0, sizeof(L.to_world));
    L.to_world[0] = s.x; L.to_world[1] = s.y; L.to_world[2] = s.z;
    L.to_world[4] = t.x; L.to_world[5] = t.y; L.to_world[6] = t.z;
    L.to_world[8] = n.x; L.to_world[9] = n.y; L.to_world[10] = n.z;
}

// SpotLight::SpotLight (SceneTypes/Light.cu:268-277): width = cutoff angle, fall = beam width, both in degrees
uint32_t scene_builder::add_spot_light(const float position[3], const float target[3], const float intensity[3], float cutoff_deg, float beam_deg) {
    ctl_light L{};
    L.type = CTL_LIGHT_SPOT;
    for (int i = 0; i < 3; i++) { L.position[i] = position[i]; L.direction[i] = target[i]; L.radiance[i] = intensity[i]; }
    L.cutoff_angle = cutoff_deg * (kPi / 180.0f); L.beam_width = beam_deg * (kPi / 180.0f);
    L.cos_beam_width = cosf(L.beam_width); L.cos_cutoff_angle = cosf(L.cutoff_angle);
    L.inv_transition_width = 1.0f / (L.cutoff_angle - L.beam_width);
    store_frame(L, normalize(f3(target[0], target[1], target[2]) - f3(position[0], position[1], position[2])));
    lights.push_back(L);
    return (uint32_t)lights.size() - 1;
}

// DistantLight(L, d, r) (SceneTypes/Light.h:155-163): radius = r * 1.1 (the Mitsuba loader passes r = 1, ObjectParser.h:530)
uint32_t scene_builder::add_distant_light(const float direction[3], const float irradiance[3], float scene_radius) {
    ctl_light L{};
    L.type = CTL_LIGHT_DISTANT;
    for (int i = 0; i < 3; i++) { L.direction[i] = direction[i]; L.radiance[i] = irradiance[i]; }
    store_frame(L, normalize(f3(direction[0], direction[1], direction[2])));
    L.bsphere_radius = scene_radius * 1.1f;
    lights.push_back(L);
    return (uint32_t)lights.size() - 1;
}

uint32_t scene_builder::add_image(const uint32_t* texels, uint32_t w, uint32_t h, uint32_t texel_type, uint32_t wrap, uint32_t filter) {
    if (!texels || w == 0 || h == 0) throw std::runtime_error("ctl_builder_add_image: empty image");
    if (texel_type > CTL_TEXEL_RGBCOL || wrap > CTL_WRAP_BLACK || filter > CTL_FILTER_TRILINEAR) throw std::runtime_error("ctl_builder_add_image: bad texel type / wrap / filter mode");
    image_texels.emplace_back(texels, texels + (size_t)w * h);
    ctl_mipmap m{}; m.width = w; m.height = h; m.texel_type = texel_type; m.wrap_mode = wrap; m.filter_mode = filter;
    images.push_back(m);
    return (uint32_t)images.size() - 1;
}

static f3 texel_decode(uint32_t v, uint32_t type) {   // SpectrumConverter::RGBEToFloat3 / COLORREFToFloat3 (Math/Spectrum.h:528-565)
    unsigned x = v & 0xff, y = (v >> 8) & 0xff, z = (v >> 16) & 0xff, w = v >> 24;
    if (type == CTL_TEXEL_RGBE) {
        if (!w) return f3(0.0f);
        float e = ldexpf(1.0f, int(w) - (128 + 8));
        return f3(x * e, y * e, z * e);
    }
    return f3(float(x) / 255.0f, float(y) / 255.0f, float(z) / 255.0f);
}

// DynamicScene::setEnvironementMap (Engine/DynamicScene.cpp:846-859) + InfiniteLight::InfiniteLight (SceneTypes/Light.cpp:10-61)
uint32_t scene_builder::set_environment_map(uint32_t image, const float scale[3], const ctl_float4x4* to_world) {
    if (env_light != 0xffffffffu) throw std::runtime_error("Can't set environment map when it is already set!");
    if (image >= images.size()) throw std::runtime_error("ctl_builder_set_environment_map: bad image index");
    const ctl_mipmap& M = images[image]; const std::vector<uint32_t>& tx = image_texels[image];
    const uint32_t W = M.width, H = M.height;
    auto align_to = [&](size_t a) { while (anim.size() % a) anim.push_back(0); };
    align_to(16); uint32_t cols_off = (uint32_t)anim.size(); anim.resize(anim.size() + (size_t)(W + 1) * H * sizeof(float));
    align_to(16); uint32_t rows_off = (uint32_t)anim.size(); anim.resize(anim.size() + (size_t)(H + 1) * sizeof(float));
    align_to(16); uint32_t wts_off = (uint32_t)anim.size(); anim.resize(anim.size() + (size_t)H * sizeof(float));
    std::vector<float> cdfCols((size_t)(W + 1) * H), cdfRows(H + 1), rowWeights(H);
    const float sizeX = (float)W, sizeY = (float)H;
    size_t colPos = 0, rowPos = 0;
    float rowSum = 0.0f;
    cdfRows[rowPos++] = 0;
    for (uint32_t y = 0; y < H; ++y) {
        float colSum = 0;
        cdfCols[colPos++] = 0;
        for (uint32_t x = 0; x < W; ++x) {
            f3 v = texel_decode(tx[(size_t)y * W + x], M.texel_type);
            colSum += v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f;   // Spectrum::getLuminance
            cdfCols[colPos++] = colSum;
        }
        float normalization = 1.0f / colSum;
        for (uint32_t x = 1; x < W; ++x) cdfCols[colPos - x - 1] *= normalization;
        cdfCols[colPos - 1] = 1.0f;
        float weight = sinf((y + 0.5f) * kPi / sizeY);
        rowWeights[y] = weight;
        rowSum += colSum * weight;
        cdfRows[rowPos++] = rowSum;
    }
    float normalization = 1.0f / rowSum;
    for (uint32_t y = 1; y < H; ++y) cdfRows[rowPos - y - 1] *= normalization;
    cdfRows[rowPos - 1] = 1.0f;
    std::memcpy(anim.data() + cols_off, cdfCols.data(), cdfCols.size() * sizeof(float));
    std::memcpy(anim.data() + rows_off, cdfRows.data(), cdfRows.size() * sizeof(float));
    std::memcpy(anim.data() + wts_off, rowWeights.data(), rowWeights.size() * sizeof(float));
    ctl_light L{};
    L.type = CTL_LIGHT_INFINITE;
    L.env_image = image;
    for (int i = 0; i < 3; i++) L.env_scale[i] = scale[i];
    L.cdf_cols_index = cols_off; L.cdf_rows_index = rows_off; L.row_weights_index = wts_off;
    L.normalization = 1.0f / (rowSum * (2 * kPi / sizeX) * (kPi / sizeY));
    static const float ident[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    std::memcpy(L.to_world, to_world ? to_world->m : ident, 64);
    lights.push_back(L);
    env_light = (uint32_t)lights.size() - 1;
    return env_light;
}

// RoughTransmittanceManager::StaticInitialize (Engine/RoughTransmittance.cu:124-131): one slot
void scene_builder::set_rough_transmittance(uint32_t slot, const ctl_rough_transmittance& t) {
    if (slot >= 3) throw std::runtime_error("ctl_builder_set_rough_transmittance: slot must be 0..2");
    if (!t.trans || !t.diff_trans || t.eta_samples < 2 || t.alpha_samples < 2 || t.theta_samples < 2) throw std::runtime_error("ctl_builder_set_rough_transmittance: bad table");
    const size_t nt = (size_t)2 * t.eta_samples * t.alpha_samples * t.theta_samples, nd = (size_t)2 * t.eta_samples * t.alpha_samples;
    rt_trans[slot].assign(t.trans, t.trans + nt); rt_diff[slot].assign(t.diff_trans, t.diff_trans + nd);
    rt[slot] = t; have_rt = true;
}
// RoughTransmittance::RoughTransmittance(name) (Engine/RoughTransmittance.cu:8-45): "MTS_TRANSMITTANCE", three size_t counts,
// four float bounds, then per (eta, alpha): theta_samples transmittances followed by one diffuse transmittance
void scene_builder::load_rough_transmittance(uint32_t slot, const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("Could not open file : ") + path);
    std::vector<unsigned char> buf;
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); throw std::runtime_error("short read"); }
    std::fclose(f);
    const char header[] = "MTS_TRANSMITTANCE"; const size_t hl = sizeof(header) - 1;
    if (buf.size() < hl + 3 * 8 + 4 * 4 || std::memcmp(buf.data(), header, hl) != 0) throw std::runtime_error("Invalid filetype for rough transmittance!");
    size_t pos = hl; uint64_t cnt[3]; float bnd[4];
    std::memcpy(cnt, buf.data() + pos, 24); pos += 24;
    std::memcpy(bnd, buf.data() + pos, 16); pos += 16;
    const size_t E = cnt[0], A = cnt[1], T = cnt[2], nt = 2 * E * A * T, nd = 2 * E * A;
    if (buf.size() != pos + (nt + nd) * sizeof(float)) throw std::runtime_error("RoughTransmittance: unexpected file size");
    const float* ptr = (const float*)(buf.data() + pos);
    std::vector<float> trans(nt), diff(nd); size_t de = 0, fe = 0;
    for (size_t i = 0; i < 2 * E; ++i) for (size_t j = 0; j < A; ++j) { for (size_t k = 0; k < T; ++k) trans[de++] = *ptr++; diff[fe++] = *ptr++; }
    ctl_rough_transmittance t{}; t.trans = trans.data(); t.diff_trans = diff.data();
    t.eta_samples = (uint32_t)E; t.alpha_samples = (uint32_t)A; t.theta_samples = (uint32_t)T;
    t.eta_min = bnd[0]; t.eta_max = bnd[1]; t.alpha_min = bnd[2]; t.alpha_max = bnd[3];
    set_rough_transmittance(slot, t);
}

// Sensor::SetToWorld(pos, tar, up) (SceneTypes/Sensor.cu:691-699) with the loader's frame reconstruction
// (ObjectParser.h:292-297, Sensor.cu:682-689): r = f x up, u = r x f, columns (r, u, f), translation pos.
void scene_builder::set_camera_lookat(const float pos[3], const float target[3], const float up_in[3], float fov_degrees, uint32_t w, uint32_t h) {
    f3 p(pos[0], pos[1], pos[2]), tar(target[0], target[1], target[2]), u(up_in[0], up_in[1], up_in[2]);
    f3 f = normalize(tar - p);
    f3 r = normalize(cross(f, u));
    std::memset(&camera, 0, sizeof(camera));
    camera.type = CTL_SENSOR_PERSPECTIVE;
    float rot[16] = { r.x, u.x, f.x, 0, r.y, u.y, f.y, 0, r.z, u.z, f.z, 0, 0, 0, 0, 1 };
    float tr[16] = { 1, 0, 0, p.x, 0, 1, 0, p.y, 0, 0, 1, p.z, 0, 0, 0, 1 };
    mat_mul(tr, rot, camera.to_world);   // float4x4::Translate(pos) % rot (Sensor.cu:672-678)
    camera.fov = fov_degrees * (kPi / 180.0f);   // math::Radians (SceneTypes/Sensor.h:72-76)
    camera.near_depth = 1e-2f; camera.far_depth = 1e4f;   // loader defaults (ObjectParser.h:242)
    camera.resolution[0] = (float)w; camera.resolution[1] = (float)h;
    have_camera = true;
}
void scene_builder::set_camera(const ctl_sensor& s) { camera = s; have_camera = true; }

// AABB::Transform (Math/AABB.h:32-45)
static aabb box_transform(const aabb& b, const float* m) {
    aabb o;
    for (int i = 0; i < 3; i++) {
        float lo = m[i * 4 + 3], hi = m[i * 4 + 3];   // + Translation()
        float s_lo = 0, s_hi = 0;
        for (int k = 0; k < 3; k++) { float a = m[i * 4 + k] * b.lo[k], c = m[i * 4 + k] * b.hi[k]; s_lo += min2(a, c); s_hi += max2(a, c); }
        o.lo[i] = s_lo + lo; o.hi[i] = s_hi + hi;
    }
    return o;
}

void scene_builder::finalize(ctl_scene_desc& out) {
    if (nodes.empty()) throw std::runtime_error("ctl_builder_finalize: scene has no nodes");
    if (!have_camera) throw std::runtime_error("ctl_builder_finalize: no camera set");
    if (nodes.size() != xf.size() || nodes.size() != ixf.size()) throw std::runtime_error("ctl_builder_finalize: node / transform arrays out of step");
    // SceneBVH::Build (Engine/SceneBVH.cpp:11-54): one scene-BVH leaf per node (BVHRebuilder.cpp:380-383)
    std::vector<aabb> nb(nodes.size());
    aabb scene; scene.reset();
    for (size_t i = 0; i < nodes.size(); i++) { nb[i] = box_transform(mesh_info[nodes[i].mesh_index].box, xf[i].m); scene.grow(nb[i]); }
    bvh_result R;
    build_bvh(nb, 1, false, 20, R);
    scene_bvh = R.nodes;
    // leaves ~k index leaf_prims (one entry each) -> encode ~nodeIdx directly as the reference does
    auto fix = [&](int c) { return (c < 0) ? ~(int)R.leaf_prims[~c] : c; };
    for (auto& n : scene_bvh) { n.child0 = fix(n.child0); n.child1 = fix(n.child1); }
    int start = R.root < 0 ? ~(int)R.leaf_prims[~R.root] : R.root;
    int top_depth = R.max_depth, bottom_depth = 0;
    for (auto& m : mesh_info) bottom_depth = std::max(bottom_depth, m.max_depth);
    if (top_depth + bottom_depth + 4 > 64) throw std::runtime_error("ctl_builder_finalize: BVH too deep for the traversal stack");

    std::memset(&out, 0, sizeof(out));
    out.tri_data = tri.data(); out.n_tri_data = (uint32_t)tri.size();
    out.woop = woop.data(); out.n_woop = (uint32_t)woop.size(); out.woop_index = widx.data();
    out.bvh_nodes = bvh.data(); out.n_bvh_nodes = (uint32_t)bvh.size();
    out.meshes = meshes.data(); out.n_meshes = (uint32_t)meshes.size();
    out.nodes = nodes.data(); out.n_nodes = (uint32_t)nodes.size();
    out.materials = mats.data(); out.n_materials = (uint32_t)mats.size();
    out.lights = lights.data(); out.n_lights_buf = (uint32_t)lights.size();
    out.anim = anim.data(); out.n_anim_bytes = (uint32_t)anim.size();
    out.scene_start_node = start;
    out.scene_bvh_nodes = scene_bvh.data(); out.n_scene_bvh_nodes = (uint32_t)scene_bvh.size();
    out.node_transforms = xf.data(); out.node_inv_transforms = ixf.data();
    out.env_map_index = env_light;
    for (int i = 0; i < 3; i++) { out.box_min[i] = scene.lo[i]; out.box_max[i] = scene.hi[i]; }
    for (size_t i = 0; i < images.size(); i++) images[i].texels = image_texels[i].data();
    out.images = images.data(); out.n_images = (uint32_t)images.size();
    {   // BSDF::Update() now that every image is there (MaterialStream::UpdateMaterialsPhase2 after LoadTextures, Engine/DynamicScene.cpp:74-89): the sampling weights that come
        // from the average of an IMAGE texture were made with the bitmap counted as white (material_textures.h)
        image_set I; I.images = images.data(); I.n = (uint32_t)images.size();
        std::vector<float> avg_cache(4 * images.size() + 4, 0.0f); I.cache = reinterpret_cast<float (*)[4]>(avg_cache.data());
        for (auto& m : mats) material_update_textures(m, I);
    }
    for (int i = 0; i < 3; i++) { rt[i].trans = rt_trans[i].empty() ? nullptr : rt_trans[i].data(); rt[i].diff_trans = rt_diff[i].empty() ? nullptr : rt_diff[i].data(); }
    out.rough_transmittance = have_rt ? rt : nullptr;
    {   // the light that depends on the scene box: InfiniteLight::Update (SceneTypes/Light.h:318-325)
        f3 lo(scene.lo[0], scene.lo[1], scene.lo[2]), hi(scene.hi[0], scene.hi[1], scene.hi[2]);
        f3 center = (lo + hi) * 0.5f;   // AABB::Center (Math/AABB.h)
        f3 size = hi - lo;
        for (auto& L : lights) {
            if (L.type == CTL_LIGHT_INFINITE) { L.bsphere_center[0] = center.x; L.bsphere_center[1] = center.y; L.bsphere_center[2] = center.z; L.bsphere_radius = length(size) / 1.5f; }
        }
    }
    out.camera = camera;
    // LightStream::fillDeviceData (Engine/DynamicScene.cpp:172-196): uniform weights, first MAX_NUM_LIGHTS lights
    out.num_lights = std::min<uint32_t>(CTL_MAX_NUM_LIGHTS, (uint32_t)lights.size());
    float accum = 0; for (size_t i = 0; i < lights.size(); i++) accum += 1.0f;
    for (uint32_t i = 0; i < out.num_lights; i++) {
        out.light_indices[i] = i;
        float pdf = 1.0f / accum;
        out.light_cdf[i] = (i > 0 ? out.light_cdf[i - 1] : 0.0f) + pdf;
    }
    // m_rayTraceEps = MIN_RAYTRACE_DISTANCE_RELATIVE * |box.Size()| (Engine/DynamicScene.cpp:587)
    f3 sz(scene.hi[0] - scene.lo[0], scene.hi[1] - scene.lo[1], scene.hi[2] - scene.lo[2]);
    out.ray_trace_eps = 1e-4f * length(sz);
}

} // namespace ctl
