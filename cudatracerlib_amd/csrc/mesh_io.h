// mesh_io.h — triangle-mesh front-ends of the scene loader: Wavefront OBJ (+MTL), PLY, Mitsuba .serialized, and the analytic
// primitives the loader tessellates (rectangle, cube, sphere, disk, cylinder).
// Reference: Engine/MeshLoader/{ObjParser,PlyParser}.cpp, Engine/SceneLoader/Mitsuba/ObjectParser.cpp:9-204, Primitives.h.
#pragma once
#include "../../include/ctl_amd.h"
#include <string>
#include <vector>

namespace ctl {

struct mesh_data {
    std::vector<float> positions;          // 3 per vertex
    std::vector<float> normals;            // 3 per vertex or empty (-> Mesh::ComputeVertexNormals)
    std::vector<float> uvs;                // 2 per vertex or empty
    std::vector<uint32_t> indices;         // 3 per triangle, already in the reference's (reversed) winding
    std::vector<uint8_t> tri_material;     // per triangle
    std::vector<ctl_material> materials;   // >= 1
    std::vector<float> emission;           // 3 per material (OBJ "Ke"), zero = none
    std::vector<std::string> image_files;  // bitmaps named by the .mtl: CTL_TEX_IMAGE textures of `materials` carry an index into this list
    // Mesh::CompileMesh options of the .serialized path
    bool flip_normals = false, face_normals = false; float max_smooth_angle = 0.0f;
    uint32_t n_vertices() const { return (uint32_t)(positions.size() / 3); }
    uint32_t n_triangles() const { return (uint32_t)(indices.size() / 3); }
};

mesh_data load_obj(const std::string& path);
mesh_data load_ply(const std::string& path);
mesh_data load_serialized(const std::string& path, int shape_index);
// dispatch on the file extension (Mesh::CompileMesh callers, Engine/DynamicScene.cpp CreateNode)
mesh_data load_mesh_file(const std::string& path);

// unit primitives in the frames the loader's `local` matrices expect (ObjectParser.h:1201-1251): these are this build's own
// tessellations, not the OBJ texts of Primitives.h
mesh_data make_plane();      // [-1,1]^2 in the xz plane, normal +y (the loader rotates it into xy)
mesh_data make_cube();       // [0,1]^3
mesh_data make_sphere();     // radius 1 about the origin
mesh_data make_disk();       // radius 1 in the xy plane, normal +z
mesh_data make_cylinder();   // radius 1 about the z axis, z in [0, 2], no caps

} // namespace ctl
