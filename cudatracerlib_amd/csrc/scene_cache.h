// scene_cache.h — on-disk cache of compiled geometry: the role of the reference's `.xmsh` files (Engine/Mesh.cpp:46-98 reads what
// Mesh::CompileMesh :199-290 wrote; DynamicScene::CreateNode compiles a mesh only when its .xmsh is missing or stale).
//
// The reference dumps its structs raw, vtable pointers of the 3344-B Material included, so a file is tied to one compiler.  This
// cache keeps only plain arrays and is keyed by CONTENT: the file name is a 128-bit hash of every input of the compile step
// (vertex data, options, format version), so a stale entry cannot be picked up and no time stamps are compared.  Two kinds:
//   mesh_<hash>.ctlc   one compiled mesh: TriangleData[], BVHNodeData[], Woop rows, index words, box (scene_builder.cpp add_mesh)
//   flat_<hash>.ctlc   the flattened world-space BVH of a scene: flat nodes + leaf entries (flatten.cpp)
// File: "CTLC" u32 version u32 n_sections, then per section u64 byte count + bytes, then a 16-byte checksum (content_hash of every
// section's count and bytes).  Little endian, written to a unique temporary (mkstemp) and renamed, so a concurrent reader (other ranks
// of a multi-GPU job share the directory) sees a whole file or none; a reader must call verify() after its last section and drop
// what it read when that fails (torn or corrupted entry).
// Off unless a directory is set (ctl_set_cache_dir / CTL_CACHE_DIR).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace ctl {

void set_cache_dir(const char* dir);   // nullptr / "" disables
std::string cache_dir();               // "" = disabled

// 128-bit content hash (two multiply-rotate lanes over 8-byte words; not cryptographic — it only has to tell inputs apart)
class content_hash {
public:
    void add(const void* p, size_t n);
    template <typename T> void add_value(const T& v) { add(&v, sizeof(T)); }
    template <typename T> void add_vector(const std::vector<T>& v) { const uint64_t n = v.size(); add_value(n); if (n) add(v.data(), n * sizeof(T)); }
    std::string hex() const;
    void digest(uint64_t out[2]) const;
private:
    uint64_t a_ = 0x9E3779B185EBCA87ull, b_ = 0xC2B2AE3D27D4EB4Full, len_ = 0;
    void word(uint64_t w);
};

class cache_writer {
public:
    cache_writer(const std::string& kind, const std::string& hash_hex);   // does nothing when the cache is disabled
    ~cache_writer();
    bool active() const { return f_ != nullptr; }
    void section(const void* p, size_t bytes);
    template <typename T> void vector(const std::vector<T>& v) { section(v.data(), v.size() * sizeof(T)); }
    template <typename T> void value(const T& v) { section(&v, sizeof(T)); }
    void commit();   // rename into place; without it the temporary file is removed
private:
    FILE* f_ = nullptr; std::string tmp_, final_; uint32_t n_ = 0; bool ok_ = true; content_hash sum_;
};

class cache_reader {
public:
    cache_reader(const std::string& kind, const std::string& hash_hex);
    ~cache_reader();
    bool found() const { return f_ != nullptr; }
    template <typename T> bool vector(std::vector<T>& v) {
        uint64_t bytes;
        if (!next(bytes) || bytes % sizeof(T)) return false;
        v.resize(bytes / sizeof(T));
        return bytes == 0 || body(v.data(), bytes);
    }
    template <typename T> bool value(T& v) { uint64_t bytes; return next(bytes) && bytes == sizeof(T) && body(&v, sizeof(T)); }
    bool verify();   // every section was read and the trailing checksum matches what was read
private:
    FILE* f_ = nullptr; uint32_t left_ = 0; content_hash sum_; bool ok_ = true;
    bool next(uint64_t& bytes);
    bool body(void* p, uint64_t bytes);
};

} // namespace ctl
