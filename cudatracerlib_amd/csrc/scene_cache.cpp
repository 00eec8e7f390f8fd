// scene_cache.cpp — see scene_cache.h
#include "scene_cache.h"
#include <cstdlib>
#include <mutex>
#include <unistd.h>
#include <sys/stat.h>

namespace ctl {

namespace {
constexpr uint32_t kMagic = 0x434C5443u;   // "CTLC"
constexpr uint32_t kVersion = 2;   // 2: trailing checksum
std::mutex g_mu; std::string g_dir; bool g_dir_set = false;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
}

void set_cache_dir(const char* dir) { std::lock_guard<std::mutex> l(g_mu); g_dir = dir ? dir : ""; g_dir_set = true; }
std::string cache_dir() {
    std::lock_guard<std::mutex> l(g_mu);
    if (!g_dir_set) { const char* e = std::getenv("CTL_CACHE_DIR"); g_dir = e ? e : ""; g_dir_set = true; }
    return g_dir;
}

void content_hash::word(uint64_t w) {
    a_ = rotl(a_ ^ (w * 0xC2B2AE3D27D4EB4Full), 31) * 0x9E3779B185EBCA87ull;
    b_ = rotl(b_ + w, 27) * 0x165667B19E3779F9ull + 0x85EBCA77C2B2AE63ull;
}
void content_hash::add(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    len_ += n;
    while (n >= 8) { uint64_t w; std::memcpy(&w, c, 8); word(w); c += 8; n -= 8; }
    if (n) { uint64_t w = 0; std::memcpy(&w, c, n); word(w ^ ((uint64_t)n << 56)); }
}
void content_hash::digest(uint64_t out[2]) const {
    uint64_t a = a_ ^ len_, b = b_ + len_;
    for (int i = 0; i < 2; i++) {   // avalanche
        a ^= a >> 33; a *= 0xFF51AFD7ED558CCDull; a ^= b; b ^= b >> 29; b *= 0xC4CEB9FE1A85EC53ull; b ^= a;
    }
    out[0] = a; out[1] = b;
}
std::string content_hash::hex() const {
    uint64_t d[2]; digest(d);
    const uint64_t a = d[0], b = d[1];
    char s[33]; std::snprintf(s, sizeof(s), "%016llx%016llx", (unsigned long long)a, (unsigned long long)b);
    return s;
}

static std::string entry_path(const std::string& dir, const std::string& kind, const std::string& hash_hex) { return dir + "/" + kind + "_" + hash_hex + ".ctlc"; }

cache_writer::cache_writer(const std::string& kind, const std::string& hash_hex) {
    const std::string dir = cache_dir();
    if (dir.empty()) return;
    ::mkdir(dir.c_str(), 0777);   // one level; an existing directory is fine
    final_ = entry_path(dir, kind, hash_hex);
    // unique per writer (mkstemp): ranks on different hosts or in different containers share pids, threads of one process always do
    tmp_ = final_ + ".tmpXXXXXX";
    const int fd = ::mkstemp(&tmp_[0]);
    if (fd < 0) { tmp_.clear(); return; }
    f_ = ::fdopen(fd, "wb");
    if (!f_) { ::close(fd); std::remove(tmp_.c_str()); tmp_.clear(); return; }
    const uint32_t head[3] = { kMagic, kVersion, 0 };
    ok_ = std::fwrite(head, 1, sizeof(head), f_) == sizeof(head);
}
void cache_writer::section(const void* p, size_t bytes) {
    if (!f_ || !ok_) return;
    const uint64_t b = bytes;
    ok_ = std::fwrite(&b, 1, 8, f_) == 8 && (bytes == 0 || std::fwrite(p, 1, bytes, f_) == bytes);
    sum_.add_value(b); if (bytes) sum_.add(p, bytes);
    n_++;
}
void cache_writer::commit() {
    if (!f_) return;
    if (ok_) { uint64_t d[2]; sum_.digest(d); ok_ = std::fwrite(d, 1, 16, f_) == 16; }
    if (ok_) { ok_ = std::fseek(f_, 8, SEEK_SET) == 0 && std::fwrite(&n_, 1, 4, f_) == 4; }
    ok_ = (std::fclose(f_) == 0) && ok_; f_ = nullptr;
    if (ok_) ok_ = std::rename(tmp_.c_str(), final_.c_str()) == 0;
    if (!ok_) std::remove(tmp_.c_str());
    tmp_.clear();
}
cache_writer::~cache_writer() { if (f_) { std::fclose(f_); f_ = nullptr; } if (!tmp_.empty()) std::remove(tmp_.c_str()); }

cache_reader::cache_reader(const std::string& kind, const std::string& hash_hex) {
    const std::string dir = cache_dir();
    if (dir.empty()) return;
    f_ = std::fopen(entry_path(dir, kind, hash_hex).c_str(), "rb");
    if (!f_) return;
    uint32_t head[3];
    if (std::fread(head, 1, sizeof(head), f_) != sizeof(head) || head[0] != kMagic || head[1] != kVersion) { std::fclose(f_); f_ = nullptr; return; }
    left_ = head[2];
}
cache_reader::~cache_reader() { if (f_) std::fclose(f_); }
bool cache_reader::next(uint64_t& bytes) {
    if (!f_ || left_ == 0) return false;
    left_--;
    if (!(std::fread(&bytes, 1, 8, f_) == 8 && bytes < ((uint64_t)1 << 40))) { ok_ = false; return false; }
    sum_.add_value(bytes);
    return true;
}
bool cache_reader::body(void* p, uint64_t bytes) {
    if (std::fread(p, 1, bytes, f_) != bytes) { ok_ = false; return false; }
    sum_.add(p, bytes);
    return true;
}
bool cache_reader::verify() {
    if (!f_ || !ok_ || left_ != 0) return false;
    uint64_t want[2], got[2];
    if (std::fread(want, 1, 16, f_) != 16) return false;
    sum_.digest(got);
    return want[0] == got[0] && want[1] == got[1];
}

} // namespace ctl
