// image_io.cpp — see image_io.h.
#include "image_io.h"
#include "mitsuba_loader.h"   // io_error / unsupported_error
#include "../../include/ctl_amd.h"
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>

namespace ctl {

static std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw io_error("Could not open file : " + path);
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); throw io_error("short read : " + path); }
    std::fclose(f);
    return buf;
}
static std::string lower_ext(const std::string& path) {
    size_t d = path.find_last_of('.');
    std::string e = d == std::string::npos ? "" : path.substr(d + 1);
    for (auto& c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}
// A header may not promise more pixels than a texture can have (2^28) or than the bytes that follow could possibly hold: a 60-byte file that declares
// 2^31 x 2^31 pixels must be refused before anything is allocated for it (found by tools/fuzz_loaders.py).  min_bytes_per_pixel: the format's densest encoding.
static void check_image_size(uint64_t w, uint64_t h, size_t bytes_left, double min_bytes_per_pixel, const char* what, const std::string& path) {
    if (w == 0 || h == 0 || w > 65536 || h > 65536 || w * h > (1ull << 28)) throw io_error(std::string("unreasonable ") + what + " dimensions : " + path);
    if ((double)w * (double)h * min_bytes_per_pixel > (double)bytes_left) throw io_error(std::string("truncated ") + what + " (the header promises more pixels than the file holds) : " + path);
}
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
static uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ------------------------------------------------------------------------------------------------ PNG
static decoded_image decode_png(const std::vector<uint8_t>& d, const std::string& path) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw io_error("not a PNG file : " + path);
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    for (size_t p = 8; p + 12 <= d.size();) {
        const uint32_t len = be32(&d[p]); const char* t = (const char*)&d[p + 4];
        if (p + 12 + len > d.size()) throw io_error("truncated PNG : " + path);
        const uint8_t* c = &d[p + 8];
        if (!std::memcmp(t, "IHDR", 4)) { if (len != 13) throw io_error("bad PNG IHDR : " + path); w = be32(c); h = be32(c + 4); depth = c[8]; ctype = c[9]; interlace = c[12]; }
        else if (!std::memcmp(t, "PLTE", 4)) plte.assign(c, c + len);
        else if (!std::memcmp(t, "tRNS", 4)) trns.assign(c, c + len);
        else if (!std::memcmp(t, "IDAT", 4)) idat.insert(idat.end(), c, c + len);
        else if (!std::memcmp(t, "IEND", 4)) break;
        p += 12 + len;
    }
    if (!w || !h) throw io_error("PNG without IHDR : " + path);
    if (interlace) throw unsupported_error("interlaced PNG is not supported : " + path);
    {   // PNG spec table 11.1: grey 1,2,4,8,16; palette 1,2,4,8; every other colour type 8 or 16
        const bool sub = depth == 1 || depth == 2 || depth == 4;
        const bool ok = ctype == 0 ? (sub || depth == 8 || depth == 16) : ctype == 3 ? (sub || depth == 8) : (depth == 8 || depth == 16);
        if (!ok) throw io_error("bad PNG bit depth : " + path);
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) throw io_error("bad PNG colour type : " + path);
    check_image_size(w, h, idat.size(), (double)channels * depth / 8.0 / 1032.0, "PNG", path);   // deflate expands by at most ~1032 : 1
    const size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) throw io_error("PNG inflate failed : " + path);
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    decoded_image img; img.width = w; img.height = h; img.rgba8.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &raw[(stride + 1) * y]; const int ft = line[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
            int v;
            switch (ft) {
            case 0: v = x; break; case 1: v = x + a; break; case 2: v = x + b; break; case 3: v = x + ((a + b) >> 1); break;
            case 4: { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); break; }
            default: throw io_error("bad PNG filter : " + path);
            }
            cur[i] = (uint8_t)v;
        }
        auto sample = [&](uint32_t x, int ch) -> int {   // 8-bit value of channel ch of pixel x
            if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
            if (depth == 8) return cur[(size_t)x * channels + ch];
            const size_t bit = (size_t)x * depth; const int v = (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
            return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
        };
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* o = &img.rgba8[((size_t)y * w + x) * 4];
            if (ctype == 3) { const int i = sample(x, 0); if ((size_t)i * 3 + 2 >= plte.size()) throw io_error("PNG palette index out of range : " + path);
                              o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2]; o[3] = (size_t)i < trns.size() ? trns[i] : 255; }
            else if (ctype == 0) { o[0] = o[1] = o[2] = (uint8_t)sample(x, 0); o[3] = 255; }
            else if (ctype == 4) { o[0] = o[1] = o[2] = (uint8_t)sample(x, 0); o[3] = (uint8_t)sample(x, 1); }
            else { o[0] = (uint8_t)sample(x, 0); o[1] = (uint8_t)sample(x, 1); o[2] = (uint8_t)sample(x, 2); o[3] = ctype == 6 ? (uint8_t)sample(x, 3) : 255; }
        }
        std::swap(prev, cur);
    }
    return img;
}

// ------------------------------------------------------------------------------------------------ BMP / TGA / PNM / PFM / HDR
static decoded_image decode_bmp(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') throw io_error("not a BMP file : " + path);
    const uint32_t off = le32(&d[10]); const int32_t w = (int32_t)le32(&d[18]), hs = (int32_t)le32(&d[22]); const int bpp = le16(&d[28]); const uint32_t comp = le32(&d[30]);
    if ((bpp != 24 && bpp != 32) || (comp != 0 && comp != 3) || w <= 0 || hs == 0) throw unsupported_error("only uncompressed 24/32-bit BMP is supported : " + path);
    if (hs == INT32_MIN) throw io_error("unreasonable BMP dimensions : " + path);
    const uint32_t h = (uint32_t)std::abs(hs); const size_t stride = ((size_t)w * bpp / 8 + 3) & ~(size_t)3;
    check_image_size((uint64_t)w, h, d.size(), bpp / 8.0, "BMP", path);
    if ((size_t)off + stride * h > d.size()) throw io_error("truncated BMP : " + path);
    decoded_image img; img.width = (uint32_t)w; img.height = h; img.rgba8.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &d[off + stride * (hs > 0 ? h - 1 - y : y)];
        for (int32_t x = 0; x < w; x++) { const uint8_t* s = line + (size_t)x * bpp / 8; uint8_t* o = &img.rgba8[((size_t)y * w + x) * 4]; o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = bpp == 32 ? s[3] : 255; }
    }
    return img;
}
static decoded_image decode_tga(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 18) throw io_error("truncated TGA : " + path);
    const int idlen = d[0], cmap = d[1], type = d[2], bpp = d[16], desc = d[17]; const uint32_t w = le16(&d[12]), h = le16(&d[14]);
    if (cmap || (type != 2 && type != 3 && type != 10 && type != 11) || (bpp != 8 && bpp != 24 && bpp != 32) || !w || !h) throw unsupported_error("TGA variant not supported : " + path);
    const size_t px = bpp / 8; size_t p = 18 + idlen;
    check_image_size(w, h, d.size(), (type == 2 || type == 3) ? (double)px : (double)(px + 1) / 128.0, "TGA", path);   // a run-length packet of 1 + px bytes covers at most 128 pixels
    std::vector<uint8_t> raw((size_t)w * h * px);
    if (type == 2 || type == 3) { if (p + raw.size() > d.size()) throw io_error("truncated TGA : " + path); std::memcpy(raw.data(), &d[p], raw.size()); }
    else for (size_t o = 0; o < raw.size();) {
        if (p >= d.size()) throw io_error("truncated TGA : " + path);
        const int c = d[p++]; const size_t n = (size_t)(c & 127) + 1;
        if (c & 128) { if (p + px > d.size()) throw io_error("truncated TGA : " + path); for (size_t i = 0; i < n && o < raw.size(); i++, o += px) std::memcpy(&raw[o], &d[p], px); p += px; }
        else { if (p + n * px > d.size()) throw io_error("truncated TGA : " + path); const size_t m = std::min(n * px, raw.size() - o); std::memcpy(&raw[o], &d[p], m); o += m; p += n * px; }
    }
    decoded_image img; img.width = w; img.height = h; img.rgba8.resize((size_t)w * h * 4);
    const bool top = (desc & 0x20) != 0;
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
        const uint8_t* s = &raw[((size_t)(top ? y : h - 1 - y) * w + x) * px]; uint8_t* o = &img.rgba8[((size_t)y * w + x) * 4];
        if (px == 1) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; } else { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = px == 4 ? s[3] : 255; }
    }
    return img;
}
static bool pnm_token(const std::vector<uint8_t>& d, size_t& p, std::string& tok) {
    for (;;) {
        while (p < d.size() && std::isspace(d[p])) p++;
        if (p < d.size() && d[p] == '#') { while (p < d.size() && d[p] != '\n') p++; continue; }
        break;
    }
    tok.clear(); while (p < d.size() && !std::isspace(d[p])) tok += (char)d[p++];
    return !tok.empty();
}
static decoded_image decode_pnm(const std::vector<uint8_t>& d, const std::string& path) {
    size_t p = 0; std::string magic, t;
    if (!pnm_token(d, p, magic)) throw io_error("empty PNM : " + path);
    const bool pfm = magic == "PF" || magic == "Pf";
    if (!pfm && magic != "P2" && magic != "P3" && magic != "P5" && magic != "P6") throw unsupported_error("PNM variant not supported : " + path);
    if (!pnm_token(d, p, t)) throw io_error("bad PNM header : " + path); const uint32_t w = (uint32_t)std::stoul(t);
    if (!pnm_token(d, p, t)) throw io_error("bad PNM header : " + path); const uint32_t h = (uint32_t)std::stoul(t);
    if (!pnm_token(d, p, t)) throw io_error("bad PNM header : " + path); const double maxv = std::stod(t);
    if (magic != "PF" && magic != "Pf" && !(maxv >= 1 && maxv <= 65535)) throw io_error("bad PNM maximum value : " + path);
    {
        const int ch_ = (magic == "P3" || magic == "P6" || magic == "PF") ? 3 : 1;
        const double per_sample = pfm ? 4.0 : ((magic == "P2" || magic == "P3") ? 2.0 : (maxv > 255 ? 2.0 : 1.0));   // ASCII: a digit and a separator at least
        check_image_size(w, h, d.size() > p ? d.size() - p : 0, per_sample * ch_ - ((magic == "P2" || magic == "P3") ? 1.0 : 0.0), pfm ? "PFM" : "PNM", path);
    }
    decoded_image img; img.width = w; img.height = h;
    if (pfm) {
        p++;   // the single whitespace after the scale
        const int ch = magic == "PF" ? 3 : 1; const bool little = maxv < 0;
        if (p + (size_t)w * h * ch * 4 > d.size()) throw io_error("truncated PFM : " + path);
        img.is_float = true; img.rgb.resize((size_t)w * h * 3);
        for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) for (int c = 0; c < 3; c++) {
            const uint8_t* s = &d[p + (((size_t)(h - 1 - y) * w + x) * ch + (ch == 3 ? c : 0)) * 4];   // PFM rows are bottom-up
            uint32_t u = little ? le32(s) : be32(s); float f; std::memcpy(&f, &u, 4);
            img.rgb[((size_t)y * w + x) * 3 + c] = f;
        }
        return img;
    }
    const int ch = (magic == "P3" || magic == "P6") ? 3 : 1; const bool ascii = magic == "P2" || magic == "P3";
    img.rgba8.resize((size_t)w * h * 4);
    if (!ascii) p++;
    const int bytes = maxv > 255 ? 2 : 1;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int v[3];
        for (int c = 0; c < ch; c++) {
            if (ascii) { if (!pnm_token(d, p, t)) throw io_error("truncated PNM : " + path); v[c] = (int)(std::stod(t) * 255.0 / maxv + 0.5); }
            else { if (p + bytes > d.size()) throw io_error("truncated PNM : " + path); const int raw = bytes == 2 ? (d[p] << 8 | d[p + 1]) : d[p]; p += bytes; v[c] = (int)(raw * 255.0 / maxv + 0.5); }
        }
        uint8_t* o = &img.rgba8[i * 4];
        o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[ch == 3 ? 1 : 0]; o[2] = (uint8_t)v[ch == 3 ? 2 : 0]; o[3] = 255;
    }
    return img;
}
static decoded_image decode_hdr(const std::vector<uint8_t>& d, const std::string& path) {
    size_t p = 0; auto line = [&]() { std::string l; while (p < d.size() && d[p] != '\n') l += (char)d[p++]; p++; return l; };
    std::string l = line();
    if (l.rfind("#?", 0) != 0) throw io_error("not a Radiance HDR file : " + path);
    for (;;) { if (p >= d.size()) throw io_error("truncated HDR header : " + path); l = line(); if (l.empty() || l == "\r") break; }
    l = line();
    int h = 0, w = 0; char ys = 0, xs = 0;
    if (std::sscanf(l.c_str(), "%cY %d %cX %d", &ys, &h, &xs, &w) != 4 || w <= 0 || h <= 0) throw unsupported_error("HDR orientation not supported : " + path);
    check_image_size((uint64_t)w, (uint64_t)h, d.size() > p ? d.size() - p : 0, 8.0 / 127.0, "HDR", path);   // run-length scanlines: 2 bytes per channel per run of at most 127
    if ((uint64_t)h * 4 > d.size() - std::min(p, d.size())) throw io_error("truncated HDR : " + path);              // and every scanline starts with 4 bytes
    decoded_image img; img.width = (uint32_t)w; img.height = (uint32_t)h; img.is_float = true; img.rgb.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        if (p + 4 > d.size()) throw io_error("truncated HDR : " + path);
        if (w >= 8 && w < 32768 && d[p] == 2 && d[p + 1] == 2 && !(d[p + 2] & 0x80)) {   // new RLE: channels stored separately
            if (((d[p + 2] << 8) | d[p + 3]) != w) throw io_error("bad HDR scanline : " + path);
            p += 4;
            for (int c = 0; c < 4; c++) for (int x = 0; x < w;) {
                if (p >= d.size()) throw io_error("truncated HDR : " + path);
                int n = d[p++];
                if (n > 128) { n -= 128; if (p >= d.size() || x + n > w) throw io_error("bad HDR run : " + path); const uint8_t v = d[p++]; while (n--) scan[(size_t)x++ * 4 + c] = v; }
                else { if (p + n > d.size() || x + n > w || n == 0) throw io_error("bad HDR run : " + path); while (n--) scan[(size_t)x++ * 4 + c] = d[p++]; }
            }
        } else { if (p + (size_t)w * 4 > d.size()) throw io_error("truncated HDR : " + path); std::memcpy(scan.data(), &d[p], (size_t)w * 4); p += (size_t)w * 4; }
        const int yy = ys == '-' ? y : h - 1 - y;
        for (int x = 0; x < w; x++) {
            const uint8_t* s = &scan[(size_t)(xs == '+' ? x : w - 1 - x) * 4]; float* o = &img.rgb[((size_t)yy * w + x) * 3];
            if (s[3]) { const float e = std::ldexp(1.0f, (int)s[3] - (128 + 8)); o[0] = s[0] * e; o[1] = s[1] * e; o[2] = s[2] * e; }   // rgbe.c convention (no +0.5), as FreeImage's HDR plugin else o[0] = o[1] = o[2] = 0.0f;
        }
    }
    return img;
}

// ------------------------------------------------------------------------------------------------ OpenEXR
// Single-part scanline files (OpenEXR file layout: magic, version, attribute list, chunk offset table, chunks of 1 / 16 scanlines), channels
// R, G, B (or Y alone) of type HALF, FLOAT or UINT with 1:1 sampling, compression NONE, RLE, ZIPS or ZIP.  Tiled, multi-part and deep files and
// the PIZ / PXR24 / B44 / DWA codecs are rejected with a message naming what was found.
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; sh++; } bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}
static void exr_unpredict_and_interleave(std::vector<uint8_t>& buf, std::vector<uint8_t>& tmp) {
    // the writer splits the bytes of a block into even / odd halves and stores differences + 128 (ImfZip.cpp / ImfRle.cpp)
    for (size_t i = 1; i < buf.size(); i++) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
    tmp.resize(buf.size());
    const size_t half = (buf.size() + 1) / 2;
    for (size_t i = 0; i < buf.size(); i++) tmp[i] = (i & 1) ? buf[half + i / 2] : buf[i / 2];
    buf.swap(tmp);
}
static decoded_image decode_exr(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 8 || le32(&d[0]) != 20000630u) throw io_error("not an OpenEXR file : " + path);
    const uint32_t version = le32(&d[4]);
    if ((version & 0xff) != 2) throw unsupported_error("OpenEXR file format version " + std::to_string(version & 0xff) + " : " + path);
    if (version & 0x200) throw unsupported_error("tiled OpenEXR files are not read (re-save as scanline) : " + path);
    if (version & 0x1800) throw unsupported_error("multi-part / deep OpenEXR files are not read : " + path);
    struct channel { std::string name; int type; int xs, ys; size_t offset_in_line; };
    std::vector<channel> chans; int compression = -1, line_order = 0; int dw[4] = { 0, 0, -1, -1 }; bool have_dw = false;
    size_t p = 8;
    auto cstr = [&](size_t limit) { std::string t; while (p < d.size() && d[p] && t.size() < limit) t += (char)d[p++]; if (p >= d.size() || d[p]) throw io_error("corrupt OpenEXR header : " + path); p++; return t; };
    for (;;) {
        if (p >= d.size()) throw io_error("truncated OpenEXR header : " + path);
        if (d[p] == 0) { p++; break; }
        const std::string name = cstr(255), type = cstr(255);
        if (p + 4 > d.size()) throw io_error("truncated OpenEXR header : " + path);
        const uint32_t size = le32(&d[p]); p += 4;
        if (p + size > d.size()) throw io_error("truncated OpenEXR header : " + path);
        const size_t a = p; p += size;
        if (name == "channels" && type == "chlist") {
            size_t q = a;
            while (q < a + size && d[q]) {
                channel c; while (q < a + size && d[q]) c.name += (char)d[q++]; q++;
                if (q + 16 > a + size) throw io_error("corrupt OpenEXR channel list : " + path);
                c.type = (int)le32(&d[q]); c.xs = (int)le32(&d[q + 8]); c.ys = (int)le32(&d[q + 12]); c.offset_in_line = 0; q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression" && size >= 1) compression = d[a];
        else if (name == "dataWindow" && size >= 16) { for (int k = 0; k < 4; k++) dw[k] = (int)le32(&d[a + 4 * k]); have_dw = true; }
        else if (name == "lineOrder" && size >= 1) line_order = d[a];
    }
    if (chans.empty() || !have_dw || compression < 0) throw io_error("OpenEXR header lacks channels / dataWindow / compression : " + path);
    static const char* kCodec[] = { "NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB" };
    if (compression > 3) throw unsupported_error(std::string("OpenEXR compression ") + (compression < 10 ? kCodec[compression] : "?") + " is not read (NONE, RLE, ZIPS and ZIP are; re-save with one of them) : " + path);
    const long w = (long)dw[2] - dw[0] + 1, h = (long)dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) throw io_error("corrupt OpenEXR dataWindow : " + path);
    check_image_size((uint64_t)w, (uint64_t)h, d.size(), 2.0 / 1032.0, "OpenEXR", path);   // at least one HALF channel, deflated at best ~1032 : 1
    size_t line_bytes = 0; int idx[3] = { -1, -1, -1 }, lum = -1;
    for (size_t k = 0; k < chans.size(); k++) {
        channel& c = chans[k];
        if (c.xs != 1 || c.ys != 1) throw unsupported_error("sub-sampled OpenEXR channel " + c.name + " : " + path);
        if (c.type < 0 || c.type > 2) throw io_error("corrupt OpenEXR channel type : " + path);
        c.offset_in_line = line_bytes; line_bytes += (size_t)w * (c.type == 1 ? 2 : 4);
        if (c.name == "R") idx[0] = (int)k; else if (c.name == "G") idx[1] = (int)k; else if (c.name == "B") idx[2] = (int)k; else if (c.name == "Y") lum = (int)k;
    }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) { if (lum < 0) throw unsupported_error("OpenEXR file without R, G, B or Y channels : " + path); idx[0] = idx[1] = idx[2] = lum; }
    const int lines_per_block = compression == 3 ? 16 : 1;
    const size_t n_blocks = (size_t)(h + lines_per_block - 1) / lines_per_block;
    if (p + 8 * n_blocks > d.size()) throw io_error("truncated OpenEXR offset table : " + path);
    decoded_image img; img.width = (uint32_t)w; img.height = (uint32_t)h; img.is_float = true; img.rgb.assign((size_t)w * h * 3, 0.0f);
    std::vector<uint8_t> buf, tmp;
    for (size_t b = 0; b < n_blocks; b++) {
        uint64_t off = 0; for (int k = 7; k >= 0; k--) off = (off << 8) | d[p + 8 * b + k];
        // compared without additions: `off` comes from the file and off + 8 wraps for offsets near 2^64
        if (off > d.size() || d.size() - off < 8) throw io_error("truncated OpenEXR (chunk offset) : " + path);
        const long y0 = (long)(int)le32(&d[off]) - dw[1]; const uint32_t size = le32(&d[off + 4]);
        if (y0 < 0 || y0 >= h || size > d.size() - off - 8) throw io_error("corrupt OpenEXR chunk : " + path);
        const long lines = std::min<long>(lines_per_block, h - y0);
        const size_t raw = line_bytes * (size_t)lines;
        const uint8_t* src = &d[off + 8];
        if (size == raw || compression == 0) { if (size < raw) throw io_error("corrupt OpenEXR chunk : " + path); buf.assign(src, src + raw); }   // stored: the codec did not shrink it
        else if (compression == 1) {
            buf.clear(); buf.reserve(raw);
            for (size_t q = 0; q < size;) {
                const int n = (int8_t)src[q++];
                if (n < 0) { if (q + (size_t)-n > size) throw io_error("corrupt OpenEXR RLE data : " + path); buf.insert(buf.end(), src + q, src + q - n); q += (size_t)-n; }
                else { if (q >= size) throw io_error("corrupt OpenEXR RLE data : " + path); buf.insert(buf.end(), (size_t)n + 1, src[q++]); }
                if (buf.size() > raw) throw io_error("corrupt OpenEXR RLE data : " + path);
            }
            if (buf.size() != raw) throw io_error("corrupt OpenEXR RLE data : " + path);
            exr_unpredict_and_interleave(buf, tmp);
        } else {
            buf.resize(raw); uLongf got = (uLongf)raw;
            if (uncompress(buf.data(), &got, src, size) != Z_OK || got != raw) throw io_error("corrupt OpenEXR ZIP data : " + path);
            exr_unpredict_and_interleave(buf, tmp);
        }
        for (long l = 0; l < lines; l++) {
            const uint8_t* line = &buf[line_bytes * (size_t)l];
            float* o = &img.rgb[(size_t)(y0 + l) * w * 3];
            for (int c = 0; c < 3; c++) {
                const channel& ch = chans[idx[c]]; const uint8_t* s = line + ch.offset_in_line;
                for (long x = 0; x < w; x++) {
                    float v;
                    if (ch.type == 1) v = half_bits_to_float(le16(s + 2 * x));
                    else if (ch.type == 2) { const uint32_t u = le32(s + 4 * x); std::memcpy(&v, &u, 4); }
                    else v = (float)le32(s + 4 * x);
                    o[x * 3 + c] = v;
                }
            }
        }
    }
    (void)line_order;   // every chunk names its own y: increasing, decreasing and random order read the same way
    return img;
}

decoded_image load_image_file(const std::string& path) {
    const std::string ext = lower_ext(path);
    if (ext == "jpg" || ext == "jpeg" || ext == "jpe") return decode_jpeg(read_file(path), path);
    const std::vector<uint8_t> d = read_file(path);
    if (ext == "png") return decode_png(d, path);
    if (ext == "bmp") return decode_bmp(d, path);
    if (ext == "tga") return decode_tga(d, path);
    if (ext == "ppm" || ext == "pgm" || ext == "pnm" || ext == "pfm") return decode_pnm(d, path);
    if (ext == "hdr" || ext == "rgbe" || ext == "pic") return decode_hdr(d, path);
    if (ext == "exr") return decode_exr(d, path);
    throw unsupported_error("image format not supported : " + path);
}

uint32_t float3_to_rgbe(float r, float g, float b) {
    float max_ = std::max(r, std::max(g, b));
    if (max_ < 1e-32) return 0;
    int e;
    max_ = (float)std::frexp((double)max_, &e) * 256.0f / max_;
    const uint32_t x = (uint8_t)(r * max_), y = (uint8_t)(g * max_), z = (uint8_t)(b * max_), w = (uint8_t)(e + 128);
    return x | (y << 8) | (z << 16) | (w << 24);
}
uint32_t float3_to_rgbcol(float r, float g, float b) {
    auto to_int = [](float x) { return (uint32_t)(uint8_t)(std::min(1.0f, std::max(0.0f, x)) * 255.0f); };
    return to_int(r) | (to_int(g) << 8) | (to_int(b) << 16) | (255u << 24);
}
uint32_t image_to_texels(const decoded_image& img, std::vector<uint32_t>& texels) {
    const uint32_t w = img.width, h = img.height;
    texels.resize((size_t)w * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint32_t sy = h - 1 - y;   // FreeImage scanline y (bottom-up) = picture row h-1-y
        for (uint32_t x = 0; x < w; x++) {
            if (img.is_float) { const float* s = &img.rgb[((size_t)sy * w + x) * 3]; texels[(size_t)y * w + x] = float3_to_rgbe(s[0], s[1], s[2]); }
            else { const uint8_t* s = &img.rgba8[((size_t)sy * w + x) * 4]; texels[(size_t)y * w + x] = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24); }
        }
    }
    return img.is_float ? CTL_TEXEL_RGBE : CTL_TEXEL_RGBCOL;
}

// ------------------------------------------------------------------------------------------------ writers
static void write_all(const std::string& path, const std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw io_error("Could not open file for writing : " + path);
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    if (!ok) throw io_error("short write : " + path);
}
void write_png(const std::string& path, const float* rgb, uint32_t w, uint32_t h, bool srgb_gamma) {
    std::vector<uint8_t> raw(((size_t)w * 3 + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* line = &raw[((size_t)w * 3 + 1) * y]; line[0] = 0;
        for (size_t i = 0; i < (size_t)w * 3; i++) {
            float v = rgb[(size_t)y * w * 3 + i]; v = std::isfinite(v) ? std::min(1.0f, std::max(0.0f, v)) : 0.0f;
            if (srgb_gamma) v = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
            line[1 + i] = (uint8_t)(v * 255.0f + 0.5f);
        }
    }
    uLongf clen = compressBound((uLong)raw.size()); std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw io_error("PNG deflate failed : " + path);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        const uint8_t l[4] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len };
        out.insert(out.end(), l, l + 4);
        const size_t s = out.size(); out.insert(out.end(), type, type + 4); if (len) out.insert(out.end(), data, data + len);
        const uint32_t c = (uint32_t)crc32(0, &out[s], (uInt)(len + 4));
        const uint8_t cb[4] = { (uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c };
        out.insert(out.end(), cb, cb + 4);
    };
    const uint8_t ihdr[13] = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0 };
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), (uint32_t)clen); chunk("IEND", nullptr, 0);
    write_all(path, out);
}
void write_hdr(const std::string& path, const float* rgb, uint32_t w, uint32_t h) {
    const std::string head = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(h) + " +X " + std::to_string(w) + "\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint32_t t = float3_to_rgbe(std::max(0.0f, rgb[i * 3]), std::max(0.0f, rgb[i * 3 + 1]), std::max(0.0f, rgb[i * 3 + 2]));
        out.push_back((uint8_t)t); out.push_back((uint8_t)(t >> 8)); out.push_back((uint8_t)(t >> 16)); out.push_back((uint8_t)(t >> 24));
    }
    write_all(path, out);
}
void write_pfm(const std::string& path, const float* rgb, uint32_t w, uint32_t h) {
    const std::string head = "PF\n" + std::to_string(w) + " " + std::to_string(h) + "\n-1.0\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    out.resize(head.size() + (size_t)w * h * 12);
    for (uint32_t y = 0; y < h; y++) std::memcpy(&out[head.size() + (size_t)y * w * 12], &rgb[(size_t)(h - 1 - y) * w * 3], (size_t)w * 12);
    write_all(path, out);
}

} // namespace ctl
