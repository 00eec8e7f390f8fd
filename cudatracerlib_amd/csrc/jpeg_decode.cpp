// jpeg_decode.cpp — Huffman JPEG (ITU T.81; SOF0 baseline, SOF1 extended sequential, SOF2 progressive; 8-bit samples) for texture files.
// The reference reads JPEG through FreeImage -> libjpeg; this decoder follows the standard's decoding procedures (F.2 sequential, G.1.2
// progressive: spectral selection + successive approximation, any number of scans) into a coefficient buffer, then a float inverse DCT and
// libjpeg's default "fancy" (triangle) chroma up-sampling, so texels agree with libjpeg's to about one 8-bit step (its integer IDCT rounds
// differently).  Arithmetic-coded, lossless, hierarchical and 12-bit files are rejected with a message.
#include "image_io.h"
#include "mitsuba_loader.h"   // io_error / unsupported_error
#include <cmath>
#include <cstring>
#include <algorithm>

namespace ctl {
namespace {

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct huff_table {   // canonical code tables of Annex C / F.2.2.3
    bool present = false; uint8_t vals[256]; int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t counts[16], const uint8_t* symbols) {
        present = true;
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < counts[l - 1]; i++) vals[k++] = *symbols++;
            code += counts[l - 1];
            maxcode[l] = counts[l - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int bw = 0, bh = 0;            // plane size in samples, padded to whole MCUs
    int nbx = 0, nby = 0;          // blocks that carry picture (a non-interleaved scan codes exactly these, A.2.3)
    std::vector<int16_t> coef;     // [block row][block column][64], natural order, not yet dequantised
    std::vector<uint8_t> plane;
    int16_t* block(int bx, int by) { return &coef[((size_t)by * (bw / 8) + bx) * 64]; }
};

struct bit_reader {
    const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0; bool hit_marker = false;
    int bit() {
        if (n == 0) {
            if (p >= end || hit_marker) { acc = 0; n = 8; }   // past the data: feed zeros (truncated files decode to grey instead of crashing)
            else {
                uint8_t b = *p++;
                if (b == 0xff) { if (p < end && *p == 0) p++; else { hit_marker = true; p--; b = 0; } }
                acc = b; n = 8;
            }
        }
        n--; return (acc >> n) & 1;
    }
    int bits(int c) { int v = 0; while (c--) v = (v << 1) | bit(); return v; }
    void restart() { n = 0; hit_marker = false; }
};

int decode_symbol(bit_reader& br, const huff_table& h, const std::string& path) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    throw io_error("corrupt JPEG (bad Huffman code) : " + path);
}
inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }   // F.2.2.1

void idct8x8(const float* in, uint8_t* out, int stride) {   // separable inverse DCT, level shift, clamp
    static float c[8][8]; static bool init = false;
    if (!init) { for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f); init = true; }
    float tmp[64];
    for (int v = 0; v < 8; v++) for (int x = 0; x < 8; x++) { float s = 0; for (int u = 0; u < 8; u++) s += c[x][u] * in[v * 8 + u]; tmp[v * 8 + x] = s; }
    for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {
        float s = 0; for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
        const int q = (int)std::floor(s + 128.5f);
        out[y * stride + x] = (uint8_t)std::min(255, std::max(0, q));
    }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// ---- per-block entropy decoding: F.2.2 (sequential) and G.1.2 (progressive)
struct scan_params { int ss = 0, se = 63, ah = 0, al = 0; int eobrun = 0; bool progressive = false; };

static void decode_block(bit_reader& br, component& c, int16_t* blk, scan_params& sp, const huff_table* dc, const huff_table* ac, const std::string& path) {
    if (!sp.progressive) {   // DC difference + the 63 AC coefficients of one block
        const int t = decode_symbol(br, dc[c.td], path);
        if (t > 11) throw io_error("corrupt JPEG (DC size) : " + path);
        c.pred += t ? extend(br.bits(t), t) : 0;
        blk[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, ac[c.ta], path), r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) throw io_error("corrupt JPEG (AC run) : " + path);
            blk[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
            k++;
        }
        return;
    }
    if (sp.ss == 0) {        // DC scan: first pass codes the difference of the point-transformed value, a refinement pass one more bit (G.1.2.1)
        if (sp.ah == 0) {
            const int t = decode_symbol(br, dc[c.td], path);
            if (t > 11) throw io_error("corrupt JPEG (DC size) : " + path);
            c.pred += t ? extend(br.bits(t), t) : 0;
            blk[0] = (int16_t)(c.pred * (1 << sp.al));
        } else if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << sp.al));
        return;
    }
    const int p1 = 1 << sp.al, m1 = -(1 << sp.al);
    if (sp.ah == 0) {        // AC first pass of a band (G.1.2.2): run / size symbols plus end-of-band runs that span blocks
        if (sp.eobrun > 0) { sp.eobrun--; return; }
        for (int k = sp.ss; k <= sp.se; k++) {
            const int rs = decode_symbol(br, ac[c.ta], path), r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                if (k > sp.se) throw io_error("corrupt JPEG (AC run) : " + path);
                blk[kZigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * p1);
            } else if (r == 15) k += 15;
            else { sp.eobrun = (1 << r) - 1 + (r ? br.bits(r) : 0); break; }
        }
        return;
    }
    // AC refinement pass (G.1.2.3): every coefficient that is already non-zero gets one correction bit as it is passed; zero-history
    // coefficients are counted by the run lengths, and a newly non-zero coefficient is +-1 << Al
    int k = sp.ss;
    if (sp.eobrun == 0) {
        for (; k <= sp.se; k++) {
            const int rs = decode_symbol(br, ac[c.ta], path); int r = rs >> 4; const int sz = rs & 15; int value = 0;
            if (sz) {
                if (sz != 1) throw io_error("corrupt JPEG (refinement size) : " + path);
                value = br.bit() ? p1 : m1;
            } else if (r != 15) { sp.eobrun = (1 << r) + (r ? br.bits(r) : 0); break; }
            for (; k <= sp.se; k++) {
                int16_t& co = blk[kZigzag[k]];
                if (co != 0) { if (br.bit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1)); }
                else { if (r == 0) break; r--; }
            }
            if (value && k <= sp.se) blk[kZigzag[k]] = (int16_t)value;
        }
    }
    if (sp.eobrun > 0) {
        for (; k <= sp.se; k++) {
            int16_t& co = blk[kZigzag[k]];
            if (co != 0 && br.bit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
        }
        sp.eobrun--;
    }
}

}  // namespace

decoded_image decode_jpeg(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) throw io_error("not a JPEG file : " + path);
    uint16_t qt[4][64] = {}; bool have_qt[4] = {};
    huff_table dc[4], ac[4];
    std::vector<component> comps; int width = 0, height = 0, restart_interval = 0, adobe_transform = -1; bool have_frame = false, progressive = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0, n_scans = 0;
    size_t pos = 2;
    while (pos + 4 <= d.size()) {
        if (d[pos] != 0xff) { pos++; continue; }
        const uint8_t m = d[pos + 1];
        if (m == 0xff) { pos++; continue; }
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01 || m == 0x00) { pos += 2; continue; }
        if (m == 0xd9) break;
        const size_t len = be16(&d[pos + 2]);
        if (len < 2 || pos + 2 + len > d.size()) throw io_error("corrupt JPEG (segment length) : " + path);
        const uint8_t* s = &d[pos + 4]; const size_t n = len - 2;
        if (m == 0xdb) {   // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15; i++;
                if (tq > 3 || i + (pq ? 128 : 64) > n) throw io_error("corrupt JPEG (DQT) : " + path);
                for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? be16(&s[i + 2 * k]) : s[i + k]; }
                have_qt[tq] = true; i += pq ? 128 : 64;
            }
        } else if (m == 0xc4) {   // DHT
            size_t i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15; int total = 0;
                for (int k = 0; k < 16; k++) total += s[i + 1 + k];
                if (th > 3 || tc > 1 || total > 256 || i + 17 + total > n) throw io_error("corrupt JPEG (DHT) : " + path);
                (tc ? ac : dc)[th].build(&s[i + 1], &s[i + 17]);
                i += 17 + total;
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {   // SOF0 / SOF1 / SOF2
            if (have_frame) throw io_error("corrupt JPEG (two frame headers) : " + path);
            if (n < 6 || s[0] != 8) throw unsupported_error("JPEG with a sample precision other than 8 bits : " + path);
            progressive = m == 0xc2;
            height = be16(&s[1]); width = be16(&s[3]);
            const int nc = s[5];
            if ((uint64_t)width * (uint64_t)height > (1ull << 28) || (double)width * height / 4096.0 > (double)d.size()) throw io_error("unreasonable JPEG dimensions for a file of this size : " + path);   // before the coefficient planes are allocated; an all-DC scan spends about a bit per 8x8 block
            if ((nc != 1 && nc != 3) || n < (size_t)(6 + 3 * nc) || width <= 0 || height <= 0) throw unsupported_error("JPEG with " + std::to_string(nc) + " components (only greyscale and YCbCr / RGB are read) : " + path);
            comps.resize(nc);
            for (int k = 0; k < nc; k++) { comps[k].id = s[6 + 3 * k]; comps[k].h = s[7 + 3 * k] >> 4; comps[k].v = s[7 + 3 * k] & 15; comps[k].tq = s[8 + 3 * k];
                if (comps[k].h < 1 || comps[k].h > 4 || comps[k].v < 1 || comps[k].v > 4 || comps[k].tq > 3) throw io_error("corrupt JPEG (SOF) : " + path); }
            for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
            mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8;
                c.nbx = ((width * c.h + hmax - 1) / hmax + 7) / 8; c.nby = ((height * c.v + vmax - 1) / vmax + 7) / 8;
                c.coef.assign((size_t)c.bw * c.bh, 0);
            }
            have_frame = true;
        } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) throw unsupported_error("lossless / hierarchical / arithmetic-coded JPEG is not read : " + path);
        else if (m == 0xdd) { if (n >= 2) restart_interval = be16(s); }
        else if (m == 0xee) { if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11]; }
        else if (m == 0xda) {   // SOS: scan header, then the entropy-coded segment up to the next marker that is not RSTn
            if (!have_frame) throw io_error("corrupt JPEG (scan before frame header) : " + path);
            const int ns = n ? s[0] : 0;
            if (ns < 1 || ns > (int)comps.size() || n < (size_t)(1 + 2 * ns + 3)) throw io_error("corrupt JPEG (SOS) : " + path);
            std::vector<component*> in_scan;
            for (int k = 0; k < ns; k++) {
                component* found = nullptr;
                for (auto& c : comps) if (c.id == s[1 + 2 * k]) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = &c; }
                if (!found || found->td > 3 || found->ta > 3) throw io_error("corrupt JPEG (SOS component) : " + path);
                in_scan.push_back(found);
            }
            scan_params sp; sp.progressive = progressive;
            sp.ss = s[1 + 2 * ns]; sp.se = s[2 + 2 * ns]; sp.ah = s[3 + 2 * ns] >> 4; sp.al = s[3 + 2 * ns] & 15;
            if (!progressive) { sp.ss = 0; sp.se = 63; sp.ah = sp.al = 0; }
            else if (sp.ss > sp.se || sp.se > 63 || sp.al > 13 || (sp.ss == 0 && sp.se != 0) || (sp.ss > 0 && ns != 1)) throw io_error("corrupt JPEG (progressive scan parameters) : " + path);
            for (component* c : in_scan) {
                const bool need_dc = !progressive || (sp.ss == 0 && sp.ah == 0), need_ac = !progressive || sp.ss > 0;
                if ((need_dc && !dc[c->td].present) || (need_ac && !ac[c->ta].present)) throw io_error("corrupt JPEG (missing table) : " + path);
                c->pred = 0;
            }
            const uint8_t* scan_begin = &d[pos + 2 + len];
            const uint8_t* scan_end = scan_begin;   // the next marker other than RSTn / a stuffed zero
            while (scan_end + 1 < d.data() + d.size() && !(scan_end[0] == 0xff && scan_end[1] != 0 && scan_end[1] != 0xff && !(scan_end[1] >= 0xd0 && scan_end[1] <= 0xd7))) scan_end++;
            if (scan_end + 1 >= d.data() + d.size()) scan_end = d.data() + d.size();
            bit_reader br{ scan_begin, scan_end };
            int to_restart = restart_interval;
            auto maybe_restart = [&]() {
                if (!restart_interval) return;
                if (to_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictors and the end-of-band run
                    const uint8_t* q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
                    if (q + 1 < br.end) br.p = q + 2;
                    br.restart();
                    for (component* c : in_scan) c->pred = 0;
                    sp.eobrun = 0;
                    to_restart = restart_interval;
                }
                to_restart--;
            };
            if (ns == 1) {   // non-interleaved: the component's own blocks in raster order, one block per MCU
                component& c = *in_scan[0];
                for (int by = 0; by < c.nby; by++)
                    for (int bx = 0; bx < c.nbx; bx++) { maybe_restart(); decode_block(br, c, c.block(bx, by), sp, dc, ac, path); }
            } else {
                for (int my = 0; my < mcuy; my++)
                    for (int mx = 0; mx < mcux; mx++) {
                        maybe_restart();
                        for (component* c : in_scan)
                            for (int by = 0; by < c->v; by++)
                                for (int bx = 0; bx < c->h; bx++) decode_block(br, *c, c->block(mx * c->h + bx, my * c->v + by), sp, dc, ac, path);
                    }
            }
            // every scan walks all blocks of its components: bound their number (libjpeg-turbo's own limit is configurable; real progressive files have ~10)
            if (++n_scans > 256) throw io_error("corrupt JPEG (more than 256 scans) : " + path);
            pos = (size_t)(scan_end - d.data());
            continue;
        }
        pos += 2 + len;
    }
    if (!n_scans) throw io_error("corrupt JPEG (no scan) : " + path);
    for (auto& c : comps) {   // dequantise + inverse DCT of every block
        if (!have_qt[c.tq]) throw io_error("corrupt JPEG (missing table) : " + path);
        c.plane.assign((size_t)c.bw * c.bh, 128);
        for (int by = 0; by < c.bh / 8; by++)
            for (int bx = 0; bx < c.bw / 8; bx++) {
                const int16_t* co = c.block(bx, by); float blk[64];
                for (int k = 0; k < 64; k++) blk[k] = (float)co[k] * qt[c.tq][k];
                idct8x8(blk, &c.plane[(size_t)(by * 8) * c.bw + bx * 8], c.bw);
            }
    }
    // up-sample every component to full resolution: libjpeg's "fancy" triangle filters for 2:1, replication otherwise
    auto sample = [&](const component& c, int x, int y) -> float {
        const int sx = hmax / c.h, sy = vmax / c.v;
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;   // the component's true size
        auto at = [&](int i, int j) { i = std::min(std::max(i, 0), cw - 1); j = std::min(std::max(j, 0), ch - 1); return (float)c.plane[(size_t)j * c.bw + i]; };
        if (sx == 1 && sy == 1) return at(x, y);
        if (sx == 2 && (sy == 1 || sy == 2) && hmax % c.h == 0 && vmax % c.v == 0) {
            const int i = x >> 1, i2 = (x & 1) ? i + 1 : i - 1;
            if (sy == 1) return 0.75f * at(i, y) + 0.25f * at(i2, y);
            const int j = y >> 1, j2 = (y & 1) ? j + 1 : j - 1;
            return 0.5625f * at(i, j) + 0.1875f * at(i2, j) + 0.1875f * at(i, j2) + 0.0625f * at(i2, j2);
        }
        return at(x * c.h / hmax, y * c.v / vmax);
    };
    decoded_image img; img.width = (uint32_t)width; img.height = (uint32_t)height; img.rgba8.resize((size_t)width * height * 4);
    auto u8 = [](float v) { return (uint8_t)std::min(255.0f, std::max(0.0f, std::floor(v + 0.5f))); };
    const bool ycc = comps.size() == 3 && adobe_transform != 0 && !(comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            uint8_t* o = &img.rgba8[((size_t)y * width + x) * 4];
            if (comps.size() == 1) { o[0] = o[1] = o[2] = u8(sample(comps[0], x, y)); }
            else {
                const float a = sample(comps[0], x, y), b = sample(comps[1], x, y), c = sample(comps[2], x, y);
                if (ycc) { o[0] = u8(a + 1.402f * (c - 128.0f)); o[1] = u8(a - 0.344136f * (b - 128.0f) - 0.714136f * (c - 128.0f)); o[2] = u8(a + 1.772f * (b - 128.0f)); }
                else { o[0] = u8(a); o[1] = u8(b); o[2] = u8(c); }
            }
            o[3] = 255;
        }
    return img;
}

}  // namespace ctl
