// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_mipmap_driver.cpp — extern "C" driver around the reference's own KernelMIPMap (Engine/MIPMap.cu:13-278: Texel, triangle, evalEWA, Sample, SampleAlpha,
// eval).  MIPMap.cu as a whole needs FreeImage for its host half, so `make ref` extracts that line range at BUILD time into oracle/_ref/gen/ (git-ignored;
// no reference text is committed) behind `#include <Engine/MIPMap.h>`, together with the four Spectrum <-> RGBE / RGBCOL methods of Math/Spectrum.cu:179-187,260-284.
// This file contains no reference source: it includes the reference header and calls its functions.  Texel / triangle / evalEWA are private members there;
// the `#define private public` below opens them for this translation unit only (the class layout is unchanged).
#define private public
#include <Engine/MIPMap_device.h>
#undef private
#include <Math/Spectrum.h>
#include <cstdint>
#include <cstring>

using namespace CudaTracerLib;

extern "C" {

// hdr = {width, height, texel type (0 RGBE, 1 RGBCOL), wrap mode, filter mode, levels}; offsets16 = m_sOffsets; texels = all levels; lut64 = m_weightLut.
// what: 0 Texel(level, uv)   1 triangle(level, uv)   2 evalEWA(level, uv, A, B, C)   3 eval(uv, d0, d1)   4 Sample(uv)   5 Sample(uv, width)
//       6 SampleAlpha(uv) (out[0])   7 Sample(width, x, y)
// args: 8 floats per query = {u, v, p0, p1, p2, p3, level, 0}: (p0,p1) = d0 / (A,B) / width / (width,x) ; (p2,p3) = d1 / (C,-) / - / (y,-)
void ref_mipmap_query(const uint32_t* texels, const uint32_t* hdr, const uint32_t* offsets16, const float* lut64, int what, int n, const float* args, float* out3) {
    KernelMIPMap K; std::memset(&K, 0, sizeof(K));
    K.m_pHostData = const_cast<unsigned int*>(texels); K.m_pDeviceData = nullptr;
    K.m_uWidth = hdr[0]; K.m_uHeight = hdr[1]; K.m_fDim = Vec2f((float)hdr[0], (float)hdr[1]);
    K.m_uType = (Texture_DataType)hdr[2]; K.m_uWrapMode = (ImageWrap)hdr[3]; K.m_uFilterMode = (ImageFilter)hdr[4]; K.m_uLevels = hdr[5];
    std::memcpy(K.m_sOffsets, offsets16, sizeof(K.m_sOffsets)); std::memcpy(K.m_weightLut, lut64, sizeof(K.m_weightLut));
    for (int i = 0; i < n; i++) {
        const float* a = args + 8 * i; const Vec2f uv(a[0], a[1]); const unsigned level = (unsigned)a[6];
        Spectrum s(0.0f);
        switch (what) {
        case 0: s = K.Texel(level, uv); break;
        case 1: s = K.triangle(level, uv); break;
        case 2: s = K.evalEWA(level, uv, a[2], a[3], a[4]); break;
        case 3: s = K.eval(uv, Vec2f(a[2], a[3]), Vec2f(a[4], a[5])); break;
        case 4: s = K.Sample(uv); break;
        case 5: s = K.Sample(uv, a[2]); break;
        case 6: s = Spectrum(K.SampleAlpha(uv)); break;
        case 7: s = K.Sample(a[2], (int)a[3], (int)a[4]); break;
        }
        float r, g, b; s.toLinearRGB(r, g, b);
        out3[3 * i] = r; out3[3 * i + 1] = g; out3[3 * i + 2] = b;
    }
}

}  // extern "C"
