// image_io.h — image file decoding / encoding for textures, environment maps and render output.
// The reference goes through FreeImage (Engine/MIPMap.cu:542-592, Engine/Image.cpp:67-75), which is not vendored; this reads
// PNG (zlib), JPEG (baseline, extended sequential and progressive Huffman), BMP, TGA, PPM/PGM, PFM, Radiance HDR and OpenEXR (single-part scanline files,
// HALF / FLOAT / UINT channels, NONE / RLE / ZIPS / ZIP) and writes PNG, HDR and PFM.  PIZ and the lossy EXR codecs, tiled EXR and arithmetic-coded JPEG are rejected.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ctl {

struct decoded_image {
    uint32_t width = 0, height = 0;
    bool is_float = false;            // true: `rgb` holds 3 floats per pixel; false: `rgba8` holds 4 bytes per pixel
    std::vector<float> rgb;           // rows top-down
    std::vector<uint8_t> rgba8;       // rows top-down
};

// throws io_error (missing / corrupt file) or unsupported_error (format not built in)
decoded_image load_image_file(const std::string& path);
decoded_image decode_jpeg(const std::vector<uint8_t>& bytes, const std::string& path);   // jpeg_decode.cpp: sequential and progressive Huffman JPEG

// Level 0 of a KernelMIPMap as parseImage builds it (Engine/MIPMap.cu:565-586): FreeImage scanlines are bottom-up, so texel row 0
// is the BOTTOM row of the picture; float images become RGBE (SpectrumConverter::Float3ToRGBE), others RGBCOL.
// Returns CTL_TEXEL_RGBE or CTL_TEXEL_RGBCOL.
uint32_t image_to_texels(const decoded_image& img, std::vector<uint32_t>& texels);

uint32_t float3_to_rgbe(float r, float g, float b);     // Math/Spectrum.h:534-555
uint32_t float3_to_rgbcol(float r, float g, float b);   // Math/Spectrum.h:521-526

// output (rows top-down, 3 floats per pixel)
void write_png(const std::string& path, const float* rgb, uint32_t w, uint32_t h, bool srgb_gamma);
void write_hdr(const std::string& path, const float* rgb, uint32_t w, uint32_t h);
void write_pfm(const std::string& path, const float* rgb, uint32_t w, uint32_t h);

} // namespace ctl
