// ORACLE — TEST INFRASTRUCTURE ONLY.
// ref_image_driver.cpp — extern "C" driver around the reference's own Image::AddSample (Engine/Image.cu:22-44, host branch): Spectrum::clampNegative, Floor2Int of the film
// position, the bounds / isNaN / isValid test that DROPS a sample, the += of rgb and weightSum.  `make ref` compiles lines 1-44 of Engine/Image.cu (its includes, the splat
// helper and AddSample) through a build-time extract under oracle/_ref/gen/ (git-ignored) behind the reference's own Engine/Image.h; the rest of the file (Splat, the
// FreeImage writers, Clear, the luminance reduction) needs FreeImage and the CUDA runtime.  The class's only constructor allocates through CUDA_MALLOC and its members are
// private, so the driver lays the object out in raw storage in the member order of Engine/Image.h:31-91 / Base/SynchronizedBuffer.h:18-56,160-165 (checked against sizeof, and
// at run time against the class's own public accessors getWidth / getHeight / getPixelData) and calls the member function on it.  This file contains no reference source.
#include <Engine/Image.h>
#include <cstdint>
#include <cstring>

using namespace CudaTracerLib;

namespace {
struct sync_buffer_layout { void* vptr; int location; unsigned length; void* host; void* device; };                    // SynchronizedBuffer<T>: ISynchronizedBuffer {vptr, m_location} + m_length, m_hostData, m_deviceData
struct image_layout { void* vptr; int location; void* buffers[3]; int xres, yres; sync_buffer_layout pixels; void* filtered; bool owns; void* view; };   // ISynchronizedBufferParent {vptr, m_location, std::vector} + Image's members
static_assert(sizeof(sync_buffer_layout) == sizeof(SynchronizedBuffer<PixelData>), "member layout of SynchronizedBuffer<PixelData>");
static_assert(sizeof(image_layout) == sizeof(Image), "member layout of Image");
static_assert(sizeof(PixelData) == 28, "PixelData is seven floats");
}  // namespace

extern "C" {

// pixels: w x h PixelData (rgb[3], rgbSplat[3], weightSum), added to; samples: n x {sx, sy, r, g, b}.  Returns 0, or -2 when the raw layout does not answer the class's accessors.
int ref_image_add_samples(void* pixels, int w, int h, int n, const float* samples) {
    if (w < 2 || h < 1) return -1;
    alignas(16) unsigned char raw[sizeof(Image)];
    image_layout L; std::memset(&L, 0, sizeof L);
    L.location = DataLocation::Synchronized; L.xres = w; L.yres = h;
    L.pixels.location = DataLocation::Synchronized; L.pixels.length = (unsigned)(w * h); L.pixels.host = pixels;
    std::memcpy(raw, &L, sizeof L);
    Image* img = reinterpret_cast<Image*>(raw);
    if ((int)img->getWidth() != w || (int)img->getHeight() != h || &img->getPixelData(1, h - 1) != (PixelData*)pixels + ((h - 1) * w + 1)) return -2;
    for (int i = 0; i < n; i++) img->AddSample(samples[5 * i], samples[5 * i + 1], Spectrum(samples[5 * i + 2], samples[5 * i + 3], samples[5 * i + 4]));
    return 0;
}

}  // extern "C"
