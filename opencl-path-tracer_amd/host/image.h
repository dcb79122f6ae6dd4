// Radiance RGBE (.hdr) and PNG readers + the texture-array preparation of the reference's CLTextureArray::loadImage
// (src/opencl/texture.cpp:72-120): decode, rescale to the array's fixed layer size (Lanczos-3, like
// FreeImage_Rescale(..., FILTER_LANCZOS3)), expand to RGBA32F with alpha 1 and apply the brightness multiplier.
// Rows come out BOTTOM-UP like FreeImage_GetBits hands them to enqueueWriteImage, so an image loaded here lands
// in the texture array the way the reference's does.  Not bit-identical to FreeImage's resampler (its kernel
// normalisation and edge handling are its own: "parity unpinned at the FreeImage boundary");
// same-size images pass through exactly.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace raytracer {

struct ImageRGBA8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba; // height * width * 4, row 0 = TOP row of the picture (file order)
};

struct ImageRGBAF {
    uint32_t width = 0, height = 0;
    std::vector<float> rgba; // height * width * 4, row 0 = bottom row of the picture
};

// throws std::runtime_error on malformed files
ImageRGBAF loadRadianceHDR(const std::string& path);
// Lanczos-3 separable resampling (clamped edges); identity when the size is unchanged
ImageRGBAF rescaleLanczos3(const ImageRGBAF& in, uint32_t width, uint32_t height, bool keepAlpha = false);
// loadImage(filePath, isLinear = true, brightnessMultiplier) for a float array of layer size width x height
ImageRGBAF loadSkydomeLayer(const std::string& path, uint32_t width, uint32_t height, float brightnessMultiplier);

// PNG decoder: all colour types and bit depths (16-bit samples keep their high byte), PLTE / tRNS, Adam7 interlacing,
// CRC-checked chunks; inflate is zlib's.  Throws std::runtime_error on malformed files.
ImageRGBA8 loadPNG(const std::string& path);
// loadImage(filePath, isLinear, 1.0) for the 8-bit material array (texture.cpp:84-92,112-131): decode, Lanczos-3 rescale to
// the layer size (0 = keep), back to 8 bits, FreeImage_AdjustGamma(1 / 2.2) on the colour channels unless isLinear,
// alpha kept (alpha-0 texels are cut-outs, shading.cl:587-601).  Returns what read_imagef yields for those bytes
// (value / 255), rows bottom-up.
ImageRGBAF loadMaterialLayer(const std::string& path, uint32_t width, uint32_t height, bool isLinear);

} // namespace raytracer
