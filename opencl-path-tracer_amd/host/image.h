// Radiance RGBE (.hdr) reader + the texture-array preparation of the reference's CLTextureArray::loadImage
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

struct ImageRGBAF {
    uint32_t width = 0, height = 0;
    std::vector<float> rgba; // height * width * 4, row 0 = bottom row of the picture
};

// throws std::runtime_error on malformed files
ImageRGBAF loadRadianceHDR(const std::string& path);
// Lanczos-3 separable resampling (clamped edges); identity when the size is unchanged
ImageRGBAF rescaleLanczos3(const ImageRGBAF& in, uint32_t width, uint32_t height);
// loadImage(filePath, isLinear = true, brightnessMultiplier) for a float array of layer size width x height
ImageRGBAF loadSkydomeLayer(const std::string& path, uint32_t width, uint32_t height, float brightnessMultiplier);

} // namespace raytracer
