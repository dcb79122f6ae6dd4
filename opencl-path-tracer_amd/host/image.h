// Radiance RGBE (.hdr) and PNG readers + the texture-array preparation of the reference's CLTextureArray::loadImage
// (src/opencl/texture.cpp:72-120): decode, rescale to the array's fixed layer size (Lanczos-3, like
// FreeImage_Rescale(..., FILTER_LANCZOS3)), expand to RGBA32F with alpha 1 and apply the brightness multiplier.
// Rows come out BOTTOM-UP like FreeImage_GetBits hands them to enqueueWriteImage, so an image loaded here lands
// in the texture array the way the reference's does.  Not bit-identical to FreeImage's resampler (its kernel
// normalisation and edge handling are its own: "parity unpinned at the FreeImage boundary");
// same-size images pass through exactly.
#pragma once
#include <cstdint>
#include <map>
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
// ... and as the bytes themselves: b, g, r, a per texel, rows bottom-up (the FreeImage 32-bit bitmap the reference uploads into
// its CL_BGRA / CL_UNORM_INT8 image array; `rgba` holds them in that order)
ImageRGBA8 loadMaterialLayerBGRA8(const std::string& path, uint32_t width, uint32_t height, bool isLinear);

// UniqueTextureArray (src/opencl/texture.h:18-31, texture.cpp:9-24): the files a scene's materials refer to, each
// once, in first-use order; the index is the material's tex_id = the layer of the texture array.
struct TextureFile {
    std::string path;
    bool isLinear = false;
    float brightnessMultiplier = 1.0f;
};
class UniqueTextureFiles {
public:
    int add(const std::string& path, bool isLinear = false, float brightnessMultiplier = 1.0f)
    {
        auto it = m_lookup.find(path);
        if (it != m_lookup.end())
            return it->second;
        m_files.push_back({ path, isLinear, brightnessMultiplier });
        return m_lookup[path] = (int)m_files.size() - 1;
    }
    const std::vector<TextureFile>& files() const { return m_files; }

private:
    std::vector<TextureFile> m_files;
    std::map<std::string, int> m_lookup;
};

} // namespace raytracer
