// RayTracer with the reference's public surface (src/raytracer.h:20-30) on top of the HIP C-ABI
// (include/ptamd.h).  Differences a caller sees: no GL target (the image is fetched with
// getOutput()/getAccumulator() -- the GPU box is headless), texture arrays are plain float RGBA
// arrays (filled from memory or from Radiance .hdr files, host/image.h) instead of FreeImage-loaded files.
#pragma once
#include "../../include/ptamd.h"
#include "camera.h"
#include "image.h"
#include "scene.h"
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace raytracer {

// what UniqueTextureArray / CLTextureArray hold after loading (src/opencl/texture.h:18-49):
// `layers` RGBA float images of one size, already linear and brightness-scaled
struct TextureArray {
    uint32_t width = 0, height = 0, layers = 0;
    std::vector<float> rgba; // storeAsFloat = true (skydome): r g b a floats
    std::vector<uint8_t> bgra8; // storeAsFloat = false (material textures): b g r a bytes, as the reference uploads them
    bool storeAsFloat() const { return bgra8.empty(); }
    const void* data() const { return storeAsFloat() ? (const void*)rgba.data() : (const void*)bgra8.data(); }
    int add(const float* texels, uint32_t w, uint32_t h)
    {
        if (layers == 0)
            width = w, height = h;
        if (w != width || h != height)
            throw std::invalid_argument("TextureArray: all layers must have the same size");
        if (!bgra8.empty())
            throw std::invalid_argument("TextureArray: float and 8-bit layers cannot be mixed");
        rgba.insert(rgba.end(), texels, texels + (size_t)w * h * 4);
        return (int)layers++;
    }
    // UniqueTextureArray::add(filePath, isLinear = true, brightnessMultiplier) for Radiance .hdr files
    // (src/opencl/texture.cpp:18-49,72-120): the first file fixes the layer size unless one was set, later ones
    // are rescaled to it
    int add(const std::string& hdrFile, float brightnessMultiplier = 1.0f)
    {
        ImageRGBAF img = loadRadianceHDR(hdrFile);
        if (layers == 0 && width == 0)
            width = img.width, height = img.height;
        img = loadSkydomeLayer(hdrFile, width, height, brightnessMultiplier);
        return add(img.rgba.data(), img.width, img.height);
    }
    // CLTextureArray(files, ..., width, height, storeAsFloat = false): every registered file as a layer of that size
    static TextureArray fromFiles(const UniqueTextureFiles& files, uint32_t w, uint32_t h)
    {
        TextureArray t;
        t.width = w, t.height = h;
        for (const TextureFile& f : files.files())
            t.addMaterial(f.path, f.isLinear);
        return t;
    }
    // UniqueTextureArray::add(filePath, isLinear) for the 8-bit material array (PNG files: every texture the reference ships)
    int addMaterial(const std::string& pngFile, bool isLinear = false)
    {
        if (!rgba.empty())
            throw std::invalid_argument("TextureArray: float and 8-bit layers cannot be mixed");
        const ImageRGBA8 img = loadMaterialLayerBGRA8(pngFile, width, height, isLinear); // 0 x 0: the first file fixes the layer size
        if (layers == 0)
            width = img.width, height = img.height;
        if (img.width != width || img.height != height)
            throw std::invalid_argument("TextureArray: all layers must have the same size");
        bgra8.insert(bgra8.end(), img.rgba.begin(), img.rgba.end());
        return (int)layers++;
    }
};

class RayTracer {
public:
    RayTracer(int width, int height, std::shared_ptr<Scene> scene, const TextureArray& materialTextures, const TextureArray& skydomeTextures,
        int device = 0, uint32_t seed = 1);
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    void rayTrace(const Camera& camera); // one sample per pixel; resets the accumulation when the camera changed
    void frameTick(); // re-flatten lights + top-level BVH after scene-graph transforms changed (asynchronous upload + flip)
    void updateGeometry(); // after Mesh::refit: new vertices (the device refits its trees); takes effect with the next frameTick.  Not between rebuildGeometry() and
                           // its frameTick(): the device library refuses a refit while a rebuilt scene waits to be adopted (PT_ERR_STATE -> std::runtime_error)
    void rebuildGeometry(); // after meshes of the scene were REPLACED (a new tree per frame, MeshSequence's other branch): converted and copied beside the scene that is rendering; adopted by the next frameTick
    int getSamplesPerPixel() const;
    int getMaxSamplesPerPixel() const { return 20000000; } // MAX_SAMPLES_PER_PIXEL, src/raytracer.cpp:38

    std::vector<float> getOutput(); // width*height RGBA in [0,1]: the `accumulate` kernel's image
    std::vector<float> getAccumulator(); // width*height float4 HDR sums
    pt_stats getStats();
    pt_ctx* context() { return m_ctx; }

private:
    void check(int rc, const char* what);
    void upload(const TextureArray& materialTextures, const TextureArray& skydomeTextures); // constructor body proper
    pt_ctx* m_ctx = nullptr;
    std::shared_ptr<Scene> m_scene;
    FlattenedScene m_flat;
    CameraData m_prevCamera;
    bool m_haveCamera = false;
    int m_width, m_height;
};

} // namespace raytracer
