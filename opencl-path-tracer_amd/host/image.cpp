#include "image.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace raytracer {

namespace {
inline void rgbeToFloat(const uint8_t p[4], float* out)
{
    if (p[3] == 0) {
        out[0] = out[1] = out[2] = 0.0f;
    } else {
        const float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
        out[0] = p[0] * f, out[1] = p[1] * f, out[2] = p[2] * f;
    }
    out[3] = 1.0f;
}
} // namespace

ImageRGBAF loadRadianceHDR(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    std::string line;
    if (!std::getline(in, line) || (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0))
        throw std::runtime_error(path + ": not a Radiance picture");
    bool rgbe = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (line.empty())
            break; // end of header
        if (line.rfind("FORMAT=", 0) == 0)
            rgbe = line == "FORMAT=32-bit_rle_rgbe";
    }
    if (!rgbe)
        throw std::runtime_error(path + ": only FORMAT=32-bit_rle_rgbe is supported");
    if (!std::getline(in, line))
        throw std::runtime_error(path + ": missing resolution line");
    char sy = 0, sx = 0, ay = 0, ax = 0;
    long h = 0, w = 0;
    if (std::sscanf(line.c_str(), "%c%c %ld %c%c %ld", &sy, &ay, &h, &sx, &ax, &w) != 6 || ay != 'Y' || ax != 'X' || h <= 0 || w <= 0 || h > 65535 || w > 65535)
        throw std::runtime_error(path + ": unsupported resolution line '" + line + "'");
    const bool topDown = sy == '-', leftToRight = sx == '+';
    ImageRGBAF img;
    img.width = (uint32_t)w, img.height = (uint32_t)h;
    img.rgba.resize((size_t)w * h * 4);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (long row = 0; row < h; row++) {
        uint8_t head[4];
        in.read((char*)head, 4);
        if (!in)
            throw std::runtime_error(path + ": truncated");
        if (w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && (head[2] & 0x80) == 0) {
            // adaptive run-length encoding, one component plane at a time
            if ((((long)head[2]) << 8 | head[3]) != w)
                throw std::runtime_error(path + ": scanline length mismatch");
            for (int comp = 0; comp < 4; comp++) {
                long x = 0;
                while (x < w) {
                    int count = in.get();
                    if (count < 0)
                        throw std::runtime_error(path + ": truncated");
                    if (count > 128) { // run
                        count -= 128;
                        const int value = in.get();
                        if (value < 0 || x + count > w)
                            throw std::runtime_error(path + ": bad run");
                        for (int k = 0; k < count; k++)
                            scan[(size_t)(x++) * 4 + comp] = (uint8_t)value;
                    } else { // literal
                        if (count == 0 || x + count > w)
                            throw std::runtime_error(path + ": bad literal run");
                        for (int k = 0; k < count; k++) {
                            const int value = in.get();
                            if (value < 0)
                                throw std::runtime_error(path + ": truncated");
                            scan[(size_t)(x++) * 4 + comp] = (uint8_t)value;
                        }
                    }
                }
            }
        } else { // flat scanline (the 4 bytes read are its first pixel)
            std::memcpy(scan.data(), head, 4);
            in.read((char*)scan.data() + 4, (std::streamsize)(w - 1) * 4);
            if (!in)
                throw std::runtime_error(path + ": truncated");
        }
        const long pictureRow = topDown ? row : h - 1 - row; // 0 = top of the picture
        float* dst = &img.rgba[(size_t)(h - 1 - pictureRow) * w * 4]; // stored bottom-up
        for (long x = 0; x < w; x++)
            rgbeToFloat(&scan[(size_t)x * 4], dst + (size_t)(leftToRight ? x : w - 1 - x) * 4);
    }
    return img;
}

namespace {
inline double sinc(double x)
{
    if (x == 0.0)
        return 1.0;
    x *= 3.14159265358979323846;
    return std::sin(x) / x;
}
inline double lanczos3(double x) { return std::fabs(x) < 3.0 ? sinc(x) * sinc(x / 3.0) : 0.0; }

// one axis: n -> m samples for `lines` lines of `stride` floats per sample step
void resampleAxis(const float* src, float* dst, uint32_t n, uint32_t m, size_t sampleStrideSrc, size_t sampleStrideDst, size_t lines, size_t lineStrideSrc,
    size_t lineStrideDst)
{
    const double scale = (double)n / m, support = 3.0 * std::max(1.0, scale), inv = 1.0 / std::max(1.0, scale);
    for (uint32_t j = 0; j < m; j++) {
        const double centre = (j + 0.5) * scale;
        const long lo = (long)std::floor(centre - support), hi = (long)std::ceil(centre + support);
        std::vector<double> wgt;
        std::vector<uint32_t> idx;
        double sum = 0.0;
        for (long k = lo; k <= hi; k++) {
            const double wk = lanczos3((k + 0.5 - centre) * inv);
            if (wk == 0.0)
                continue;
            wgt.push_back(wk);
            idx.push_back((uint32_t)std::min<long>(std::max<long>(k, 0), (long)n - 1)); // clamped edge
            sum += wk;
        }
        for (size_t l = 0; l < lines; l++)
            for (int c = 0; c < 4; c++) {
                double acc = 0.0;
                for (size_t t = 0; t < wgt.size(); t++)
                    acc += wgt[t] * src[l * lineStrideSrc + idx[t] * sampleStrideSrc + c];
                dst[l * lineStrideDst + j * sampleStrideDst + c] = (float)(acc / sum);
            }
    }
}
} // namespace

ImageRGBAF rescaleLanczos3(const ImageRGBAF& in, uint32_t width, uint32_t height)
{
    if (width == 0 || height == 0)
        throw std::runtime_error("rescaleLanczos3: empty target");
    if (in.width == width && in.height == height)
        return in;
    ImageRGBAF tmp; // horizontal pass
    tmp.width = width, tmp.height = in.height;
    tmp.rgba.resize((size_t)width * in.height * 4);
    resampleAxis(in.rgba.data(), tmp.rgba.data(), in.width, width, 4, 4, in.height, (size_t)in.width * 4, (size_t)width * 4);
    ImageRGBAF out; // vertical pass
    out.width = width, out.height = height;
    out.rgba.resize((size_t)width * height * 4);
    resampleAxis(tmp.rgba.data(), out.rgba.data(), in.height, height, (size_t)width * 4, (size_t)width * 4, width, 4, 4);
    for (size_t i = 3; i < out.rgba.size(); i += 4)
        out.rgba[i] = 1.0f;
    return out;
}

ImageRGBAF loadSkydomeLayer(const std::string& path, uint32_t width, uint32_t height, float brightnessMultiplier)
{
    ImageRGBAF img = rescaleLanczos3(loadRadianceHDR(path), width, height);
    for (size_t i = 0; i < img.rgba.size(); i += 4) { // texture.cpp:101-108
        img.rgba[i] *= brightnessMultiplier, img.rgba[i + 1] *= brightnessMultiplier, img.rgba[i + 2] *= brightnessMultiplier;
        img.rgba[i + 3] = 1.0f;
    }
    return img;
}

} // namespace raytracer
