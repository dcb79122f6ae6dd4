#include "image.h"
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <zlib.h>

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
    {   // the resolution line is untrusted: every scanline costs at least its 4-byte head and 2 bytes per run of <= 127 pixels and component
        // (run-length form), or 4 bytes per pixel (flat form) -- a picture the file cannot hold is refused before 16 bytes per pixel are set aside
        const std::streamoff here = in.tellg();
        in.seekg(0, std::ios::end);
        const uint64_t left = here < 0 ? 0 : (uint64_t)(in.tellg() - here);
        in.seekg(here);
        const uint64_t rowMin = (w >= 8 && w < 32768) ? std::min<uint64_t>(4 + 8 * (((uint64_t)w + 126) / 127), 4 * (uint64_t)w) : 4 * (uint64_t)w;
        if ((uint64_t)h * rowMin > left)
            throw std::runtime_error(path + ": truncated (the resolution line announces more scanlines than the file holds)");
    }
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

ImageRGBAF rescaleLanczos3(const ImageRGBAF& in, uint32_t width, uint32_t height, bool keepAlpha)
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
    if (!keepAlpha)
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

// ---- PNG (ISO/IEC 15948): every colour type and bit depth, tRNS, Adam7; inflate by zlib ----------------------------
namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c)
{
    const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// reverse the scanline filters of one (sub-)image in place; returns a pointer past its data
const uint8_t* unfilter(uint8_t* data, size_t avail, uint32_t w, uint32_t h, uint32_t bitsPerPixel, std::vector<uint8_t>& rows)
{
    const size_t rowBytes = ((size_t)w * bitsPerPixel + 7) / 8, bpp = std::max<size_t>(1, bitsPerPixel / 8);
    if (w == 0 || h == 0)
        return data;
    if (avail < (rowBytes + 1) * h)
        throw std::runtime_error("PNG: image data too short");
    rows.assign(rowBytes * h, 0);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* in = data + (rowBytes + 1) * y;
        const int filter = in[0];
        in++;
        uint8_t* out = rows.data() + rowBytes * y;
        const uint8_t* up = y ? out - rowBytes : nullptr;
        for (size_t i = 0; i < rowBytes; i++) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = in[i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: throw std::runtime_error("PNG: unknown filter type");
            }
            out[i] = (uint8_t)v;
        }
    }
    return data + (rowBytes + 1) * h;
}
} // namespace

ImageRGBA8 loadPNG(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0)
        throw std::runtime_error(path + ": not a PNG file");
    uint32_t w = 0, h = 0, depth = 0, colour = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool haveHeader = false, done = false;
    for (size_t pos = 8; !done;) {
        if (pos + 12 > file.size())
            throw std::runtime_error(path + ": truncated PNG chunk");
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if ((size_t)len > file.size() - pos - 12)
            throw std::runtime_error(path + ": truncated PNG chunk");
        const uint8_t* data = &file[pos + 8];
        if (crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len))
            throw std::runtime_error(path + ": PNG chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13)
                throw std::runtime_error(path + ": bad IHDR");
            w = be32(data), h = be32(data + 4), depth = data[8], colour = data[9], interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1 || w == 0 || h == 0 || w > (1u << 15) || h > (1u << 15))
                throw std::runtime_error(path + ": unsupported PNG header");
            haveHeader = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        } else if (!(type[0] & 0x20)) {
            throw std::runtime_error(path + ": unknown critical PNG chunk");
        }
        pos += 12 + (size_t)len;
    }
    if (!haveHeader)
        throw std::runtime_error(path + ": PNG without IHDR");
    uint32_t channels = 0;
    switch (colour) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error(path + ": bad PNG colour type");
    }
    const bool depthOk = colour == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
        : colour == 3                ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (depth == 8 || depth == 16);
    if (!depthOk || (colour == 3 && (palette.empty() || palette.size() % 3)))
        throw std::runtime_error(path + ": bad PNG bit depth / palette");
    const uint32_t bitsPerPixel = channels * depth;
    // Adam7 passes: (x0, y0, dx, dy); a non-interlaced image is the single pass (0, 0, 1, 1)
    static const uint32_t adam7[7][4] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    const uint32_t whole[1][4] = { { 0, 0, 1, 1 } };
    const uint32_t(*passes)[4] = interlace ? adam7 : whole;
    const int numPasses = interlace ? 7 : 1;
    size_t rawSize = 0;
    for (int p = 0; p < numPasses; p++) {
        const uint32_t pw = (w - passes[p][0] + passes[p][2] - 1) / passes[p][2], ph = (h - passes[p][1] + passes[p][3] - 1) / passes[p][3];
        if (w > passes[p][0] && h > passes[p][1])
            rawSize += (((size_t)pw * bitsPerPixel + 7) / 8 + 1) * ph;
    }
    // deflate expands by a factor of 1032 at most: a header that implies more than the image data can inflate to is refused before
    // anything is sized by it (a 100-byte file may announce 32 768 x 32 768 pixels: 8 GB of scanlines)
    if (rawSize > idat.size() * 1032 + 1024)
        throw std::runtime_error(path + ": PNG image data does not inflate to the size the header implies");
    std::vector<uint8_t> raw(rawSize);
    uLongf got = (uLongf)rawSize;
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != rawSize)
        throw std::runtime_error(path + ": PNG image data does not inflate to the size the header implies");

    ImageRGBA8 img;
    img.width = w, img.height = h;
    img.rgba.assign((size_t)w * h * 4, 255);
    std::vector<uint8_t> rows;
    uint8_t* cursor = raw.data();
    const uint32_t maxv = (1u << std::min(depth, 8u)) - 1u;
    for (int p = 0; p < numPasses; p++) {
        if (!(w > passes[p][0] && h > passes[p][1]))
            continue;
        const uint32_t pw = (w - passes[p][0] + passes[p][2] - 1) / passes[p][2], ph = (h - passes[p][1] + passes[p][3] - 1) / passes[p][3];
        cursor = const_cast<uint8_t*>(unfilter(cursor, (size_t)(raw.data() + raw.size() - cursor), pw, ph, bitsPerPixel, rows));
        const size_t rowBytes = ((size_t)pw * bitsPerPixel + 7) / 8;
        for (uint32_t y = 0; y < ph; y++)
            for (uint32_t x = 0; x < pw; x++) {
                const uint8_t* row = rows.data() + rowBytes * y;
                uint32_t s[4] = { 0, 0, 0, 0 }; // samples of this pixel, 8-bit (16-bit: high byte) or raw sub-byte value
                uint32_t s16[4] = { 0, 0, 0, 0 };
                for (uint32_t c = 0; c < channels; c++) {
                    if (depth == 16) {
                        const size_t o = ((size_t)x * channels + c) * 2;
                        s16[c] = ((uint32_t)row[o] << 8) | row[o + 1];
                        s[c] = row[o];
                    } else if (depth == 8) {
                        s16[c] = s[c] = row[(size_t)x * channels + c];
                    } else {
                        const size_t bit = (size_t)x * depth; // one channel only below 8 bits
                        s16[c] = s[c] = (row[bit / 8] >> (8 - depth - bit % 8)) & maxv;
                    }
                }
                uint8_t px[4] = { 0, 0, 0, 255 };
                switch (colour) {
                case 0: {
                    const uint8_t g = depth < 8 ? (uint8_t)(s[0] * 255u / maxv) : (uint8_t)s[0];
                    px[0] = px[1] = px[2] = g;
                    if (trns.size() >= 2 && s16[0] == (((uint32_t)trns[0] << 8) | trns[1]))
                        px[3] = 0;
                    break;
                }
                case 2:
                    px[0] = (uint8_t)s[0], px[1] = (uint8_t)s[1], px[2] = (uint8_t)s[2];
                    if (trns.size() >= 6 && s16[0] == (((uint32_t)trns[0] << 8) | trns[1]) && s16[1] == (((uint32_t)trns[2] << 8) | trns[3])
                        && s16[2] == (((uint32_t)trns[4] << 8) | trns[5]))
                        px[3] = 0;
                    break;
                case 3:
                    if ((size_t)s[0] * 3 + 2 >= palette.size())
                        throw std::runtime_error(path + ": PNG palette index out of range");
                    px[0] = palette[s[0] * 3], px[1] = palette[s[0] * 3 + 1], px[2] = palette[s[0] * 3 + 2];
                    px[3] = s[0] < trns.size() ? trns[s[0]] : 255;
                    break;
                case 4:
                    px[0] = px[1] = px[2] = (uint8_t)s[0], px[3] = (uint8_t)s[1];
                    break;
                default:
                    px[0] = (uint8_t)s[0], px[1] = (uint8_t)s[1], px[2] = (uint8_t)s[2], px[3] = (uint8_t)s[3];
                }
                const size_t X = passes[p][0] + (size_t)x * passes[p][2], Y = passes[p][1] + (size_t)y * passes[p][3];
                std::memcpy(&img.rgba[(Y * w + X) * 4], px, 4);
            }
    }
    return img;
}

// the same layer as the 32-bit bitmap the reference uploads into its CL_BGRA / CL_UNORM_INT8 array (texture.cpp:112-131):
// bytes b, g, r, a per texel, rows bottom-up
ImageRGBA8 loadMaterialLayerBGRA8(const std::string& path, uint32_t width, uint32_t height, bool isLinear)
{
    const ImageRGBAF f = loadMaterialLayer(path, width, height, isLinear);
    ImageRGBA8 out;
    out.width = f.width, out.height = f.height;
    out.rgba.resize(f.rgba.size());
    for (size_t t = 0; t < f.rgba.size() / 4; t++) {
        const uint8_t r = (uint8_t)std::lround(f.rgba[4 * t] * 255.0f), g = (uint8_t)std::lround(f.rgba[4 * t + 1] * 255.0f),
                      b = (uint8_t)std::lround(f.rgba[4 * t + 2] * 255.0f), a = (uint8_t)std::lround(f.rgba[4 * t + 3] * 255.0f);
        out.rgba[4 * t] = b, out.rgba[4 * t + 1] = g, out.rgba[4 * t + 2] = r, out.rgba[4 * t + 3] = a;
    }
    return out;
}

ImageRGBAF loadMaterialLayer(const std::string& path, uint32_t width, uint32_t height, bool isLinear)
{
    const ImageRGBA8 png = loadPNG(path);
    ImageRGBAF f; // bottom-up like a FreeImage bitmap, samples 0..255
    f.width = png.width, f.height = png.height;
    f.rgba.resize(png.rgba.size());
    for (uint32_t y = 0; y < png.height; y++)
        for (size_t i = 0; i < (size_t)png.width * 4; i++)
            f.rgba[(size_t)(png.height - 1 - y) * png.width * 4 + i] = (float)png.rgba[(size_t)y * png.width * 4 + i];
    f = rescaleLanczos3(f, width ? width : png.width, height ? height : png.height, true);
    // FreeImage_AdjustGamma(dib, 1 / 2.2): lut[i] = 255 * (i / 255) ^ (1 / gamma) + 0.5, colour channels only
    uint8_t lut[256];
    for (int i = 0; i < 256; i++) {
        const double v = isLinear ? (double)i : 255.0 * std::pow(i / 255.0, 2.2) + 0.5;
        lut[i] = (uint8_t)std::min(255.0, std::max(0.0, std::floor(v)));
    }
    for (size_t i = 0; i < f.rgba.size(); i++) {
        const int b = (int)std::min(255.0f, std::max(0.0f, std::floor(f.rgba[i] + 0.5f))); // the rescaled image is 8-bit again
        f.rgba[i] = (float)((i & 3) == 3 ? b : lut[b]) / 255.0f; // what read_imagef returns for a UNORM_INT8 texel
    }
    return f;
}

} // namespace raytracer
