// image_io.cpp -- the image files of `bitmap` textures (src/textures/bitmap.cpp reads them through src/core/bitmap.cpp + libpng):
// PNG, 8 bits per sample, gray / gray+alpha / RGB / RGBA / palette, non-interlaced; the chunk stream is parsed here, the IDAT
// payload is inflated with zlib, the scanline filters (PNG specification, section 9) are undone in place.
#include "dtof_scene.h"
#include <zlib.h>
#include <cstring>
#include <fstream>

namespace dtof {

static uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

// -> pixels: height * width * channels bytes, channels = 1 (gray) or 3 (RGB; alpha is dropped, palettes are expanded)
void read_png(const std::string &path, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, uint32_t &channels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("could not open \"" + path + "\"");
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0)
        throw std::runtime_error("bitmap: \"" + path + "\" is not a PNG file (this build reads PNG textures only)");
    uint32_t w = 0, h = 0, depth = 0, color = 0, interlace = 0; bool have_ihdr = false;
    std::vector<uint8_t> idat, palette;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]); const char *tag = (const char *) &file[pos + 4];
        if (pos + 12 + (size_t) len > file.size()) throw std::runtime_error("bitmap: truncated PNG chunk in \"" + path + "\"");
        const uint8_t *data = &file[pos + 8];
        if (!memcmp(tag, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error("bitmap: bad IHDR in \"" + path + "\"");
            w = be32(data); h = be32(data + 4); depth = data[8]; color = data[9]; interlace = data[12]; have_ihdr = true;
        } else if (!memcmp(tag, "PLTE", 4)) palette.assign(data, data + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t) len;
    }
    if (!have_ihdr || w == 0 || h == 0) throw std::runtime_error("bitmap: \"" + path + "\" has no image header");
    const bool packed = depth < 8 && (color == 0 || color == 3) && (depth == 1 || depth == 2 || depth == 4);   // gray / palette indices of 1, 2, 4 bits
    if ((depth != 8 && !packed) || interlace != 0 || (color != 0 && color != 2 && color != 3 && color != 4 && color != 6))
        throw std::runtime_error("bitmap: \"" + path + "\": only non-interlaced PNG files with at most 8 bits per sample are supported");
    if ((uint64_t) w * h > (1ull << 28)) throw std::runtime_error("bitmap: \"" + path + "\" is too large");
    const uint32_t spp = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;   // samples per pixel in the file
    const size_t stride = packed ? ((size_t) w * depth + 7) / 8 : (size_t) w * spp;   // bytes per scanline; the filters work on bytes, with a
    const uint32_t fbpp = packed ? 1 : spp;                                            // distance of one pixel, at least one byte
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf) raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong) idat.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error("bitmap: could not inflate the image data of \"" + path + "\"");
    // undo the scanline filters
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &img[stride * y]; const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= fbpp ? dst[i - fbpp] : 0, b = up ? up[i] : 0, c = (up && i >= fbpp) ? up[i - fbpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); } break;
                default: throw std::runtime_error("bitmap: bad scanline filter in \"" + path + "\"");
            }
            dst[i] = (uint8_t) (src[i] + pred);
        }
    }
    if (packed) {   // unpack the samples (most significant bits first); gray levels are scaled to 0 .. 255
        std::vector<uint8_t> un((size_t) w * h);
        const uint32_t maxv = (1u << depth) - 1u;
        for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
            const size_t bit = (size_t) x * depth; const uint8_t byte = img[stride * y + bit / 8];
            const uint32_t v = (byte >> (8 - depth - (bit % 8))) & maxv;
            un[(size_t) y * w + x] = (uint8_t) (color == 0 ? v * 255u / maxv : v);
        }
        img.swap(un);
    }
    channels = (color == 0 || color == 4) ? 1 : 3;
    pixels.resize((size_t) w * h * channels);
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const uint8_t *px = &img[i * spp];
        if (color == 3) {
            if ((size_t) px[0] * 3 + 2 >= palette.size()) throw std::runtime_error("bitmap: palette index out of range in \"" + path + "\"");
            memcpy(&pixels[i * 3], &palette[(size_t) px[0] * 3], 3);
        } else if (channels == 1) pixels[i] = px[0];
        else memcpy(&pixels[i * 3], px, 3);
    }
    width = w; height = h;
}

}  // namespace dtof
