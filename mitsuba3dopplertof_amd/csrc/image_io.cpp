// image_io.cpp -- the image files of `bitmap` textures (src/textures/bitmap.cpp reads them through src/core/bitmap.cpp + libpng):
// PNG, 8 bits per sample, gray / gray+alpha / RGB / RGBA / palette, non-interlaced; the chunk stream is parsed here, the IDAT
// payload is inflated with zlib, the scanline filters (PNG specification, section 9) are undone in place.
#include "dtof_scene.h"
#include <zlib.h>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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


// ---------------------------------------------------------------------------- radiance maps (envmap)
// PFM (src/core/bitmap.cpp:2164-2217): "PF" | "Pf", width, height, scale-and-byte-order, then float rows BOTTOM row first.
// RGBE (:1988-2096): "#?..." header lines up to "-Y h +X w", flat or new-style run-length encoded scanlines, value = mantissa * 2^(e - 136).
// Returns float RGB, top row first.
static void read_pfm(const std::vector<uint8_t> &file, const std::string &path, std::vector<float> &rgb, uint32_t &w, uint32_t &h) {
    size_t pos = 2; std::string tok[3];
    for (int k = 0; k < 3; ++k) {
        while (pos < file.size() && isspace(file[pos])) ++pos;
        while (pos < file.size() && !isspace(file[pos])) tok[k] += (char) file[pos++];
    }
    ++pos;   // the single whitespace byte after the header
    char *end = nullptr;
    const unsigned long ww = strtoul(tok[0].c_str(), &end, 10), hh = strtoul(tok[1].c_str(), nullptr, 10);
    const double so = strtod(tok[2].c_str(), nullptr);
    if (tok[2].empty() || ww == 0 || hh == 0 || ww > 65536 || hh > 65536) throw std::runtime_error("Could not parse PFM header! (\"" + path + "\")");
    const uint32_t ch = file[1] == 'F' ? 3 : 1;
    const size_t n = (size_t) ww * hh * ch;
    if (pos + n * 4 > file.size()) throw std::runtime_error("read_pfm(): \"" + path + "\" is truncated");
    w = (uint32_t) ww; h = (uint32_t) hh; rgb.resize((size_t) w * h * 3);
    const bool big = !((float) so <= 0.f); const float scale = std::fabs((float) so);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x)
            for (uint32_t c = 0; c < 3; ++c) {
                const uint8_t *p = &file[pos + 4 * (((size_t) (h - 1 - y) * w + x) * ch + (ch == 3 ? c : 0))];
                uint32_t bits = big ? ((uint32_t) p[0] << 24 | (uint32_t) p[1] << 16 | (uint32_t) p[2] << 8 | p[3]) : ((uint32_t) p[3] << 24 | (uint32_t) p[2] << 16 | (uint32_t) p[1] << 8 | p[0]);
                float v; memcpy(&v, &bits, 4);
                rgb[((size_t) y * w + x) * 3 + c] = scale != 1.f ? v * scale : v;
            }
}
static void read_rgbe(const std::vector<uint8_t> &file, const std::string &path, std::vector<float> &rgb, uint32_t &w, uint32_t &h) {
    size_t pos = 0; bool ok = false; w = h = 0;
    auto line = [&]() { std::string l; while (pos < file.size() && file[pos] != '\n') l += (char) file[pos++]; ++pos; return l; };
    line();
    while (true) {
        if (pos >= file.size()) throw std::runtime_error("read_rgbe(): Invalid header! (\"" + path + "\")");
        const std::string l = line();
        if (l.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) ok = true;
        else if (l.rfind("FORMAT=32-bit_rle_xyze", 0) == 0) throw std::runtime_error("read_rgbe(): XYZE files are not supported (\"" + path + "\")");
        else {
            unsigned long hh = 0, ww = 0; char a[8] = { 0 }, b[8] = { 0 };
            if (sscanf(l.c_str(), "%7s %lu %7s %lu", a, &hh, b, &ww) == 4 && !strcmp(a, "-Y") && !strcmp(b, "+X")) { h = (uint32_t) hh; w = (uint32_t) ww; break; }
        }
    }
    if (!ok) throw std::runtime_error("read_rgbe(): unrecognized format!");
    // a damaged header must not be allocated before the first read fails: at most 64 Mpixels, and no more than run-length coding can pack into the file
    if (w == 0 || h == 0 || (uint64_t) w * h > (1ull << 26) || (uint64_t) w * h / 64 > file.size()) throw std::runtime_error("read_rgbe(): implausible size in \"" + path + "\"");
    std::vector<uint8_t> px((size_t) w * h * 4);
    auto need = [&](size_t n) { if (pos + n > file.size()) throw std::runtime_error("read_rgbe(): \"" + path + "\" is truncated"); };
    auto flat_from = [&](size_t first_pixel) { const size_t n = ((size_t) w * h - first_pixel) * 4; need(n); memcpy(&px[first_pixel * 4], &file[pos], n); pos += n; };
    if (w < 8 || w > 0x7fff) flat_from(0);
    else {
        std::vector<uint8_t> row((size_t) w * 4);
        for (uint32_t y = 0; y < h; ++y) {
            need(4);
            const uint8_t *r = &file[pos];
            if (r[0] != 2 || r[1] != 2 || (r[2] & 0x80)) { flat_from((size_t) y * w); break; }   // not run-length encoded from here on
            if ((size_t) ((r[2] << 8) | r[3]) != w) throw std::runtime_error("read_rgbe(): wrong scanline width!");
            pos += 4;
            size_t at = 0;
            for (uint32_t c = 0; c < 4; ++c) {
                const size_t stop = (size_t) (c + 1) * w;
                while (at < stop) {
                    need(2);
                    const uint32_t n0 = file[pos], v = file[pos + 1]; pos += 2;
                    if (n0 > 128) {
                        const size_t n = n0 - 128;
                        if (n == 0 || n > stop - at) throw std::runtime_error("read_rgbe(): bad scanline data!");
                        memset(&row[at], (int) v, n); at += n;
                    } else {
                        const size_t n = n0;
                        if (n == 0 || n > stop - at) throw std::runtime_error("read_rgbe(): bad scanline data!");
                        row[at++] = (uint8_t) v;
                        if (n > 1) { need(n - 1); memcpy(&row[at], &file[pos], n - 1); pos += n - 1; at += n - 1; }
                    }
                }
            }
            for (uint32_t x = 0; x < w; ++x) for (uint32_t c = 0; c < 4; ++c) px[((size_t) y * w + x) * 4 + c] = row[(size_t) c * w + x];
        }
    }
    rgb.resize((size_t) w * h * 3);
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const uint8_t *q = &px[i * 4];
        const float f = q[3] ? std::ldexp(1.f, (int) q[3] - (128 + 8)) : 0.f;
        for (int c = 0; c < 3; ++c) rgb[i * 3 + c] = q[3] ? (float) q[c] * f : 0.f;
    }
}
void read_radiance_image(const std::string &path, std::vector<float> &rgb, uint32_t &width, uint32_t &height, float (*srgb_to_linear_u8)(uint32_t)) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("could not open \"" + path + "\"");
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() >= 8 && file[0] == 'P' && (file[1] == 'F' || file[1] == 'f')) return read_pfm(file, path, rgb, width, height);
    if (file.size() >= 8 && file[0] == '#' && file[1] == '?') return read_rgbe(file, path, rgb, width, height);
    if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') {   // 8-bit PNG: sRGB -> linear (Bitmap::convert to Float32 with srgb_gamma = false)
        std::vector<uint8_t> px; uint32_t ch;
        read_png(path, px, width, height, ch);
        float lut[256]; for (uint32_t i = 0; i < 256; ++i) lut[i] = srgb_to_linear_u8(i);
        rgb.resize((size_t) width * height * 3);
        for (size_t i = 0; i < (size_t) width * height; ++i) for (uint32_t c = 0; c < 3; ++c) rgb[i * 3 + c] = lut[px[i * ch + (ch == 3 ? c : 0)]];
        return;
    }
    throw std::runtime_error("envmap: \"" + path + "\": unsupported image format (this build reads RGBE .hdr, PFM and 8-bit PNG radiance maps)");
}

}  // namespace dtof
