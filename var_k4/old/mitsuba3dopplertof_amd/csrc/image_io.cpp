// image_io.cpp -- the image files of `bitmap` textures (src/textures/bitmap.cpp reads them through src/core/bitmap.cpp + libpng):
// PNG, 8 bits per sample, gray / gray+alpha / RGB / RGBA / palette, non-interlaced; the chunk stream is parsed here, the IDAT
// payload is inflated with zlib, the scanline filters (PNG specification, section 9) are undone in place.
#include "dtof_scene.h"
#include <zlib.h>
#include <algorithm>
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


// ---------------------------------------------------------------------------- JPEG (baseline sequential DCT, Huffman, 8 bit)
// What src/core/bitmap.cpp reads through libjpeg with its default settings (JDCT_ISLOW, fancy upsampling): the entropy decoder of ITU T.81
// annex F, the "slow but accurate" integer inverse DCT of the IJG library (jidctint.c: 13-bit constants, 2 extra bits after the column pass),
// its triangle-filter chroma upsampling for 2x1 and 2x2 subsampled components (jdsample.c: h2v1_fancy / h2v2_fancy) and its fixed-point
// YCbCr -> RGB tables (jdcolor.c) -- integer arithmetic throughout, so the samples are those of libjpeg-turbo / libjpeg 6b bit for bit.
// Progressive, arithmetic-coded, 12-bit, CMYK and other subsampling layouts are refused.
namespace {
struct JpegComponent { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int dc_pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; };
struct JpegHuff { uint8_t bits[17] = { 0 }; uint8_t vals[256] = { 0 }; int mincode[17], maxcode[18], valptr[17]; bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) { valptr[l] = k; mincode[l] = code; code += bits[l]; k += bits[l]; maxcode[l] = bits[l] ? code - 1 : -1; code <<= 1; }
        maxcode[17] = 0x7fffffff;
    } };
struct JpegBits {
    const uint8_t *p, *end; uint32_t acc = 0; int n = 0; bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            uint32_t b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xff) { if (p + 1 < end && p[1] == 0) p += 2; else { hit_marker = true; b = 0; } }
                else ++p;
            }
            acc |= b << (24 - n); n += 8;
        }
    }
    int get(int count) { if (count == 0) return 0; if (n < count) fill(); const int v = (int) (acc >> (32 - count)); acc <<= count; n -= count; return v; }
    int decode(const JpegHuff &h) {
        int code = 0;
        for (int l = 1; l <= 16; ++l) { code = (code << 1) | get(1); if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]]; }
        throw std::runtime_error("read_jpeg(): corrupt Huffman code");
    }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};
inline int jpeg_extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }
inline uint8_t jpeg_clamp(int v) { return (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v); }
const int kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30,
                          37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
// jpeg_idct_islow (jidctint.c): coef = dequantised coefficients in natural order; out = 8 x 8 samples
void jpeg_idct(const int *coef, uint8_t *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; ++c) {
        const int *in = coef + c;
        if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) { const long dc = (long) in[0] * (1L << P1); for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc; continue; }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3; const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390; z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long *w = ws + r * 8; uint8_t *o = out + (size_t) r * stride;
        if (!w[1] && !w[2] && !w[3] && !w[4] && !w[5] && !w[6] && !w[7]) { const uint8_t dc = jpeg_clamp((int) descale(w[0], P1 + 3) + 128); for (int c = 0; c < 8; ++c) o[c] = dc; continue; }
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3; const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390; z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = jpeg_clamp((int) descale(tmp10 + tmp3, S) + 128); o[7] = jpeg_clamp((int) descale(tmp10 - tmp3, S) + 128);
        o[1] = jpeg_clamp((int) descale(tmp11 + tmp2, S) + 128); o[6] = jpeg_clamp((int) descale(tmp11 - tmp2, S) + 128);
        o[2] = jpeg_clamp((int) descale(tmp12 + tmp1, S) + 128); o[5] = jpeg_clamp((int) descale(tmp12 - tmp1, S) + 128);
        o[3] = jpeg_clamp((int) descale(tmp13 + tmp0, S) + 128); o[4] = jpeg_clamp((int) descale(tmp13 - tmp0, S) + 128);
    }
}
}  // namespace

// -> pixels: height * width * channels bytes, channels = 1 (grayscale file) or 3 (RGB)
void read_jpeg(const std::string &path, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, uint32_t &channels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("could not open \"" + path + "\"");
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto fail = [&](const std::string &m) { throw std::runtime_error("read_jpeg(): \"" + path + "\": " + m); };
    if (file.size() < 4 || file[0] != 0xff || file[1] != 0xd8) fail("not a JPEG file");
    int qt[4][64]; bool have_qt[4] = { false, false, false, false }; JpegHuff dc[4], ac[4];
    std::vector<JpegComponent> comp; uint32_t w = 0, h = 0; int restart = 0; bool adobe = false; int adobe_transform = -1;
    size_t pos = 2;
    auto be16 = [&](size_t at) { if (at + 2 > file.size()) fail("truncated"); return (int) (file[at] << 8 | file[at + 1]); };
    for (;;) {
        while (pos < file.size() && file[pos] != 0xff) ++pos;
        while (pos < file.size() && file[pos] == 0xff) ++pos;
        if (pos >= file.size()) fail("no image data");
        const int marker = file[pos++];
        if (marker == 0xd8 || (marker >= 0xd0 && marker <= 0xd7) || marker == 0x01) continue;
        if (marker == 0xd9) fail("no image data");
        const int len = be16(pos); if (len < 2 || pos + len > file.size()) fail("truncated segment");
        const uint8_t *seg = &file[pos + 2]; const int n = len - 2;
        if (marker == 0xdb) {           // DQT
            for (int k = 0; k < n;) {
                const int pq = seg[k] >> 4, tq = seg[k] & 15; ++k;
                if (tq > 3 || k + (pq ? 128 : 64) > n) fail("bad quantisation table");
                for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (seg[k] << 8 | seg[k + 1]) : seg[k]; k += pq ? 2 : 1; }
                have_qt[tq] = true;
            }
        } else if (marker == 0xc4) {    // DHT
            for (int k = 0; k < n;) {
                if (k + 17 > n) fail("bad Huffman table");
                const int tc = seg[k] >> 4, th = seg[k] & 15; ++k;
                if (tc > 1 || th > 3) fail("bad Huffman table");
                JpegHuff &t = tc ? ac[th] : dc[th]; int total = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = seg[k++]; total += t.bits[l]; }
                if (total > 256 || k + total > n) fail("bad Huffman table");
                for (int i = 0; i < total; ++i) t.vals[i] = seg[k++];
                t.build(); t.present = true;
            }
        } else if (marker == 0xc0 || marker == 0xc1) {   // SOF0 / SOF1: baseline / extended sequential, Huffman
            if (n < 6 || seg[0] != 8) fail("only 8-bit JPEG files are supported");
            h = (uint32_t) (seg[1] << 8 | seg[2]); w = (uint32_t) (seg[3] << 8 | seg[4]);
            const int nc = seg[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3 * nc) fail("only grayscale and YCbCr JPEG files are supported");
            comp.resize(nc);
            for (int c = 0; c < nc; ++c) { comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c]; if (comp[c].tq > 3) fail("bad frame header"); }
        } else if (marker == 0xc2 || (marker >= 0xc3 && marker <= 0xcf && marker != 0xc4 && marker != 0xc8 && marker != 0xcc)) {
            fail("progressive, lossless and arithmetic-coded JPEG files are not supported (baseline sequential only)");
        } else if (marker == 0xdd) { if (n < 2) fail("bad DRI"); restart = seg[0] << 8 | seg[1]; }
        else if (marker == 0xee && n >= 12 && !memcmp(seg, "Adobe", 5)) { adobe = true; adobe_transform = seg[11]; }
        else if (marker == 0xda) {      // SOS: the one scan of a baseline file
            if (comp.empty() || w == 0 || h == 0) fail("scan before frame header");
            if (n < 1 || seg[0] != (int) comp.size() || n < 1 + 2 * (int) comp.size() + 3) fail("only single-scan (interleaved) baseline files are supported");
            for (size_t c = 0; c < comp.size(); ++c) {
                size_t which = comp.size();
                for (size_t j = 0; j < comp.size(); ++j) if (comp[j].id == seg[1 + 2 * c]) which = j;
                if (which != c) fail("unexpected component order in the scan");
                comp[c].td = seg[2 + 2 * c] >> 4; comp[c].ta = seg[2 + 2 * c] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present || !have_qt[comp[c].tq]) fail("scan refers to a missing table");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    if ((uint64_t) w * h > (1ull << 28)) fail("image too large");
    const bool colour = comp.size() == 3;
    if (colour && adobe && adobe_transform == 0) fail("RGB-coded (Adobe transform 0) JPEG files are not supported");
    int hmax = 1, vmax = 1; for (auto &c : comp) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
    if (!colour) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }   // a single-component scan is not interleaved: its sampling factors do not matter
    else {
        if (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1 || !((comp[0].h == 1 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 2)))
            fail("only 4:4:4, 4:2:2 (2x1) and 4:2:0 (2x2) chroma subsampling are supported");
    }
    const uint32_t mcux = (w + 8 * hmax - 1) / (8 * hmax), mcuy = (h + 8 * vmax - 1) / (8 * vmax);
    for (auto &c : comp) { c.bw = (int) mcux * c.h * 8; c.bh = (int) mcuy * c.v * 8; c.plane.assign((size_t) c.bw * c.bh, 0); c.dc_pred = 0; }
    JpegBits br; br.p = &file[pos]; br.end = file.data() + file.size();
    int coef[64], until_restart = restart;
    for (uint32_t my = 0; my < mcuy; ++my) for (uint32_t mx = 0; mx < mcux; ++mx) {
        if (restart && until_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictors
            br.reset();
            while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) ++br.p;
            if (br.p + 1 >= br.end) fail("missing restart marker");
            br.p += 2; for (auto &c : comp) c.dc_pred = 0; until_restart = restart;
        }
        for (auto &c : comp) for (int by = 0; by < c.v; ++by) for (int bx = 0; bx < c.h; ++bx) {
            memset(coef, 0, sizeof coef);
            const int t = br.decode(dc[c.td]);
            if (t > 11) fail("corrupt DC coefficient");
            c.dc_pred = (int) ((unsigned) c.dc_pred + (unsigned) jpeg_extend(br.get(t), t));   // wraps instead of overflowing on a damaged stream
            coef[0] = (int) ((unsigned) c.dc_pred * (unsigned) qt[c.tq][0]);
            for (int k = 1; k < 64;) {
                const int rs = br.decode(ac[c.ta]), r = rs >> 4, s2 = rs & 15;
                if (s2 == 0) { if (r == 15) { k += 16; continue; } break; }
                k += r; if (k > 63) fail("corrupt AC coefficients");
                coef[kZigzag[k]] = jpeg_extend(br.get(s2), s2) * qt[c.tq][kZigzag[k]]; ++k;
            }
            jpeg_idct(coef, &c.plane[(size_t) ((my * c.v + by) * 8) * c.bw + (size_t) (mx * c.h + bx) * 8], c.bw);
        }
        if (restart) --until_restart;
    }
    width = w; height = h; channels = colour ? 3 : 1;
    pixels.resize((size_t) w * h * channels);
    if (!colour) { for (uint32_t y = 0; y < h; ++y) memcpy(&pixels[(size_t) y * w], &comp[0].plane[(size_t) y * comp[0].bw], w); return; }
    // chroma to full resolution (jdsample.c); cw / chh = the downsampled dimensions libjpeg works on (ceil(w * h_i / hmax), likewise the height)
    std::vector<uint8_t> up[2];
    for (int ci = 1; ci <= 2; ++ci) {
        const JpegComponent &c = comp[ci]; std::vector<uint8_t> &o = up[ci - 1];
        const int cw = (int) ((w + hmax - 1) / hmax), chh = (int) ((h + vmax - 1) / vmax);
        o.assign((size_t) w * h, 0);
        auto row = [&](int y) { y = y < 0 ? 0 : y >= chh ? chh - 1 : y; return &c.plane[(size_t) y * c.bw]; };
        if (hmax == 1 && vmax == 1) { for (uint32_t y = 0; y < h; ++y) memcpy(&o[(size_t) y * w], row((int) y), w); continue; }
        std::vector<uint8_t> line((size_t) cw * 2 + 2);
        for (uint32_t y = 0; y < h; ++y) {
            if (cw <= 2) {              // jinit_upsampler: the fancy (triangle) filters need more than two columns, else h2v1_upsample / h2v2_upsample replicate
                const uint8_t *in = row((int) y / vmax);
                for (int x = 0; x < cw; ++x) line[2 * x] = line[2 * x + 1] = in[x];
            } else if (vmax == 1) {     // h2v1_fancy_upsample
                const uint8_t *in = row((int) y);
                if (cw == 1) { line[0] = line[1] = in[0]; }
                else {
                    line[0] = in[0]; line[1] = (uint8_t) ((in[0] * 3 + in[1] + 2) >> 2);
                    for (int x = 1; x < cw - 1; ++x) { const int v = in[x] * 3; line[2 * x] = (uint8_t) ((v + in[x - 1] + 1) >> 2); line[2 * x + 1] = (uint8_t) ((v + in[x + 1] + 2) >> 2); }
                    line[2 * cw - 2] = (uint8_t) ((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2); line[2 * cw - 1] = in[cw - 1];
                }
            } else {                    // h2v2_fancy_upsample: the nearer neighbour row is the one above for even output rows, below for odd ones
                const int cy = (int) y / 2; const uint8_t *in0 = row(cy), *in1 = row((y & 1) ? cy + 1 : cy - 1);
                if (cw == 1) { const int s = in0[0] * 3 + in1[0]; line[0] = (uint8_t) ((s * 4 + 8) >> 4); line[1] = (uint8_t) ((s * 4 + 7) >> 4); }
                else {
                    int thiss = in0[0] * 3 + in1[0], nexts = in0[1] * 3 + in1[1], lasts;
                    line[0] = (uint8_t) ((thiss * 4 + 8) >> 4); line[1] = (uint8_t) ((thiss * 3 + nexts + 7) >> 4);
                    lasts = thiss; thiss = nexts;
                    for (int x = 1; x < cw - 1; ++x) {
                        nexts = in0[x + 1] * 3 + in1[x + 1];
                        line[2 * x] = (uint8_t) ((thiss * 3 + lasts + 8) >> 4); line[2 * x + 1] = (uint8_t) ((thiss * 3 + nexts + 7) >> 4);
                        lasts = thiss; thiss = nexts;
                    }
                    line[2 * cw - 2] = (uint8_t) ((thiss * 3 + lasts + 8) >> 4); line[2 * cw - 1] = (uint8_t) ((thiss * 4 + 7) >> 4);
                }
            }
            memcpy(&o[(size_t) y * w], line.data(), w);
        }
    }
    // ycc_rgb_convert (jdcolor.c): 16-bit fixed point, ONE_HALF folded into the Cb-to-G table
    for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
        const int Y = comp[0].plane[(size_t) y * comp[0].bw + x], cb = up[0][(size_t) y * w + x] - 128, cr = up[1][(size_t) y * w + x] - 128;
        uint8_t *o = &pixels[((size_t) y * w + x) * 3];
        o[0] = jpeg_clamp(Y + (int) ((91881L * cr + 32768) >> 16));
        o[1] = jpeg_clamp(Y + (int) ((-22554L * cb + 32768 - 46802L * cr) >> 16));
        o[2] = jpeg_clamp(Y + (int) ((116130L * cb + 32768) >> 16));
    }
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
// OpenEXR (Bitmap::read_exr, src/core/bitmap.cpp, through the OpenEXR library): single-part scan-line files, compression NONE, ZIPS (one
// line per chunk), ZIP (16 lines) or PIZ (32 lines), channels R, G, B (or Y) as HALF, FLOAT or UINT; other channels (A, ...) are skipped.  ZIP
// chunks are zlib streams of the byte-reordered (even bytes, then odd bytes), delta-coded scan lines (ImfZipCompressor); PIZ: below.  RLE / PXR24 /
// B44 / DWA files, tiles and deep data are refused.
static float half_to_float(uint16_t hbits) {
    const uint32_t sign = (uint32_t) (hbits & 0x8000u) << 16, e = (hbits >> 10) & 31u, m = hbits & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } bits = sign | ((uint32_t) (113 - sh) << 23) | ((mm & 1023u) << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
// PIZ chunks (ImfPizCompressor): a bitmap of the 16-bit values that occur -> LUT, canonical Huffman coding of the wavelet coefficients (code
// lengths packed with zero runs, one symbol = "repeat the previous value n times"; ImfHuf), 2-D Haar-like wavelet per channel in a 14-bit or a
// 16-bit variant (ImfWav).  Decodes into the chunk's channel-planar uint16 buffer.
namespace {
struct PizBits { const uint8_t *d; size_t p, end; uint64_t c = 0; int lc = 0;
    uint32_t get(int n) { while (lc < n) { if (p >= end) throw std::runtime_error("read_exr(): truncated PIZ data"); c = (c << 8) | d[p++]; lc += 8; } lc -= n; return (uint32_t) ((c >> lc) & ((1ull << n) - 1)); } };
void piz_huf_uncompress(const uint8_t *data, size_t size, uint16_t *out, size_t n_raw) {
    if (size < 20) throw std::runtime_error("read_exr(): truncated PIZ data");
    uint32_t hd[5]; memcpy(hd, data, 20);
    const uint32_t im = hd[0], iM = hd[1], n_bits = hd[3];
    constexpr uint32_t kEnc = (1u << 16) + 1;
    if (im >= kEnc || iM >= kEnc || im > iM) throw std::runtime_error("read_exr(): bad PIZ Huffman table");
    std::vector<uint8_t> len(kEnc, 0);
    PizBits br { data, 20, size };
    for (uint32_t i = im; i <= iM;) {                    // hufUnpackEncTable
        const uint32_t l = br.get(6);
        if (l == 63) i += br.get(8) + 6;                 // LONG_ZEROCODE_RUN: 8 more bits + SHORTEST_LONG_RUN
        else if (l >= 59) i += l - 59 + 2;               // SHORT_ZEROCODE_RUN
        else len[i++] = (uint8_t) l;
    }
    const size_t table_end = br.p;
    uint64_t count[59] = { 0 }, base[59];
    for (uint32_t i = im; i <= iM; ++i) ++count[len[i]];
    { uint64_t c = 0; for (int k = 58; k > 0; --k) { const uint64_t nc = (c + count[k]) >> 1; base[k] = c; c = nc; } }   // hufCanonicalCodeTable
    std::vector<uint32_t> syms; uint64_t first_index[59] = { 0 };
    { uint64_t at = 0; for (int k = 1; k <= 58; ++k) { first_index[k] = at; at += count[k]; } syms.resize((size_t) at); uint64_t fill[59]; memcpy(fill, first_index, sizeof fill);
      for (uint32_t i = im; i <= iM; ++i) if (len[i]) syms[(size_t) fill[len[i]]++] = i; }
    PizBits bits { data, table_end, size };
    size_t pos = 0; uint64_t code = 0; int length = 0; uint64_t used = 0;
    while (used < n_bits && pos < n_raw) {
        code = (code << 1) | bits.get(1); ++length; ++used;
        if (length > 58) throw std::runtime_error("read_exr(): invalid PIZ Huffman code");
        if (count[length] == 0 || code < base[length] || code - base[length] >= count[length]) continue;
        const uint32_t sym = syms[(size_t) (first_index[length] + (code - base[length]))];
        if (sym == iM) {                                 // the run-length symbol
            const uint32_t run = bits.get(8); used += 8;
            if (pos == 0 || pos + run > n_raw) throw std::runtime_error("read_exr(): invalid PIZ run");
            for (uint32_t r = 0; r < run; ++r) out[pos + r] = out[pos - 1];
            pos += run;
        } else out[pos++] = (uint16_t) sym;
        code = 0; length = 0;
    }
    if (pos != n_raw) throw std::runtime_error("read_exr(): PIZ chunk decodes to the wrong size");
}
inline void wdec14(uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) { const int ls = (int16_t) l, hs = (int16_t) hh; const int ai = ls + (hs & 1) + (hs >> 1); a = (uint16_t) ai; b = (uint16_t) (ai - hs); }
inline void wdec16(uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) { const int m = l, d = hh; const int bb = (m - (d >> 1)) & 0xffff; const int aa = (d + bb - 0x8000) & 0xffff; b = (uint16_t) bb; a = (uint16_t) aa; }
void piz_wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {   // ImfWav wav2Decode
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx; int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    auto dec = [&](uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) { if (w14) wdec14(l, hh, a, b); else wdec16(l, hh, a, b); };
    while (p >= 1) {
        uint16_t *py = in; uint16_t *const ey = in + (ptrdiff_t) oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t) oy * p, oy2 = (ptrdiff_t) oy * p2, ox1 = (ptrdiff_t) ox * p, ox2 = (ptrdiff_t) ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py; uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10); dec(*p01, *p11, i01, i11); dec(i00, i01, *px, *p01); dec(i10, i11, *p10, *p11);
            }
            if (nx & p) { uint16_t *p10 = px + oy1; dec(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t *px = py; uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t *p01 = px + ox1; dec(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}
}  // namespace
static void read_exr(const std::vector<uint8_t> &file, const std::string &path, std::vector<float> &rgb, uint32_t &w, uint32_t &h) {
    auto fail = [&](const std::string &m) { throw std::runtime_error("read_exr(): \"" + path + "\": " + m); };
    auto need = [&](size_t at, size_t n) { if (at > file.size() || file.size() - at < n) fail("truncated file"); };
    auto rd32 = [&](size_t at) { need(at, 4); uint32_t v; memcpy(&v, &file[at], 4); return v; };
    need(0, 8);
    const uint32_t version = rd32(4);
    if ((version & 0xff) != 2 || (version & 0x1a00)) fail("only single-part scan-line OpenEXR 2 files are supported (no tiles, deep data or multi-part files)");
    size_t pos = 8;
    struct Chan { std::string name; int type; }; std::vector<Chan> chans;
    int comp = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    for (;;) {
        need(pos, 1);
        if (file[pos] == 0) { ++pos; break; }
        std::string name, type;
        while (pos < file.size() && file[pos]) name += (char) file[pos++];
        ++pos;
        while (pos < file.size() && file[pos]) type += (char) file[pos++];
        ++pos;
        const uint32_t size = rd32(pos); pos += 4; need(pos, size);
        if (name == "compression" && size >= 1) comp = file[pos];
        else if (name == "dataWindow" && size >= 16) { x0 = (int) rd32(pos); y0 = (int) rd32(pos + 4); x1 = (int) rd32(pos + 8); y1 = (int) rd32(pos + 12); }
        else if (name == "channels") {
            size_t p = pos; const size_t end = pos + size;
            while (p < end && file[p]) {
                Chan c; while (p < end && file[p]) c.name += (char) file[p++];
                if (p + 17 > end) fail("bad channel list");
                c.type = (int) rd32(p + 1); p += 17;
                if (c.type < 0 || c.type > 2) fail("bad channel type");
                if (rd32(p - 8) != 1 || rd32(p - 4) != 1) fail("subsampled channels are not supported");
                chans.push_back(c);
            }
        }
        pos += size;
    }
    if (comp != 0 && comp != 2 && comp != 3 && comp != 4) fail("only uncompressed, ZIP- and PIZ-compressed files are supported (not RLE, PXR24, B44 or DWA)");
    if (x1 < x0 || y1 < y0 || chans.empty()) fail("bad header");
    const uint64_t W = (uint64_t) ((int64_t) x1 - x0 + 1), H = (uint64_t) ((int64_t) y1 - y0 + 1);
    if (W > 65536 || H > 65536 || W * H > (1ull << 26)) fail("implausible size");
    int ir = -1, ig = -1, ib = -1, iy = -1;
    for (size_t c = 0; c < chans.size(); ++c) { if (chans[c].name == "R") ir = (int) c; if (chans[c].name == "G") ig = (int) c; if (chans[c].name == "B") ib = (int) c; if (chans[c].name == "Y") iy = (int) c; }
    const bool colour = ir >= 0 && ig >= 0 && ib >= 0;
    if (!colour && iy < 0) fail("no R, G, B or Y channels");
    size_t line_bytes = 0; for (auto &c : chans) line_bytes += (c.type == 1 ? 2 : 4) * (size_t) W;
    const uint32_t lines = comp == 3 ? 16 : comp == 4 ? 32 : 1; const size_t n_chunks = (size_t) ((H + lines - 1) / lines);
    need(pos, n_chunks * 8);
    if (line_bytes * H / 1024 > file.size() && comp == 0) fail("truncated file");     // a damaged size must not be allocated before the first read fails
    w = (uint32_t) W; h = (uint32_t) H; rgb.assign((size_t) W * H * 3, 0.f);
    std::vector<uint8_t> raw, tmp;
    for (size_t k = 0; k < n_chunks; ++k) {
        uint64_t off; memcpy(&off, &file[pos + 8 * k], 8);
        if (off > file.size()) fail("bad chunk offset");
        need((size_t) off, 8);
        const int y = (int) rd32((size_t) off); const uint32_t size = rd32((size_t) off + 4);
        need((size_t) off + 8, size);
        if (y < y0 || y > y1) fail("bad chunk");
        const uint32_t ny = (uint32_t) std::min<int64_t>(lines, (int64_t) y1 - y + 1);
        const size_t raw_len = line_bytes * ny;
        const uint8_t *src = &file[(size_t) off + 8];
        bool planar = false;   // PIZ leaves the chunk channel by channel (all rows of the first channel, then the next)
        if (comp == 4 && size < raw_len) {
            if (size < 8) fail("truncated PIZ chunk");
            uint16_t mn, mxv; memcpy(&mn, src, 2); memcpy(&mxv, src + 2, 2);
            size_t p = 4;
            std::vector<uint8_t> bitmap(8192, 0);
            if (mn <= mxv) { if (mxv >= 8192 || p + (size_t) (mxv - mn + 1) > size) fail("bad PIZ bitmap"); memcpy(&bitmap[mn], src + p, (size_t) (mxv - mn + 1)); p += (size_t) (mxv - mn + 1); }
            std::vector<uint16_t> lut; lut.reserve(65536);
            for (uint32_t v = 0; v < 65536; ++v) if (v == 0 || (bitmap[v >> 3] & (1u << (v & 7)))) lut.push_back((uint16_t) v);
            const uint16_t max_value = (uint16_t) (lut.size() - 1);
            if (p + 4 > size) fail("truncated PIZ chunk");
            int32_t length; memcpy(&length, src + p, 4); p += 4;
            if (length < 0 || p + (size_t) length > size) fail("bad PIZ chunk");
            std::vector<uint16_t> sym(raw_len / 2);
            piz_huf_uncompress(src + p, (size_t) length, sym.data(), sym.size());
            size_t q = 0;
            for (auto &c : chans) {
                const int sz = c.type == 1 ? 1 : 2;
                for (int j = 0; j < sz; ++j) piz_wav2_decode(sym.data() + q + j, (int) W, sz, (int) ny, (int) W * sz, max_value);
                q += (size_t) W * ny * sz;
            }
            for (uint16_t &v : sym) v = v < lut.size() ? lut[v] : 0;
            raw.resize(raw_len); memcpy(raw.data(), sym.data(), raw_len);
            src = raw.data(); planar = true;
        } else if (comp != 0 && comp != 4 && size < raw_len) {
            tmp.resize(raw_len); uLongf got = (uLongf) raw_len;
            if (uncompress(tmp.data(), &got, src, size) != Z_OK || got != raw_len) fail("corrupt ZIP chunk");
            for (size_t i = 1; i < raw_len; ++i) tmp[i] = (uint8_t) (tmp[i - 1] + tmp[i] - 128);      // undo the predictor
            raw.resize(raw_len);
            const size_t half = (raw_len + 1) / 2;
            for (size_t i = 0; i < raw_len; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];    // re-interleave
            src = raw.data();
        } else if (size != raw_len) fail("bad chunk size");
        size_t q = 0;
        for (uint32_t rr = 0; rr < (planar ? 1u : ny); ++rr) for (size_t c = 0; c < chans.size(); ++c) for (uint32_t r = planar ? 0 : rr; r < (planar ? ny : rr + 1); ++r) {
            const int t = chans[c].type; const size_t bpp = t == 1 ? 2 : 4;
            int dst = -1; if (colour) dst = (int) c == ir ? 0 : (int) c == ig ? 1 : (int) c == ib ? 2 : -1; else if ((int) c == iy) dst = 3;
            if (dst >= 0) for (uint64_t x = 0; x < W; ++x) {
                float v;
                if (t == 1) { uint16_t hb; memcpy(&hb, src + q + 2 * x, 2); v = half_to_float(hb); }
                else if (t == 2) memcpy(&v, src + q + 4 * x, 4);
                else { uint32_t u; memcpy(&u, src + q + 4 * x, 4); v = (float) u; }
                float *px = &rgb[((size_t) (y - y0 + (int) r) * W + x) * 3];
                if (dst == 3) px[0] = px[1] = px[2] = v; else px[dst] = v;
            }
            q += bpp * (size_t) W;
        }
    }
}
void read_radiance_image(const std::string &path, std::vector<float> &rgb, uint32_t &width, uint32_t &height, float (*srgb_to_linear_u8)(uint32_t)) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("could not open \"" + path + "\"");
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() >= 8 && file[0] == 'P' && (file[1] == 'F' || file[1] == 'f')) return read_pfm(file, path, rgb, width, height);
    if (file.size() >= 8 && file[0] == '#' && file[1] == '?') return read_rgbe(file, path, rgb, width, height);
    if (file.size() >= 8 && file[0] == 0x76 && file[1] == 0x2f && file[2] == 0x31 && file[3] == 0x01) return read_exr(file, path, rgb, width, height);
    const bool jpeg = file.size() >= 8 && file[0] == 0xff && file[1] == 0xd8;
    if (jpeg || (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P')) {   // 8-bit PNG / JPEG: sRGB -> linear (Bitmap::convert to Float32 with srgb_gamma = false)
        std::vector<uint8_t> px; uint32_t ch;
        if (jpeg) read_jpeg(path, px, width, height, ch); else read_png(path, px, width, height, ch);
        float lut[256]; for (uint32_t i = 0; i < 256; ++i) lut[i] = srgb_to_linear_u8(i);
        rgb.resize((size_t) width * height * 3);
        for (size_t i = 0; i < (size_t) width * height; ++i) for (uint32_t c = 0; c < 3; ++c) rgb[i * 3 + c] = lut[px[i * ch + (ch == 3 ? c : 0)]];
        return;
    }
    throw std::runtime_error("envmap: \"" + path + "\": unsupported image format (this build reads RGBE .hdr, PFM, OpenEXR without or with ZIP compression, 8-bit PNG and baseline JPEG radiance maps)");
}

}  // namespace dtof
