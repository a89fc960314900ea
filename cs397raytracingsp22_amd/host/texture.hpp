// texture.hpp — C++ host mirror of src/util/texture.rs: `Texture` holds a decoded RGB8 image.
// Texture::sample (texture.rs:26-32) runs on the GPU.  Texture::load_from_file (texture.rs:16-25) is load-time work:
// `image::open(file)` then, per sample, `get_pixel(..).to_rgb()` — i.e. whatever the file holds, the path sees RGB8.
// This mirror decodes what the reference's assets use, by content (magic bytes), and returns an empty optional on any
// failure, as the reference returns None (texture.rs:22-24):
//   PNG   every colour type at 1/2/4/8 bits per channel, interlaced or not, CRC-checked (inflate: zlib) — byte-exact
//         against the `image` crate's RGB8 (palette -> RGB, grey -> replicated, alpha dropped; grey < 8 bits scaled to 0..255)
//   TGA   true-colour 24/32 bpp, grey 8 bpp, colour-mapped (24/32-bit entries), raw or RLE, either origin — byte-exact
//   PPM   binary P6 (what the tests and tools exchange)
//   JPEG  8-bit baseline / extended-sequential AND progressive Huffman, grey or YCbCr, sampling 1x1 / 2x1 / 2x2 (what the
//         reference's earthmap.jpg, normal_test.jpg and the progressive magenta.jpg use), restart intervals.  The arithmetic is
//         the classic one — 13-bit fixed-point "islow" inverse DCT, triangle-filter ("fancy") chroma upsampling, 16-bit
//         fixed-point YCbCr -> RGB — so the result equals libjpeg's (PIL's) bit for bit on the reference's files; the `image`
//         crate's jpeg-decoder 0.1.22 is a different implementation of the same standard and may differ from both by +-1-2 LSB
//         (SURVEY.md section 8c says so and ships textures pre-decoded for parity runs).
//   BMP / GIF / TIFF  the common sub-formats: texture_formats.hpp
// Not decoded (nullopt): 16-bit PNG channels (the crate's 16 -> 8 bit conversion could not be pinned without its source),
// arithmetic-coded / lossless / 12-bit / CMYK JPEG, exotic sampling factors.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <string>
#include <vector>
#include <zlib.h>

#include "texture_formats.hpp"

namespace cs397 {

struct Texture {                                       // texture.rs:12-14
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;                          // width*height*3, row 0 = top

    static std::optional<Texture> load_from_file(const std::string& file_name) {
        std::vector<uint8_t> d;
        {
            FILE* f = fopen(file_name.c_str(), "rb");
            if (!f) return std::nullopt;
            uint8_t buf[65536]; size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
            fclose(f);
        }
        static const uint8_t png_sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
        if (d.size() >= 8 && memcmp(d.data(), png_sig, 8) == 0) return decode_png(d);
        if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') return decode_ppm(d);
        if (d.size() >= 2 && d[0] == 0xff && d[1] == 0xd8) return decode_jpeg(d);
        {   // the rarer containers image::open reads as well (texture_formats.hpp)
            const bool bmp = d.size() >= 2 && d[0] == 'B' && d[1] == 'M', gif = d.size() >= 4 && memcmp(d.data(), "GIF8", 4) == 0;
            const bool tiff = d.size() >= 4 && (memcmp(d.data(), "II*\0", 4) == 0 || memcmp(d.data(), "MM\0*", 4) == 0);
            if (bmp || gif || tiff) {
                Texture t;
                const bool ok = bmp ? formats::decode_bmp(d, t.width, t.height, t.rgb) : gif ? formats::decode_gif(d, t.width, t.height, t.rgb)
                                                                                                : formats::decode_tiff(d, t.width, t.height, t.rgb);
                if (ok) return t;
                return std::nullopt;
            }
        }
        return decode_tga(d);                                                            // TGA has no magic: try it last
    }

  private:
    static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

    static std::optional<Texture> decode_ppm(const std::vector<uint8_t>& d) {
        size_t i = 2; int vals[3], got = 0;
        while (got < 3 && i < d.size()) {
            while (i < d.size() && (d[i] == ' ' || d[i] == '\n' || d[i] == '\r' || d[i] == '\t')) i++;
            if (i < d.size() && d[i] == '#') { while (i < d.size() && d[i] != '\n') i++; continue; }
            int v = 0; bool any = false;
            while (i < d.size() && d[i] >= '0' && d[i] <= '9') { v = v * 10 + (d[i] - '0'); i++; any = true; if (v > (1 << 24)) return std::nullopt; }
            if (!any) return std::nullopt;
            vals[got++] = v;
        }
        if (got != 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0 || i >= d.size()) return std::nullopt;
        i++;                                                   // the single whitespace after maxval
        Texture t; t.width = vals[0]; t.height = vals[1];
        const size_t n = (size_t)t.width * t.height * 3;
        if (d.size() - i < n) return std::nullopt;
        t.rgb.assign(d.begin() + (long)i, d.begin() + (long)(i + n));
        return t;
    }

    // ---------------------------------------------------------------- PNG (ISO/IEC 15948)
    static uint8_t paeth(int a, int b, int c) {
        const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
        return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
    }
    // undo the scanline filters of one (sub)image in place: rows of 1 + stride bytes
    static bool unfilter(uint8_t* p, size_t rows, size_t stride, size_t bpp) {
        std::vector<uint8_t> zero(stride, 0);
        const uint8_t* prev = zero.data();
        for (size_t y = 0; y < rows; y++) {
            uint8_t* row = p + y * (stride + 1);
            const uint8_t ft = row[0];
            uint8_t* cur = row + 1;
            for (size_t x = 0; x < stride; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                switch (ft) {
                case 0: break;
                case 1: cur[x] = (uint8_t)(cur[x] + a); break;
                case 2: cur[x] = (uint8_t)(cur[x] + b); break;
                case 3: cur[x] = (uint8_t)(cur[x] + ((a + b) >> 1)); break;
                case 4: cur[x] = (uint8_t)(cur[x] + paeth(a, b, c)); break;
                default: return false;
                }
            }
            prev = cur;
        }
        return true;
    }
    // Untrusted input: no decoder allocates for an image of more than kMaxPixels, nor for one whose file is too short to
    // hold it (a few header bytes must not cost gigabytes); such files return None like any undecodable file (texture.rs:22-24).
    static constexpr size_t kMaxPixels = (size_t)1 << 27;              // 134 M pixels (16384 x 8192)
    static std::optional<Texture> decode_png(const std::vector<uint8_t>& d) {
        size_t i = 8;
        uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
        std::vector<uint8_t> idat, plte;
        bool have_ihdr = false, end = false;
        while (!end) {
            if (d.size() - i < 12) return std::nullopt;
            const uint32_t len = be32(&d[i]);
            if (len > d.size() - i - 12) return std::nullopt;
            const uint8_t* type = &d[i + 4];
            const uint8_t* data = &d[i + 8];
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len)) return std::nullopt;
            if (!memcmp(type, "IHDR", 4)) {
                if (len != 13 || have_ihdr) return std::nullopt;
                W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
                if (data[10] != 0 || data[11] != 0 || interlace > 1 || W == 0 || H == 0 || W > (1u << 15) || H > (1u << 15)) return std::nullopt;
                have_ihdr = true;
            } else if (!have_ihdr) return std::nullopt;
            else if (!memcmp(type, "PLTE", 4)) { if (len % 3 || len > 768) return std::nullopt; plte.assign(data, data + len); }
            else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
            else if (!memcmp(type, "IEND", 4)) end = true;
            else if (!(type[0] & 0x20)) return std::nullopt;          // unknown CRITICAL chunk
            i += 12 + (size_t)len;
        }
        int channels;
        switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: return std::nullopt; }
        if (!(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return std::nullopt;   // 16-bit: see header
        if (ctype == 3 && plte.empty()) return std::nullopt;
        const size_t bits_pp = (size_t)channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
        // (sub)images: the whole image, or the seven Adam7 passes
        struct Pass { uint32_t x0, y0, dx, dy; };
        std::vector<Pass> passes;
        if (!interlace) passes.push_back(Pass{ 0, 0, 1, 1 });
        else { static const Pass a7[7] = { {0,0,8,8}, {4,0,8,8}, {0,4,4,8}, {2,0,4,4}, {0,2,2,4}, {1,0,2,2}, {0,1,1,2} }; passes.assign(a7, a7 + 7); }
        size_t raw_size = 0;
        for (const Pass& p : passes) {
            const size_t pw = (W > p.x0) ? (W - p.x0 + p.dx - 1) / p.dx : 0, ph = (H > p.y0) ? (H - p.y0 + p.dy - 1) / p.dy : 0;
            if (pw && ph) raw_size += ph * (1 + (pw * bits_pp + 7) / 8);
        }
        if ((size_t)W * H > kMaxPixels || raw_size / 1032 > idat.size() + 64) return std::nullopt;     // deflate expands at most 1032 : 1
        std::vector<uint8_t> raw(raw_size);
        uLongf got = (uLongf)raw_size;
        if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size) return std::nullopt;
        Texture t; t.width = (int)W; t.height = (int)H; t.rgb.assign((size_t)W * H * 3, 0);
        size_t off = 0;
        const int maxv = (1 << depth) - 1;
        for (const Pass& p : passes) {
            const size_t pw = (W > p.x0) ? (W - p.x0 + p.dx - 1) / p.dx : 0, ph = (H > p.y0) ? (H - p.y0 + p.dy - 1) / p.dy : 0;
            if (!pw || !ph) continue;
            const size_t stride = (pw * bits_pp + 7) / 8;
            if (!unfilter(raw.data() + off, ph, stride, bpp)) return std::nullopt;
            for (size_t y = 0; y < ph; y++) {
                const uint8_t* row = raw.data() + off + y * (stride + 1) + 1;
                for (size_t x = 0; x < pw; x++) {
                    uint8_t* o = &t.rgb[(((size_t)p.y0 + y * p.dy) * W + p.x0 + x * p.dx) * 3];
                    if (depth == 8) {
                        const uint8_t* s = row + x * (size_t)channels;
                        if (ctype == 2 || ctype == 6) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
                        else if (ctype == 0 || ctype == 4) { o[0] = o[1] = o[2] = s[0]; }
                        else { if ((size_t)s[0] * 3 + 2 >= plte.size()) return std::nullopt; o[0] = plte[s[0] * 3]; o[1] = plte[s[0] * 3 + 1]; o[2] = plte[s[0] * 3 + 2]; }
                    } else {
                        const size_t bit = x * (size_t)depth;
                        const int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
                        if (ctype == 0) { o[0] = o[1] = o[2] = (uint8_t)(v * 255 / maxv); }
                        else { if ((size_t)v * 3 + 2 >= plte.size()) return std::nullopt; o[0] = plte[v * 3]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3 + 2]; }
                    }
                }
            }
            off += ph * (stride + 1);
        }
        return t;
    }

    // ---------------------------------------------------------------- JPEG (ITU-T T.81), Huffman, 8 bit
    struct JHuff { uint8_t bits[17]; uint8_t vals[256]; int mincode[17], maxcode[18], valptr[17]; bool ok = false; };
    struct JComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int bw = 0, bh = 0;            // blocks per row / column (MCU-padded)
                   int w = 0, hgt = 0;                                                              // blocks that carry image data
                   std::vector<int16_t> coef; std::vector<uint8_t> plane; };
    struct JBits {
        const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0; bool hit_marker = false;
        void fill() {
            while (n <= 24) {
                int b = 0;
                if (!hit_marker && p < end) {
                    b = *p;
                    if (b == 0xff) {
                        if (p + 1 < end && p[1] == 0x00) p += 2;
                        else { hit_marker = true; b = 0; }                 // a marker: feed zeros, leave p on it
                    } else p++;
                }
                acc |= (uint32_t)b << (24 - n); n += 8;
            }
        }
        int get(int k) { if (k == 0) return 0; if (n < k) fill(); const int v = (int)(acc >> (32 - k)); acc <<= k; n -= k; return v; }
        int bit() { return get(1); }
        void reset() { acc = 0; n = 0; hit_marker = false; }
    };
    static bool jhuff_build(JHuff& h) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            h.valptr[l] = k; h.mincode[l] = code;
            code += h.bits[l]; k += h.bits[l];
            h.maxcode[l] = h.bits[l] ? code - 1 : -1;
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        h.maxcode[17] = 0x7fffffff; h.ok = k <= 256;
        return h.ok;
    }
    static int jhuff_decode(JBits& b, const JHuff& h) {
        int code = b.bit();
        for (int l = 1; l <= 16; l++) {
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
            code = (code << 1) | b.bit();
        }
        return -1;
    }
    static int jextend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    static const uint8_t* jzigzag() {
        static const uint8_t z[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
        return z;
    }
    static uint8_t jclamp(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    // inverse DCT, 13-bit fixed point, two passes (columns then rows): dequantised block in natural order -> 8x8 samples
    static void jidct(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
        const int CB = 13, P1 = 2;
        int ws[64];
        auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
        for (int c = 0; c < 8; c++) {
            auto D = [&](int r) { return (long)in[r * 8 + c] * q[r * 8 + c]; };
            long z2 = D(2), z3 = D(6);
            long z1 = (z2 + z3) * 4433;
            long tmp2 = z1 + z3 * -15137, tmp3 = z1 + z2 * 6270;
            z2 = D(0); z3 = D(4);
            long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = D(7); tmp1 = D(5); tmp2 = D(3); tmp3 = D(1);
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * 9633;
            tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; r++) {
            const int* w = ws + r * 8;
            long z2 = w[2], z3 = w[6];
            long z1 = (z2 + z3) * 4433;
            long tmp2 = z1 + z3 * -15137, tmp3 = z1 + z2 * 6270;
            long tmp0 = ((long)w[0] + w[4]) << CB, tmp1 = ((long)w[0] - w[4]) << CB;
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * 9633;
            tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            uint8_t* o = out + r * stride;
            const int S = CB + P1 + 3;
            o[0] = jclamp(descale(tmp10 + tmp3, S) + 128); o[7] = jclamp(descale(tmp10 - tmp3, S) + 128);
            o[1] = jclamp(descale(tmp11 + tmp2, S) + 128); o[6] = jclamp(descale(tmp11 - tmp2, S) + 128);
            o[2] = jclamp(descale(tmp12 + tmp1, S) + 128); o[5] = jclamp(descale(tmp12 - tmp1, S) + 128);
            o[3] = jclamp(descale(tmp13 + tmp0, S) + 128); o[4] = jclamp(descale(tmp13 - tmp0, S) + 128);
        }
    }
    static std::optional<Texture> decode_jpeg(const std::vector<uint8_t>& d) {
        const uint8_t* Z = jzigzag();
        uint16_t qt[4][64]; bool have_q[4] = { false, false, false, false };
        JHuff hdc[4], hac[4];
        std::vector<JComp> comp;
        int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, adobe_transform = -1;
        bool progressive = false, have_sof = false;
        int mcux = 0, mcuy = 0;
        size_t i = 2;
        auto u16 = [&](size_t k) { return (int)((d[k] << 8) | d[k + 1]); };
        while (true) {
            while (i < d.size() && d[i] != 0xff) i++;                     // (garbage between segments is tolerated)
            while (i < d.size() && d[i] == 0xff) i++;
            if (i >= d.size()) return std::nullopt;
            const int m = d[i++];
            if (m == 0xd9) break;                                         // EOI
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
            if (i + 2 > d.size()) return std::nullopt;
            const int len = u16(i);
            if (len < 2 || i + (size_t)len > d.size()) return std::nullopt;
            const size_t seg = i + 2, seg_end = i + (size_t)len;
            if (m == 0xdb) {                                              // DQT
                size_t k = seg;
                while (k < seg_end) {
                    const int pq = d[k] >> 4, tq = d[k] & 15; k++;
                    if (tq > 3 || pq > 1 || k + (size_t)(pq ? 128 : 64) > seg_end) return std::nullopt;
                    for (int z = 0; z < 64; z++) { qt[tq][Z[z]] = (uint16_t)(pq ? u16(k) : d[k]); k += pq ? 2 : 1; }
                    have_q[tq] = true;
                }
            } else if (m == 0xc4) {                                       // DHT
                size_t k = seg;
                while (k < seg_end) {
                    const int tc = d[k] >> 4, th = d[k] & 15; k++;
                    if (tc > 1 || th > 3 || k + 16 > seg_end) return std::nullopt;
                    JHuff& h = tc ? hac[th] : hdc[th];
                    int total = 0; h.bits[0] = 0;
                    for (int l = 1; l <= 16; l++) { h.bits[l] = d[k++]; total += h.bits[l]; }
                    if (total > 256 || k + (size_t)total > seg_end) return std::nullopt;
                    for (int v = 0; v < total; v++) h.vals[v] = d[k++];
                    if (!jhuff_build(h)) return std::nullopt;
                }
            } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {             // SOF0 / SOF1 / SOF2
                if (have_sof || len < 8 || d[seg] != 8) return std::nullopt;
                H = u16(seg + 1); W = u16(seg + 3);
                const int nc = d[seg + 5];
                if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || len < 8 + 3 * nc) return std::nullopt;
                // every 8x8 block costs at least one bit of entropy-coded data in a first scan: a file shorter than that is truncated
                if ((size_t)W * H > kMaxPixels || ((size_t)W * H) / 64 / 8 > d.size()) return std::nullopt;
                progressive = m == 0xc2; have_sof = true;
                comp.resize((size_t)nc);
                for (int c = 0; c < nc; c++) {
                    JComp& C = comp[(size_t)c];
                    C.id = d[seg + 6 + 3 * c]; C.h = d[seg + 7 + 3 * c] >> 4; C.v = d[seg + 7 + 3 * c] & 15; C.tq = d[seg + 8 + 3 * c];
                    if (C.h < 1 || C.h > 2 || C.v < 1 || C.v > 2 || C.tq > 3) return std::nullopt;
                    hmax = C.h > hmax ? C.h : hmax; vmax = C.v > vmax ? C.v : vmax;
                }
                if (nc == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
                mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
                for (JComp& C : comp) {
                    if (hmax % C.h || vmax % C.v) return std::nullopt;
                    if (C.id != comp[0].id && (C.h != 1 || C.v != 1)) return std::nullopt;           // chroma finer than 1x1: not handled
                    C.bw = mcux * C.h; C.bh = mcuy * C.v;
                    const int cw = (W * C.h + hmax - 1) / hmax, ch = (H * C.v + vmax - 1) / vmax;
                    C.w = (cw + 7) / 8; C.hgt = (ch + 7) / 8;
                    C.coef.assign((size_t)C.bw * C.bh * 64, 0);
                }
            } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
                return std::nullopt;                                     // lossless / hierarchical / arithmetic coding
            } else if (m == 0xdd) {                                       // DRI
                if (len < 4) return std::nullopt;
                restart = u16(seg);
            } else if (m == 0xee && len >= 14 && !memcmp(&d[seg], "Adobe", 5)) {
                adobe_transform = d[seg + 11];
            } else if (m == 0xda) {                                       // SOS + entropy-coded data
                if (!have_sof || len < 3) return std::nullopt;            // (len >= 3: the segment holds the component count at all)
                const int ns = d[seg];
                if (ns < 1 || ns > (int)comp.size() || len < 6 + 2 * ns) return std::nullopt;
                int sc[3];
                for (int k = 0; k < ns; k++) {
                    sc[k] = -1;
                    for (size_t c = 0; c < comp.size(); c++) if (comp[c].id == d[seg + 1 + 2 * k]) sc[k] = (int)c;
                    if (sc[k] < 0) return std::nullopt;
                    comp[(size_t)sc[k]].td = d[seg + 2 + 2 * k] >> 4; comp[(size_t)sc[k]].ta = d[seg + 2 + 2 * k] & 15;
                    if (comp[(size_t)sc[k]].td > 3 || comp[(size_t)sc[k]].ta > 3) return std::nullopt;
                }
                int Ss = d[seg + 1 + 2 * ns], Se = d[seg + 2 + 2 * ns], Ah = d[seg + 3 + 2 * ns] >> 4, Al = d[seg + 3 + 2 * ns] & 15;
                if (!progressive) { Ss = 0; Se = 63; Ah = Al = 0; }
                if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0 && progressive) || (Ss > 0 && ns != 1) || Al > 13) return std::nullopt;
                JBits b; b.p = d.data() + seg_end; b.end = d.data() + d.size();
                int pred[3] = { 0, 0, 0 }, eobrun = 0, rst_left = restart;
                const bool interleaved = ns > 1;
                const int nx = interleaved ? mcux : comp[(size_t)sc[0]].w, ny = interleaved ? mcuy : comp[(size_t)sc[0]].hgt;
                for (int my = 0; my < ny; my++) for (int mx = 0; mx < nx; mx++) {
                    if (restart && rst_left == 0) {
                        // expect RSTn: skip to it, reset predictors
                        b.reset();
                        while (b.p + 1 < b.end && !(b.p[0] == 0xff && b.p[1] >= 0xd0 && b.p[1] <= 0xd7)) b.p++;
                        if (b.p + 1 >= b.end) return std::nullopt;
                        b.p += 2;
                        pred[0] = pred[1] = pred[2] = 0; eobrun = 0; rst_left = restart;
                    }
                    if (restart) rst_left--;
                    for (int k = 0; k < ns; k++) {
                        JComp& C = comp[(size_t)sc[k]];
                        const int bh_n = interleaved ? C.h : 1, bv_n = interleaved ? C.v : 1;
                        for (int by = 0; by < bv_n; by++) for (int bx = 0; bx < bh_n; bx++) {
                            const int X = interleaved ? mx * C.h + bx : mx, Y = interleaved ? my * C.v + by : my;
                            int16_t* blk = &C.coef[((size_t)Y * C.bw + X) * 64];
                            if (!progressive) {
                                if (!hdc[C.td].ok || !hac[C.ta].ok) return std::nullopt;
                                const int t = jhuff_decode(b, hdc[C.td]);
                                if (t < 0 || t > 11) return std::nullopt;
                                pred[k] += t ? jextend(b.get(t), t) : 0;
                                blk[0] = (int16_t)pred[k];
                                for (int z = 1; z < 64;) {
                                    const int rs = jhuff_decode(b, hac[C.ta]);
                                    if (rs < 0) return std::nullopt;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { z += 16; continue; } break; }
                                    z += r;
                                    if (z > 63) return std::nullopt;
                                    blk[Z[z]] = (int16_t)jextend(b.get(sz), sz);
                                    z++;
                                }
                            } else if (Ss == 0) {
                                if (Ah == 0) {                             // DC first
                                    if (!hdc[C.td].ok) return std::nullopt;
                                    const int t = jhuff_decode(b, hdc[C.td]);
                                    if (t < 0 || t > 11) return std::nullopt;
                                    pred[k] += t ? jextend(b.get(t), t) : 0;
                                    blk[0] = (int16_t)(pred[k] * (1 << Al));
                                } else if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << Al));   // DC refinement
                            } else if (Ah == 0) {                          // AC first
                                if (!hac[C.ta].ok) return std::nullopt;
                                if (eobrun > 0) { eobrun--; continue; }
                                for (int z = Ss; z <= Se;) {
                                    const int rs = jhuff_decode(b, hac[C.ta]);
                                    if (rs < 0) return std::nullopt;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { z += 16; continue; }
                                        eobrun = (1 << r) - 1;
                                        if (r) eobrun += b.get(r);
                                        break;
                                    }
                                    z += r;
                                    if (z > 63) return std::nullopt;
                                    blk[Z[z]] = (int16_t)(jextend(b.get(sz), sz) * (1 << Al));
                                    z++;
                                }
                            } else {                                       // AC refinement (T.81 G.1.2.3)
                                if (!hac[C.ta].ok) return std::nullopt;
                                const int p1 = 1 << Al, m1 = -(1 << Al);
                                int z = Ss;
                                if (eobrun == 0) {
                                    for (; z <= Se;) {
                                        const int rs = jhuff_decode(b, hac[C.ta]);
                                        if (rs < 0) return std::nullopt;
                                        int r = rs >> 4; const int sz = rs & 15;
                                        int val = 0;
                                        if (sz) { if (sz != 1) return std::nullopt; val = b.bit() ? p1 : m1; }
                                        else if (r != 15) { eobrun = 1 << r; if (r) eobrun += b.get(r); break; }
                                        // advance over already-nonzero coefficients (correcting them) and r zero-history ones
                                        for (; z <= Se; z++) {
                                            int16_t& cf = blk[Z[z]];
                                            if (cf != 0) {
                                                if (b.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                                            } else { if (--r < 0) break; }
                                        }
                                        if (val && z <= Se) blk[Z[z]] = (int16_t)val;
                                        z++;
                                    }
                                }
                                if (eobrun > 0) {
                                    for (; z <= Se; z++) {
                                        int16_t& cf = blk[Z[z]];
                                        if (cf != 0 && b.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                                    }
                                    eobrun--;
                                }
                            }
                        }
                    }
                }
                // continue the marker scan behind the entropy-coded data
                size_t k = (size_t)(b.p - d.data());
                if (!b.hit_marker) {            // the reader stopped early: find the next real marker
                    while (k + 1 < d.size() && !(d[k] == 0xff && d[k + 1] != 0x00 && !(d[k + 1] >= 0xd0 && d[k + 1] <= 0xd7))) k++;
                }
                i = k;
                continue;
            }
            i = seg_end;
        }
        if (!have_sof) return std::nullopt;
        // dequantise + inverse DCT into per-component planes (MCU-padded)
        for (JComp& C : comp) {
            if (!have_q[C.tq]) return std::nullopt;
            C.plane.assign((size_t)C.bw * 8 * C.bh * 8, 0);
            const int stride = C.bw * 8;
            for (int by = 0; by < C.bh; by++) for (int bx = 0; bx < C.bw; bx++)
                jidct(&C.coef[((size_t)by * C.bw + bx) * 64], qt[C.tq], &C.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
        Texture t; t.width = W; t.height = H; t.rgb.assign((size_t)W * H * 3, 0);
        // chroma upsampling ("fancy": triangle filter, as libjpeg) to full resolution
        auto upsample = [&](const JComp& C, std::vector<uint8_t>& out) -> bool {
            out.assign((size_t)W * H, 0);
            const int stride = C.bw * 8;
            const int cw = (W * C.h + hmax - 1) / hmax, ch = (H * C.v + vmax - 1) / vmax;        // valid samples of this component
            const int fx = hmax / C.h, fy = vmax / C.v;
            if (fx == 1 && fy == 1) { for (int y = 0; y < H; y++) memcpy(&out[(size_t)y * W], &C.plane[(size_t)y * stride], (size_t)W); return true; }
            // libjpeg replicates the right / bottom edge of a component before upsampling: reads beyond cw / ch see the edge sample
            auto S = [&](int x, int y) { x = x < 0 ? 0 : (x >= cw ? cw - 1 : x); y = y < 0 ? 0 : (y >= ch ? ch - 1 : y); return (int)C.plane[(size_t)y * stride + x]; };
            if (fx == 2 && fy == 1) {
                for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
                    const int i0 = x >> 1;
                    int v;
                    if (cw == 1) v = S(0, y);
                    else if (x == 0) v = S(0, y);
                    else if (x == 2 * cw - 1) v = S(cw - 1, y);
                    else if (x & 1) v = (3 * S(i0, y) + S(i0 + 1, y) + 2) >> 2;
                    else v = (3 * S(i0, y) + S(i0 - 1, y) + 1) >> 2;
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
                return true;
            }
            if (fx == 2 && fy == 2) {
                for (int y = 0; y < H; y++) {
                    const int r = y >> 1, rn = (y & 1) ? r + 1 : r - 1;           // nearer neighbour row (edge rows replicate)
                    auto colsum = [&](int i1) { return 3 * S(i1, r) + S(i1, rn < 0 ? 0 : (rn >= ch ? ch - 1 : rn)); };
                    for (int x = 0; x < W; x++) {
                        const int i0 = x >> 1;
                        int v;
                        if (cw == 1) v = (colsum(0) * 4 + 8) >> 4;
                        else if (x == 0) v = (colsum(0) * 4 + 8) >> 4;
                        else if (x == 2 * cw - 1) v = (colsum(cw - 1) * 4 + 7) >> 4;
                        else if (x & 1) v = (colsum(i0) * 3 + colsum(i0 + 1) + 7) >> 4;
                        else v = (colsum(i0) * 3 + colsum(i0 - 1) + 8) >> 4;
                        out[(size_t)y * W + x] = (uint8_t)v;
                    }
                }
                return true;
            }
            return false;
        };
        if (comp.size() == 1) {
            const int stride = comp[0].bw * 8;
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const uint8_t g = comp[0].plane[(size_t)y * stride + x]; uint8_t* o = &t.rgb[((size_t)y * W + x) * 3]; o[0] = o[1] = o[2] = g; }
            return t;
        }
        std::vector<uint8_t> P[3];
        for (int c = 0; c < 3; c++) if (!upsample(comp[(size_t)c], P[c])) return std::nullopt;
        const bool ycc = adobe_transform != 0;                            // Adobe transform 0 = the three components ARE R, G, B
        for (size_t k = 0; k < (size_t)W * H; k++) {
            uint8_t* o = &t.rgb[k * 3];
            if (!ycc) { o[0] = P[0][k]; o[1] = P[1][k]; o[2] = P[2][k]; continue; }
            const int y = P[0][k], cb = P[1][k] - 128, cr = P[2][k] - 128;
            // 16-bit fixed point: 1.40200, 1.77200, 0.71414, 0.34414 (x 65536, rounded), ONE_HALF = 32768
            const int r = y + ((91881 * cr + 32768) >> 16);
            const int bl = y + ((116130 * cb + 32768) >> 16);
            const int g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
            o[0] = jclamp(r); o[1] = jclamp(g); o[2] = jclamp(bl);
        }
        return t;
    }

    // ---------------------------------------------------------------- TGA (Truevision, v1 fields)
    static std::optional<Texture> decode_tga(const std::vector<uint8_t>& d) {
        if (d.size() < 18) return std::nullopt;
        const int id_len = d[0], cmap_type = d[1], itype = d[2];
        const int cmap_first = d[3] | (d[4] << 8), cmap_len = d[5] | (d[6] << 8), cmap_bits = d[7];
        const int W = d[12] | (d[13] << 8), H = d[14] | (d[15] << 8), bpp = d[16], desc = d[17];
        const bool rle = itype == 9 || itype == 10 || itype == 11;
        const int base = rle ? itype - 8 : itype;                     // 1 colour-mapped, 2 true-colour, 3 grey
        if (W <= 0 || H <= 0 || cmap_type > 1 || !(base == 1 || base == 2 || base == 3)) return std::nullopt;
        if (base == 1 && !(cmap_type == 1 && bpp == 8 && (cmap_bits == 24 || cmap_bits == 32))) return std::nullopt;
        if (base == 2 && !(bpp == 24 || bpp == 32)) return std::nullopt;
        if (base == 3 && bpp != 8) return std::nullopt;
        size_t i = 18 + (size_t)id_len;
        const size_t cmap_bytes = cmap_type ? (size_t)cmap_len * ((size_t)(cmap_bits + 7) / 8) : 0;
        if (d.size() < i + cmap_bytes) return std::nullopt;
        const uint8_t* cmap = d.data() + i;
        i += cmap_bytes;
        const size_t px = (size_t)bpp / 8, n = (size_t)W * H;
        // raw: every pixel is in the file; RLE: a packet header byte stands for at most 128 pixels
        if (n > kMaxPixels || (!rle && d.size() - i < n * px) || (rle && n / 128 > d.size() - i)) return std::nullopt;
        std::vector<uint8_t> pix(n * px);
        if (!rle) {
            if (d.size() - i < n * px) return std::nullopt;
            memcpy(pix.data(), d.data() + i, n * px);
        } else {
            size_t o = 0;
            while (o < n) {
                if (i >= d.size()) return std::nullopt;
                const int hdr = d[i++]; const size_t cnt = (size_t)(hdr & 127) + 1;
                if (o + cnt > n) return std::nullopt;
                if (hdr & 128) {
                    if (d.size() - i < px) return std::nullopt;
                    for (size_t k = 0; k < cnt; k++) memcpy(&pix[(o + k) * px], &d[i], px);
                    i += px;
                } else {
                    if (d.size() - i < cnt * px) return std::nullopt;
                    memcpy(&pix[o * px], &d[i], cnt * px);
                    i += cnt * px;
                }
                o += cnt;
            }
        }
        Texture t; t.width = W; t.height = H; t.rgb.resize(n * 3);
        const bool top = (desc & 0x20) != 0, right = (desc & 0x10) != 0;
        const size_t ce = (size_t)(cmap_bits + 7) / 8;
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
            const uint8_t* s = &pix[((size_t)y * W + x) * px];
            uint8_t* o = &t.rgb[((size_t)(top ? y : H - 1 - y) * W + (right ? W - 1 - x : x)) * 3];
            if (base == 2) { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }            // stored B, G, R(, A)
            else if (base == 3) { o[0] = o[1] = o[2] = s[0]; }
            else {
                const int idx = (int)s[0] - cmap_first;
                if (idx < 0 || idx >= cmap_len) return std::nullopt;
                const uint8_t* e = cmap + (size_t)idx * ce;
                o[0] = e[2]; o[1] = e[1]; o[2] = e[0];
            }
        }
        return t;
    }
};

}  // namespace cs397
