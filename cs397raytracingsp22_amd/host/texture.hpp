// texture.hpp — C++ host mirror of src/util/texture.rs: `Texture` holds a decoded RGB8 image.
// Texture::sample (texture.rs:26-32) runs on the GPU.  Texture::load_from_file (texture.rs:16-25) is load-time work:
// `image::open(file)` then, per sample, `get_pixel(..).to_rgb()` — i.e. whatever the file holds, the path sees RGB8.
// This mirror decodes what the reference's assets use, by content (magic bytes), and returns an empty optional on any
// failure, as the reference returns None (texture.rs:22-24):
//   PNG   every colour type at 1/2/4/8 bits per channel, interlaced or not, CRC-checked (inflate: zlib) — byte-exact
//         against the `image` crate's RGB8 (palette -> RGB, grey -> replicated, alpha dropped; grey < 8 bits scaled to 0..255)
//   TGA   true-colour 24/32 bpp, grey 8 bpp, colour-mapped (24/32-bit entries), raw or RLE, either origin — byte-exact
//   PPM   binary P6 (what the tests and tools exchange)
// JPEG is not decoded here (the reference's three .jpg files include a progressive one; a decoder that must agree with
// jpeg-decoder 0.1.22 to +-2 LSB is a project of its own): load_from_file returns nullopt and the caller supplies
// decoded texels (scenes do that through the Python mirror, texture.py).  16-bit PNG channels: nullopt (the crate's
// 16 -> 8 bit conversion could not be pinned without its source).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <string>
#include <vector>
#include <zlib.h>

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
        if (d.size() >= 2 && d[0] == 0xff && d[1] == 0xd8) return std::nullopt;          // JPEG: see the header comment
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
