// texture_formats.hpp — the rarer containers `image::open` (texture.rs:17) also reads: BMP, GIF and TIFF, decoded to RGB8 the way
// `get_pixel(..).to_rgb()` sees them (palette -> RGB, grey replicated, alpha dropped).  No asset of the reference uses them; they
// are here so that a scene file that names one does not lose its texture in the C++ mirror.  Pinned byte for byte against PIL's
// `.convert("RGB")` (tests/test_host_texture_decode.py) — for these lossless containers that IS the stored data — except where a
// line below says "unpinned".  Every function returns false for anything it does not decode (the reference's None) and never
// reads outside `d` or allocates more than the file can fill.
//   BMP   1 / 4 / 8 bpp paletted (raw, RLE4, RLE8), 24 bpp, 32 bpp (BI_RGB, or BI_BITFIELDS with the standard byte masks);
//         bottom-up or top-down; OS/2 core headers.  Not decoded: 16 bpp (the 5/6-bit -> 8-bit scaling of the crate is unpinned).
//   GIF   87a / 89a, first frame, global or local colour table, interlaced or not; transparency is dropped (the palette colour
//         stays, as to_rgb does).  A first frame smaller than the logical screen leaves the rest black (unpinned: PIL fills it
//         with the background colour instead); a screen far larger than the frame's data could fill is refused.
//   TIFF  II / MM, first IFD, strips, chunky planar configuration, 8 bits per sample: grey (black- or white-is-zero), RGB, RGB +
//         extra samples, palette (16-bit entries / 256: unpinned for the crate, PIL's rule); compression none, PackBits, LZW,
//         Deflate; horizontal predictor.  Not decoded: tiles, 1 / 4 / 16-bit samples, separate planes, JPEG-in-TIFF, CMYK / YCbCr.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include <zlib.h>

namespace cs397 {
namespace formats {

static constexpr size_t kMaxPixels = (size_t)1 << 27;      // as texture.hpp

struct View {
    const uint8_t* p; size_t n;
    bool has(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint16_t le16(size_t o) const { return (uint16_t)(p[o] | (p[o + 1] << 8)); }
    uint32_t le32(size_t o) const { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); }
    uint16_t be16(size_t o) const { return (uint16_t)((p[o] << 8) | p[o + 1]); }
    uint32_t be32(size_t o) const { return ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | (uint32_t)p[o + 3]; }
};

// ---------------------------------------------------------------- BMP
inline bool decode_bmp(const std::vector<uint8_t>& d, int& W, int& H, std::vector<uint8_t>& rgb) {
    const View v{ d.data(), d.size() };
    if (!v.has(0, 26) || d[0] != 'B' || d[1] != 'M') return false;
    const uint32_t data_off = v.le32(10), hsize = v.le32(14);
    int64_t w, h; uint32_t bpp, comp = 0, clr_used = 0;
    const bool core = hsize == 12;
    if (core) { w = v.le16(18); h = v.le16(20); bpp = v.le16(24); }
    else {
        if (hsize < 40 || !v.has(14, hsize)) return false;
        w = (int32_t)v.le32(18); h = (int32_t)v.le32(22); bpp = v.le16(28); comp = v.le32(30); clr_used = v.le32(46);
    }
    const bool top_down = h < 0;
    if (top_down) h = -h;
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (size_t)w * (size_t)h > kMaxPixels) return false;
    if (bpp != 1 && bpp != 4 && bpp != 8 && bpp != 24 && bpp != 32) return false;
    if (comp == 3) {                                        // BI_BITFIELDS: only the layout BI_RGB has anyway
        if (bpp != 32) return false;
        const size_t m = 14 + 40;                           // the masks follow a 40-byte header, or are part of a larger one
        if (!v.has(m, 12)) return false;
        if (v.le32(m) != 0x00ff0000u || v.le32(m + 4) != 0x0000ff00u || v.le32(m + 8) != 0x000000ffu) return false;
    } else if (!(comp == 0 || (comp == 1 && bpp == 8) || (comp == 2 && bpp == 4))) return false;
    if (top_down && comp != 0 && comp != 3) return false;   // RLE bitmaps are bottom-up by definition
    // palette
    uint8_t pal[256][3];
    memset(pal, 0, sizeof pal);
    if (bpp <= 8) {
        const size_t esz = core ? 3 : 4;
        size_t n = clr_used ? clr_used : ((size_t)1 << bpp);
        if (n > 256) return false;
        const size_t poff = 14 + (size_t)hsize;
        if (data_off >= poff) n = std::min(n, (size_t)(data_off - poff) / esz);
        if (!v.has(poff, n * esz)) return false;
        for (size_t i = 0; i < n; i++) { pal[i][0] = d[poff + i * esz + 2]; pal[i][1] = d[poff + i * esz + 1]; pal[i][2] = d[poff + i * esz]; }
    }
    if (data_off > d.size()) return false;
    const size_t Wz = (size_t)w, Hz = (size_t)h;
    const uint8_t* src = d.data() + data_off; const size_t avail = d.size() - data_off;
    if (comp == 0 || comp == 3) {
        const size_t stride = ((Wz * bpp + 31) / 32) * 4;
        if (stride * Hz > avail) return false;
        rgb.assign(Wz * Hz * 3, 0);
        for (size_t y = 0; y < Hz; y++) {
            const uint8_t* row = src + (top_down ? y : Hz - 1 - y) * stride;
            uint8_t* o = &rgb[y * Wz * 3];
            for (size_t x = 0; x < Wz; x++) {
                if (bpp == 24 || bpp == 32) { const uint8_t* q = row + x * (bpp / 8); o[0] = q[2]; o[1] = q[1]; o[2] = q[0]; }
                else {
                    unsigned idx;
                    if (bpp == 8) idx = row[x];
                    else if (bpp == 4) idx = (row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15u;
                    else idx = (row[x >> 3] >> (7 - (x & 7))) & 1u;
                    o[0] = pal[idx][0]; o[1] = pal[idx][1]; o[2] = pal[idx][2];
                }
                o += 3;
            }
        }
    } else {                                                // RLE8 / RLE4 into an index plane (unwritten pixels keep index 0)
        if (avail < 2 || Wz * Hz / 128 > avail + 512) return false;         // a run codes at most 255 pixels in 2 bytes: the file must be able to fill the plane
        std::vector<uint8_t> idx(Wz * Hz, 0);
        size_t i = 0, x = 0, y = 0;                         // y counts from the bottom row
        auto put = [&](unsigned val) { if (x < Wz && y < Hz) idx[(Hz - 1 - y) * Wz + x] = (uint8_t)val; x++; };
        bool done = false;
        while (!done && i + 1 < avail) {
            const unsigned n = src[i], b = src[i + 1]; i += 2;
            if (n) {
                for (unsigned k = 0; k < n; k++) put(bpp == 8 ? b : ((k & 1) ? (b & 15u) : (b >> 4)));
            } else if (b == 0) { x = 0; y++; }
            else if (b == 1) done = true;
            else if (b == 2) { if (i + 1 >= avail) return false; x += src[i]; y += src[i + 1]; i += 2; }
            else {
                const size_t bytes = bpp == 8 ? b : (b + 1u) / 2u;
                if (bytes > avail - i) return false;
                for (unsigned k = 0; k < b; k++) put(bpp == 8 ? src[i + k] : ((k & 1) ? (src[i + k / 2] & 15u) : (src[i + k / 2] >> 4)));
                i += bytes + (bytes & 1);
            }
            if (y > Hz) return false;
        }
        rgb.resize(Wz * Hz * 3);
        for (size_t k = 0; k < Wz * Hz; k++) { rgb[3 * k] = pal[idx[k]][0]; rgb[3 * k + 1] = pal[idx[k]][1]; rgb[3 * k + 2] = pal[idx[k]][2]; }
    }
    W = (int)w; H = (int)h;
    return true;
}

// ---------------------------------------------------------------- LZW (GIF: LSB-first codes; TIFF: MSB-first, early change)
// Appends at most `limit` bytes to `out`; false on a malformed stream.  A stream that simply ends is not an error here: the
// callers check the amount they got.
inline bool lzw_decode(const uint8_t* s, size_t n, int min_bits, bool msb_first, size_t limit, std::vector<uint8_t>& out) {
    if (min_bits < 2 || min_bits > 11) return false;
    const int clear = 1 << min_bits, eoi = clear + 1;
    std::vector<uint16_t> prefix(4096); std::vector<uint8_t> suffix(4096), first(4096);
    std::vector<uint8_t> stack(4097);
    for (int i = 0; i < clear; i++) { prefix[i] = 0xffff; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; }
    int next = eoi + 1, bits = min_bits + 1, prev = -1;
    uint32_t acc = 0; int nacc = 0; size_t i = 0;
    const size_t start = out.size();
    for (;;) {
        while (nacc < bits && i < n) {
            if (msb_first) acc = (acc << 8) | s[i]; else acc |= (uint32_t)s[i] << nacc;
            i++; nacc += 8;
        }
        if (nacc < bits) return true;                       // ran out of data
        int code;
        if (msb_first) { code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1u)); nacc -= bits; acc &= (1u << nacc) - 1u; }
        else { code = (int)(acc & ((1u << bits) - 1u)); acc >>= bits; nacc -= bits; }
        if (code == clear) { next = eoi + 1; bits = min_bits + 1; prev = -1; continue; }
        if (code == eoi) return true;
        if (prev < 0) {
            if (code >= clear) return false;
            if (out.size() - start >= limit) return true;
            out.push_back((uint8_t)code); prev = code; continue;
        }
        int cur = code; size_t sp = 0;
        if (code > next || (code == next && next >= 4096)) return false;
        if (code == next) { stack[sp++] = first[prev]; cur = prev; }          // KwKwK
        while (cur >= clear) { if (sp >= 4096) return false; stack[sp++] = suffix[cur]; cur = prefix[cur]; }
        stack[sp++] = (uint8_t)cur;
        const uint8_t f = (uint8_t)cur;
        while (sp > 0) { if (out.size() - start >= limit) return true; out.push_back(stack[--sp]); }
        if (next < 4096) {
            prefix[next] = (uint16_t)prev; suffix[next] = f; first[next] = first[prev];
            next++;
            const int grow_at = msb_first ? (1 << bits) - 1 : (1 << bits);     // TIFF switches one code early
            if (next >= grow_at && bits < 12) bits++;
        }
        prev = code;
    }
}

// ---------------------------------------------------------------- GIF
inline bool decode_gif(const std::vector<uint8_t>& d, int& W, int& H, std::vector<uint8_t>& rgb) {
    const View v{ d.data(), d.size() };
    if (!v.has(0, 13) || memcmp(d.data(), "GIF8", 4) != 0 || (d[4] != '7' && d[4] != '9') || d[5] != 'a') return false;
    const size_t sw = v.le16(6), sh = v.le16(8);
    if (sw == 0 || sh == 0 || sw * sh > kMaxPixels) return false;
    size_t i = 13;
    const uint8_t* gct = nullptr; size_t gct_n = 0;
    if (d[10] & 0x80) { gct_n = (size_t)2 << (d[10] & 7); if (!v.has(i, gct_n * 3)) return false; gct = d.data() + i; i += gct_n * 3; }
    auto skip_blocks = [&]() { while (i < d.size()) { const size_t n = d[i++]; if (n == 0) return true; if (n > d.size() - i) return false; i += n; } return false; };
    for (;;) {
        if (i >= d.size()) return false;
        const uint8_t tag = d[i++];
        if (tag == 0x3b) return false;                      // trailer before any image
        if (tag == 0x21) { if (i >= d.size()) return false; i++; if (!skip_blocks()) return false; continue; }
        if (tag != 0x2c) return false;
        if (!v.has(i, 9)) return false;
        const size_t fx = v.le16(i), fy = v.le16(i + 2), fw = v.le16(i + 4), fh = v.le16(i + 6);
        const uint8_t packed = d[i + 8]; i += 9;
        const uint8_t* ct = gct; size_t ct_n = gct_n;
        if (packed & 0x80) { ct_n = (size_t)2 << (packed & 7); if (!v.has(i, ct_n * 3)) return false; ct = d.data() + i; i += ct_n * 3; }
        if (!ct || fw == 0 || fh == 0 || fx + fw > sw || fy + fh > sh) return false;
        if (i >= d.size()) return false;
        const int min_bits = d[i++];
        std::vector<uint8_t> data;
        while (i < d.size()) { const size_t n = d[i++]; if (n == 0) break; if (n > d.size() - i) return false; data.insert(data.end(), d.begin() + (long)i, d.begin() + (long)(i + n)); i += n; }
        // the data must be able to fill the frame (a 12-bit code stands for at most 4096 pixels) — and the screen: a large canvas
        // around a tiny first frame is refused rather than allocated
        if (fw * fh > data.size() * 2731 + 4096 || sw * sh > data.size() * 2731 + 4096) return false;
        std::vector<uint8_t> idx;
        idx.reserve(fw * fh);
        if (!lzw_decode(data.data(), data.size(), min_bits, false, fw * fh, idx) || idx.size() != fw * fh) return false;
        rgb.assign(sw * sh * 3, 0);
        const bool interlaced = (packed & 0x40) != 0;
        size_t row = 0;
        static const int start[4] = { 0, 4, 2, 1 }, step[4] = { 8, 8, 4, 2 };
        for (int pass = 0; pass < (interlaced ? 4 : 1); pass++) {
            for (size_t y = interlaced ? (size_t)start[pass] : 0; y < fh; y += interlaced ? (size_t)step[pass] : 1, row++) {
                const uint8_t* s = &idx[row * fw];
                uint8_t* o = &rgb[((fy + y) * sw + fx) * 3];
                for (size_t x = 0; x < fw; x++, o += 3) {
                    const size_t c = s[x];
                    if (c < ct_n) { o[0] = ct[3 * c]; o[1] = ct[3 * c + 1]; o[2] = ct[3 * c + 2]; }
                }
            }
        }
        W = (int)sw; H = (int)sh;
        return true;
    }
}

// ---------------------------------------------------------------- TIFF
inline bool decode_tiff(const std::vector<uint8_t>& d, int& W, int& H, std::vector<uint8_t>& rgb) {
    const View v{ d.data(), d.size() };
    if (!v.has(0, 8)) return false;
    const bool le = d[0] == 'I' && d[1] == 'I', be = d[0] == 'M' && d[1] == 'M';
    if (!le && !be) return false;
    auto u16 = [&](size_t o) { return le ? v.le16(o) : v.be16(o); };
    auto u32 = [&](size_t o) { return le ? v.le32(o) : v.be32(o); };
    if (u16(2) != 42) return false;
    const size_t ifd = u32(4);
    if (!v.has(ifd, 2)) return false;
    const size_t n_ent = u16(ifd);
    if (!v.has(ifd + 2, n_ent * 12)) return false;
    struct Field { uint16_t type = 0; uint32_t count = 0; size_t off = 0; bool ok = false; };
    auto find = [&](uint16_t tag) {
        Field f;
        for (size_t k = 0; k < n_ent; k++) {
            const size_t e = ifd + 2 + k * 12;
            if (u16(e) != tag) continue;
            f.type = u16(e + 2); f.count = u32(e + 4);
            const size_t esz = f.type == 3 ? 2 : (f.type == 4 ? 4 : (f.type == 1 ? 1 : 0));
            if (esz == 0 || f.count == 0 || f.count > (1u << 28)) return f;
            const size_t bytes = esz * (size_t)f.count;
            f.off = bytes <= 4 ? e + 8 : (size_t)u32(e + 8);
            f.ok = v.has(f.off, bytes);
            return f;
        }
        return f;
    };
    auto val = [&](const Field& f, size_t k) -> uint32_t { return f.type == 3 ? u16(f.off + 2 * k) : (f.type == 4 ? u32(f.off + 4 * k) : d[f.off + k]); };
    auto scalar = [&](uint16_t tag, uint32_t dflt) { const Field f = find(tag); return f.ok ? val(f, 0) : dflt; };
    const size_t w = scalar(256, 0), h = scalar(257, 0);
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || w * h > kMaxPixels) return false;
    const uint32_t comp = scalar(259, 1), photo = scalar(262, 0xffff), spp = scalar(277, 1), planar = scalar(284, 1), pred = scalar(317, 1);
    if (find(322).ok || find(324).ok) return false;         // tiles
    if (planar != 1 || spp < 1 || spp > 8 || (pred != 1 && pred != 2)) return false;
    if (w * h * spp > kMaxPixels * 4) return false;         // decoded samples, before the conversion to RGB8
    { const Field b = find(258); if (!b.ok) return false; for (uint32_t k = 0; k < b.count && k < spp; k++) if (val(b, k) != 8) return false; }      // absent = 1 bit per sample
    if (!((photo <= 1 && spp >= 1) || (photo == 2 && spp >= 3) || (photo == 3 && spp >= 1))) return false;
    if (comp != 1 && comp != 5 && comp != 8 && comp != 32946 && comp != 32773) return false;
    const Field so = find(273), sc = find(279);
    if (!so.ok || !sc.ok || so.count != sc.count) return false;
    size_t rps = scalar(278, 0xffffffffu);
    if (rps == 0) return false;
    if (rps > h) rps = h;
    if ((size_t)so.count != (h + rps - 1) / rps) return false;
    uint8_t pal[256][3];
    if (photo == 3) {
        const Field cm = find(320);
        if (!cm.ok || cm.type != 3 || cm.count != 768) return false;
        for (int k = 0; k < 256; k++) for (int c = 0; c < 3; c++) pal[k][c] = (uint8_t)(val(cm, (size_t)c * 256 + k) / 256);
    }
    const size_t row_bytes = w * spp;
    // the file must be able to hold the image: an uncompressed strip does byte for byte, a compressed one at these codecs' best ratios
    rgb.clear();
    std::vector<uint8_t> pix;
    std::vector<uint8_t> strip;
    size_t out_rows = 0;
    for (size_t s = 0; s < so.count; s++) {
        const size_t off = val(so, s), cnt = val(sc, s);
        if (!v.has(off, cnt)) return false;
        const size_t rows = std::min(rps, h - out_rows), want = rows * row_bytes;
        strip.clear();
        if (comp == 1) { if (cnt < want) return false; strip.assign(d.begin() + (long)off, d.begin() + (long)(off + want)); }
        else if (comp == 32773) {                           // PackBits (expands at most 128:1)
            if (cnt < want / 128) return false;
            size_t i = off; const size_t end = off + cnt;
            while (i < end && strip.size() < want) {
                const int8_t n = (int8_t)d[i++];
                if (n >= 0) { const size_t c = (size_t)n + 1; if (c > end - i) return false; strip.insert(strip.end(), d.begin() + (long)i, d.begin() + (long)(i + c)); i += c; }
                else if (n != -128) { if (i >= end) return false; strip.insert(strip.end(), (size_t)(1 - n), d[i++]); }
            }
            if (strip.size() < want) return false;
            strip.resize(want);
        } else if (comp == 5) {
            if (cnt < want / 4096) return false;
            strip.reserve(want);
            if (!lzw_decode(d.data() + off, cnt, 8, true, want, strip) || strip.size() != want) return false;
        } else {                                            // Deflate (zlib stream)
            if (cnt < want / 1032) return false;            // deflate cannot expand more than 1032:1
            strip.resize(want);
            uLongf got = (uLongf)want;
            const int rc = uncompress(strip.data(), &got, d.data() + off, (uLong)cnt);
            if ((rc != Z_OK && rc != Z_BUF_ERROR) || got != want) return false;
        }
        if (pred == 2) for (size_t r = 0; r < rows; r++) { uint8_t* p = &strip[r * row_bytes]; for (size_t k = spp; k < row_bytes; k++) p[k] = (uint8_t)(p[k] + p[k - spp]); }
        // straight into the RGB8 output, strip by strip: the output only ever grows by what a strip really decoded to
        const size_t k0 = rgb.size() / 3, nk = rows * w;
        rgb.resize((k0 + nk) * 3);
        for (size_t k = 0; k < nk; k++) {
            const uint8_t* p = &strip[k * spp];
            uint8_t* o = &rgb[3 * (k0 + k)];
            if (photo == 2) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
            else if (photo == 3) { o[0] = pal[p[0]][0]; o[1] = pal[p[0]][1]; o[2] = pal[p[0]][2]; }
            else { const uint8_t g = photo == 0 ? (uint8_t)(255 - p[0]) : p[0]; o[0] = o[1] = o[2] = g; }
        }
        out_rows += rows;
    }
    if (out_rows != h || rgb.size() != w * h * 3) return false;
    W = (int)w; H = (int)h;
    return true;
}

}  // namespace formats
}  // namespace cs397
