// formats.cpp -- the other containers `image::open` reads in the reference (main.rs:164; the image crate picks the
// decoder from the file name's extension, this build from the magic bytes, TGA alone by extension because it has none):
// GIF (first frame on the logical screen), TIFF (strips; uncompressed / PackBits / LZW / deflate; 8- and 16-bit grey,
// palette, RGB, RGBA; horizontal predictor), TGA (types 1 2 3 9 10 11) and ICO (largest entry, PNG or DIB inside).
// Everything becomes RGBA8 like the rest of png.hpp; alpha is dropped later by img_to_data (main.rs:170), so how each
// container's transparency is treated never reaches the network.  Parsers of untrusted bytes: every offset and count is
// checked against the buffer before use, sizes are capped before allocation (tests/test_decoder_robustness.py runs them
// under ASAN + UBSan).
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "png.hpp"

namespace srpng {
namespace {

constexpr uint64_t kMaxPixels = (uint64_t)1 << 28;  // as png.cpp

bool size_ok(long w, long h) { return w > 0 && h > 0 && w <= (1 << 20) && h <= (1 << 20) && (uint64_t)w * (uint64_t)h <= kMaxPixels; }

// ------------------------------------------------------------------------------------------------ GIF
// LZW as GIF packs it: codes LSB-first, width min+1 .. 12 bits, clear and end codes, data in sub-blocks
bool gif_lzw(const std::vector<uint8_t>& data, int min_bits, std::vector<uint8_t>& out, size_t want) {
    if (min_bits < 2 || min_bits > 8) return false;
    const int clear = 1 << min_bits, eoi = clear + 1;
    int width = min_bits + 1, next = eoi + 1, prev = -1;
    uint16_t prefix[4096];
    uint8_t suffix[4096], first[4096], stack[4097];
    for (int i = 0; i < clear; ++i) { prefix[i] = 0xffff; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; }
    uint32_t acc = 0;
    int nbits = 0;
    size_t pos = 0;
    out.clear();
    out.reserve(want);
    while (out.size() < want) {
        while (nbits < width) {
            if (pos >= data.size()) return out.size() == want;
            acc |= (uint32_t)data[pos++] << nbits;
            nbits += 8;
        }
        const int code = (int)(acc & ((1u << width) - 1));
        acc >>= width; nbits -= width;
        if (code == clear) { width = min_bits + 1; next = eoi + 1; prev = -1; continue; }
        if (code == eoi) break;
        int cur = code, sp = 0;
        if (code >= next) {  // the one code that may be ahead of the table: prev + first(prev)
            if (code != next || prev < 0) return false;
            stack[sp++] = first[prev];
            cur = prev;
        }
        while (cur >= clear) {
            if (cur >= 4096 || sp >= 4096) return false;
            stack[sp++] = suffix[cur];
            cur = prefix[cur];
        }
        if (cur < 0) return false;
        stack[sp++] = (uint8_t)cur;
        const uint8_t f = (uint8_t)cur;
        while (sp > 0 && out.size() < want) out.push_back(stack[--sp]);
        if (prev >= 0 && next < 4096) {
            prefix[next] = (uint16_t)prev; suffix[next] = f; first[next] = first[prev];
            ++next;
            if (next == (1 << width) && width < 12) ++width;
        }
        prev = code;
    }
    return true;
}

bool decode_gif(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 13) { err = "truncated GIF"; return false; }
    const int sw = d[6] | d[7] << 8, sh = d[8] | d[9] << 8;
    if (!size_ok(sw, sh)) { err = "bad GIF screen size"; return false; }
    size_t pos = 13;
    const uint8_t* gct = nullptr;
    int gct_n = 0;
    if (d[10] & 0x80) {
        gct_n = 2 << (d[10] & 7);
        if (pos + (size_t)gct_n * 3 > len) { err = "truncated GIF"; return false; }
        gct = d + pos;
        pos += (size_t)gct_n * 3;
    }
    int transparent = -1;
    for (;;) {
        if (pos >= len) { err = "GIF has no image"; return false; }
        const uint8_t b = d[pos++];
        if (b == 0x3b) { err = "GIF has no image"; return false; }
        if (b == 0x21) {  // extension: label, then sub-blocks
            if (pos >= len) { err = "truncated GIF"; return false; }
            const uint8_t label = d[pos++];
            bool first_block = true;
            for (;;) {
                if (pos >= len) { err = "truncated GIF"; return false; }
                const size_t n = d[pos++];
                if (n == 0) break;
                if (pos + n > len) { err = "truncated GIF"; return false; }
                if (label == 0xf9 && first_block && n >= 4 && (d[pos] & 1)) transparent = d[pos + 3];
                first_block = false;
                pos += n;
            }
            continue;
        }
        if (b != 0x2c) { err = "bad GIF block"; return false; }
        if (pos + 9 > len) { err = "truncated GIF"; return false; }
        const int fx = d[pos] | d[pos + 1] << 8, fy = d[pos + 2] | d[pos + 3] << 8;
        const int fw = d[pos + 4] | d[pos + 5] << 8, fh = d[pos + 6] | d[pos + 7] << 8;
        const uint8_t flags = d[pos + 8];
        pos += 9;
        if (fw <= 0 || fh <= 0 || fx + fw > sw || fy + fh > sh) { err = "GIF frame outside the screen"; return false; }
        // a few bytes must not buy a gigabyte: the screen may exceed the first frame, but not by orders of magnitude
        // (16 M pixels = 64 MB of RGBA are allowed whatever the frame; beyond that the screen may be at most 4x the frame, whose
        // own pixels the LZW stream has to pay for)
        if ((uint64_t)sw * sh > ((uint64_t)1 << 24) && (uint64_t)sw * sh > (uint64_t)fw * fh * 4) { err = "GIF screen far larger than its frame"; return false; }
        const uint8_t* ct = gct;
        int ct_n = gct_n;
        if (flags & 0x80) {
            ct_n = 2 << (flags & 7);
            if (pos + (size_t)ct_n * 3 > len) { err = "truncated GIF"; return false; }
            ct = d + pos;
            pos += (size_t)ct_n * 3;
        }
        if (!ct) { err = "GIF has no colour table"; return false; }
        if (pos >= len) { err = "truncated GIF"; return false; }
        const int min_bits = d[pos++];
        std::vector<uint8_t> data;
        for (;;) {
            if (pos >= len) { err = "truncated GIF"; return false; }
            const size_t n = d[pos++];
            if (n == 0) break;
            if (pos + n > len) { err = "truncated GIF"; return false; }
            data.insert(data.end(), d + pos, d + pos + n);
            pos += n;
        }
        std::vector<uint8_t> idx;
        const size_t want = (size_t)fw * fh;
        if (want / 4096 > data.size() + 1) { err = "GIF data too short"; return false; }  // LZW cannot expand further
        if (!gif_lzw(data, min_bits, idx, want) || idx.size() != want) { err = "bad GIF LZW data"; return false; }
        out.w = sw; out.h = sh;
        out.rgba.assign((size_t)sw * sh * 4, 0);  // uncovered screen: transparent black
        static const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
        int row = 0;
        for (int pass = 0; pass < ((flags & 0x40) ? 4 : 1); ++pass)
            for (int y = (flags & 0x40) ? start[pass] : 0; y < fh; y += (flags & 0x40) ? step[pass] : 1, ++row) {
                const uint8_t* src = idx.data() + (size_t)row * fw;
                uint8_t* dst = out.rgba.data() + ((size_t)(fy + y) * sw + fx) * 4;
                for (int x = 0; x < fw; ++x) {
                    const int i = src[x];
                    if (i >= ct_n) { err = "GIF colour index out of range"; return false; }
                    dst[4 * x] = ct[3 * i]; dst[4 * x + 1] = ct[3 * i + 1]; dst[4 * x + 2] = ct[3 * i + 2];
                    dst[4 * x + 3] = i == transparent ? 0 : 255;
                }
            }
        return true;
    }
}

// ------------------------------------------------------------------------------------------------ TIFF
struct TiffReader {
    const uint8_t* d; size_t len; bool be;
    bool ok(size_t off, size_t n) const { return off <= len && n <= len - off; }
    uint32_t u16(size_t o) const { return be ? (uint32_t)d[o] << 8 | d[o + 1] : (uint32_t)d[o + 1] << 8 | d[o]; }
    uint32_t u32(size_t o) const {
        return be ? (uint32_t)d[o] << 24 | (uint32_t)d[o + 1] << 16 | (uint32_t)d[o + 2] << 8 | d[o + 3]
                  : (uint32_t)d[o + 3] << 24 | (uint32_t)d[o + 2] << 16 | (uint32_t)d[o + 1] << 8 | d[o];
    }
};

// values of one IFD entry (SHORT or LONG), at most `cap` of them
bool tiff_values(const TiffReader& r, size_t entry, std::vector<uint32_t>& v, size_t cap) {
    const uint32_t type = r.u16(entry + 2), count = r.u32(entry + 4);
    const size_t sz = type == 3 ? 2 : type == 4 ? 4 : type == 1 ? 1 : 0;
    if (!sz || count == 0 || count > cap) return false;
    size_t off = entry + 8;
    if ((size_t)count * sz > 4) {
        off = r.u32(entry + 8);
        if (!r.ok(off, (size_t)count * sz)) return false;
    }
    v.resize(count);
    for (uint32_t i = 0; i < count; ++i) v[i] = sz == 2 ? r.u16(off + 2 * i) : sz == 4 ? r.u32(off + 4 * i) : r.d[off + i];
    return true;
}

// TIFF LZW: codes MSB-first, 9 .. 12 bits, "early change" (the width grows one code early), clear = 256, end = 257
bool tiff_lzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t want) {
    uint16_t prefix[4096];
    uint8_t suffix[4096], first[4096], stack[4097];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0xffff; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; }
    int width = 9, next = 258, prev = -1;
    uint32_t acc = 0;
    int nbits = 0;
    size_t pos = 0;
    const size_t start = out.size();
    while (out.size() - start < want) {
        while (nbits < width) {
            if (pos >= n) return true;  // short strip: the caller checks the count
            acc = acc << 8 | src[pos++];
            nbits += 8;
        }
        const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 256) { width = 9; next = 258; prev = -1; continue; }
        if (code == 257) break;
        int cur = code, sp = 0;
        if (code >= next) {
            if (code != next || prev < 0) return false;
            stack[sp++] = first[prev];
            cur = prev;
        }
        while (cur >= 256) {
            if (cur >= 4096 || sp >= 4096) return false;
            stack[sp++] = suffix[cur];
            cur = prefix[cur];
        }
        stack[sp++] = (uint8_t)cur;
        const uint8_t f = (uint8_t)cur;
        while (sp > 0 && out.size() - start < want) out.push_back(stack[--sp]);
        if (prev >= 0 && next < 4096) {
            prefix[next] = (uint16_t)prev; suffix[next] = f; first[next] = first[prev];
            ++next;
            if (next + 1 >= (1 << width) && width < 12) ++width;
        }
        prev = code;
    }
    return true;
}

bool tiff_packbits(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t want) {
    const size_t start = out.size();
    size_t pos = 0;
    while (pos < n && out.size() - start < want) {
        const int8_t c = (int8_t)src[pos++];
        if (c >= 0) {
            const size_t m = (size_t)c + 1;
            if (pos + m > n) return false;
            out.insert(out.end(), src + pos, src + pos + std::min(m, want - (out.size() - start)));
            pos += m;
        } else if (c != -128) {
            if (pos >= n) return false;
            out.insert(out.end(), std::min((size_t)(1 - c), want - (out.size() - start)), src[pos++]);
        }
    }
    return true;
}

bool decode_tiff(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 8) { err = "truncated TIFF"; return false; }
    TiffReader r{d, len, d[0] == 'M'};
    if (r.u16(2) != 42) { err = "not a TIFF file"; return false; }
    const size_t ifd = r.u32(4);
    if (!r.ok(ifd, 2)) { err = "truncated TIFF"; return false; }
    const size_t n_entries = r.u16(ifd);
    if (!r.ok(ifd + 2, n_entries * 12)) { err = "truncated TIFF"; return false; }
    long w = 0, h = 0;
    uint32_t compression = 1, photometric = 1, spp = 1, rows_per_strip = 0xffffffffu, planar = 1, predictor = 1;
    std::vector<uint32_t> bits = {1}, offsets, counts, cmap, v;
    for (size_t e = 0; e < n_entries; ++e) {
        const size_t at = ifd + 2 + e * 12;
        const uint32_t tag = r.u16(at);
        const bool got = tiff_values(r, at, v, tag == 320 ? 3 * 65536 : tag == 273 || tag == 279 ? (size_t)1 << 20 : 8);
        if (!got) {
            if (tag == 256 || tag == 257 || tag == 258 || tag == 259 || tag == 262 || tag == 273 || tag == 277 || tag == 279 ||
                tag == 284 || tag == 317 || tag == 320) { err = "bad TIFF field"; return false; }
            continue;
        }
        switch (tag) {
            case 256: w = v[0]; break;
            case 257: h = v[0]; break;
            case 258: bits = v; break;
            case 259: compression = v[0]; break;
            case 262: photometric = v[0]; break;
            case 273: offsets = v; break;
            case 277: spp = v[0]; break;
            case 278: rows_per_strip = v[0]; break;
            case 279: counts = v; break;
            case 284: planar = v[0]; break;
            case 317: predictor = v[0]; break;
            case 320: cmap = v; break;
            case 322: case 323: case 324: case 325: err = "tiled TIFF is not supported"; return false;
            default: break;
        }
    }
    if (!size_ok(w, h)) { err = "bad TIFF dimensions"; return false; }
    const uint32_t depth = bits[0];
    for (uint32_t b : bits) if (b != depth) { err = "unsupported TIFF sample layout"; return false; }
    if (spp < 1 || spp > 4 || (bits.size() != spp && !(bits.size() == 1 && spp == 1)) || planar != 1 ||
        (depth != 8 && depth != 16 && !(depth == 1 && spp == 1 && photometric <= 1)) || (predictor != 1 && predictor != 2) ||
        (predictor == 2 && depth == 1)) { err = "unsupported TIFF sample layout"; return false; }
    if (photometric > 3 || (photometric == 2 && spp < 3) || (photometric == 3 && (spp != 1 || depth != 8 || cmap.size() != 768)) ||
        (photometric <= 1 && spp > 2)) { err = "unsupported TIFF photometric interpretation"; return false; }
    if (compression != 1 && compression != 5 && compression != 8 && compression != 32946 && compression != 32773) {
        err = "unsupported TIFF compression"; return false;
    }
    if (offsets.empty() || offsets.size() != counts.size()) { err = "bad TIFF strips"; return false; }
    if (rows_per_strip == 0) { err = "bad TIFF strips"; return false; }
    if (rows_per_strip > (uint32_t)h) rows_per_strip = (uint32_t)h;
    const size_t n_strips = ((size_t)h + rows_per_strip - 1) / rows_per_strip;
    if (offsets.size() < n_strips) { err = "bad TIFF strips"; return false; }
    const size_t bps = depth / 8, row_bytes = depth == 1 ? ((size_t)w + 7) / 8 : (size_t)w * spp * bps;
    uint64_t have = 0;
    for (size_t s = 0; s < n_strips; ++s) {
        if (!r.ok(offsets[s], counts[s])) { err = "truncated TIFF"; return false; }
        have += counts[s];
    }
    // refuse a header the data cannot back BEFORE allocating; the bound is the codec's best case: LZW reaches ~1360:1 on a
    // long run in a single strip (a 4000x4000 solid grey LZW TIFF from libtiff is 13 KB), deflate 1032:1, PackBits 64:1
    const uint64_t max_ratio = compression == 5 ? 1500 : compression == 1 ? 1 : compression == 32773 ? 128 : 1100;
    if ((uint64_t)row_bytes * h / max_ratio > have + 16) { err = "TIFF data too short"; return false; }
    out.w = (int)w; out.h = (int)h;
    out.rgba.assign((size_t)w * h * 4, 255);
    std::vector<uint8_t> strip;
    for (size_t s = 0; s < n_strips; ++s) {
        const size_t y0 = s * rows_per_strip, rows = std::min((size_t)rows_per_strip, (size_t)h - y0), want = rows * row_bytes;
        const uint8_t* src = d + offsets[s];
        const size_t n = counts[s];
        strip.clear();
        bool ok = true;
        if (compression == 1) { if (n < want) ok = false; else strip.assign(src, src + want); }
        else if (compression == 5) ok = tiff_lzw(src, n, strip, want);
        else if (compression == 32773) ok = tiff_packbits(src, n, strip, want);
        else {
            strip.resize(want);
            uLongf got = (uLongf)want;
            const int zr = uncompress(strip.data(), &got, src, (uLong)n);
            ok = (zr == Z_OK || zr == Z_BUF_ERROR) && got == want;
        }
        if (!ok || strip.size() != want) { err = "bad TIFF strip data"; return false; }
        for (size_t y = 0; y < rows; ++y) {
            uint8_t* row = strip.data() + y * row_bytes;
            if (predictor == 2) {  // horizontal differencing, per sample
                if (bps == 1) for (size_t i = spp; i < row_bytes; ++i) row[i] = (uint8_t)(row[i] + row[i - spp]);
                else for (size_t i = spp; i < (size_t)w * spp; ++i) {
                    const size_t p = i * 2, q = (i - spp) * 2;
                    const uint32_t cur = r.be ? (uint32_t)row[p] << 8 | row[p + 1] : (uint32_t)row[p + 1] << 8 | row[p];
                    const uint32_t left = r.be ? (uint32_t)row[q] << 8 | row[q + 1] : (uint32_t)row[q + 1] << 8 | row[q];
                    const uint32_t sum = (cur + left) & 0xffff;
                    if (r.be) { row[p] = (uint8_t)(sum >> 8); row[p + 1] = (uint8_t)sum; } else { row[p + 1] = (uint8_t)(sum >> 8); row[p] = (uint8_t)sum; }
                }
            }
            uint8_t* dst = out.rgba.data() + (y0 + y) * (size_t)w * 4;
            auto sample = [&](size_t x, uint32_t c) -> uint8_t {  // 16-bit samples keep their high byte, like png.cpp
                const size_t o = (x * spp + c) * bps;
                return bps == 1 ? row[o] : (r.be ? row[o] : row[o + 1]);
            };
            for (size_t x = 0; x < (size_t)w; ++x) {
                uint8_t* o = dst + 4 * x;
                if (depth == 1) {
                    const int bit = (row[x / 8] >> (7 - x % 8)) & 1;
                    o[0] = o[1] = o[2] = (bit ^ (photometric == 0)) ? 255 : 0;
                } else if (photometric == 3) {
                    const uint32_t i = row[x];
                    o[0] = (uint8_t)(cmap[i] >> 8); o[1] = (uint8_t)(cmap[256 + i] >> 8); o[2] = (uint8_t)(cmap[512 + i] >> 8);
                } else if (photometric == 2) {
                    o[0] = sample(x, 0); o[1] = sample(x, 1); o[2] = sample(x, 2);
                    if (spp == 4) o[3] = sample(x, 3);
                } else {
                    const uint8_t g = sample(x, 0);
                    o[0] = o[1] = o[2] = photometric == 0 ? (uint8_t)(255 - g) : g;
                    if (spp == 2) o[3] = sample(x, 1);
                }
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ TGA
bool decode_tga(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 18) { err = "truncated TGA"; return false; }
    const int id_len = d[0], cmap_type = d[1], type = d[2];
    const int cm_first = d[3] | d[4] << 8, cm_len = d[5] | d[6] << 8, cm_bits = d[7];
    const int w = d[12] | d[13] << 8, h = d[14] | d[15] << 8, bpp = d[16], desc = d[17];
    const bool rle = type >= 9;
    const int base = rle ? type - 8 : type;
    if (base < 1 || base > 3 || cmap_type > 1 || !size_ok(w, h)) { err = "unsupported TGA"; return false; }
    if ((base == 1 && (cmap_type != 1 || bpp != 8 || (cm_bits != 15 && cm_bits != 16 && cm_bits != 24 && cm_bits != 32))) ||
        (base == 2 && bpp != 15 && bpp != 16 && bpp != 24 && bpp != 32) || (base == 3 && bpp != 8)) { err = "unsupported TGA"; return false; }
    size_t pos = 18 + (size_t)id_len;
    const size_t cm_bytes = cmap_type ? (size_t)cm_len * ((cm_bits + 7) / 8) : 0;
    if (pos + cm_bytes > len) { err = "truncated TGA"; return false; }
    const uint8_t* cm = d + pos;
    pos += cm_bytes;
    const size_t px_bytes = (size_t)(bpp + 7) / 8, npx = (size_t)w * h;
    if (!rle && pos + npx * px_bytes > len) { err = "truncated TGA"; return false; }
    if (rle && npx / 128 > len - pos + 1) { err = "truncated TGA"; return false; }  // a packet covers at most 128 pixels
    auto colour = [&](const uint8_t* p, int bits, uint8_t* o) {
        if (bits == 8) { o[0] = o[1] = o[2] = p[0]; o[3] = 255; }
        else if (bits == 15 || bits == 16) {
            const int v = p[0] | p[1] << 8;
            o[0] = (uint8_t)(((v >> 10) & 31) * 255 / 31); o[1] = (uint8_t)(((v >> 5) & 31) * 255 / 31); o[2] = (uint8_t)((v & 31) * 255 / 31); o[3] = 255;
        } else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = bits == 32 ? p[3] : 255; }
    };
    out.w = w; out.h = h;
    out.rgba.assign(npx * 4, 255);
    auto put = [&](size_t i, const uint8_t* p) -> bool {
        const size_t y = i / w, x = i % w;
        const size_t oy = (desc & 0x20) ? y : (size_t)h - 1 - y, ox = (desc & 0x10) ? (size_t)w - 1 - x : x;
        uint8_t* o = out.rgba.data() + (oy * w + ox) * 4;
        if (base == 1) {
            const int idx = p[0] - cm_first;
            if (idx < 0 || idx >= cm_len) return false;
            colour(cm + (size_t)idx * ((cm_bits + 7) / 8), cm_bits, o);
        } else colour(p, base == 3 ? 8 : bpp, o);
        return true;
    };
    size_t i = 0;
    if (!rle) {
        for (; i < npx; ++i) if (!put(i, d + pos + i * px_bytes)) { err = "TGA colour index out of range"; return false; }
        return true;
    }
    while (i < npx) {
        if (pos >= len) { err = "truncated TGA"; return false; }
        const int c = d[pos++], run = (c & 127) + 1;
        if (c & 128) {
            if (pos + px_bytes > len) { err = "truncated TGA"; return false; }
            for (int k = 0; k < run && i < npx; ++k, ++i) if (!put(i, d + pos)) { err = "TGA colour index out of range"; return false; }
            pos += px_bytes;
        } else {
            if (pos + (size_t)run * px_bytes > len) { err = "truncated TGA"; return false; }
            for (int k = 0; k < run && i < npx; ++k, ++i) if (!put(i, d + pos + (size_t)k * px_bytes)) { err = "TGA colour index out of range"; return false; }
            pos += (size_t)run * px_bytes;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ ICO
bool decode_ico(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 6) { err = "truncated ICO"; return false; }
    const int count = d[4] | d[5] << 8;
    if (count <= 0 || 6 + (size_t)count * 16 > len) { err = "truncated ICO"; return false; }
    size_t best = 0;
    long best_area = -1;
    for (int k = 0; k < count; ++k) {  // the largest entry, then the deepest
        const uint8_t* e = d + 6 + (size_t)k * 16;
        const long ew = e[0] ? e[0] : 256, eh = e[1] ? e[1] : 256, bits = e[6] | e[7] << 8;
        const long score = ew * eh * 64 + bits;
        if (score > best_area) { best_area = score; best = (size_t)k; }
    }
    const uint8_t* e = d + 6 + best * 16;
    const size_t size = (uint32_t)e[8] | e[9] << 8 | e[10] << 16 | (uint32_t)e[11] << 24;
    const size_t off = (uint32_t)e[12] | e[13] << 8 | e[14] << 16 | (uint32_t)e[15] << 24;
    if (off > len || size > len - off || size < 40) { err = "truncated ICO"; return false; }
    const uint8_t* p = d + off;
    if (size >= 8 && p[0] == 0x89 && p[1] == 'P') return decode_memory(p, size, out, err);
    // a DIB without file header whose height counts the XOR image and the 1-bit AND mask: rebuild a BMP around the image
    auto le32 = [&](size_t o) { return (uint32_t)p[o] | p[o + 1] << 8 | p[o + 2] << 16 | (uint32_t)p[o + 3] << 24; };
    const uint32_t hdr = le32(0);
    const int32_t w = (int32_t)le32(4), h2 = (int32_t)le32(8);
    const int bpp = p[14] | p[15] << 8;
    if (hdr < 40 || hdr > size || w <= 0 || h2 <= 1 || w > 4096 || h2 > 8192 || le32(16) != 0 ||
        !(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32)) { err = "unsupported ICO image"; return false; }
    const int h = h2 / 2;
    uint32_t ncol = bpp <= 8 ? le32(32) : 0;
    if (bpp <= 8 && (ncol == 0 || ncol > (1u << bpp))) ncol = 1u << bpp;
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4, pal = (size_t)hdr, pix = pal + (size_t)ncol * 4;
    if (pix + stride * h > size) { err = "truncated ICO"; return false; }
    const size_t mask_stride = (((size_t)w + 31) / 32) * 4, mask = pix + stride * h;
    const bool have_mask = mask + mask_stride * h <= size;
    out.w = w; out.h = h;
    out.rgba.assign((size_t)w * h * 4, 255);
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = p + pix + stride * (size_t)(h - 1 - y);
        const uint8_t* mrow = have_mask ? p + mask + mask_stride * (size_t)(h - 1 - y) : nullptr;
        for (int x = 0; x < w; ++x) {
            uint8_t* o = out.rgba.data() + ((size_t)y * w + x) * 4;
            if (bpp <= 8) {
                const uint32_t idx = bpp == 8 ? row[x] : bpp == 4 ? (row[x / 2] >> ((x & 1) ? 0 : 4)) & 15 : (row[x / 8] >> (7 - x % 8)) & 1;
                if (idx >= ncol) { err = "ICO palette index out of range"; return false; }
                const uint8_t* c = p + pal + (size_t)idx * 4;
                o[0] = c[2]; o[1] = c[1]; o[2] = c[0];
            } else {
                const uint8_t* c = row + (size_t)x * bpp / 8;
                o[0] = c[2]; o[1] = c[1]; o[2] = c[0];
                if (bpp == 32) o[3] = c[3];
            }
            if (mrow && bpp != 32 && ((mrow[x / 8] >> (7 - x % 8)) & 1)) o[3] = 0;
        }
    }
    return true;
}

}  // namespace

bool decode_more_formats(const uint8_t* d, size_t len, bool tga_by_name, Image& out, std::string& err, bool& recognised) {
    recognised = true;
    if (len >= 6 && !memcmp(d, "GIF8", 4) && (d[4] == '7' || d[4] == '9') && d[5] == 'a') return decode_gif(d, len, out, err);
    if (len >= 4 && ((d[0] == 'I' && d[1] == 'I' && d[2] == 42 && d[3] == 0) || (d[0] == 'M' && d[1] == 'M' && d[2] == 0 && d[3] == 42)))
        return decode_tiff(d, len, out, err);
    if (len >= 6 && d[0] == 0 && d[1] == 0 && d[2] == 1 && d[3] == 0 && !tga_by_name) return decode_ico(d, len, out, err);
    if (tga_by_name) return decode_tga(d, len, out, err);
    recognised = false;
    return false;
}

}  // namespace srpng
