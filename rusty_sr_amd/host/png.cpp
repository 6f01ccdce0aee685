#include "png.hpp"

#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <climits>
#include <exception>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <sched.h>

namespace srpng {
namespace {
// Largest image any decoder here will allocate for: 2^28 px = 1 GiB of RGBA8 (16384 x 16384).  The reference's
// `image` crate has comparable built-in limits; without one a 30-byte header can ask for terabytes.
constexpr uint64_t kMaxPixels = (uint64_t)1 << 28;

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo PNG filtering in place: `rows` scanlines of `stride` bytes, each preceded by its filter byte.  One loop per
// filter type with the first pixel (no left neighbour) peeled off, so the inner loops carry no tests; the bytes-per-pixel
// count is a compile-time constant, so the B independent channel chains of Sub / Average / Paeth are unrolled side by side
// (each byte depends on the byte B positions back: latency-bound when done one after the other).
template <size_t B>
bool unfilter_bpp(uint8_t* d, int rows, size_t stride, std::string& err) {
    const uint8_t* prev = nullptr;
    const size_t head = std::min(B, stride);
    for (int y = 0; y < rows; ++y) {
        uint8_t* line = d + (size_t)y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        switch (ft) {
            case 0: break;
            case 1: {
                size_t i = B;
                for (; i + B <= stride; i += B)
                    for (size_t c = 0; c < B; ++c) cur[i + c] = (uint8_t)(cur[i + c] + cur[i + c - B]);
                for (; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - B]);
                break;
            }
            case 2:
                if (prev) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
                break;
            case 3:
                if (prev) {
                    for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(cur[i] + (prev[i] >> 1));
                    size_t i = B;
                    for (; i + B <= stride; i += B)
                        for (size_t c = 0; c < B; ++c) cur[i + c] = (uint8_t)(cur[i + c] + ((cur[i + c - B] + prev[i + c]) >> 1));
                    for (; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - B] + prev[i]) >> 1));
                } else {
                    for (size_t i = B; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + (cur[i - B] >> 1));
                }
                break;
            case 4:
                if (prev) {
                    for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);  // paeth(0, b, 0) = b
                    auto one = [&](size_t i) {
                        // the predictor as a running minimum (ties keep the earlier of a, b, c: the PNG rule) -- conditional
                        // moves, where the textbook if-chain mispredicts on every other byte of a photograph
                        const int a = cur[i - B], bb = prev[i], c = prev[i - B];
                        const int pa = abs(bb - c), pb = abs(a - c), pc = abs(a + bb - 2 * c);
                        int pred = a, best = pa;
                        pred = pb < best ? bb : pred;
                        best = pb < best ? pb : best;
                        pred = pc < best ? c : pred;
                        cur[i] = (uint8_t)(cur[i] + pred);
                    };
                    size_t i = B;
                    for (; i + B <= stride; i += B)
                        for (size_t c = 0; c < B; ++c) one(i + c);
                    for (; i < stride; ++i) one(i);
                } else {
                    for (size_t i = B; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - B]);  // paeth(a, 0, 0) = a
                }
                break;
            default: err = "bad PNG filter type"; return false;
        }
        prev = cur;
    }
    return true;
}

bool unfilter(uint8_t* d, int rows, size_t stride, int bpp, std::string& err) {
    switch (bpp) {
        case 1: return unfilter_bpp<1>(d, rows, stride, err);
        case 2: return unfilter_bpp<2>(d, rows, stride, err);
        case 3: return unfilter_bpp<3>(d, rows, stride, err);
        case 4: return unfilter_bpp<4>(d, rows, stride, err);
        case 6: return unfilter_bpp<6>(d, rows, stride, err);
        case 8: return unfilter_bpp<8>(d, rows, stride, err);
        default: err = "unsupported PNG pixel size"; return false;
    }
}

struct Hdr { int w, h, depth, ctype, interlace; };

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// expand one unfiltered scanline of `n` pixels to RGBA8 at out[(x0 + k*dx)]
void expand_row(const Hdr& H, const uint8_t* row, int n, const std::vector<uint8_t>& plte,
                const std::vector<uint8_t>& trns, uint8_t* out, int x0, int dx) {
    const int ch = channels_of(H.ctype);
    if (H.depth == 8 && dx == 1 && (H.ctype == 2 || H.ctype == 6)) {  // the common cases, without the per-sample tests
        uint8_t* o = out + (size_t)x0 * 4;
        if (H.ctype == 6) { memcpy(o, row, (size_t)n * 4); return; }
        for (int k = 0; k < n; ++k) { o[4 * k] = row[3 * k]; o[4 * k + 1] = row[3 * k + 1]; o[4 * k + 2] = row[3 * k + 2]; o[4 * k + 3] = 255; }
        return;
    }
    for (int k = 0; k < n; ++k) {
        int s[4] = {0, 0, 0, 255};
        for (int c = 0; c < ch; ++c) {
            int v;
            if (H.depth == 8) v = row[k * ch + c];
            else if (H.depth == 16) v = row[(k * ch + c) * 2];  // high byte
            else {
                const int bit = (k * ch + c) * H.depth;
                v = (row[bit >> 3] >> (8 - H.depth - (bit & 7))) & ((1 << H.depth) - 1);
                if (H.ctype != 3) v = v * 255 / ((1 << H.depth) - 1);
            }
            s[c] = v;
        }
        uint8_t* o = out + (size_t)(x0 + k * dx) * 4;
        switch (H.ctype) {
            case 0: o[0] = o[1] = o[2] = (uint8_t)s[0]; o[3] = 255; break;
            case 2: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; break;
            case 3: {
                const size_t idx = (size_t)s[0];
                o[0] = idx * 3 + 2 < plte.size() ? plte[idx * 3] : 0;
                o[1] = idx * 3 + 2 < plte.size() ? plte[idx * 3 + 1] : 0;
                o[2] = idx * 3 + 2 < plte.size() ? plte[idx * 3 + 2] : 0;
                o[3] = idx < trns.size() ? trns[idx] : 255;
                break;
            }
            case 4: o[0] = o[1] = o[2] = (uint8_t)s[0]; o[3] = (uint8_t)s[1]; break;
            default: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; break;
        }
    }
}

}  // namespace

bool decode_memory(const uint8_t* p, size_t len, Image& out, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (len < 8 || memcmp(p, sig, 8)) { err = "not a PNG file"; return false; }
    Hdr H{0, 0, 0, 0, 0};
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool have_hdr = false, end = false;
    while (!end && pos + 12 <= len) {
        const uint32_t n = be32(p + pos);
        const uint8_t* type = p + pos + 4;
        const uint8_t* data = p + pos + 8;
        if (pos + 12 + (size_t)n > len) { err = "truncated PNG chunk"; return false; }
        if (crc32(crc32(0, type, 4), data, n) != be32(data + n)) { err = "PNG chunk CRC mismatch"; return false; }
        if (!memcmp(type, "IHDR", 4) && n == 13) {
            H = {(int)be32(data), (int)be32(data + 4), data[8], data[9], data[12]};
            have_hdr = true;
            if (data[10] != 0 || data[11] != 0 || H.interlace > 1) { err = "unsupported PNG compression/filter/interlace method"; return false; }
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + n);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + n);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + n);
        else if (!memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)n;
    }
    if (!have_hdr || H.w <= 0 || H.h <= 0 || idat.empty()) { err = "PNG has no image data"; return false; }
    if ((uint64_t)H.w * (uint64_t)H.h > kMaxPixels) { err = "PNG dimensions too large"; return false; }
    const bool depth_ok = H.depth == 8 || H.depth == 16 || ((H.ctype == 0 || H.ctype == 3) && (H.depth == 1 || H.depth == 2 || H.depth == 4));
    if (!depth_ok || (H.ctype != 0 && H.ctype != 2 && H.ctype != 3 && H.ctype != 4 && H.ctype != 6) || (H.ctype == 3 && H.depth == 16)) {
        err = "unsupported PNG colour type / bit depth"; return false;
    }
    if (H.ctype == 3 && (plte.empty() || plte.size() % 3 != 0 || plte.size() > 768)) { err = "PNG palette missing or malformed"; return false; }
    const int bits_pp = channels_of(H.ctype) * H.depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    auto stride_of = [&](int w) { return ((size_t)w * bits_pp + 7) / 8; };
    // pass geometry: non-interlaced = one pass; Adam7 = seven
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dxs[7] = {8, 8, 4, 4, 2, 2, 1}, dys[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = H.interlace ? 7 : 1;
    size_t raw_len = 0;
    int pw[7], ph[7];
    for (int k = 0; k < npass; ++k) {
        pw[k] = H.interlace ? (H.w - xs[k] + dxs[k] - 1) / dxs[k] : H.w;
        ph[k] = H.interlace ? (H.h - ys[k] + dys[k] - 1) / dys[k] : H.h;
        if (pw[k] > 0 && ph[k] > 0) raw_len += (size_t)ph[k] * (stride_of(pw[k]) + 1);
    }
    // deflate cannot expand more than ~1032:1: a header that promises more than the IDAT data can hold is refused
    // BEFORE anything of that size is allocated
    if (raw_len / 1032 > idat.size() + 16) { err = "PNG inflate failed"; return false; }
    std::vector<uint8_t> raw(raw_len);
    uLongf got = (uLongf)raw_len;
    const int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || got != raw_len) { err = "PNG inflate failed"; return false; }
    out.w = H.w; out.h = H.h;
    out.rgba.assign((size_t)H.w * H.h * 4, 0);
    size_t off = 0;
    for (int k = 0; k < npass; ++k) {
        if (pw[k] <= 0 || ph[k] <= 0) continue;
        const size_t st = stride_of(pw[k]);
        if (!unfilter(raw.data() + off, ph[k], st, bpp, err)) return false;
        for (int y = 0; y < ph[k]; ++y) {
            const int oy = H.interlace ? ys[k] + y * dys[k] : y;
            expand_row(H, raw.data() + off + (size_t)y * (st + 1) + 1, pw[k], plte, trns,
                       out.rgba.data() + (size_t)oy * H.w * 4, H.interlace ? xs[k] : 0, H.interlace ? dxs[k] : 1);
        }
        off += (size_t)ph[k] * (st + 1);
    }
    // grey / RGB + tRNS colour key -> alpha 0 for the key (8-bit compare on the stored sample)
    return true;
}

static bool read_all(const std::string& path, std::vector<uint8_t>& buf, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open file"; return false; }
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return true;
}

// PPM / PGM / PBM: binary (P6 / P5 / P4) and plain (P3 / P2 / P1), maxval <= 65535
static bool decode_pnm(const uint8_t* d, size_t len, Image& out, std::string& err) {
    const int kind = d[1] - '0';
    const bool plain = kind <= 3, bitmap = kind == 1 || kind == 4;
    const int ch = (kind == 3 || kind == 6) ? 3 : 1;
    size_t pos = 2;
    auto next_int = [&](long& v) {
        for (;;) {
            if (pos >= len) return false;
            if (d[pos] == '#') { while (pos < len && d[pos] != '\n') ++pos; continue; }
            if (isspace(d[pos])) { ++pos; continue; }
            break;
        }
        if (!isdigit(d[pos])) return false;
        v = 0;
        while (pos < len && isdigit(d[pos]) && v < (1L << 40)) { v = v * 10 + (d[pos] - '0'); ++pos; }
        return true;
    };
    long w = 0, h = 0, maxv = 1;
    if (!next_int(w) || !next_int(h) || (!bitmap && !next_int(maxv))) { err = "bad PNM header"; return false; }
    if (w <= 0 || h <= 0 || w > (1 << 20) || h > (1 << 20) || maxv <= 0 || maxv > 65535) { err = "bad PNM header"; return false; }
    if ((uint64_t)w * (uint64_t)h > kMaxPixels) { err = "PNM dimensions too large"; return false; }
    {   // the samples must be there before the output is allocated: plain files need >= 1 byte per sample (P1) or 2
        const uint64_t samples = (uint64_t)w * h * ch, left = len - std::min(pos, len);
        const uint64_t need = plain ? (bitmap ? samples : samples * 2 - 1) : bitmap ? ((uint64_t)(w + 7) / 8) * h
                                                                                   : samples * (maxv > 255 ? 2 : 1);
        if (need > left) { err = "truncated PNM"; return false; }
    }
    out.w = (int)w; out.h = (int)h;
    out.rgba.assign((size_t)w * h * 4, 255);
    auto put = [&](size_t p, int c, long v) {
        const uint8_t q = (uint8_t)((std::min(v, maxv) * 255 + maxv / 2) / maxv);
        if (ch == 3) out.rgba[p * 4 + c] = q; else out.rgba[p * 4] = out.rgba[p * 4 + 1] = out.rgba[p * 4 + 2] = q;
    };
    if (plain) {
        for (size_t p = 0; p < (size_t)w * h; ++p)
            for (int c = 0; c < ch; ++c) {
                long v;
                if (bitmap) {  // P1: digits need no separators, 1 = black
                    while (pos < len && (isspace(d[pos]) || d[pos] == '#')) { if (d[pos] == '#') while (pos < len && d[pos] != '\n') ++pos; else ++pos; }
                    if (pos >= len || (d[pos] != '0' && d[pos] != '1')) { err = "truncated PNM"; return false; }
                    v = d[pos++] == '0';
                } else if (!next_int(v)) { err = "truncated PNM"; return false; }
                put(p, c, v);
            }
        return true;
    }
    ++pos;  // the single whitespace after the header
    if (bitmap) {
        const size_t stride = ((size_t)w + 7) / 8;
        if (pos + stride * h > len) { err = "truncated PNM"; return false; }
        for (long y = 0; y < h; ++y)
            for (long x = 0; x < w; ++x) put((size_t)y * w + x, 0, !((d[pos + y * stride + x / 8] >> (7 - x % 8)) & 1));
        return true;
    }
    const int bps = maxv > 255 ? 2 : 1;
    if (pos + (size_t)w * h * ch * bps > len) { err = "truncated PNM"; return false; }
    for (size_t p = 0; p < (size_t)w * h; ++p)
        for (int c = 0; c < ch; ++c) {
            const uint8_t* s = d + pos + (p * ch + c) * bps;
            put(p, c, bps == 2 ? (long)(s[0] << 8 | s[1]) : (long)s[0]);
        }
    return true;
}

// uncompressed BMP: 1 / 4 / 8-bit palettised, 24 and 32 bits per pixel (BI_RGB / BI_BITFIELDS with
// the usual BGRA masks); alpha is ignored as the image crate of the reference's era does
static bool decode_bmp(const uint8_t* d, size_t len, Image& out, std::string& err) {
    auto le32 = [&](size_t o) { return (uint32_t)d[o] | d[o + 1] << 8 | d[o + 2] << 16 | (uint32_t)d[o + 3] << 24; };
    if (len < 54) { err = "truncated BMP"; return false; }
    const uint32_t off = le32(10), hdr = le32(14), comp = le32(30);
    const int32_t w = (int32_t)le32(18), hs = (int32_t)le32(22);
    const int bpp = d[28] | d[29] << 8;
    if (hs == INT32_MIN) { err = "unsupported BMP (only uncompressed 1/4/8/24/32-bit)"; return false; }
    const int h = hs < 0 ? -hs : hs;
    const bool pal = bpp == 1 || bpp == 4 || bpp == 8;
    if (hdr < 40 || w <= 0 || h <= 0 || w > (1 << 20) || h > (1 << 20) || !(pal || bpp == 24 || bpp == 32) ||
        (comp != 0 && !(comp == 3 && bpp == 32))) { err = "unsupported BMP (only uncompressed 1/4/8/24/32-bit)"; return false; }
    if ((uint64_t)w * (uint64_t)h > kMaxPixels) { err = "BMP dimensions too large"; return false; }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    uint32_t ncol = pal ? le32(46) : 0;
    if (pal && (ncol == 0 || ncol > (1u << bpp))) ncol = 1u << bpp;
    const size_t pal_off = 14 + (size_t)hdr;
    if ((size_t)off + stride * h > len || pal_off + (size_t)ncol * 4 > len) { err = "truncated BMP"; return false; }
    out.w = w; out.h = h;
    out.rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = d + off + stride * (hs < 0 ? y : h - 1 - y);
        for (int x = 0; x < w; ++x) {
            const uint8_t* px;
            if (pal) {
                const uint32_t idx = bpp == 8 ? row[x] : bpp == 4 ? (row[x / 2] >> ((x & 1) ? 0 : 4)) & 15 : (row[x / 8] >> (7 - x % 8)) & 1;
                if (idx >= ncol) { err = "BMP palette index out of range"; return false; }
                px = d + pal_off + (size_t)idx * 4;
            } else px = row + (size_t)x * bpp / 8;
            uint8_t* o = out.rgba.data() + ((size_t)y * w + x) * 4;
            o[0] = px[2]; o[1] = px[1]; o[2] = px[0]; o[3] = 255;
        }
    }
    return true;
}

bool decode_more_formats(const uint8_t* d, size_t len, bool tga_by_name, Image& out, std::string& err, bool& recognised);  // formats.cpp
static bool decode_any_memory(const std::vector<uint8_t>& buf, Image& out, std::string& err, bool tga_by_name);

bool probe_image_size(const std::string& path, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> b(65536);
    b.resize(fread(b.data(), 1, b.size(), f));
    fseek(f, 0, SEEK_END);
    const long file_size = ftell(f);
    fclose(f);
    const size_t n = b.size();
    long W = 0, Hh = 0;
    if (n >= 24 && b[0] == 0x89 && b[1] == 'P' && !memcmp(b.data() + 12, "IHDR", 4)) {
        W = be32(b.data() + 16); Hh = be32(b.data() + 20);
    } else if (n >= 26 && b[0] == 'B' && b[1] == 'M') {
        const int32_t bw = (int32_t)((uint32_t)b[18] | b[19] << 8 | b[20] << 16 | (uint32_t)b[21] << 24);
        const int32_t bh = (int32_t)((uint32_t)b[22] | b[23] << 8 | b[24] << 16 | (uint32_t)b[25] << 24);
        W = bw; Hh = bh < 0 ? -(long)bh : bh;
    } else if (n >= 7 && b[0] == 'P' && b[1] >= '1' && b[1] <= '6' && isspace(b[2])) {
        size_t pos = 2;
        long v[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            for (;;) {
                if (pos >= n) return false;
                if (b[pos] == '#') { while (pos < n && b[pos] != '\n') ++pos; continue; }
                if (isspace(b[pos])) { ++pos; continue; }
                break;
            }
            if (!isdigit(b[pos])) return false;
            while (pos < n && isdigit(b[pos]) && v[k] < (1L << 40)) { v[k] = v[k] * 10 + (b[pos] - '0'); ++pos; }
        }
        W = v[0]; Hh = v[1];
    } else if (n >= 10 && !memcmp(b.data(), "GIF8", 4)) {
        W = b[6] | b[7] << 8; Hh = b[8] | b[9] << 8;  // the logical screen, which is what the decoder returns
    } else if (n >= 4 && b[0] == 0xff && b[1] == 0xd8) {
        size_t pos = 2;
        while (pos + 9 < n) {  // marker segments up to the frame header (SOF0 / 1 / 2)
            if (b[pos] != 0xff) return false;
            const int m = b[pos + 1];
            if (m == 0xff) { ++pos; continue; }
            const size_t len = (size_t)b[pos + 2] << 8 | b[pos + 3];
            if (m == 0xc0 || m == 0xc1 || m == 0xc2) { Hh = b[pos + 5] << 8 | b[pos + 6]; W = b[pos + 7] << 8 | b[pos + 8]; break; }
            if (m == 0xda || len < 2) return false;
            pos += 2 + len;
        }
    }
    if (W <= 0 || Hh <= 0 || W > (1 << 20) || Hh > (1 << 20) || (uint64_t)W * (uint64_t)Hh > kMaxPixels) return false;
    // the caller sizes buffers with this before the decoder has seen the data: a header that promises more pixels than
    // ~1100 per byte of file (deflate's limit; no real photograph comes near it) is left for the decoder to judge
    if (file_size <= 0 || (uint64_t)W * (uint64_t)Hh > (uint64_t)file_size * 1100) return false;
    w = (int)W; h = (int)Hh;
    return true;
}

bool decode_image_file(const std::string& path, Image& out, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_all(path, buf, err)) return false;
    // TGA has no magic bytes: like the image crate (which goes by the extension for every format) it is taken by name
    bool tga = false;
    {
        const size_t dot = path.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
        for (auto& ch : ext) ch = (char)tolower((unsigned char)ch);
        tga = ext == "tga";
    }
    try {  // backstop: whatever a hostile header makes a container ask for, the caller gets an error, not a terminate()
        return decode_any_memory(buf, out, err, tga);
    } catch (const std::exception& e) {
        err = std::string("image too large or corrupt (") + e.what() + ")";
        out = Image();
        return false;
    }
}

static bool decode_any_memory(const std::vector<uint8_t>& buf, Image& out, std::string& err, bool tga_by_name) {
    if (buf.size() >= 8 && buf[0] == 0x89 && buf[1] == 'P') return decode_memory(buf.data(), buf.size(), out, err);
    if (buf.size() >= 4 && buf[0] == 0xff && buf[1] == 0xd8) return decode_jpeg_memory(buf.data(), buf.size(), out, err);
    if (buf.size() >= 7 && buf[0] == 'P' && buf[1] >= '1' && buf[1] <= '6' && isspace(buf[2])) return decode_pnm(buf.data(), buf.size(), out, err);
    if (buf.size() >= 54 && buf[0] == 'B' && buf[1] == 'M') return decode_bmp(buf.data(), buf.size(), out, err);
    bool recognised = false;
    const bool ok = decode_more_formats(buf.data(), buf.size(), tga_by_name, out, err, recognised);  // GIF, TIFF, ICO, TGA
    if (!recognised) err = "unrecognised image format (this build reads PNG, JPEG, GIF, TIFF, BMP, ICO, TGA, PPM/PGM/PBM)";
    if (!ok) out = Image();
    return ok;
}

bool decode_file(const std::string& path, Image& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open file"; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return decode_memory(buf.data(), buf.size(), out, err);
}

// CPUs this process may really use: the affinity mask, capped by the cgroup quota (a container that shows 256 logical CPUs
// may be allowed 16 -- more threads than that only fight over them).
unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = (atol(quota) + period - 1) / period;
            if (q > 0 && (unsigned)q < n) n = (unsigned)q;
        }
        fclose(f);
    }
    return n ? n : 4;
}

// One scanline, all candidate filters in ONE pass over the pixels (None / Sub / Up / Paeth; 4 bytes per pixel): the
// residuals go to four buffers, the libpng heuristic (smallest sum of |signed residual|) picks one.  Written so that
// the compiler vectorises it -- no branch depends on the data.
__attribute__((target_clones("avx2", "default")))  // picked at load time: the build has no -march
static int filter_row(const uint8_t* cur, const uint8_t* prev, size_t stride, uint8_t* cand /* 4 x stride */) {
    uint8_t* f0 = cand; uint8_t* f1 = cand + stride; uint8_t* f2 = cand + 2 * stride; uint8_t* f4 = cand + 3 * stride;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s4 = 0;
    auto mag = [](uint8_t v) -> uint32_t { return v < 128 ? v : 256u - v; };
    for (size_t i = 0; i < 4 && i < stride; ++i) {  // first pixel: no left neighbour (Paeth degenerates to Up)
        const uint8_t x = cur[i], bb = prev ? prev[i] : 0;
        f0[i] = x; f1[i] = x; f2[i] = (uint8_t)(x - bb); f4[i] = (uint8_t)(x - bb);
        s0 += mag(f0[i]); s1 += mag(f1[i]); s2 += mag(f2[i]); s4 += mag(f4[i]);
    }
    if (prev) {
        for (size_t i = 4; i < stride; ++i) {
            const int x = cur[i], a = cur[i - 4], bb = prev[i], c = prev[i - 4];
            const int pa = bb > c ? bb - c : c - bb, pb = a > c ? a - c : c - a;
            const int t = a + bb - 2 * c, pc = t < 0 ? -t : t;
            const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c);
            const uint8_t r0 = (uint8_t)x, r1 = (uint8_t)(x - a), r2 = (uint8_t)(x - bb), r4 = (uint8_t)(x - pred);
            f0[i] = r0; f1[i] = r1; f2[i] = r2; f4[i] = r4;
            s0 += r0 < 128 ? r0 : 256u - r0; s1 += r1 < 128 ? r1 : 256u - r1;
            s2 += r2 < 128 ? r2 : 256u - r2; s4 += r4 < 128 ? r4 : 256u - r4;
        }
    } else {  // first row of the image: Up = None, Paeth = Sub
        for (size_t i = 4; i < stride; ++i) {
            const uint8_t r0 = cur[i], r1 = (uint8_t)(cur[i] - cur[i - 4]);
            f0[i] = r0; f1[i] = r1; f2[i] = r0; f4[i] = r1;
            s0 += r0 < 128 ? r0 : 256u - r0; s1 += r1 < 128 ? r1 : 256u - r1;
        }
        s2 = s0; s4 = s1;
    }
    int best = 0;
    uint32_t bs = s0;
    if (s1 < bs) { bs = s1; best = 1; }
    if (s2 < bs) { bs = s2; best = 2; }
    if (s4 < bs) { bs = s4; best = 4; }
    return best;
}

// Filter + deflate rows [y0, y1) into one raw-deflate segment that ends on a byte boundary (Z_SYNC_FLUSH) or, for the
// last band, with the final block (Z_FINISH): segments of independent bands concatenate into one valid deflate stream
// (the pigz construction).  The segment comes back wrapped as a complete IDAT chunk -- length, type, data, CRC -- so the
// CRC is computed here too, in parallel; the first band's data starts with the two zlib header bytes.
static bool encode_band(const uint8_t* rgba, int w, int y0, int y1, bool first, bool last, int zlevel,
                        uint8_t* raw /* (stride + 1) * rows */, uint8_t* cand /* 4 * stride */, uint8_t* chunk, size_t cap, size_t& chunk_len,
                        uLong& adler, size_t& raw_len) {
    const size_t stride = (size_t)w * 4;
    for (int y = y0; y < y1; ++y) {
        const uint8_t* cur = rgba + (size_t)y * stride;
        const uint8_t* prev = y ? cur - stride : nullptr;  // the row above, also across band seams
        uint8_t* dst = raw + (size_t)(y - y0) * (stride + 1);
        const int ft = filter_row(cur, prev, stride, cand);
        dst[0] = (uint8_t)ft;
        memcpy(dst + 1, cand + (ft == 4 ? 3 : ft) * stride, stride);
    }
    raw_len = (stride + 1) * (size_t)(y1 - y0);
    if (raw_len > 0xfffffff0u) return false;  // zlib's uInt counters: never truncate silently
    adler = adler32(adler32(0L, Z_NULL, 0), raw, (uInt)raw_len);
    // Run-length matches + dynamic Huffman (rle_deflate.cpp; the choices of zlib's Z_RLE strategy).  On filtered residuals
    // of continuous-tone images longer LZ77 matches are rare and distort the literal statistics: measured on upscaled
    // photographs and on the reference's cartoon output such files are 3-20 % SMALLER than zlib levels 1-6.
    // zlevel <= 0 -- or a band the fast coder cannot fit -- takes zlib's general matcher (level 3): flat synthetic
    // images with repeating patterns want it.
    const size_t head = 8 + (first ? 2 : 0);
    if (cap < head + 64) return false;
    size_t n = 0;
    bool ok = false;
    if (zlevel > 0) {
        const size_t got = rle_deflate(raw, raw_len, last, chunk + head, cap - head - 4);
        if (got) { n = head - 8 + got; ok = true; }
    }
    if (!ok) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 3, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        if (cap < head + deflateBound(&zs, (uLong)raw_len) + 16 + 4) { deflateEnd(&zs); return false; }
        zs.next_in = raw; zs.avail_in = (uInt)raw_len;
        zs.next_out = chunk + head; zs.avail_out = (uInt)(cap - head - 4);
        const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
        ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
        n = head - 8 + (cap - head - 4 - zs.avail_out);  // chunk data bytes
        deflateEnd(&zs);
    }
    if (!ok || n > 0x7fffffffu) return false;
    chunk[0] = (uint8_t)(n >> 24); chunk[1] = (uint8_t)(n >> 16); chunk[2] = (uint8_t)(n >> 8); chunk[3] = (uint8_t)n;
    memcpy(chunk + 4, "IDAT", 4);
    if (first) { chunk[8] = 0x78; chunk[9] = 0x5e; }  // zlib header: deflate, 32 KB window, check bits
    const uint32_t crc = (uint32_t)crc32(0, chunk + 4, (uInt)(n + 4));
    uint8_t* t = chunk + 8 + n;
    t[0] = (uint8_t)(crc >> 24); t[1] = (uint8_t)(crc >> 16); t[2] = (uint8_t)(crc >> 8); t[3] = (uint8_t)crc;
    chunk_len = 8 + n + 4;
    return true;
}

bool encode_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err, int zlevel) {
    if (w <= 0 || h <= 0 || !rgba) { err = "empty image"; return false; }
    // At 4K-in the RGBA output is 299 MB: a single-threaded filter + deflate would dwarf the GPU time (SURVEY.md 8(f)
    // item 2).  Row bands of 1-2 MB are filtered, deflated and wrapped as IDAT chunks by a pool of worker threads that
    // take them in order; this thread writes each chunk as soon as all before it are out, so compression, CRC and the
    // file write overlap and nothing is concatenated in memory.
    const size_t stride = (size_t)w * 4, bytes = stride * h;
    if (zlevel < 0) zlevel = 1;  // run-length + Huffman (encode_band)
    // bands of 1-2 MB, their number a multiple of the worker count so that the last round of bands is a full one
    const size_t cpus = usable_cpus();
    const size_t per_thread = (bytes / cpus + ((size_t)3 << 19) - 1) / ((size_t)3 << 19);  // bands per worker at ~1.5 MB each
    const size_t target = std::max<size_t>(1, std::min<size_t>((size_t)h, cpus * std::max<size_t>(1, per_thread)));
    const int rows_per = bytes <= ((size_t)1 << 20) ? h : (int)(((size_t)h + target - 1) / target);
    const int nband = (h + rows_per - 1) / rows_per;
    const int nthr = (int)std::min<size_t>(cpus, (size_t)nband);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create file"; return false; }
    std::vector<uint8_t> head = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](std::vector<uint8_t>& out, const char* type, const uint8_t* d, size_t n) {
        put32(out, (uint32_t)n);
        const size_t s = out.size();
        out.insert(out.end(), type, type + 4);
        if (n) out.insert(out.end(), d, d + n);
        put32(out, (uint32_t)crc32(0, out.data() + s, (uInt)(n + 4)));
    };
    uint8_t ihdr[13];
    const uint32_t W = (uint32_t)w, Hh = (uint32_t)h;
    ihdr[0] = W >> 24; ihdr[1] = W >> 16; ihdr[2] = W >> 8; ihdr[3] = W;
    ihdr[4] = Hh >> 24; ihdr[5] = Hh >> 16; ihdr[6] = Hh >> 8; ihdr[7] = Hh;
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8-bit RGBA like the reference's outputs
    chunk(head, "IHDR", ihdr, 13);
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();

    // One arena for all chunks, each band at its worst-case offset: `new[]` leaves the pages untouched, so only what the
    // compressed data really covers is ever faulted in, and no allocation happens per band (a fresh process -- which the
    // CLI always is -- would send each multi-MB vector through mmap / munmap).  Filter scratch is per worker, allocated once.
    const size_t band_raw = (stride + 1) * (size_t)rows_per;
    const size_t slot = 8 + 2 + std::max((size_t)compressBound((uLong)band_raw), rle_deflate_bound(band_raw)) + 64;
    std::unique_ptr<uint8_t[]> arena(new (std::nothrow) uint8_t[slot * nband]);
    if (!arena) { fclose(f); err = "out of memory"; return false; }
    std::vector<size_t> part_len(nband, 0);
    std::vector<uLong> adl(nband);
    std::vector<size_t> rawlen(nband);
    std::vector<char> state(nband, 0);  // 0 pending, 1 done, 2 failed
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> next{0};
    auto worker = [&] {
        std::unique_ptr<uint8_t[]> raw(new (std::nothrow) uint8_t[(stride + 1) * rows_per]), cand(new (std::nothrow) uint8_t[4 * stride]);
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= nband) return;
            const int y0 = b * rows_per, y1 = std::min(h, y0 + rows_per);
            bool good = false;
            if (raw && cand)
                good = encode_band(rgba, w, y0, y1, b == 0, b == nband - 1, zlevel, raw.get(), cand.get(), arena.get() + slot * b, slot, part_len[b],
                                   adl[b], rawlen[b]);
            { std::lock_guard<std::mutex> lk(mu); state[b] = good ? 1 : 2; }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) th.emplace_back(worker);
    uLong adler = adler32(0L, Z_NULL, 0);
    bool failed = false;
    for (int b = 0; b < nband; ++b) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return state[b] != 0; }); }
        if (state[b] == 2) failed = true;
        if (!failed && ok) {
            ok = fwrite(arena.get() + slot * b, 1, part_len[b], f) == part_len[b];
            adler = adler32_combine(adler, adl[b], (z_off_t)rawlen[b]);
        }
    }
    for (auto& t : th) t.join();
    if (failed) { fclose(f); remove(path.c_str()); err = "deflate failed"; return false; }  // no half-written file stays behind
    // the zlib trailer (Adler-32 of all filtered bytes) is known only now: a last four-byte IDAT chunk (IDAT data
    // concatenates across chunks, PNG spec 11.2.4)
    std::vector<uint8_t> tail;
    const uint8_t ad[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
    chunk(tail, "IDAT", ad, 4);
    chunk(tail, "IEND", nullptr, 0);
    ok = ok && fwrite(tail.data(), 1, tail.size(), f) == tail.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path.c_str()); err = "short write"; }
    return ok;
}

// `.save(path)` by extension (reference main.rs:175)
bool encode_image_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err) {
    std::string ext;
    const size_t dot = path.find_last_of('.');
    if (dot != std::string::npos) ext = path.substr(dot + 1);
    for (auto& ch : ext) ch = (char)tolower((unsigned char)ch);
    if (ext == "png") return encode_file(path, rgba, w, h, err);
    if (ext == "jpg" || ext == "jpeg") return encode_jpeg_file(path, rgba, w, h, err);
    if (ext != "bmp" && ext != "ppm") { err = "unsupported output format '." + ext + "' (this build writes .png, .jpg, .bmp, .ppm)"; return false; }
    if (w <= 0 || h <= 0 || !rgba) { err = "empty image"; return false; }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create file"; return false; }
    bool ok = true;
    if (ext == "ppm") {
        ok = fprintf(f, "P6\n%d %d\n255\n", w, h) > 0;
        std::vector<uint8_t> row((size_t)w * 3);
        for (int y = 0; y < h && ok; ++y) {
            for (int x = 0; x < w; ++x) memcpy(&row[(size_t)x * 3], rgba + ((size_t)y * w + x) * 4, 3);
            ok = fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    } else {  // BMP: BITMAPINFOHEADER, 32 bits per pixel BGRA, bottom-up
        const uint64_t img = (uint64_t)w * h * 4;
        if (img + 54 > 0xffffffffull) { fclose(f); err = "image too large for BMP"; return false; }
        uint8_t hd[54] = {'B', 'M'};
        auto le32 = [&](int o, uint32_t v) { hd[o] = (uint8_t)v; hd[o + 1] = (uint8_t)(v >> 8); hd[o + 2] = (uint8_t)(v >> 16); hd[o + 3] = (uint8_t)(v >> 24); };
        le32(2, (uint32_t)(54 + img)); le32(10, 54); le32(14, 40); le32(18, (uint32_t)w); le32(22, (uint32_t)h);
        hd[26] = 1; hd[28] = 32; le32(34, (uint32_t)img); le32(38, 2835); le32(42, 2835);
        ok = fwrite(hd, 1, 54, f) == 54;
        std::vector<uint8_t> row((size_t)w * 4);
        for (int y = h - 1; y >= 0 && ok; --y) {
            for (int x = 0; x < w; ++x) {
                const uint8_t* p = rgba + ((size_t)y * w + x) * 4;
                uint8_t* o = &row[(size_t)x * 4];
                o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = p[3];
            }
            ok = fwrite(row.data(), 1, row.size(), f) == row.size();
        }
    }
    fclose(f);
    if (!ok) err = "short write";
    return ok;
}

}  // namespace srpng

extern "C" {
int srpng_encode_any_rgba8(const char* path, const uint8_t* rgba, int w, int h) {
    std::string err;
    return srpng::encode_image_file(path, rgba, w, h, err) ? 0 : -1;
}
int srpng_decode_rgba8(const char* path, int* w, int* h, uint8_t** rgba) {
    srpng::Image img; std::string err;
    if (!srpng::decode_file(path, img, err)) return -1;
    *w = img.w; *h = img.h;
    *rgba = (uint8_t*)malloc(img.rgba.size());
    memcpy(*rgba, img.rgba.data(), img.rgba.size());
    return 0;
}
int srpng_encode_rgba8(const char* path, const uint8_t* rgba, int w, int h) {
    std::string err;
    return srpng::encode_file(path, rgba, w, h, err) ? 0 : -1;
}
int srpng_probe_size(const char* path, int* w, int* h) { return srpng::probe_image_size(path, *w, *h) ? 0 : -1; }
void srpng_free(uint8_t* p) { free(p); }
}
