#include "png.hpp"

#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <climits>
#include <exception>
#include <thread>

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

// undo PNG filtering in place: `rows` scanlines of `stride` bytes, each preceded by its filter byte
bool unfilter(uint8_t* d, int rows, size_t stride, int bpp, std::string& err) {
    const uint8_t* prev = nullptr;
    for (int y = 0; y < rows; ++y) {
        uint8_t* line = d + (size_t)y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0,
                      c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
            int add;
            switch (ft) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = b; break;
                case 3: add = (a + b) >> 1; break;
                case 4: add = paeth(a, b, c); break;
                default: err = "bad PNG filter type"; return false;
            }
            cur[i] = (uint8_t)(cur[i] + add);
        }
        prev = cur;
    }
    return true;
}

struct Hdr { int w, h, depth, ctype, interlace; };

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// expand one unfiltered scanline of `n` pixels to RGBA8 at out[(x0 + k*dx)]
void expand_row(const Hdr& H, const uint8_t* row, int n, const std::vector<uint8_t>& plte,
                const std::vector<uint8_t>& trns, uint8_t* out, int x0, int dx) {
    const int ch = channels_of(H.ctype);
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

static bool decode_any_memory(const std::vector<uint8_t>& buf, Image& out, std::string& err);

bool decode_image_file(const std::string& path, Image& out, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_all(path, buf, err)) return false;
    try {  // backstop: whatever a hostile header makes a container ask for, the caller gets an error, not a terminate()
        return decode_any_memory(buf, out, err);
    } catch (const std::exception& e) {
        err = std::string("image too large or corrupt (") + e.what() + ")";
        out = Image();
        return false;
    }
}

static bool decode_any_memory(const std::vector<uint8_t>& buf, Image& out, std::string& err) {
    if (buf.size() >= 8 && buf[0] == 0x89 && buf[1] == 'P') return decode_memory(buf.data(), buf.size(), out, err);
    if (buf.size() >= 4 && buf[0] == 0xff && buf[1] == 0xd8) return decode_jpeg_memory(buf.data(), buf.size(), out, err);
    if (buf.size() >= 7 && buf[0] == 'P' && buf[1] >= '1' && buf[1] <= '6' && isspace(buf[2])) return decode_pnm(buf.data(), buf.size(), out, err);
    if (buf.size() >= 54 && buf[0] == 'B' && buf[1] == 'M') return decode_bmp(buf.data(), buf.size(), out, err);
    err = "unrecognised image format (this build reads PNG, baseline JPEG, PPM/PGM/PBM, BMP)";
    return false;
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

// Filter + deflate rows [y0, y1) into one raw-deflate segment that ends on a byte boundary
// (Z_SYNC_FLUSH) or, for the last band, with the final block (Z_FINISH).  Segments of
// independent bands concatenate into one valid deflate stream (the pigz construction).
static bool encode_band(const uint8_t* rgba, int w, int y0, int y1, bool last, int zlevel,
                        std::vector<uint8_t>& comp, uLong& adler, size_t& raw_len) {
    const size_t stride = (size_t)w * 4;
    std::vector<uint8_t> raw((stride + 1) * (size_t)(y1 - y0)), cand(stride);
    for (int y = y0; y < y1; ++y) {
        const uint8_t* cur = rgba + (size_t)y * stride;
        const uint8_t* prev = y ? cur - stride : nullptr;  // the row above, also across band seams
        // adaptive filter: minimum sum of absolute (signed) residuals among None/Sub/Up/Paeth
        long best = -1;
        uint8_t* dst = raw.data() + (size_t)(y - y0) * (stride + 1);
        for (int ft : {0, 1, 2, 4}) {
            long sum = 0;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= 4 ? cur[i - 4] : 0, b = prev ? prev[i] : 0, c = (prev && i >= 4) ? prev[i - 4] : 0;
                const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : paeth(a, b, c);
                const uint8_t v = (uint8_t)(cur[i] - pred);
                cand[i] = v;
                sum += v < 128 ? v : 256 - v;
            }
            if (best < 0 || sum < best) { best = sum; dst[0] = (uint8_t)ft; memcpy(dst + 1, cand.data(), stride); }
        }
    }
    raw_len = raw.size();
    if (raw.size() > 0xfffffff0u) return false;  // zlib's uInt counters: never truncate silently
    adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, zlevel, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    comp.resize(deflateBound(&zs, (uLong)raw.size()) + 16);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    const bool ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
    comp.resize(comp.size() - zs.avail_out);
    deflateEnd(&zs);
    return ok;
}

bool encode_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err, int zlevel) {
    if (w <= 0 || h <= 0 || !rgba) { err = "empty image"; return false; }
    // Row bands are filtered and deflated in parallel: at 4K-in the RGBA output is 299 MB and a
    // single-threaded deflate would dwarf the GPU time.
    const size_t bytes = (size_t)w * h * 4;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    int nband = (int)std::min<size_t>({(size_t)hw, (size_t)64, bytes / (1u << 20) + 1, (size_t)h});
    std::vector<std::vector<uint8_t>> parts(nband);
    std::vector<uLong> adl(nband);
    std::vector<size_t> rawlen(nband);
    std::vector<char> okv(nband, 0);
    std::vector<std::thread> th;
    for (int b = 0; b < nband; ++b)
        th.emplace_back([&, b] {
            const int y0 = (int)((long long)h * b / nband), y1 = (int)((long long)h * (b + 1) / nband);
            okv[b] = encode_band(rgba, w, y0, y1, b == nband - 1, zlevel, parts[b], adl[b], rawlen[b]);
        });
    for (auto& t : th) t.join();
    std::vector<uint8_t> comp = {0x78, 0x5e};  // zlib header: deflate, 32 KB window, check bits
    uLong adler = adler32(0L, Z_NULL, 0);
    for (int b = 0; b < nband; ++b) {
        if (!okv[b]) { err = "deflate failed"; return false; }
        comp.insert(comp.end(), parts[b].begin(), parts[b].end());
        adler = adler32_combine(adler, adl[b], (z_off_t)rawlen[b]);
    }
    put32(comp, (uint32_t)adler);
    const uLongf clen = (uLongf)comp.size();
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char* type, const uint8_t* d, size_t n) {
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
    chunk("IHDR", ihdr, 13);
    // a PNG chunk length is at most 2^31 - 1: a large stream goes out as several IDAT chunks (the decoder
    // concatenates them, PNG spec 11.2.4)
    const size_t kMaxIdat = (size_t)1 << 30;
    for (size_t off = 0; off < clen || off == 0; off += kMaxIdat) {
        chunk("IDAT", comp.data() + off, std::min(kMaxIdat, (size_t)clen - off));
        if (clen == 0) break;
    }
    chunk("IEND", nullptr, 0);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create file"; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write";
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
void srpng_free(uint8_t* p) { free(p); }
}
