#include "png.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>

namespace srpng {
namespace {

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
    const bool depth_ok = H.depth == 8 || H.depth == 16 || ((H.ctype == 0 || H.ctype == 3) && (H.depth == 1 || H.depth == 2 || H.depth == 4));
    if (!depth_ok || (H.ctype != 0 && H.ctype != 2 && H.ctype != 3 && H.ctype != 4 && H.ctype != 6) || (H.ctype == 3 && H.depth == 16)) {
        err = "unsupported PNG colour type / bit depth"; return false;
    }
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
    chunk("IDAT", comp.data(), clen);
    chunk("IEND", nullptr, 0);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create file"; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write";
    return ok;
}

}  // namespace srpng

extern "C" {
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
