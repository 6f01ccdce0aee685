// jpeg.cpp -- baseline (sequential Huffman, 8-bit) JPEG decoder for the rusty_sr host CLI.
// The reference opens its input with image::open (main.rs:164), which also takes JPEG; this is
// the stand-in.  Supports greyscale and YCbCr (JFIF) / RGB (Adobe transform 0) with any sampling
// factors (4:4:4, 4:2:2, 4:2:0, 4:1:1 ...), restart intervals, 8-bit precision.  Progressive,
// arithmetic-coded, lossless and 12-bit files are rejected with a message.  Chroma is upsampled
// with libjpeg's triangle filter for 2:1 ratios and the IDCT is a plain float separable transform,
// so pixels can differ from libjpeg's by a few levels -- as they do between any two JPEG decoders.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "png.hpp"

namespace srpng {
namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; int pw = 0, ph = 0; };

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t buf = 0; int cnt = 0; bool hit_marker = false;
    void fill() {
        while (cnt <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed byte
                    else { hit_marker = true; b = 0; }                   // a marker: feed zeros
                } else ++p;
            }
            buf |= (uint32_t)b << (24 - cnt);
            cnt += 8;
        }
    }
    int bit() { if (cnt == 0) fill(); const int r = buf >> 31; buf <<= 1; --cnt; return r; }
    int bits(int n) { int r = 0; for (int i = 0; i < n; ++i) r = (r << 1) | bit(); return r; }
    void reset() { buf = 0; cnt = 0; hit_marker = false; }
};

int decode_sym(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}
int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float* in, uint8_t* out, int stride) {
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            float s = 0;
            for (int u = 0; u < 8; ++u) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            float s = 0;
            for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
            const int q = (int)std::lround(s + 128.0f);
            out[y * stride + x] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
        }
}

uint8_t clamp8(float v) { const int q = (int)std::lround(v); return (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q); }

}  // namespace

bool decode_jpeg_memory(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 4 || d[0] != 0xff || d[1] != 0xd8) { err = "not a JPEG file"; return false; }
    uint16_t qt[4][64] = {};
    Huff dc[4], ac[4];
    Comp comp[4];
    int ncomp = 0, W = 0, H = 0, restart = 0, adobe_transform = -1;
    bool have_sof = false;
    size_t pos = 2;
    while (pos + 4 <= len) {
        if (d[pos] != 0xff) { ++pos; continue; }
        const int m = d[pos + 1];
        if (m == 0xff) { ++pos; continue; }
        pos += 2;
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
        if (m == 0xd9) break;
        if (pos + 2 > len) break;
        const size_t seglen = (size_t)d[pos] << 8 | d[pos + 1];
        if (seglen < 2 || pos + seglen > len) { err = "truncated JPEG segment"; return false; }
        const uint8_t* s = d + pos + 2;
        const size_t n = seglen - 2;
        if (m == 0xdb) {  // DQT
            size_t k = 0;
            while (k < n) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || k + (pq ? 128 : 64) > n) { err = "bad DQT"; return false; }
                for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (uint16_t)(s[k] << 8 | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
            }
        } else if (m == 0xc4) {  // DHT
            size_t k = 0;
            while (k + 17 <= n) {
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (th > 3 || tc > 1) { err = "bad DHT"; return false; }
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[k + l]; total += s[k + l]; }
                k += 17;
                if (total > 256 || k + total > n) { err = "bad DHT"; return false; }
                memcpy(h.vals, s + k, total);
                k += total;
                h.build();
            }
        } else if (m == 0xc0 || m == 0xc1) {  // baseline / extended sequential Huffman
            if (n < 6 || s[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            H = s[1] << 8 | s[2]; W = s[3] << 8 | s[4]; ncomp = s[5];
            if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3) || n < 6 + (size_t)3 * ncomp) { err = "unsupported JPEG frame"; return false; }
            // an 8x8 block costs at least 2 bits of entropy-coded data per component: a frame header that promises more
            // pixels than the file could possibly hold is refused before its planes are allocated (also caps at 2^28 px)
            if ((uint64_t)W * (uint64_t)H > ((uint64_t)1 << 28) || (uint64_t)W * (uint64_t)H / 256 > (uint64_t)len + 64) { err = "JPEG dimensions do not fit the file"; return false; }
            for (int i = 0; i < ncomp; ++i) { comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i] & 3;
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) { err = "bad JPEG sampling factors"; return false; } }
            have_sof = true;
        } else if (m == 0xc2) { err = "progressive JPEG is not supported by this build (re-save as baseline JPEG or PNG)"; return false; }
        else if ((m >= 0xc3 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc) { err = "unsupported JPEG coding process"; return false; }
        else if (m == 0xdd && n >= 2) restart = s[0] << 8 | s[1];
        else if (m == 0xee && n >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        else if (m == 0xda) {  // SOS: decode the single interleaved scan of a baseline file
            if (!have_sof) { err = "JPEG scan before frame header"; return false; }
            if (n < 1) { err = "bad JPEG scan header"; return false; }
            const int ns = s[0];
            if (ns != ncomp || n < 1 + (size_t)2 * ns + 3) { err = "multi-scan baseline JPEG is not supported"; return false; }
            // T.81 A.2.2: a scan with ONE component is not interleaved -- one 8x8 data unit per MCU in raster order over
            // ceil(W/8) x ceil(H/8), whatever sampling factors the frame header declares for it
            if (ncomp == 1) comp[0].h = comp[0].v = 1;
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == s[1 + 2 * i]) ci = j;
                if (ci < 0) { err = "bad JPEG scan header"; return false; }
                comp[ci].td = s[2 + 2 * i] >> 4; comp[ci].ta = s[2 + 2 * i] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3 || !dc[comp[ci].td].present || !ac[comp[ci].ta].present) { err = "missing JPEG Huffman table"; return false; }
            }
            int hmax = 1, vmax = 1;
            for (int i = 0; i < ncomp; ++i) { hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax; }
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int i = 0; i < ncomp; ++i) {
                comp[i].pw = mcux * comp[i].h * 8; comp[i].ph = mcuy * comp[i].v * 8;
                comp[i].plane.assign((size_t)comp[i].pw * comp[i].ph, 0);
                comp[i].pred = 0;
            }
            BitReader br{d + pos + seglen, d + len};
            int rst_left = restart;
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    if (restart && rst_left == 0) {  // expect RSTn: skip to it, reset predictors
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) ++br.p;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
                        rst_left = restart;
                    }
                    for (int i = 0; i < ncomp; ++i)
                        for (int by = 0; by < comp[i].v; ++by)
                            for (int bx = 0; bx < comp[i].h; ++bx) {
                                float blk[64] = {0};
                                const int t = decode_sym(br, dc[comp[i].td]);
                                if (t < 0 || t > 15) { err = "corrupt JPEG data"; return false; }
                                comp[i].pred += extend(br.bits(t), t);
                                blk[0] = (float)comp[i].pred * qt[comp[i].tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_sym(br, ac[comp[i].ta]);
                                    if (rs < 0) { err = "corrupt JPEG data"; return false; }
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) { err = "corrupt JPEG data"; return false; }
                                    blk[kZigzag[k]] = (float)extend(br.bits(sz), sz) * qt[comp[i].tq][kZigzag[k]];
                                    ++k;
                                }
                                idct8x8(blk, comp[i].plane.data() + (size_t)((my * comp[i].v + by) * 8) * comp[i].pw + (mx * comp[i].h + bx) * 8, comp[i].pw);
                            }
                    if (restart) --rst_left;
                }
            out.w = W; out.h = H;
            out.rgba.assign((size_t)W * H * 4, 255);
            // chroma upsampling: libjpeg's "fancy" triangle filter for the usual 2:1 ratios
            // (3/4 nearest + 1/4 next-nearest sample, per axis), replication otherwise
            std::vector<uint8_t> up[3];
            for (int i = 0; i < ncomp; ++i) {
                const int rx = hmax / comp[i].h, ry = vmax / comp[i].v;
                const int cw = (W * comp[i].h + hmax - 1) / hmax, chh = (H * comp[i].v + vmax - 1) / vmax;  // valid samples
                up[i].resize((size_t)W * H);
                const std::vector<uint8_t>& pl = comp[i].plane;
                const int pw = comp[i].pw;
                auto at = [&](int x, int y) { x = x < 0 ? 0 : x >= cw ? cw - 1 : x; y = y < 0 ? 0 : y >= chh ? chh - 1 : y; return (int)pl[(size_t)y * pw + x]; };
                const bool fx = rx == 2 && hmax % comp[i].h == 0, fy = ry == 2 && vmax % comp[i].v == 0;
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        const int sx = x * comp[i].h / hmax, sy = y * comp[i].v / vmax;
                        int v;
                        if (fx && fy) {
                            const int nx = sx + ((x & 1) ? 1 : -1), ny = sy + ((y & 1) ? 1 : -1);
                            v = (9 * at(sx, sy) + 3 * at(nx, sy) + 3 * at(sx, ny) + at(nx, ny) + 8) >> 4;
                        } else if (fx) {
                            v = (3 * at(sx, sy) + at(sx + ((x & 1) ? 1 : -1), sy) + ((x & 1) ? 2 : 1)) >> 2;
                        } else if (fy) {
                            v = (3 * at(sx, sy) + at(sx, sy + ((y & 1) ? 1 : -1)) + 2) >> 2;
                        } else {
                            v = at(sx, sy);
                        }
                        up[i][(size_t)y * W + x] = (uint8_t)v;
                    }
            }
            const bool ycc = ncomp == 3 && adobe_transform != 0;
            for (size_t p = 0; p < (size_t)W * H; ++p) {
                uint8_t* o = out.rgba.data() + p * 4;
                if (ncomp == 1) { o[0] = o[1] = o[2] = up[0][p]; }
                else if (ycc) {
                    const float Y = up[0][p], cb = up[1][p] - 128.0f, cr = up[2][p] - 128.0f;
                    o[0] = clamp8(Y + 1.402f * cr); o[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr); o[2] = clamp8(Y + 1.772f * cb);
                } else { o[0] = up[0][p]; o[1] = up[1][p]; o[2] = up[2][p]; }
            }
            return true;
        }
        pos += seglen;
    }
    err = "JPEG has no image data";
    return false;
}

}  // namespace srpng

extern "C" int srpng_decode_any_rgba8(const char* path, int* w, int* h, uint8_t** rgba) {
    srpng::Image img; std::string err;
    if (!srpng::decode_image_file(path, img, err)) return -1;
    *w = img.w; *h = img.h;
    *rgba = (uint8_t*)malloc(img.rgba.size());
    memcpy(*rgba, img.rgba.data(), img.rgba.size());
    return 0;
}
