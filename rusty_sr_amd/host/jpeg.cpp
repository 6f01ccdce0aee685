// jpeg.cpp -- JPEG codec for the rusty_sr host CLI: Huffman-coded 8-bit baseline, extended sequential (also multi-scan)
// and PROGRESSIVE (spectral selection + successive approximation) decoding, and a baseline encoder.
// The reference opens its input with image::open (main.rs:164), which also takes JPEG, and writes whatever the output
// extension names (main.rs:175); this is the stand-in.  Supports greyscale and YCbCr (JFIF) / RGB (Adobe transform 0) with
// any sampling factors (4:4:4, 4:2:2, 4:2:0, 4:1:1 ...), restart intervals.  Arithmetic-coded, lossless, hierarchical and
// 12-bit files are rejected with a message.  Chroma is upsampled
// with libjpeg's triangle filter for 2:1 ratios and the IDCT is a plain float separable transform,
// so pixels can differ from libjpeg's by a few levels -- as they do between any two JPEG decoders.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "png.hpp"

namespace srpng {
namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    uint16_t fast[512];  // the next 9 bits -> (length << 8 | symbol) for codes of up to 9 bits, 0 otherwise
    bool present = false;
    void build() {
        int code = 0, k = 0;
        memset(fast, 0, sizeof fast);
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            if (l <= 9)
                for (int j = 0; j < bits[l] && (code + j) < (1 << l); ++j) {
                    const int first = (code + j) << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); ++f) fast[first + f] = (uint16_t)(l << 8 | vals[(k + j) & 255]);
                }
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; int pw = 0, ph = 0;
              std::vector<int16_t> coef; };  // coef: 64 per block, natural order, accumulated over the scans

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
    int bits(int n) {  // n <= 16
        if (n == 0) return 0;
        if (cnt < n) fill();
        const int r = (int)(buf >> (32 - n));
        buf <<= n; cnt -= n;
        return r;
    }
    void reset() { buf = 0; cnt = 0; hit_marker = false; }
};

// One Huffman symbol: the next 9 bits index a table that resolves every code of up to 9 bits (nearly all of them) in one
// step; longer codes fall back to the canonical length-by-length search.  Consumes exactly the bits of the code.
int decode_sym(BitReader& br, const Huff& h) {
    if (br.cnt < 16) br.fill();
    const uint16_t e = h.fast[br.buf >> 23];
    if (e) {
        const int l = e >> 8;
        br.buf <<= l; br.cnt -= l;
        return e & 255;
    }
    const int top = (int)(br.buf >> 16);
    for (int l = 1; l <= 16; ++l) {
        const int code = top >> (16 - l);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
            br.buf <<= l; br.cnt -= l;
            return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
    }
    br.buf <<= 16; br.cnt -= 16;  // no code matches: the 16 bits are spent, as they were bit by bit
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
    // Most coefficients of most blocks are zero, and a zero term adds nothing: rows stop at their last non-zero coefficient,
    // all-zero rows are skipped in the column pass.  Every remaining term is added in the same order: results unchanged.
    float tmp[64];
    int live[8], nlive = 0;
    for (int y = 0; y < 8; ++y) {
        int last = -1;
        for (int u = 0; u < 8; ++u) if (in[y * 8 + u] != 0.0f) last = u;
        if (last < 0) { for (int x = 0; x < 8; ++x) tmp[y * 8 + x] = 0.0f; continue; }
        live[nlive++] = y;
        for (int x = 0; x < 8; ++x) {
            float s = 0;
            for (int u = 0; u <= last; ++u) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            float s = 0;
            for (int k = 0; k < nlive; ++k) s += c[y][live[k]] * tmp[live[k] * 8 + x];
            const float v = s + 128.0f;
            out[y * stride + x] = (uint8_t)(v <= 0.0f ? 0 : v >= 255.0f ? 255 : (int)(v + 0.5f));  // = lround + clamp
        }
}

// round half up on [0, 255] = std::lround there, without the libm call per sample
inline uint8_t clamp8(float v) { return (uint8_t)(v <= 0.0f ? 0 : v >= 255.0f ? 255 : (int)(v + 0.5f)); }


// One entropy-coded scan of any of the supported processes into the coefficient arrays (T.81 F.2 sequential,
// G.2 progressive: spectral selection Ss..Se, successive approximation Ah / Al).
struct Scan {
    int ns = 0, ci[4] = {0, 0, 0, 0}, Ss = 0, Se = 63, Ah = 0, Al = 0;
};

bool decode_scan(BitReader& br, Comp* comp, const Scan& sc, bool progressive, int W, int H, int hmax, int vmax, int mcux, int mcuy,
                 int restart, const Huff* dc, const Huff* ac, std::string& err) {
    int eobrun = 0;
    // one 8x8 block of component c at block coordinates (bx, by) of its (MCU-padded) grid
    auto block = [&](Comp& c, int bx, int by) -> bool {
        int16_t* co = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
        if (!progressive) {
            const int t = decode_sym(br, dc[c.td]);
            if (t < 0 || t > 15) { err = "corrupt JPEG data"; return false; }
            c.pred += extend(br.bits(t), t);
            if (c.pred < -32768 || c.pred > 32767) { err = "corrupt JPEG data"; return false; }  // (libjpeg warns and goes on; a running sum of crafted differences would overflow)
            co[0] = (int16_t)c.pred;
            for (int k = 1; k < 64;) {
                const int rs = decode_sym(br, ac[c.ta]);
                if (rs < 0) { err = "corrupt JPEG data"; return false; }
                const int r = rs >> 4, sz = rs & 15;
                if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                k += r;
                if (k > 63) { err = "corrupt JPEG data"; return false; }
                co[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
                ++k;
            }
            return true;
        }
        if (sc.Ss == 0) {  // DC scan
            if (sc.Ah == 0) {
                const int t = decode_sym(br, dc[c.td]);
                if (t < 0 || t > 15) { err = "corrupt JPEG data"; return false; }
                c.pred += extend(br.bits(t), t);
                if (c.pred < -32768 || c.pred > 32767) { err = "corrupt JPEG data"; return false; }
                co[0] = (int16_t)(c.pred * (1 << sc.Al));
            } else if (br.bit()) {
                co[0] = (int16_t)(co[0] | (1 << sc.Al));
            }
            return true;
        }
        const int p1 = 1 << sc.Al, m1 = -(1 << sc.Al);
        if (sc.Ah == 0) {  // AC, first pass of a band
            if (eobrun > 0) { --eobrun; return true; }
            for (int k = sc.Ss; k <= sc.Se;) {
                const int rs = decode_sym(br, ac[c.ta]);
                if (rs < 0) { err = "corrupt JPEG data"; return false; }
                const int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                    k += 16;
                    continue;
                }
                k += r;
                if (k > 63) { err = "corrupt JPEG data"; return false; }
                co[kZigzag[k]] = (int16_t)(extend(br.bits(sz), sz) * p1);
                ++k;
            }
            return true;
        }
        // AC refinement (T.81 G.1.2.3): correction bits for the coefficients that are already non-zero, new +-1 << Al ones
        auto refine = [&](int16_t& v) { if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1)); };
        int k = sc.Ss;
        if (eobrun == 0) {
            for (; k <= sc.Se; ++k) {
                const int rs = decode_sym(br, ac[c.ta]);
                if (rs < 0) { err = "corrupt JPEG data"; return false; }
                int r = rs >> 4, sz = rs & 15, val = 0;
                if (sz) {
                    if (sz != 1) { err = "corrupt JPEG data"; return false; }
                    val = br.bit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += br.bits(r);
                    break;
                }
                while (k <= sc.Se) {
                    int16_t& v = co[kZigzag[k]];
                    if (v != 0) refine(v);
                    else if (--r < 0) break;
                    ++k;
                }
                if (val) {
                    if (k > sc.Se) { err = "corrupt JPEG data"; return false; }
                    co[kZigzag[k]] = (int16_t)val;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= sc.Se; ++k) { int16_t& v = co[kZigzag[k]]; if (v != 0) refine(v); }
            --eobrun;
        }
        return true;
    };
    auto restart_here = [&]() {  // expect RSTn: skip to it, reset predictors and the end-of-band run
        br.reset();
        while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) {
            if (br.p[0] == 0xff && br.p[1] != 0 && br.p[1] != 0xff) return;  // another marker: the scan is over (truncated data)
            ++br.p;
        }
        if (br.p + 1 < br.end) br.p += 2;
        for (int i = 0; i < sc.ns; ++i) comp[sc.ci[i]].pred = 0;
        eobrun = 0;
    };
    for (int i = 0; i < sc.ns; ++i) comp[sc.ci[i]].pred = 0;
    int rst_left = restart;
    if (sc.ns == 1) {  // not interleaved (T.81 A.2.2): the component's own blocks in raster order
        Comp& c = comp[sc.ci[0]];
        const int nbx = ((W * c.h + hmax - 1) / hmax + 7) / 8, nby = ((H * c.v + vmax - 1) / vmax + 7) / 8;
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx) {
                if (restart && rst_left == 0) { restart_here(); rst_left = restart; }
                if (!block(c, bx, by)) return false;
                if (restart) --rst_left;
            }
        return true;
    }
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart && rst_left == 0) { restart_here(); rst_left = restart; }
            for (int i = 0; i < sc.ns; ++i) {
                Comp& c = comp[sc.ci[i]];
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx)
                        if (!block(c, mx * c.h + bx, my * c.v + by)) return false;
            }
            if (restart) --rst_left;
        }
    return true;
}

}  // namespace

bool decode_jpeg_memory(const uint8_t* d, size_t len, Image& out, std::string& err) {
    if (len < 4 || d[0] != 0xff || d[1] != 0xd8) { err = "not a JPEG file"; return false; }
    uint16_t qt[4][64] = {};
    Huff dc[4], ac[4];
    Comp comp[4];
    int ncomp = 0, W = 0, H = 0, restart = 0, adobe_transform = -1, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, scans = 0;
    bool have_sof = false, progressive = false;
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
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // baseline / extended sequential / progressive, Huffman
            if (have_sof) { err = "unsupported JPEG (more than one frame)"; return false; }
            if (n < 6 || s[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            H = s[1] << 8 | s[2]; W = s[3] << 8 | s[4]; ncomp = s[5];
            if (W <= 0 || H <= 0 || (ncomp != 1 && ncomp != 3) || n < 6 + (size_t)3 * ncomp) { err = "unsupported JPEG frame"; return false; }
            // an 8x8 block costs at least 2 bits of entropy-coded data per component: a frame header that promises more
            // pixels than the file could possibly hold is refused before anything of that size is allocated (also caps at 2^27 px: ~2.5 GB of decoder buffers at most)
            if ((uint64_t)W * (uint64_t)H > ((uint64_t)1 << 27) || (uint64_t)W * (uint64_t)H / 256 > (uint64_t)len + 64) { err = "JPEG dimensions do not fit the file"; return false; }
            for (int i = 0; i < ncomp; ++i) { comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i] & 3;
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) { err = "bad JPEG sampling factors"; return false; } }
            // T.81 A.2.2: with ONE component nothing is interleaved -- one data unit per MCU whatever factors the header declares
            if (ncomp == 1) comp[0].h = comp[0].v = 1;
            for (int i = 0; i < ncomp; ++i) { hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax; }
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int i = 0; i < ncomp; ++i) {
                comp[i].bw = mcux * comp[i].h; comp[i].bh = mcuy * comp[i].v;
                comp[i].pw = comp[i].bw * 8; comp[i].ph = comp[i].bh * 8;
                comp[i].coef.assign((size_t)comp[i].bw * comp[i].bh * 64, 0);
            }
            progressive = m == 0xc2;
            have_sof = true;
        }
        else if ((m >= 0xc3 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc) { err = "unsupported JPEG coding process (arithmetic / lossless / hierarchical)"; return false; }
        else if (m == 0xdd && n >= 2) restart = s[0] << 8 | s[1];
        else if (m == 0xee && n >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        else if (m == 0xda) {  // SOS: one scan -- the only one of a baseline file, one of several otherwise
            if (!have_sof) { err = "JPEG scan before frame header"; return false; }
            if (n < 1) { err = "bad JPEG scan header"; return false; }
            Scan sc;
            sc.ns = s[0];
            if (sc.ns < 1 || sc.ns > ncomp || n < 1 + (size_t)2 * sc.ns + 3) { err = "bad JPEG scan header"; return false; }
            for (int i = 0; i < sc.ns; ++i) {
                int ci = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == s[1 + 2 * i]) ci = j;
                for (int j = 0; j < i; ++j) if (sc.ci[j] == ci) ci = -1;
                if (ci < 0) { err = "bad JPEG scan header"; return false; }
                sc.ci[i] = ci;
                comp[ci].td = s[2 + 2 * i] >> 4; comp[ci].ta = s[2 + 2 * i] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3) { err = "bad JPEG scan header"; return false; }
            }
            sc.Ss = s[1 + 2 * sc.ns]; sc.Se = s[2 + 2 * sc.ns]; sc.Ah = s[3 + 2 * sc.ns] >> 4; sc.Al = s[3 + 2 * sc.ns] & 15;
            if (!progressive) { sc.Ss = 0; sc.Se = 63; sc.Ah = sc.Al = 0; }
            if (sc.Ss > sc.Se || sc.Se > 63 || sc.Ah > 13 || sc.Al > 13 || (progressive && sc.Ss == 0 && sc.Se != 0) ||
                (progressive && sc.Ss > 0 && sc.ns != 1)) { err = "bad JPEG scan header"; return false; }
            for (int i = 0; i < sc.ns; ++i) {
                const Comp& c = comp[sc.ci[i]];
                const bool need_dc = !progressive || (sc.Ss == 0 && sc.Ah == 0), need_ac = !progressive || sc.Ss > 0;
                if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) { err = "missing JPEG Huffman table"; return false; }
            }
            if (++scans > 1000) { err = "corrupt JPEG data"; return false; }
            BitReader br{d + pos + seglen, d + len};
            if (!decode_scan(br, comp, sc, progressive, W, H, hmax, vmax, mcux, mcuy, restart, dc, ac, err)) return false;
            pos = (size_t)(br.p - d);  // the marker search above resumes from wherever the entropy decoder stopped
            continue;
        }
        pos += seglen;
    }
    if (!have_sof || scans == 0) { err = "JPEG has no image data"; return false; }
    // dequantise + inverse DCT of every block
    for (int i = 0; i < ncomp; ++i) {
        Comp& c = comp[i];
        c.plane.assign((size_t)c.pw * c.ph, 0);
        for (int by = 0; by < c.bh; ++by)
            for (int bx = 0; bx < c.bw; ++bx) {
                const int16_t* co = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                float blk[64];
                for (int k = 0; k < 64; ++k) blk[k] = (float)co[k] * qt[c.tq][k];
                idct8x8(blk, c.plane.data() + (size_t)(by * 8) * c.pw + bx * 8, c.pw);
            }
        std::vector<int16_t>().swap(c.coef);
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
        bool fx = rx == 2 && hmax % comp[i].h == 0, fy = ry == 2 && vmax % comp[i].v == 0;
        if (fx && cw <= 2) fx = fy = false;  // libjpeg (jdsample.c): the h2v1 / h2v2 triangle filters need more than two samples per row, else replication
        // per output column: the nearest sample and the next-nearest one (clamped to the valid samples), computed once
        std::vector<int> sxs(W), nxs(W);
        for (int x = 0; x < W; ++x) {
            const int sx = std::min(x * comp[i].h / hmax, cw - 1);
            int nx = sx + ((x & 1) ? 1 : -1);
            nx = nx < 0 ? 0 : nx >= cw ? cw - 1 : nx;
            sxs[x] = sx; nxs[x] = nx;
        }
        for (int y = 0; y < H; ++y) {
            const int sy = std::min(y * comp[i].v / vmax, chh - 1);
            int ny = sy + ((y & 1) ? 1 : -1);
            ny = ny < 0 ? 0 : ny >= chh ? chh - 1 : ny;
            const uint8_t* r0 = pl.data() + (size_t)sy * pw;
            const uint8_t* r1 = pl.data() + (size_t)ny * pw;
            uint8_t* o = up[i].data() + (size_t)y * W;
            if (fx && fy) {
                for (int x = 0; x < W; ++x)
                    o[x] = (uint8_t)((9 * r0[sxs[x]] + 3 * r0[nxs[x]] + 3 * r1[sxs[x]] + r1[nxs[x]] + 8) >> 4);
            } else if (fx) {
                for (int x = 0; x < W; ++x) o[x] = (uint8_t)((3 * r0[sxs[x]] + r0[nxs[x]] + ((x & 1) ? 2 : 1)) >> 2);
            } else if (fy) {
                for (int x = 0; x < W; ++x) o[x] = (uint8_t)((3 * r0[sxs[x]] + r1[sxs[x]] + 2) >> 2);
            } else if (rx == 1) {
                memcpy(o, r0, (size_t)W);
            } else {
                for (int x = 0; x < W; ++x) o[x] = r0[sxs[x]];
            }
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

// ---------------------------------------------------------------------------
// Baseline JPEG encoder (`.save("x.jpg")`, reference main.rs:175 through the image crate): YCbCr 4:4:4, the Annex K
// quantisation tables scaled to quality 75 (the image crate's default), the Annex K Huffman tables, JFIF header.
// Alpha is dropped (JPEG has none).
// ---------------------------------------------------------------------------
namespace {
const uint8_t kQLum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                           18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
    0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
    0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
    0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
    0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
EncTable make_enc(const uint8_t* bits16, const uint8_t* vals) {
    EncTable t{};
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits16[l - 1]; ++i) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)l; ++code; ++k; }
        code <<= 1;
    }
    return t;
}
struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0; int cnt = 0;
    void put(uint32_t v, int n) {
        acc = (acc << n) | (v & ((1u << n) - 1)); cnt += n;
        while (cnt >= 8) { const uint8_t b = (uint8_t)(acc >> (cnt - 8)); out.push_back(b); if (b == 0xff) out.push_back(0); cnt -= 8; }
    }
    void flush() { if (cnt) put(0x7f, 8 - cnt); }
};
void fdct8x8(const float* in, float* out) {  // plain separable forward DCT-II (orthonormal JPEG scaling)
    struct Basis {
        float c[8][8];
        Basis() {
            for (int u = 0; u < 8; ++u)
                for (int x = 0; x < 8; ++x) c[u][x] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        }
    };
    static const Basis basis;  // initialised once, thread-safely: the encoder's workers all land here
    const auto& c = basis.c;
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int u = 0; u < 8; ++u) { float s = 0; for (int x = 0; x < 8; ++x) s += c[u][x] * in[y * 8 + x]; tmp[y * 8 + u] = s; }
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) { float s = 0; for (int y = 0; y < 8; ++y) s += c[v][y] * tmp[y * 8 + u]; out[v * 8 + u] = s; }
}
}  // namespace

bool encode_jpeg_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err, int quality) {
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || !rgba) { err = "image size not representable in JPEG"; return false; }
    quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t q[2][64];
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) { int v = ((t ? kQChr[i] : kQLum[i]) * scale + 50) / 100; q[t][i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v); }
    std::vector<uint8_t> out;
    auto put16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    auto marker = [&](int m, size_t payload) { out.push_back(0xff); out.push_back((uint8_t)m); put16((int)payload + 2); };
    out.push_back(0xff); out.push_back(0xd8);
    marker(0xe0, 14); { const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0}; out.insert(out.end(), jfif, jfif + 14); }
    for (int t = 0; t < 2; ++t) { marker(0xdb, 65); out.push_back((uint8_t)t); for (int i = 0; i < 64; ++i) out.push_back(q[t][kZigzag[i]]); }
    marker(0xc0, 15); out.push_back(8); put16(h); put16(w); out.push_back(3);
    for (int c = 0; c < 3; ++c) { out.push_back((uint8_t)(c + 1)); out.push_back(0x11); out.push_back((uint8_t)(c ? 1 : 0)); }
    auto dht = [&](int tc_th, const uint8_t* bits16, const uint8_t* vals, int nvals) {
        marker(0xc4, 17 + nvals); out.push_back((uint8_t)tc_th); out.insert(out.end(), bits16, bits16 + 16); out.insert(out.end(), vals, vals + nvals); };
    dht(0x00, kDcLumBits, kDcVals, 12); dht(0x10, kAcLumBits, kAcLumVals, 162); dht(0x01, kDcChrBits, kDcVals, 12); dht(0x11, kAcChrBits, kAcChrVals, 162);
    // One restart interval per row of 8x8 blocks (DRI): every interval starts with fresh DC predictors on a byte boundary,
    // so the rows are entropy-coded independently -- by a pool of threads, each into its own buffer -- and joined with
    // RST0..7 markers.  (The colour transform, DCT and Huffman coding of a 5760x3240 output took 283 ms on one core.)
    const int mcu_w = (w + 7) / 8, mcu_h = (h + 7) / 8;
    if (mcu_h > 1) { marker(0xdd, 2); put16(mcu_w); }
    marker(0xda, 10); out.push_back(3);
    for (int c = 0; c < 3; ++c) { out.push_back((uint8_t)(c + 1)); out.push_back((uint8_t)(c ? 0x11 : 0x00)); }
    out.push_back(0); out.push_back(63); out.push_back(0);
    const EncTable dcT[2] = {make_enc(kDcLumBits, kDcVals), make_enc(kDcChrBits, kDcVals)};
    const EncTable acT[2] = {make_enc(kAcLumBits, kAcLumVals), make_enc(kAcChrBits, kAcChrVals)};
    std::vector<std::vector<uint8_t>> rows(mcu_h);
    auto encode_row = [&](int by) {
        std::vector<uint8_t>& seg = rows[by];
        seg.reserve((size_t)mcu_w * 48);
        BitWriter bw{seg};
        int pred[3] = {0, 0, 0};
        for (int bx = 0; bx < mcu_w; ++bx) {
            float px[3][64];
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    const int sx = std::min(bx * 8 + x, w - 1), sy = std::min(by * 8 + y, h - 1);  // edge replication
                    const uint8_t* p = rgba + ((size_t)sy * w + sx) * 4;
                    const float r = p[0], g = p[1], b = p[2];
                    px[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                    px[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                    px[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                }
            for (int c = 0; c < 3; ++c) {
                float f[64];
                fdct8x8(px[c], f);
                int zz[64];
                const int t = c ? 1 : 0;
                for (int i = 0; i < 64; ++i) zz[i] = (int)std::lround(f[kZigzag[i]] / q[t][kZigzag[i]]);
                auto category = [](int v) { int a = v < 0 ? -v : v, n = 0; while (a) { ++n; a >>= 1; } return n; };
                auto put_val = [&](int v, int n) { if (n) bw.put((uint32_t)(v < 0 ? v + (1 << n) - 1 : v), n); };
                const int diff = zz[0] - pred[c];
                pred[c] = zz[0];
                int n = category(diff);
                bw.put(dcT[t].code[n], dcT[t].len[n]); put_val(diff, n);
                int run = 0;
                for (int i = 1; i < 64; ++i) {
                    if (zz[i] == 0) { ++run; continue; }
                    while (run > 15) { bw.put(acT[t].code[0xf0], acT[t].len[0xf0]); run -= 16; }
                    n = category(zz[i]);
                    const int sym = (run << 4) | n;
                    bw.put(acT[t].code[sym], acT[t].len[sym]); put_val(zz[i], n);
                    run = 0;
                }
                if (run) bw.put(acT[t].code[0], acT[t].len[0]);
            }
        }
        bw.flush();
    };
    {
        std::atomic<int> next{0};
        auto worker = [&] { for (int by; (by = next.fetch_add(1)) < mcu_h;) encode_row(by); };
        const int nthr = (int)std::min<size_t>(usable_cpus(), (size_t)std::max(1, mcu_h / 4));
        std::vector<std::thread> th;
        for (int t = 1; t < nthr; ++t) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
    }
    for (int by = 0; by < mcu_h; ++by) {
        out.insert(out.end(), rows[by].begin(), rows[by].end());
        std::vector<uint8_t>().swap(rows[by]);
        if (by + 1 < mcu_h) { out.push_back(0xff); out.push_back((uint8_t)(0xd0 + (by & 7))); }
    }
    out.push_back(0xff); out.push_back(0xd9);
    FILE* fo = fopen(path.c_str(), "wb");
    if (!fo) { err = "cannot create file"; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), fo) == out.size();
    fclose(fo);
    if (!ok) err = "short write";
    return ok;
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
