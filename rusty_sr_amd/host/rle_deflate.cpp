// rle_deflate.cpp -- a small deflate (RFC 1951) encoder for PNG-filtered pixels: run-length matches (distance 1) and
// dynamic Huffman codes, nothing else.  zlib's Z_RLE strategy makes the same choices and -- on filtered residuals of
// continuous-tone images, where longer LZ77 matches are rare and only distort the literal statistics -- already gives
// smaller files than its general matcher (png.cpp encode_band); this encoder writes that format several times faster
// because it does nothing general: one pass counts a block's bytes and notes its runs, one builds the two code tables,
// one writes the bits through a 64-bit buffer.  The output is a plain deflate stream any inflate reads
// (tests/test_host_cli.py decodes it with zlib and Pillow).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "png.hpp"

namespace srpng {
namespace {

struct BitSink {
    uint8_t* p;
    uint64_t acc = 0;
    int n = 0;
    inline void put(uint32_t bits, int len) {  // len <= 32, LSB first
        acc |= (uint64_t)bits << n;
        n += len;
        if (n >= 32) { memcpy(p, &acc, 4); p += 4; acc >>= 32; n -= 32; }  // little-endian host (x86-64)
    }
    inline void align() {
        while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
        n = 0; acc = 0;
    }
};

inline uint32_t reverse_bits(uint32_t v, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// Code lengths (<= max_len) for `n` symbols with the given frequencies; unused symbols get 0.  Huffman's algorithm on
// the sorted symbols, then the classic repair when the tree is deeper than allowed (move leaves up until Kraft's sum fits).
void huffman_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len) {
    struct Node { uint64_t w; int left, right; };
    std::vector<int> used;
    for (int i = 0; i < n; ++i) { len[i] = 0; if (freq[i]) used.push_back(i); }
    if (used.empty()) return;
    if (used.size() == 1) { len[used[0]] = 1; return; }
    std::sort(used.begin(), used.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    const int m = (int)used.size();
    std::vector<Node> nodes(2 * m);
    for (int i = 0; i < m; ++i) nodes[i] = {freq[used[i]], -1, -1};
    int leaf = 0, inner = m, next = m;  // two queues: sorted leaves, and inner nodes in creation (= weight) order
    auto take = [&]() {
        if (leaf < m && (inner >= next || nodes[leaf].w <= nodes[inner].w)) return leaf++;
        return inner++;
    };
    for (int k = 0; k < m - 1; ++k) {
        const int a = take(), b = take();
        nodes[next] = {nodes[a].w + nodes[b].w, a, b};
        ++next;
    }
    std::vector<int> depth(2 * m, 0);
    for (int i = next - 1; i >= m; --i) { depth[nodes[i].left] = depth[i] + 1; depth[nodes[i].right] = depth[i] + 1; }
    std::vector<int> count(std::max(max_len, 64) + 1, 0);
    for (int i = 0; i < m; ++i) ++count[std::min(depth[i], max_len)];
    uint64_t total = 0;
    for (int l = 1; l <= max_len; ++l) total += (uint64_t)count[l] << (max_len - l);
    while (total > ((uint64_t)1 << max_len)) {  // too many long codes: shorten one at max_len's expense elsewhere
        --count[max_len];
        for (int l = max_len - 1; l > 0; --l)
            if (count[l]) { --count[l]; count[l + 1] += 2; break; }
        --total;
    }
    // the rarest symbols get the longest codes
    int idx = 0;
    for (int l = max_len; l >= 1; --l)
        for (int c = 0; c < count[l]; ++c) len[used[idx++]] = (uint8_t)l;
}

void canonical_codes(const uint8_t* len, int n, uint16_t* code) {
    int count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    int c = 0;
    for (int l = 1; l < 16; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)reverse_bits((uint32_t)next[len[i]]++, len[i]) : 0;
}

constexpr int kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};

struct LenCode { uint8_t sym, extra_bits; uint16_t extra; };  // for match length 3..258

const LenCode* length_table() {
    static LenCode t[259];
    static const bool init = [] {
        for (int l = 3; l <= 258; ++l) {
            int s = 28;
            while (kLenBase[s] > l) --s;
            if (l == 258) s = 28;
            t[l] = {(uint8_t)s, (uint8_t)kLenExtra[s], (uint16_t)(l - kLenBase[s])};
        }
        return true;
    }();
    (void)init;
    return t;
}

// one deflate block of src[0, n): runs and histogram, tables, bits.  false (nothing written): the block would not fit before `end`.
// Most bytes of filtered pixels are literals, so they are never turned into tokens: the first pass only counts them (four
// interleaved histograms: consecutive equal bytes would otherwise serialise on one counter) and notes where runs start;
// the last pass walks the source again and emits literals straight from it.
struct Run { uint32_t pos; uint32_t len; };  // len bytes equal to src[pos - 1], starting at pos (len >= 3)

bool encode_block(const uint8_t* src, size_t n, bool final_block, BitSink& out, const uint8_t* end, std::vector<Run>& runs) {
    const LenCode* LT = length_table();
    uint32_t freq[286] = {0};
    uint32_t h4[4][256];
    memset(h4, 0, sizeof h4);
    runs.clear();
    size_t i = 0;
    while (i + 3 < n) {
        const uint8_t b = src[i];
        uint32_t four;
        memcpy(&four, src + i, 4);
        if (four == b * 0x01010101u) {  // four equal bytes: a literal and a run of >= 3 (one rarely taken branch, no data-dependent chain)
            size_t j = i + 4;
            while (j < n && src[j] == b) ++j;
            ++h4[0][b];
            runs.push_back({(uint32_t)(i + 1), (uint32_t)(j - i - 1)});
            i = j;
        } else {
            ++h4[i & 3][b];
            ++i;
        }
    }
    for (; i < n; ++i) ++h4[0][src[i]];
    for (int v = 0; v < 256; ++v) freq[v] = h4[0][v] + h4[1][v] + h4[2][v] + h4[3][v];
    // a run goes out as matches of 3..258 bytes (never leaving a tail of 1 or 2)
    auto for_each_match = [](uint32_t len, auto&& f) {
        while (len >= 3) {
            uint32_t l = len < 258 ? len : 258;
            if (len - l > 0 && len - l < 3) l = len - 3;
            f(l);
            len -= l;
        }
    };
    for (const Run& r : runs) for_each_match(r.len, [&](uint32_t l) { ++freq[257 + LT[l].sym]; });
    freq[256] = 1;
    uint8_t ll_len[286], d_len[30] = {0};
    huffman_lengths(freq, 286, 15, ll_len);
    d_len[0] = 1;  // the only distance ever used is 1 (code 0); one 1-bit code is a legal distance tree
    uint16_t ll_code[286];
    canonical_codes(ll_len, 286, ll_code);
    int hlit = 286;
    while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
    const int hdist = 1;
    // code lengths of both alphabets, run-length coded with symbols 16 / 17 / 18
    uint8_t seq[286 + 30];
    memcpy(seq, ll_len, (size_t)hlit);
    memcpy(seq + hlit, d_len, (size_t)hdist);
    const int nseq = hlit + hdist;
    struct CL { uint8_t sym, extra_bits, extra; };
    CL cl[286 + 30];
    int ncl = 0;
    uint32_t cl_freq[19] = {0};
    for (int k = 0; k < nseq;) {
        const uint8_t v = seq[k];
        int r = 1;
        while (k + r < nseq && seq[k + r] == v) ++r;
        int left = r;
        if (v == 0) {
            while (left >= 11) { const int t = std::min(left, 138); cl[ncl++] = {18, 7, (uint8_t)(t - 11)}; ++cl_freq[18]; left -= t; }
            if (left >= 3) { cl[ncl++] = {17, 3, (uint8_t)(left - 3)}; ++cl_freq[17]; left = 0; }
            for (; left > 0; --left) { cl[ncl++] = {0, 0, 0}; ++cl_freq[0]; }
        } else {
            cl[ncl++] = {v, 0, 0}; ++cl_freq[v]; --left;
            while (left >= 3) { const int t = std::min(left, 6); cl[ncl++] = {16, 2, (uint8_t)(t - 3)}; ++cl_freq[16]; left -= t; }
            for (; left > 0; --left) { cl[ncl++] = {v, 0, 0}; ++cl_freq[v]; }
        }
        k += r;
    }
    uint8_t cl_len[19];
    uint16_t cl_code[19];
    huffman_lengths(cl_freq, 19, 7, cl_len);
    canonical_codes(cl_len, 19, cl_code);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl_len[order[hclen - 1]] == 0) --hclen;
    {   // exact size of what follows, checked against the room left (the caller sizes its buffer for ~10 bits per byte,
        // not for the 15-bit worst case of a length-limited code)
        uint64_t bits = 17 + 3 * (uint64_t)hclen + 64;
        for (int k = 0; k < ncl; ++k) bits += cl_len[cl[k].sym] + cl[k].extra_bits;
        for (int v = 0; v < 257; ++v) bits += (uint64_t)freq[v] * ll_len[v];
        for (int c = 0; c < 29; ++c) bits += (uint64_t)freq[257 + c] * (ll_len[257 + c] + kLenExtra[c] + 1);
        if (out.p + bits / 8 + 16 > end) return false;
    }
    out.put(final_block ? 1 : 0, 1);
    out.put(2, 2);  // dynamic Huffman
    out.put((uint32_t)(hlit - 257), 5);
    out.put((uint32_t)(hdist - 1), 5);
    out.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) out.put(cl_len[order[k]], 3);
    for (int k = 0; k < ncl; ++k) {
        out.put(cl_code[cl[k].sym], cl_len[cl[k].sym]);
        if (cl[k].extra_bits) out.put(cl[k].extra, cl[k].extra_bits);
    }
    // literal code + length packed per byte value; two literals go out per put (2 x 15 bits fit the 32-bit window)
    uint32_t lit[256];
    for (int v = 0; v < 256; ++v) lit[v] = (uint32_t)ll_code[v] | (uint32_t)ll_len[v] << 16;
    auto literals = [&](size_t from, size_t to) {
        size_t k = from;
        for (; k + 1 < to; k += 2) {
            const uint32_t a = lit[src[k]], b = lit[src[k + 1]];
            const int la = (int)(a >> 16), lb = (int)(b >> 16);
            out.put((a & 0xffff) | (b & 0xffff) << la, la + lb);
        }
        if (k < to) out.put(lit[src[k]] & 0xffff, (int)(lit[src[k]] >> 16));
    };
    size_t pos = 0;
    for (const Run& r : runs) {
        literals(pos, r.pos);
        for_each_match(r.len, [&](uint32_t l) {
            const LenCode& lc = LT[l];
            const int sy = 257 + lc.sym;
            // length code, its extra bits, and the 1-bit distance code (0) in one go: at most 15 + 5 + 1 bits
            out.put((uint32_t)ll_code[sy] | (uint32_t)lc.extra << ll_len[sy], ll_len[sy] + lc.extra_bits + 1);
        });
        pos = (size_t)r.pos + r.len;
    }
    literals(pos, n);
    out.put(ll_code[256], ll_len[256]);
    return true;
}

}  // namespace

// enough for any input that Huffman coding does not expand by more than a quarter (a flat 9-bit code is the worst an
// optimal code can do; rle_deflate returns 0 rather than overrun when a length-limited code does worse)
size_t rle_deflate_bound(size_t n) { return n + n / 4 + (n / 65536 + 2) * 512 + 64; }

// Raw deflate of src[0, n) into dst[0, cap): blocks of 256 KB so that the code tables follow the picture; `last` sets
// BFINAL on the final block, otherwise the segment ends on a byte boundary with an empty stored block -- exactly what
// zlib's Z_SYNC_FLUSH writes, so segments of independent row bands concatenate into one stream.  Returns the number of
// bytes written, 0 if cap is too small.
size_t rle_deflate(const uint8_t* src, size_t n, bool last, uint8_t* dst, size_t cap) {
    if (cap < 32) return 0;
    const uint8_t* end = dst + cap;
    BitSink out{dst};
    std::vector<Run> runs;
    const size_t block = (size_t)1 << 18;
    size_t pos = 0;
    do {
        const size_t m = std::min(block, n - pos);
        if (!encode_block(src + pos, m, last && pos + m == n, out, end - 8, runs)) return 0;
        pos += m;
    } while (pos < n);
    if (!last) {
        out.put(0, 3);  // BFINAL 0, BTYPE 00 (stored)
        out.align();
        const uint8_t empty[4] = {0x00, 0x00, 0xff, 0xff};
        memcpy(out.p, empty, 4);
        out.p += 4;
    } else {
        out.align();
    }
    return (size_t)(out.p - dst);
}

}  // namespace srpng

extern "C" {
// test surface (tests/test_host_cli.py): raw deflate of a buffer
size_t srpng_rle_deflate_bound(size_t n) { return srpng::rle_deflate_bound(n); }
size_t srpng_rle_deflate(const uint8_t* src, size_t n, int last, uint8_t* dst, size_t cap) { return srpng::rle_deflate(src, n, last != 0, dst, cap); }
}
