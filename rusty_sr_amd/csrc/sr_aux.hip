// sr_aux.hip -- the two parameter-free graphs of the reference's upscale(): bilinear_net (network.rs:111-123,
// `-p bilinear`) and downsample_net (network.rs:125-138, `-d`), for gfx950.
//
//   bilinear_net   : out = LinearToSrgb(LinearInterp x3(SrgbToLinear(x)))          36 B (u8 RGBA) or 108 B (f32) out per input px
//   downsample_net : out = LinearToSrgb(mean 3x3(SrgbToLinear(x)))                 27 / 108 B in per output px
//
// Both are HBM-bound by their I/O if the transfer functions cost nothing, so that is what the kernels arrange:
//  * u8 in : SrgbToLinear of a byte is a 256-entry table (built once per context with the same powf expression the
//    one-thread-per-pixel kernels of rounds 1-3 evaluated per sample: bit-identical; copied into LDS per workgroup);
//  * u8 out: data_to_img(LinearToSrgb(l)) is a monotone step function of l with 255 steps.  The steps' positions (the
//    smallest float l that quantises to k, k = 1..255) are found ONCE per context by bisection over the float bit
//    patterns with the same powf expression (threshold_kernel), and laid out as a table indexed by the top 16 bits of
//    l (exponent + 7 mantissa bits: 128 buckets per octave, at most one step per bucket -- the builder checks): a
//    lookup is one ds_read_b32 and an add instead of a powf, and gives the powf's answer;
//  * f32 in / out: powf(x, p) = exp2(p * log2(x)) on v_log_f32 / v_exp_f32 (1 ulp each; measured 1.8e-7 absolute
//    against the oracle's libm, inside the 1e-5 the tests ask of these graphs and far inside north_star's 1e-4);
//  * the interpolation runs in the reference's operation order ((1-t) a + t b, rows then columns); a work item is a 16-byte
//    chunk (4 RGBA pixels / 4 floats) of the three output rows of an input row, consecutive lanes own consecutive chunks, so a
//    store instruction writes one contiguous run (16 bytes per lane whenever the rows are 16-byte aligned -- W % 4 == 0 --
//    dwords otherwise);
//  * f32 bilinear: every input sample is linearised once per tile (staged, edge-replicated, in LDS as [pixel][4] f32), one
//    workgroup per output tile;
//  * u8 bilinear and both downsample kernels: nothing staged, no barrier after the tables -- a lane reads the few bytes its
//    item needs straight from global memory as aligned dwords (the windows of neighbouring lanes are contiguous), the next
//    item's before this item's arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "sr_kernels.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- the transfer functions, exactly as rounds 1-3 evaluated them (alumina SrgbToLinear / LinearToSrgb: IEC 61966-2-1)
__device__ __forceinline__ float srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float l) { return l <= 0.0031308f ? 12.92f * l : 1.055f * powf(l, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ uint32_t quant_u8(float v) {  // data_to_img (main.rs:175)
    return (uint32_t)fminf(fmaxf(floorf(255.0f * v + 0.5f), 0.0f), 255.0f);
}
// ... and on the hardware's log2 / exp2 (the f32 entry points: no table can help an arbitrary float)
__device__ __forceinline__ float fast_pow(float x, float p) { return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float srgb_to_linear_fast(float s) { return s <= 0.04045f ? s / 12.92f : fast_pow((s + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb_fast(float l) { return l <= 0.0031308f ? 12.92f * l : 1.055f * fast_pow(l, 1.0f / 2.4f) - 0.055f; }

// ---- quantiser table: byte = base + (l >= step) for the bucket of l, one DWORD per bucket.
// A bucket is the set of floats that share their top 16 bits, so a step inside it is known by its low 16 bits, and
//     entry = base * 2^16 + (0x10000 - step_low16)        (no step in the bucket: base * 2^16)
//     byte  = (entry + (bits(l) & 0xffff)) >> 16          (the carry out of the low half is the comparison)
// Entry 0 serves everything below 2^-13 (zero, negative values: byte 0), the last entry everything from 1.0 up (and NaN: 255).
// Rounds 1-4 kept (step, base) as 8 bytes: 64 random buckets per wave then fall on 32 bank pairs, and two thirds of the LDS array's
// busy time in the u8 bilinear kernel were bank conflicts (profiles/r4_aux_pmc.txt); a dword entry spreads them over all 64 banks with
// the same single LDS instruction per look-up (round 4's split into two dword arrays needed two: slower).
constexpr int kQExp0 = 114;                       // biased exponent of 2^-13: below it every l quantises to 0
constexpr int kQBuckets = (127 - kQExp0) * 128;   // [2^-13, 1) in 128 buckets per octave
constexpr int kQEntries = kQBuckets + 2;          // + entry 0 (below the range) + the last entry (1.0 and up)
typedef uint32_t QEntry;
// entry + low 16 bits of l: the byte is bits 16..23 of the sum (bits 24.. are zero: base <= 255)
// UNIT: the caller guarantees 0 <= l <= 1 (and not -0), so only the lower clamp is needed.  That holds for everything the u8 kernels
// quantise: convex combinations (1 - t) a + t b and means of table values in [0, 1] -- fl((1 - t) a) <= 1 - t and fl(t b) <= t by
// monotonicity of rounding, 1 - t is exact for these t, so the sum rounds to at most 1; products and sums of non-negative values are
// never -0.
template <bool UNIT = false>
__device__ __forceinline__ uint32_t quant_sum(const QEntry* tab, float l) {
    const uint32_t bits = __float_as_uint(l);
    constexpr int LO = (kQExp0 << 7) - 1;                                  // top 16 bits of the floats of entry 0
    if constexpr (UNIT) {
        const uint32_t top = max(bits >> 16, (uint32_t)LO);
        return tab[top - LO] + (bits & 0xffffu);
    } else {
        const int top = min(max((int)bits >> 16, LO), LO + kQEntries - 1);  // (arithmetic shift: negative values land below LO)
        return tab[top - LO] + (bits & 0xffffu);
    }
}
// RGBA8 pixel (alpha 255) of three channels in [0, 1]: two v_perm_b32 pick byte 2 of each sum
__device__ __forceinline__ uint32_t quant_pixel(const QEntry* tab, float r, float g, float b) {
    const uint32_t rg = __builtin_amdgcn_perm(quant_sum<true>(tab, g), quant_sum<true>(tab, r), 0x0c0c0602u);
    return __builtin_amdgcn_perm(quant_sum<true>(tab, b), rg, 0x0d060100u);
}

// (Round 4 also tried the opposite balance -- estimate 255 LinearToSrgb(l) + 0.5 with v_log_f32 / v_exp_f32 and ask the table only
// within 1e-3 of an integer, bit-identical by construction: 43 us against 32 us for the table alone, for one, two or three of a
// pixel's channels alike (profiles/r4_aux_quantiser_estimate_first.txt).  The look-ups' bank conflicts cost less than 14 more vector
// instructions per sample.  Likewise the table as two dword arrays (64 consecutive buckets over 64 banks instead of 32 over the 32
// bank pairs of 8-byte entries): 39 us, profiles/r4_aux_quantiser_split_table.txt -- two LDS instructions per look-up cost more than
// the conflicts they avoid.  Round 5: ONE dword per bucket, above.)

// smallest float l in [0, 1] with quant_u8(linear_to_srgb(l)) >= k, for k = 1 .. 255 (thread k - 1): bisection over
// the bit patterns (non-negative floats order like their bits), then a short downward scan in case the powf is not
// monotone to the last bit around the step
__global__ void threshold_kernel(float* steps) {
    const uint32_t k = threadIdx.x + 1;
    if (k > 255) return;
    uint32_t lo = 0, hi = __float_as_uint(1.0f);  // quant(lo) < k <= quant(hi)
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (quant_u8(linear_to_srgb(__uint_as_float(mid))) >= k) hi = mid; else lo = mid;
    }
    for (int i = 0; i < 64 && hi > 1; ++i)
        if (quant_u8(linear_to_srgb(__uint_as_float(hi - 1))) >= k) --hi;
    steps[k - 1] = __uint_as_float(hi);
}

// img_to_data (main.rs:170: byte / 255, a true division) then SrgbToLinear, for every byte value: 256 floats behind the step table
__global__ void lut_kernel(float* lut) { lut[threadIdx.x] = srgb_to_linear(__fdiv_rn((float)threadIdx.x, 255.0f)); }

constexpr int kBlTW = 64, kBlTH = 16;                      // bilinear: input tile; output tile 192 x 48
constexpr int kBlTWH = kBlTW + 2, kBlTHH = kBlTH + 2, kBlNPIX = kBlTWH * kBlTHH;

// Output column o of the tile (in units of the interpolated quantity: pixels) -> tile column of its first input and the
// weight of the second: phase p = o % 3 of input pixel i = o / 3 reads inputs (i-1, i) with t = 2/3 for p = 0, (i, i+1)
// with t = 0 for p = 1 and t = 1/3 for p = 2 (LinearInterp, network.rs:118: half-pixel centres; the oracle's
// linterp3_acc).  The tile carries a one-pixel edge-replicated halo, so input i sits at tile column i + 1.
__device__ __forceinline__ void phase(int o, int& i0, float& t) {
    const int i = o / 3, p = o - 3 * i;
    i0 = min(i + (p == 0 ? 0 : 1), kBlTWH - 2);  // (clamped for the padding of a row's last, partial chunk: computed, never stored)
    t = p == 0 ? 2.0f / 3.0f : (p == 1 ? 0.0f : 1.0f / 3.0f);
}

// bilinear_net, f32 in / f32 out.  One work item = one 16-byte chunk column (4 floats) of the THREE output rows of one input row y:
// the horizontal interpolation of input rows y-1, y, y+1 is done once and shared by the three output rows (vertical phases
// t = 2/3 of (y-1, y), 0 of (y, y+1), 1/3 of (y, y+1)), in the expression and order of the one-thread-per-pixel kernel
// this replaces: ra = (1-tx) v00 + tx v01, rb = ..., out = (1-ty) ra + ty rb.  Consecutive lanes own consecutive chunks
// of a row, so every store instruction writes one contiguous run.  (Rounds 4-5 ran the u8 entry points through this kernel too:
// bilinear_u8_kernel below replaced that.)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void bilinear_tile_kernel(AuxArgs a) {
    __shared__ __attribute__((aligned(16))) float s_lin[kBlNPIX * 4];
    const int tid = threadIdx.x;
    const int tiles_x = (a.W + kBlTW - 1) / kBlTW, tiles_y = (a.H + kBlTH - 1) / kBlTH;
    const long ntiles = (long)a.n * tiles_x * tiles_y;
    const int OW = 3 * a.W, OH = 3 * a.H;
    constexpr int CPR = 3 * kBlTW * 3 / 4;            // chunks per output row of a full tile (3 floats per output pixel)
    // The input pixels of a tile travel global -> registers -> LDS.  The loads of the NEXT tile are issued before this tile's
    // arithmetic and land under it: every workgroup of the launch starts at the same moment, so without this all of them sit out the
    // same load latency together, twice per tile pair.
    constexpr int PER = (kBlNPIX + 255) / 256;
    uint32_t raw[PER][3];
    auto fetch = [&](long tile) {
        const int n = (int)(tile / (tiles_x * tiles_y)), tr = (int)(tile - (long)n * tiles_x * tiles_y);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int x0 = tx * kBlTW, y0 = ty * kBlTH;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = min(tid + 256 * k, kBlNPIX - 1), py = p / kBlTWH, px = p - py * kBlTWH;
            const int gy = min(max(y0 - 1 + py, 0), a.H - 1), gx = min(max(x0 - 1 + px, 0), a.W - 1);
            const uint32_t* q = (const uint32_t*)a.img + (((size_t)n * a.H + gy) * a.W + gx) * 3;
            raw[k][0] = q[0]; raw[k][1] = q[1]; raw[k][2] = q[2];
        }
    };
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tiles_x * tiles_y)), tr = (int)(tile - (long)n * tiles_x * tiles_y);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int x0 = tx * kBlTW, y0 = ty * kBlTH;
        __syncthreads();  // the previous tile's readers are done
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = tid + 256 * k;
            if (p < kBlNPIX)
                *(f32x4*)&s_lin[p * 4] = f32x4{srgb_to_linear_fast(__uint_as_float(raw[k][0])), srgb_to_linear_fast(__uint_as_float(raw[k][1])),
                                               srgb_to_linear_fast(__uint_as_float(raw[k][2])), 0.f};
        }
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        __syncthreads();
        const int tw = min(kBlTW, a.W - x0), th = min(kBlTH, a.H - y0);  // this tile's own input pixels
        const int elems = 3 * tw * 3;                                    // floats per output row of this tile
        for (int q = tid; q < CPR * kBlTH; q += 256) {
            const int y = q / CPR, cx = q - y * CPR;
            if (y >= th || 4 * cx >= elems) continue;
            const float* row = s_lin + (y * kBlTWH) * 4;   // input row y - 1 (tile row y); y and y + 1 follow
            float h[3][4];                                  // horizontally interpolated: [input row][element of the chunk]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fi = 4 * cx + e, ox = fi / 3, c = fi - 3 * ox;   // [x][c] interleaved: 9 floats per input pixel
                int xa; float txv;
                phase(ox, xa, txv);
#pragma unroll
                for (int r = 0; r < 3; ++r) h[r][e] = (1.0f - txv) * row[(r * kBlTWH + xa) * 4 + c] + txv * row[(r * kBlTWH + xa) * 4 + 4 + c];
            }
#pragma unroll
            for (int py = 0; py < 3; ++py) {
                const float tyv = py == 0 ? 2.0f / 3.0f : (py == 1 ? 0.0f : 1.0f / 3.0f);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = linear_to_srgb_fast((1.0f - tyv) * h[py == 0 ? 0 : 1][e] + tyv * h[py == 0 ? 1 : 2][e]);
                float* dst = (float*)a.out + (((size_t)n * OH + 3 * (y0 + y) + py) * OW + 3 * x0) * 3 + 4 * cx;
                if (ALIGNED || 4 * cx + 4 <= elems) {
                    if constexpr (ALIGNED) *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
                    else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (4 * cx + e < elems) dst[e] = o[e];
                }
            }
        }
    }
}

// bilinear_net, u8 in / RGBA8 out (round 5): the same work item -- a 16-byte chunk (4 output pixels) of the three output rows of input
// row y -- but nothing staged and no barrier after the tables are in place.  A WAVE owns 64 consecutive chunks of one input row's output
// rows (every store instruction: one contiguous 1 KB run), walks such blocks with a uniform stride, and requests the next block's
// pixels before the arithmetic of this one.  The four output pixels o0 .. o0 + 3 (o0 = 4 cx, i0 = o0 / 3, p0 = o0 % 3) read the three
// input pixels ws, ws + 1, ws + 2 with ws = i0 - (p0 == 0) and no others, so a lane reads the 9 (12) bytes of that window from each of
// the rows y - 1, y, y + 1 as the aligned dwords that hold them (as downsample_net does; all of them hold a byte of the window) and
// sends each byte through the SrgbToLinear table: 27 look-ups per item instead of 24 ds_read_b128 of staged pixels.  With w0, w1, w2 the
// window's pixels the old kernel's (1 - t) a + t b become, bit for bit (0 * w = +0 and x + 0 = x: every w is finite and >= 0):
//     e = 0: (1 - t) w0 + t w1        e = 3: (1 - t) w1 + t w2         t = 2/3, 0, 1/3 for p0 = 0, 1, 2
//     e = 1: a w0 + b w1              (a, b) = (0, 1), (1 - 1/3, 1/3), (1 - 2/3, 2/3)
//     e = 2: (c0 w0 + c1 w1) + c2 w2  (c0, c1, c2) = (0, 1 - 1/3, 1/3), (1 - 2/3, 2/3, 0), (0, 1, 0)
// At the image's left and right edges the window is loaded from inside the row (columns clamp(ws, 0, W - 3) ...) and its pixels are
// re-selected after the table; only waves that hold such a lane run that code.  Images narrower than 3 pixels read byte by byte.
template <int CH> struct BlWindow {
    static constexpr int WORDS = CH == 3 ? 3 : 4;
    uint32_t w[3][WORDS];
    uint32_t mis[3];
};
template <int CH, bool ALIGNED>
__global__ __launch_bounds__(256, ALIGNED ? (CH == 3 ? 7 : 6) : 4) void bilinear_u8_kernel(AuxArgs a) {
    __shared__ float s_lut[256];
    __shared__ QEntry s_q[kQEntries];
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = a.W, H = a.H, OW = 3 * W, OH = 3 * H;
    const int chunks = (OW + 3) / 4, nb = (chunks + 63) / 64;   // 16-byte chunks per output row; blocks of 64 of them
    const int total = a.n * H * nb;                             // (the host checks that this fits)
    const int nw = (int)gridDim.x * 4;
    int wi = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (tid >> 6));
    // (image, input row, block) of this wave's item and the stride to the next, all wave-uniform
    int row = wi / nb, b = wi - row * nb, n = row / H, y = row - n * H;
    const int rstep = nw / nb, db = nw - rstep * nb, dn = rstep / H, dy = rstep - dn * H;
    typedef BlWindow<CH> Win;
    auto geometry = [&](int bb, int& cx, int& p0, int& ws) {
        cx = min(64 * bb + lane, chunks - 1);  // (lanes past the row's end redo its last chunk: same bytes to the same place, no branch)
        const int o0 = 4 * cx, i0 = (int)(__umulhi((uint32_t)o0, 0xAAAAAAABu) >> 1);  // o0 / 3, exact for every 32-bit o0 (round 5's 16-bit form was not beyond W = 32767)
        p0 = o0 - 3 * i0;
        ws = i0 - (p0 == 0 ? 1 : 0);
    };
    auto fetch = [&](int nn, int yy, int bb, Win& win) {
        int cx, p0, ws;
        geometry(bb, cx, p0, ws);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int gy = min(max(yy - 1 + r, 0), H - 1);
            // (pointer arithmetic on the kernel argument, not integers cast back: the loads stay global_load, whose returns are counted
            // in order -- a flat_load would make every wait a wait for the previous item's stores too)
            const uint8_t* rowp = (const uint8_t*)a.img + ((size_t)nn * H + gy) * W * CH;  // uniform
            if (W >= 3) {
                const uint32_t m = (uint32_t)(uintptr_t)rowp & 3u;
                const uint32_t t = m + (uint32_t)(min(max(ws, 0), W - 3) * CH);
                const uint32_t* src = (const uint32_t*)((rowp - m) + (t & ~3u));
                win.mis[r] = t & 3u;
                win.w[r][0] = src[0];
                win.w[r][1] = src[1];
                win.w[r][2] = src[2];  // (byte 8 of the window lies in it for every misalignment)
                if constexpr (CH == 4) win.w[r][3] = (t & 3u) ? src[3] : 0u;
            } else {  // W = 1 or 2: the window's pixels one by one, already in place
                win.mis[r] = 0;
#pragma unroll
                for (int i = 0; i < Win::WORDS; ++i) win.w[r][i] = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint8_t* q = rowp + min(max(ws + k, 0), W - 1) * CH;
#pragma unroll
                    for (int c = 0; c < 3; ++c) win.w[r][(CH * k + c) >> 2] |= (uint32_t)q[c] << (8 * ((CH * k + c) & 3));
                }
            }
        }
    };
    // The next window is requested as soon as this item's window has gone through the table -- into the same registers -- and consumed
    // (as far as the compiler's wait-count bookkeeping goes) at the END of the pass, after the item's stores and on a path without
    // branches: the wait placed there is "all but the last 3 memory operations".  Consumed at the loop's head it would be "all" -- the
    // first pass arrives there with nothing but loads in flight -- and every pass would sit out the write acknowledgements of the one
    // before.
    auto arrived = [](Win& w) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int i = 0; i < Win::WORDS; ++i) asm volatile("" : "+v"(w.w[r][i]));
        }
    };
    Win win;
    if (wi < total) fetch(n, y, b, win);
    s_lut[tid] = ((const float*)((const QEntry*)a.qtab + kQEntries))[tid];  // SrgbToLinear(byte / 255): lut_kernel
    for (int k = tid; k < kQEntries; k += 256) s_q[k] = ((const QEntry*)a.qtab)[k];
    arrived(win);
    __syncthreads();  // the tables: the only barrier
    while (wi < total) {
        const int cn = n, cy = y;
        int cx, p0, ws;
        geometry(b, cx, p0, ws);
        // the window's pixels, linear: [row][pixel] as (R, G) and B
        f32x2 wxy[3][3];
        float wz[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint32_t al[3];
            al[0] = __builtin_amdgcn_alignbyte(win.w[r][1], win.w[r][0], win.mis[r]);
            al[1] = __builtin_amdgcn_alignbyte(win.w[r][2], win.w[r][1], win.mis[r]);
            al[2] = __builtin_amdgcn_alignbyte(CH == 4 ? win.w[r][3] : 0u, win.w[r][2], win.mis[r]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int bi = CH * k + c;
                    v[c] = s_lut[(al[bi >> 2] >> (8 * (bi & 3))) & 0xffu];
                }
                wxy[r][k] = f32x2{v[0], v[1]};
                wz[r][k] = v[2];
            }
        }
        wi += nw;
        b += db;
        if (b >= nb) { b -= nb; ++y; }
        y += dy;
        if (y >= H) { y -= H; ++n; }
        n += dn;
        if (wi < total) fetch(n, y, b, win);
        if (W >= 3) {
            const int wl = min(max(ws, 0), W - 3);
            if (__builtin_amdgcn_ballot_w64(ws != wl) != 0) {  // a lane at the left or right edge: its pixels by clamped column
                const int s0 = min(max(ws, 0), W - 1) - wl, s1 = min(max(ws + 1, 0), W - 1) - wl, s2 = min(max(ws + 2, 0), W - 1) - wl;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x2 x0 = wxy[r][0], x1 = wxy[r][1], x2 = wxy[r][2];
                    const float z0 = wz[r][0], z1 = wz[r][1], z2 = wz[r][2];
                    wxy[r][0] = s0 == 0 ? x0 : (s0 == 1 ? x1 : x2);
                    wxy[r][1] = s1 == 0 ? x0 : (s1 == 1 ? x1 : x2);
                    wxy[r][2] = s2 == 0 ? x0 : (s2 == 1 ? x1 : x2);
                    wz[r][0] = s0 == 0 ? z0 : (s0 == 1 ? z1 : z2);
                    wz[r][1] = s1 == 0 ? z0 : (s1 == 1 ? z1 : z2);
                    wz[r][2] = s2 == 0 ? z0 : (s2 == 1 ? z1 : z2);
                }
            }
        }
        constexpr float T23 = 2.0f / 3.0f, T13 = 1.0f / 3.0f;
        const float t03 = p0 == 0 ? T23 : (p0 == 1 ? 0.0f : T13), u03 = 1.0f - t03;
        const float b1 = p0 == 0 ? 1.0f : (p0 == 1 ? T13 : T23), a1 = p0 == 0 ? 0.0f : 1.0f - b1;
        const float c0 = p0 == 1 ? 1.0f - T23 : 0.0f, c1 = p0 == 0 ? 1.0f - T13 : (p0 == 1 ? T23 : 1.0f), c2 = p0 == 0 ? T13 : 0.0f;
        uint32_t px4[3][4];
        // Two output pixels at a time -- (e0, e3), then (e1, e2) -- so that the B channel of the two travels as a packed pair like
        // (R, G) of each does; the expressions and their order are the ones above.
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            f32x2 hxyA[3], hxyB[3], hzAB[3];   // [row]: (R, G) of the pair's first and second pixel, B of both
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (pair == 0) {
                    hxyA[r] = u03 * wxy[r][0] + t03 * wxy[r][1];
                    hxyB[r] = u03 * wxy[r][1] + t03 * wxy[r][2];
                    hzAB[r] = u03 * f32x2{wz[r][0], wz[r][1]} + t03 * f32x2{wz[r][1], wz[r][2]};
                } else {
                    hxyA[r] = a1 * wxy[r][0] + b1 * wxy[r][1];
                    hxyB[r] = (c0 * wxy[r][0] + c1 * wxy[r][1]) + c2 * wxy[r][2];
                    hzAB[r] = f32x2{a1, c0} * wz[r][0] + f32x2{b1, c1} * wz[r][1];
                    hzAB[r].y = hzAB[r].y + c2 * wz[r][2];
                }
            }
            // vertical phases: t = 2/3 of (y - 1, y); 0 of (y, y + 1): (1 - 0) a + 0 b = a exactly; 1/3 of (y, y + 1)
            const f32x2 o0A = (1.0f - T23) * hxyA[0] + T23 * hxyA[1], o2A = (1.0f - T13) * hxyA[1] + T13 * hxyA[2];
            const f32x2 o0B = (1.0f - T23) * hxyB[0] + T23 * hxyB[1], o2B = (1.0f - T13) * hxyB[1] + T13 * hxyB[2];
            const f32x2 o0z = (1.0f - T23) * hzAB[0] + T23 * hzAB[1], o2z = (1.0f - T13) * hzAB[1] + T13 * hzAB[2];
            const int eA = pair == 0 ? 0 : 1, eB = pair == 0 ? 3 : 2;
            px4[0][eA] = quant_pixel(s_q, o0A.x, o0A.y, o0z.x);
            px4[1][eA] = quant_pixel(s_q, hxyA[1].x, hxyA[1].y, hzAB[1].x);
            px4[2][eA] = quant_pixel(s_q, o2A.x, o2A.y, o2z.x);
            px4[0][eB] = quant_pixel(s_q, o0B.x, o0B.y, o0z.y);
            px4[1][eB] = quant_pixel(s_q, hxyB[1].x, hxyB[1].y, hzAB[1].y);
            px4[2][eB] = quant_pixel(s_q, o2B.x, o2B.y, o2z.y);
        }
#pragma unroll
        for (int py = 0; py < 3; ++py) {
            uint32_t* rowo = (uint32_t*)a.out + ((size_t)cn * OH + 3 * cy + py) * OW;  // uniform
            uint32_t* dst = rowo + 4 * cx;
            if constexpr (ALIGNED) {
                *(u32x4*)dst = u32x4{px4[py][0], px4[py][1], px4[py][2], px4[py][3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (4 * cx + e < OW) dst[e] = px4[py][e];
            }
        }
        arrived(win);
    }
}

// downsample_net: a thread per output pixel, a workgroup walks tiles of 64 x 4 output pixels.  The 3x3 windows do not overlap, so
// no input sample is needed twice and nothing is staged: a thread reads the 3 x (3 px) of its window straight from global memory --
// consecutive lanes read consecutive 9 / 12 / 36-byte pieces of a row, so a wave's load covers one contiguous run.  A u8 piece starts
// at any byte: it is read as the aligned dwords that hold it (all of them hold a byte of the piece: nothing outside the image's words
// is touched) and shifted into place with v_alignbyte; the channel count is a template parameter, so that every sample's byte
// position is a constant.  The 9 samples are summed in the reference's order (rows, then columns) and divided by 9.  Rounds 3-4 staged
// the raw rows through LDS (two barriers and 27 ds_read_u8 per pixel on top of the 27 table look-ups): 51 us at 5760 x 3240.
constexpr int kDsTW = 64, kDsTH = 4;
struct __attribute__((packed, aligned(4))) DsF3 { float v[3]; };
template <bool IMG_U8, int CH> struct DsWindow {
    static constexpr int WORDS = IMG_U8 ? (CH == 3 ? 3 : 4) : 9;
    uint32_t w[3][WORDS];
    uint32_t mis[3];
};
template <bool IMG_U8, bool OUT_U8, int CH>
__global__ __launch_bounds__(256) void downsample_tile_kernel(AuxArgs a) {
    __shared__ float s_lut[IMG_U8 ? 256 : 1];
    __shared__ QEntry s_q[OUT_U8 ? kQEntries : 1];
    const int tid = threadIdx.x;
    if constexpr (IMG_U8) s_lut[tid] = ((const float*)((const QEntry*)a.qtab + kQEntries))[tid];
    if constexpr (OUT_U8) {
        const QEntry* q = (const QEntry*)a.qtab;
        for (int k = tid; k < kQEntries; k += 256) s_q[k] = q[k];
    }
    const int OH = a.H / 3, OW = a.W / 3;  // remainder rows / columns dropped (unpinned, see oracle)
    const int tiles_x = (OW + kDsTW - 1) / kDsTW, tiles_y = (OH + kDsTH - 1) / kDsTH;
    const int per_img = tiles_x * tiles_y;
    const long ntiles = (long)a.n * per_img;
    const int lx = tid & (kDsTW - 1), ly = tid / kDsTW;
    typedef DsWindow<IMG_U8, CH> Win;
    // output pixel of this thread in a tile (threads past the image's edge redo the edge's pixel: same bytes to the same place, and no
    // branch around the store -- see `arrived` in bilinear_u8_kernel) and its window, requested one tile ahead
    auto place = [&](long tile, int& n, int& ox, int& oy) {
        n = (int)(tile / per_img);
        const int tr = (int)(tile - (long)n * per_img), ty = tr / tiles_x;
        ox = min((tr - ty * tiles_x) * kDsTW + lx, OW - 1);
        oy = min(ty * kDsTH + ly, OH - 1);
    };
    auto fetch = [&](long tile, Win& win) {
        int n, ox, oy;
        place(tile, n, ox, oy);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if constexpr (IMG_U8) {
                const uint8_t* rowp = (const uint8_t*)a.img + (((size_t)n * a.H + 3 * oy + dy) * a.W) * CH;  // (global pointer arithmetic:
                const uint32_t m = (uint32_t)(uintptr_t)rowp & 3u;                                         //  see bilinear_u8_kernel)
                const uint32_t t = m + (uint32_t)(3 * ox * CH);
                const uint32_t* src = (const uint32_t*)((rowp - m) + (t & ~3u));
                win.mis[dy] = t & 3u;
                win.w[dy][0] = src[0];
                win.w[dy][1] = src[1];
                win.w[dy][2] = src[2];  // (byte 8 of the piece lies in it for every misalignment)
                if constexpr (CH == 4) win.w[dy][3] = win.mis[dy] ? src[3] : 0u;
            } else {
                const DsF3* src = (const DsF3*)((const float*)a.img + (((size_t)n * a.H + 3 * oy + dy) * a.W + 3 * ox) * 3);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const DsF3 v = src[dx];
#pragma unroll
                    for (int c = 0; c < 3; ++c) win.w[dy][3 * dx + c] = __float_as_uint(v.v[c]);
                }
            }
        }
    };
    auto arrived = [](Win& win) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int i = 0; i < Win::WORDS; ++i) asm volatile("" : "+v"(win.w[r][i]));
        }
    };
    Win cur, nxt;
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x, nxt);
    arrived(nxt);
    __syncthreads();  // the tables
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cur = nxt;
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, nxt);
        int n, ox, oy;
        place(tile, n, ox, oy);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            uint32_t al[3];
            if constexpr (IMG_U8) {  // the piece's bytes 0 .. 11 in place
                al[0] = __builtin_amdgcn_alignbyte(cur.w[dy][1], cur.w[dy][0], cur.mis[dy]);
                al[1] = __builtin_amdgcn_alignbyte(cur.w[dy][2], cur.w[dy][1], cur.mis[dy]);
                al[2] = __builtin_amdgcn_alignbyte(CH == 4 ? cur.w[dy][3] : 0u, cur.w[dy][2], cur.mis[dy]);
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v;
                    if constexpr (IMG_U8) {
                        const int b = CH * dx + c;
                        v = s_lut[(al[b >> 2] >> (8 * (b & 3))) & 0xffu];
                    } else {
                        v = srgb_to_linear_fast(__uint_as_float(cur.w[dy][3 * dx + c]));
                    }
                    acc[c] += v;
                }
        }
        const size_t op = ((size_t)n * OH + oy) * OW + ox;
        if constexpr (OUT_U8) {
            ((uint32_t*)a.out)[op] = quant_pixel(s_q, acc[0] / 9.0f, acc[1] / 9.0f, acc[2] / 9.0f);
        } else {
            DsF3 o;
#pragma unroll
            for (int c = 0; c < 3; ++c) o.v[c] = linear_to_srgb_fast(acc[c] / 9.0f);
            ((DsF3*)a.out)[op] = o;
        }
        arrived(nxt);
    }
}

}  // namespace

// The quantiser table of a context of one of these graphs (sr_create_graph): 255 step positions from the device's own
// powf, laid out by bucket on the host.  *d_tab: kQEntries dword entries, then the 256 floats of the input table
// (lut_kernel), in device memory (hipFree'd by sr_destroy).
hipError_t sr_aux_build_tables(void** d_tab) {
    *d_tab = nullptr;
    float* d_steps = nullptr;
    hipError_t e = hipMalloc((void**)&d_steps, 255 * sizeof(float));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(threshold_kernel, dim3(1), dim3(256), 0, nullptr, d_steps);
    float steps[255];
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(steps, d_steps, sizeof(steps), hipMemcpyDeviceToHost);
    (void)hipFree(d_steps);
    if (e != hipSuccess) return e;
    std::vector<QEntry> tab(kQEntries);
    auto bucket_lo = [](int b) {  // smallest float of bucket b (0 .. kQBuckets: the last is 1.0)
        const uint32_t bits = ((uint32_t)(kQExp0 << 7) + (uint32_t)b) << 16;
        float f;
        memcpy(&f, &bits, 4);
        return f;
    };
    // byte(l) = #{k : steps[k] <= l}.  For a bucket [lo, hi): base = #{steps < lo}; the one step inside it (lo <= step < hi), if any,
    // enters with its low 16 bits.  Entry 0: every l below 2^-13; entry kQBuckets + 1: every l from 1.0 up.
    for (int i = 1; i < 255; ++i)
        if (!(steps[i] > steps[i - 1])) return hipErrorUnknown;
    if (steps[0] < bucket_lo(0) || !(steps[254] < 1.0f)) return hipErrorUnknown;  // a step outside [2^-13, 1): the layout does not hold
    tab[0] = 0;
    tab[kQEntries - 1] = 255u << 16;
    int k = 0;
    for (int b = 0; b < kQBuckets; ++b) {
        const float lo = bucket_lo(b), hi = bucket_lo(b + 1);
        while (k < 255 && steps[k] < lo) ++k;
        uint32_t e = (uint32_t)k << 16;
        if (k < 255 && steps[k] < hi) {
            uint32_t sb;
            memcpy(&sb, &steps[k], 4);
            e += 0x10000u - (sb & 0xffffu);
            if (k + 1 < 255 && steps[k + 1] < hi) return hipErrorUnknown;  // two steps in one bucket: cannot happen (see the header)
        }
        tab[b + 1] = e;
    }
    e = hipMalloc(d_tab, tab.size() * sizeof(QEntry) + 256 * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d_tab, tab.data(), tab.size() * sizeof(QEntry), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lut_kernel, dim3(1), dim3(256), 0, nullptr, (float*)((QEntry*)*d_tab + tab.size()));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // (the launch's own stream: other streams of the process are not waited for)
    }
    if (e != hipSuccess) { (void)hipFree(*d_tab); *d_tab = nullptr; }
    return e;
}

// Workgroups for `units` equal pieces of work (a workgroup walks every grid-th piece) when `resident` workgroups fit on the chip at
// once: never more than that (a second round of a few would run with most of the chip idle), and of the few candidates above the
// minimum number of pieces per workgroup the one whose busiest CU -- ceil(grid / cus) workgroups -- gets clearly less work.  (1920x1080
// through bilinear_u8_kernel: 4 blocks per wave make 1553 workgroups = 7 on some CUs, 6 on the others, 112 blocks on the busiest; 5
// per wave make 1242 = 5 on each, 100.)
static int balanced_grid(long units, long resident, int cus) {
    const long per0 = std::max(1L, (units + resident - 1) / resident);
    long load0 = 0, best_load = -1, best_grid = 1;
    for (long per = per0; per < per0 + 8; ++per) {
        const long grid = (units + per - 1) / per, load = ((grid + cus - 1) / cus) * per;
        if (per == per0) load0 = load;
        if (best_load < 0 || load < best_load) { best_load = load; best_grid = grid; }
    }
    // (fewer, longer workgroups also mean fewer waves in flight: 5760x3240 lost 4 % to a 0.5 % better balance, so it has to pay)
    return (int)(100 * best_load <= 97 * load0 ? best_grid : (units + per0 - 1) / per0);
}

hipError_t sr_launch_aux(int graph, const AuxArgs& a, bool img_u8, bool out_u8, hipStream_t s) {
    if (img_u8 != out_u8) return hipErrorInvalidValue;
    if (out_u8 && !a.qtab) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (graph == 1) {
        const long tiles = (long)a.n * ((a.W + kBlTW - 1) / kBlTW) * ((a.H + kBlTH - 1) / kBlTH);
        if (tiles == 0) return hipSuccess;
        // a few workgroups per CU walk the tiles -- the tables are read once per workgroup, not once per tile -- and every workgroup
        // gets the same number of them (+-1); never more workgroups than are resident at once (a second round of a few would
        // run with most of the chip idle)
        const bool aligned = a.W % 4 == 0 && ((uintptr_t)a.out & 15) == 0;
        if (img_u8) {  // bilinear_u8_kernel: a wave per block of 64 chunks of one input row, every wave the same number of blocks (+-1)
            if (a.img_ch != 3 && a.img_ch != 4) return hipErrorInvalidValue;
            const long items = (long)a.n * a.H * (((3L * a.W + 3) / 4 + 63) / 64);
            if (items >= (1L << 31) - (1L << 20)) return hipErrorInvalidValue;  // (a wave's item counter runs up to items + 4 x grid before it stops)
            const void* fn = a.img_ch == 3 ? (aligned ? (const void*)bilinear_u8_kernel<3, true> : (const void*)bilinear_u8_kernel<3, false>)
                                           : (aligned ? (const void*)bilinear_u8_kernel<4, true> : (const void*)bilinear_u8_kernel<4, false>);
            int resident = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, fn, 256, 0) != hipSuccess || resident < 1) resident = 4;
            const int grid = balanced_grid((items + 3) / 4, (long)resident * cus, cus);  // (a workgroup: 4 waves, 4 blocks at a time)
            if (a.img_ch == 3) {
                if (aligned) hipLaunchKernelGGL((bilinear_u8_kernel<3, true>), dim3(grid), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((bilinear_u8_kernel<3, false>), dim3(grid), dim3(256), 0, s, a);
            } else {
                if (aligned) hipLaunchKernelGGL((bilinear_u8_kernel<4, true>), dim3(grid), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((bilinear_u8_kernel<4, false>), dim3(grid), dim3(256), 0, s, a);
            }
            return hipGetLastError();
        }
        const void* fn = aligned ? (const void*)bilinear_tile_kernel<true> : (const void*)bilinear_tile_kernel<false>;
        int resident = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, fn, 256, 0) != hipSuccess || resident < 1) resident = 4;
        const int grid = balanced_grid(tiles, (long)std::min(resident, 6) * cus, cus);
        if (aligned) hipLaunchKernelGGL((bilinear_tile_kernel<true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((bilinear_tile_kernel<false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        const int OH = a.H / 3, OW = a.W / 3;
        const long tiles = (long)a.n * ((OW + kDsTW - 1) / kDsTW) * ((OH + kDsTH - 1) / kDsTH);
        if (tiles == 0) return hipSuccess;
        const int grid = balanced_grid(tiles, 8L * cus, cus);  // (8 workgroups per CU: 3 / 4 / 6 measured 10-60 % slower at 5760x3240)
        if (img_u8 && a.img_ch == 3) hipLaunchKernelGGL((downsample_tile_kernel<true, true, 3>), dim3(grid), dim3(256), 0, s, a);
        else if (img_u8 && a.img_ch == 4) hipLaunchKernelGGL((downsample_tile_kernel<true, true, 4>), dim3(grid), dim3(256), 0, s, a);
        else if (img_u8) return hipErrorInvalidValue;
        else hipLaunchKernelGGL((downsample_tile_kernel<false, false, 3>), dim3(grid), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
