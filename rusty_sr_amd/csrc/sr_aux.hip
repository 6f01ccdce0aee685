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
//    lookup is one ds_read_b64, one compare and one add instead of a powf, and gives the powf's answer;
//  * f32 in / out: powf(x, p) = exp2(p * log2(x)) on v_log_f32 / v_exp_f32 (1 ulp each; measured 1.8e-7 absolute
//    against the oracle's libm, inside the 1e-5 the tests ask of these graphs and far inside north_star's 1e-4);
//  * every input sample is linearised once per tile (staged, edge-replicated, in LDS as [pixel][4] f32), the
//    interpolation runs in the reference's operation order ((1-t) a + t b, rows then columns), one workgroup owns an
//    output tile and a thread stores 16 bytes at a time (4 RGBA pixels / 4 floats) whenever the rows are 16-byte aligned
//    (W % 4 == 0), dwords otherwise.
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
__device__ __forceinline__ uint32_t quant_lookup(const QEntry* tab, float l) {
    const uint32_t bits = __float_as_uint(l);
    const int idx = min(max(((int)bits >> 16) - (kQExp0 << 7) + 1, 0), kQEntries - 1);  // (arithmetic shift: negative values land below 0)
    return (tab[idx] + (bits & 0xffffu)) >> 16;
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

// One work item = one 16-byte chunk column (4 RGBA pixels / 4 floats) of the THREE output rows of one input row y: the
// horizontal interpolation of input rows y-1, y, y+1 is done once and shared by the three output rows (vertical phases
// t = 2/3 of (y-1, y), 0 of (y, y+1), 1/3 of (y, y+1)), in the expression and order of the one-thread-per-pixel kernel
// this replaces: ra = (1-tx) v00 + tx v01, rb = ..., out = (1-ty) ra + ty rb.  Consecutive lanes own consecutive chunks
// of a row, so every store instruction writes one contiguous run.
template <bool IMG_U8, bool OUT_U8, bool ALIGNED>
__global__ __launch_bounds__(256) void bilinear_tile_kernel(AuxArgs a) {
    __shared__ __attribute__((aligned(16))) float s_lin[kBlNPIX * 4];
    __shared__ float s_lut[IMG_U8 ? 256 : 1];
    __shared__ QEntry s_q[OUT_U8 ? kQEntries : 1];
    const int tid = threadIdx.x;
    if constexpr (IMG_U8) s_lut[tid] = ((const float*)((const QEntry*)a.qtab + kQEntries))[tid];  // SrgbToLinear(byte / 255): lut_kernel
    if constexpr (OUT_U8) {
        const QEntry* q = (const QEntry*)a.qtab;
        for (int k = tid; k < kQEntries; k += 256) s_q[k] = q[k];
    }
    const int tiles_x = (a.W + kBlTW - 1) / kBlTW, tiles_y = (a.H + kBlTH - 1) / kBlTH;
    const long ntiles = (long)a.n * tiles_x * tiles_y;
    const int OW = 3 * a.W, OH = 3 * a.H;
    constexpr int EPP = OUT_U8 ? 1 : 3;               // 4-byte elements per output pixel
    constexpr int CPR = 3 * kBlTW * EPP / 4;          // chunks per output row of a full tile
    // The input pixels of a tile travel global -> registers -> (table) -> LDS.  The loads of the NEXT tile are issued before this tile's
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
            const size_t gp = ((size_t)n * a.H + gy) * a.W + gx;
            if constexpr (IMG_U8) {
                const uint8_t* q = (const uint8_t*)a.img + gp * a.img_ch;
                raw[k][0] = q[0]; raw[k][1] = q[1]; raw[k][2] = q[2];
            } else {
                const uint32_t* q = (const uint32_t*)a.img + gp * 3;
                raw[k][0] = q[0]; raw[k][1] = q[1]; raw[k][2] = q[2];
            }
        }
    };
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tiles_x * tiles_y)), tr = (int)(tile - (long)n * tiles_x * tiles_y);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int x0 = tx * kBlTW, y0 = ty * kBlTH;
        __syncthreads();  // the previous tile's readers are done (and the tables are in place)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = tid + 256 * k;
            if (p < kBlNPIX) {
                f32x4 v;
                if constexpr (IMG_U8) v = f32x4{s_lut[raw[k][0]], s_lut[raw[k][1]], s_lut[raw[k][2]], 0.f};
                else v = f32x4{srgb_to_linear_fast(__uint_as_float(raw[k][0])), srgb_to_linear_fast(__uint_as_float(raw[k][1])),
                               srgb_to_linear_fast(__uint_as_float(raw[k][2])), 0.f};
                *(f32x4*)&s_lin[p * 4] = v;
            }
        }
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        __syncthreads();
        const int tw = min(kBlTW, a.W - x0), th = min(kBlTH, a.H - y0);  // this tile's own input pixels
        const int elems = 3 * tw * EPP;                                  // 4-byte elements per output row of this tile
        for (int q = tid; q < CPR * kBlTH; q += 256) {
            const int y = q / CPR, cx = q - y * CPR;
            if (y >= th || 4 * cx >= elems) continue;
            const float* row = s_lin + (y * kBlTWH) * 4;   // input row y - 1 (tile row y); y and y + 1 follow
            float h[3][4];                                  // horizontally interpolated: [input row][element of the chunk]
            if constexpr (OUT_U8) {
                uint32_t px4[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int xa; float txv;
                    phase(4 * cx + e, xa, txv);
                    // (R, G) as a packed pair, B alone: the fourth lane of the staged pixel is padding
                    f32x2 hxy[3]; float hz[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const f32x4 v0 = *(const f32x4*)(row + (r * kBlTWH + xa) * 4), v1 = *(const f32x4*)(row + (r * kBlTWH + xa) * 4 + 4);
                        hxy[r] = (1.0f - txv) * f32x2{v0.x, v0.y} + txv * f32x2{v1.x, v1.y};
                        hz[r] = (1.0f - txv) * v0.z + txv * v1.z;
                    }
#pragma unroll
                    for (int py = 0; py < 3; ++py) {
                        f32x2 oxy; float oz;
                        if (py == 1) {  // t = 0: (1 - 0) a + 0 b = a exactly (every value here is finite and >= 0)
                            oxy = hxy[1]; oz = hz[1];
                        } else {
                            const float tyv = py == 0 ? 2.0f / 3.0f : 1.0f / 3.0f;
                            oxy = (1.0f - tyv) * hxy[py == 0 ? 0 : 1] + tyv * hxy[py == 0 ? 1 : 2];
                            oz = (1.0f - tyv) * hz[py == 0 ? 0 : 1] + tyv * hz[py == 0 ? 1 : 2];
                        }
                        px4[py][e] = 0xff000000u | quant_lookup(s_q, oxy.x) | (quant_lookup(s_q, oxy.y) << 8) | (quant_lookup(s_q, oz) << 16);
                    }
                }
                (void)h;
#pragma unroll
                for (int py = 0; py < 3; ++py) {
                    uint32_t* dst = (uint32_t*)a.out + ((size_t)n * OH + 3 * (y0 + y) + py) * OW + 3 * x0 + 4 * cx;
                    if (ALIGNED || 4 * cx + 4 <= elems) {
                        if constexpr (ALIGNED) *(u32x4*)dst = u32x4{px4[py][0], px4[py][1], px4[py][2], px4[py][3]};
                        else { dst[0] = px4[py][0]; dst[1] = px4[py][1]; dst[2] = px4[py][2]; dst[3] = px4[py][3]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (4 * cx + e < elems) dst[e] = px4[py][e];
                    }
                }
            } else {
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
    // output pixel of this thread in a tile (-1: outside the image) and its window, requested one tile ahead
    auto place = [&](long tile, int& n, int& ox, int& oy) {
        n = (int)(tile / per_img);
        const int tr = (int)(tile - (long)n * per_img), ty = tr / tiles_x;
        ox = (tr - ty * tiles_x) * kDsTW + lx;
        oy = ty * kDsTH + ly;
        return ox < OW && oy < OH;
    };
    auto fetch = [&](long tile, Win& win) {
        int n, ox, oy;
        if (!place(tile, n, ox, oy)) return;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const size_t px = ((size_t)n * a.H + 3 * oy + dy) * a.W + 3 * ox;
            if constexpr (IMG_U8) {
                const uintptr_t addr = (uintptr_t)a.img + px * CH;
                const uint32_t* src = (const uint32_t*)(addr & ~(uintptr_t)3);
                win.mis[dy] = (uint32_t)(addr & 3);
                win.w[dy][0] = src[0];
                win.w[dy][1] = src[1];
                win.w[dy][2] = src[2];  // (byte 8 of the piece lies in it for every misalignment)
                if constexpr (CH == 4) win.w[dy][3] = win.mis[dy] ? src[3] : 0u;
            } else {
                const DsF3* src = (const DsF3*)((const float*)a.img + px * 3);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const DsF3 v = src[dx];
#pragma unroll
                    for (int c = 0; c < 3; ++c) win.w[dy][3 * dx + c] = __float_as_uint(v.v[c]);
                }
            }
        }
    };
    Win cur, nxt;
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x, nxt);
    __syncthreads();  // the tables
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        cur = nxt;
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, nxt);
        int n, ox, oy;
        if (!place(tile, n, ox, oy)) continue;
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
            uint32_t pk = 0xff000000u;
#pragma unroll
            for (int c = 0; c < 3; ++c) pk |= quant_lookup(s_q, acc[c] / 9.0f) << (8 * c);
            ((uint32_t*)a.out)[op] = pk;
        } else {
            DsF3 o;
#pragma unroll
            for (int c = 0; c < 3; ++c) o.v[c] = linear_to_srgb_fast(acc[c] / 9.0f);
            ((DsF3*)a.out)[op] = o;
        }
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

hipError_t sr_launch_aux(int graph, const AuxArgs& a, bool img_u8, bool out_u8, hipStream_t s) {
    if (img_u8 != out_u8) return hipErrorInvalidValue;
    if (out_u8 && !a.qtab) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (graph == 1) {
        const long tiles = (long)a.n * ((a.W + kBlTW - 1) / kBlTW) * ((a.H + kBlTH - 1) / kBlTH);
        if (tiles == 0) return hipSuccess;
        // a few workgroups per CU walk the tiles -- the tables are read once per workgroup, not once per tile -- and every workgroup
        // gets the same number of them (+-1)
        const long per = (tiles + 6L * cus - 1) / (6L * cus);
        const int grid = (int)((tiles + per - 1) / per);
        const bool aligned = a.W % 4 == 0 && ((uintptr_t)a.out & 15) == 0;
        if (img_u8) {
            if (aligned) hipLaunchKernelGGL((bilinear_tile_kernel<true, true, true>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((bilinear_tile_kernel<true, true, false>), dim3(grid), dim3(256), 0, s, a);
        } else {
            if (aligned) hipLaunchKernelGGL((bilinear_tile_kernel<false, false, true>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((bilinear_tile_kernel<false, false, false>), dim3(grid), dim3(256), 0, s, a);
        }
    } else {
        const int OH = a.H / 3, OW = a.W / 3;
        const long tiles = (long)a.n * ((OW + kDsTW - 1) / kDsTW) * ((OH + kDsTH - 1) / kDsTH);
        if (tiles == 0) return hipSuccess;
        const long per = (tiles + 8L * cus - 1) / (8L * cus);  // (8 workgroups per CU: 3 / 4 / 6 measured 10-60 % slower at 5760x3240)
        const int grid = (int)((tiles + per - 1) / per);
        if (img_u8 && a.img_ch == 3) hipLaunchKernelGGL((downsample_tile_kernel<true, true, 3>), dim3(grid), dim3(256), 0, s, a);
        else if (img_u8 && a.img_ch == 4) hipLaunchKernelGGL((downsample_tile_kernel<true, true, 4>), dim3(grid), dim3(256), 0, s, a);
        else if (img_u8) return hipErrorInvalidValue;
        else hipLaunchKernelGGL((downsample_tile_kernel<false, false, 3>), dim3(grid), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
