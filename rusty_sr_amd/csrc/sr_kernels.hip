// sr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the rusty_sr conv stack.
//
// What is computed (reference src/network.rs:27-72, semantics SURVEY.md 8(a)):
//   f  = BeLU(conv0(x)                       + b)   stage 0  (network.rs:33-35)
//   l1 = BeLU(conv1(f)                       + b)   stage 1  (:60,50,54)
//   l2 = BeLU(conv2(f)+conv5(l1)             + b)   stage 2  (:61,65,51,55)
//   l3 = BeLU(conv3(f)+conv6(l1)+conv8(l2)   + b)   stage 3  (:62,66,69,52,56)
//   e  =      conv7(l1)+conv9(l2)+conv10(l3) + b    stage 4  (:67,70,72,38)
//   out[3y+dy][3x+dx][c] = bilinear3(x) + e[y][x][(dy*3+dx)*3+c]   (:27,39)
//
// How (MI355X-first, nothing here is a translation of alumina's CPU loops):
//  * each stage is ONE implicit-GEMM kernel: M = 32 pixels of a tile row,
//    N = 32 output channels, K = concatenation of every tap of every source
//    feeding the node (the reference's "ops accumulate into a node" becomes
//    K-concatenation; accumulators never leave registers).  Even the bilinear
//    x3 residual (LinearInterp, network.rs:27) is K-concatenated into the last
//    stage: it is a fixed-weight 3x3 convolution of the edge-replicated input
//    onto the 27 expand channels;
//  * v_mfma_f32_32x32x2_f32 -- exact f32 (bitwise an fmaf chain), so parity with
//    the f32 CPU path holds to rounding-order noise (~1e-6), far inside 1e-4;
//  * MEASURED on MI355X (scripts/ubench_mfma.hip): the f32-input MFMA shares the
//    SIMD's vector ALU -- a co-resident wave's VALU instruction costs ~10 cycles
//    of MFMA time, nothing overlaps.  So the kernels are written to execute as
//    few VALU instructions as possible: tile staging uses wave-uniform (SGPR) row
//    bases + immediates, LDS operand addresses are immediates off one per-lane
//    base, weights arrive by LDS-DMA, the epilogue stores at immediate offsets;
//  * a workgroup (4 waves) owns a TH x 32 pixel tile; the source tile + halo is
//    staged in LDS in a channel-group-planar layout [cin/4][pixel][4 f32] so
//    that every A-operand fetch is one conflict-free ds_read_b128;
//  * weights are pre-packed on the host into 4 KB chunks, one per STEP (two taps x 16 input channels), in the
//    exact order the B-operand ds_read_b128 wants, and streamed by LDS-DMA through a 5-slot ring four steps
//    ahead of use, one barrier per step;
//  * stages 1-4 run the PIPE form (conv_stage_pipe_kernel: half tiles double-buffered in LDS, gather DMA of the
//    next half / the next tile's first half spread under the current half's MFMAs, persistent, 8-row tiles and,
//    at the end of every XCD's queue, 4-row tiles).  The FIRST form (conv_stage_kernel: whole source tile
//    resident, one tile class per launch) shares the matrix loops, the step order and the weight chunks, is
//    bit-identical, and is kept as the in-library cross-check (sr_set_experiment "pipe" = "none");
//  * LDS per workgroup 76-78 KB -> 2 workgroups per CU;
//  * a second arithmetic mode (split-half: activations and weights as hi + lo/2048 half pairs, three
//    v_mfma_f32_32x32x16_f16 per product, f32 accumulation) runs the same structure on the f16 matrix cores;
//  * bias + BeLU (or bias + depth-to-space [+ u8 RGBA quantisation]) are fused
//    into the epilogue: no elementwise kernel exists.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "sr_kernels.h"

// The reference CPU path (rustc/LLVM) never fuses a*b+c; neither may we, or the two
// instantiations of one kernel can differ in the last bit (packed-math vs fma selection).
#pragma clang fp contract(off)

typedef float f32x2 __attribute__((ext_vector_type(2)));
// A pair of values worked on with TWO scalar instructions per operation.  The split-half mode's epilogues run beside the partner wave's f16
// MFMAs, where a packed f32 instruction (v_pk_fma_f32 ...) costs more than the two scalar ones it replaces (MI355X_MICROARCH.md "price of
// one filler beside MFMAs"; profiles/r6_ab_split_transposed.txt: -1.2 % at 1080p).  The file is built with -fno-slp-vectorize so that the
// compiler does not pack them again.  (The exact mode keeps f32x2: there every vector instruction is paid in f32-MFMA time, so fewer is better.)
struct f32p { float x, y; };
__device__ __forceinline__ f32p operator+(f32p a, f32p b) { return f32p{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32p operator-(f32p a, f32p b) { return f32p{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32p operator*(f32p a, f32p b) { return f32p{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32p p_fma(f32p a, f32p b, f32p c) { return f32p{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
// 8 bytes of a split-half map: four hi (or lo) halves of one pixel.  (Plain stores: non-temporal and write-through ones were measured again after
// the stores became whole 16-byte cells per lane pair -- 1.478 -> 1.506 / 1.496 ms, profiles/r6_ab_split_transposed.txt.)
__device__ __forceinline__ void store_map8(char* p, uint32_t a, uint32_t b) { *(uint2*)p = make_uint2(a, b); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Feature maps in HBM.  Exact-f32 maps are PIXEL-MAJOR: [y][x][128 B].  Split-half maps are ROW-PLANAR: [y][c][x][16 B], c = the eight
// 16-byte channel groups of a pixel (4 groups of 8 hi halves, then their 4 lo groups), one row of one group = pitch x 16 B contiguous.
// A tile gather (one LDS-DMA instruction = one channel group of 64 consecutive tile pixels) then reads two or three contiguous
// runs of up to 576 B -- ~11 cache lines, every byte used -- instead of 64 lines of which it uses 16 B each and which the other
// three waves' and the other half's instructions fetch again.  The split-half mode is bound by the BOARD'S POWER, not by issue slots
// (scripts/experiments/ubench_split_floor.hip, profiles/r5_ubench_split_floor.txt: the same step loop sustains 1 170 TFLOP/s with
// line-per-lane gathers and 1 269 with contiguous ones at the same 1 370 W), so what the memory system moves per MFMA is what counts.
// The exact-f32 mode runs at the full clock and keeps the layout whose stores are whole 128-B lines.  Measured, interleaved A/B at 1080p
// (profiles/r5_ab_planar.txt): 1.683 -> 1.625 ms per frame (stages 2 / 3 / 4: -4 / -4.6 / -6 %).  (The pixel-major split maps, the
// 32x32x16 form of stages 1-3 and the f32-MFMA stage 0 of the split-half mode were compile-time variants until round 6; their A/B
// records are profiles/r5_ab_planar.txt, r5_ab_mfma16.txt, r5_ab_conv0_split.txt.)
template <int PREC>
constexpr bool kPlanar = PREC == 1;
// Stages 1-3 of the split-half mode run their products on v_mfma_f32_16x16x32_f16 (half_steps_h16 below) instead of 32x32x16: the same
// FLOPs per cycle, but 17 % more of them per watt -- a stream of random operands sustains 2 090 TFLOP/s in that shape against 1 780 in
// the other at the board's 1.37 kW (profiles/r5_ubench_split_floor.txt), and this mode is bound by exactly that.  The last stage
// (whose residual taps and depth-to-space epilogue are written for 32x32 accumulators) keeps half_steps_h.
template <int PREC, bool FINAL>
constexpr bool kH16 = PREC == 1 && !FINAL;

namespace {

constexpr int kThreads = 256;
constexpr int kTW = 32;            // tile width = one MFMA M-tile of pixels
constexpr int kChunkFloats = 1024; // one tap: 32 cin x 32 cout
constexpr int kRingSlots = 5;      // 4 KB weight chunks: one being read, up to four in flight
constexpr int kRingAhead = 4;      // tap t requests the chunk of tap t + kRingAhead
constexpr int kRingBytes = kRingSlots * 4096;

__device__ __forceinline__ float belu(float v, float beta) {
    // alumina BeLU (network.rs:35,54-56): beta*x + sqrt(x*x+1) - 1, same operation
    // order as the reference, no contraction.  sqrt is the
    // hardware v_sqrt_f32 (1 ulp; argument >= 1): the correctly-rounded expansion
    // costs ~12 more VALU instructions per value, each ~10 cycles of f32-MFMA time.
    return __fadd_rn(__fadd_rn(__fmul_rn(beta, v), __builtin_amdgcn_sqrtf(__fadd_rn(__fmul_rn(v, v), 1.0f))), -1.0f);
}

// Two values per instruction (v_pk_mul_f32 / v_pk_add_f32): every VALU instruction
// of the epilogue is paid for in f32-MFMA time, so halve their number.
__device__ __forceinline__ f32x2 belu2(f32x2 v, float beta) {
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 t = v * v + one;
    const f32x2 s = {__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
    const f32x2 b = {beta, beta};
    return (b * v + s) - one;
}

// The split-half mode's epilogues: the node's bias is folded into the main accumulators' initial value, and the arithmetic uses explicit
// FMAs -- v = accm + accx / 2048 and BeLU are 4 operations + 1 v_sqrt_f32 per value instead of 8 + 1.  (This mode is
// held to the 1e-4 bar against the oracle, not to bit-identity with the exact mode, whose belu2 keeps the reference's unfused order; every
// form and band of THIS mode runs the same code and stays bit-identical to the others.)
__device__ __forceinline__ f32p split_value(f32p accm, f32p accx) {
    return p_fma(accx, f32p{1.0f / 2048.0f, 1.0f / 2048.0f}, accm);
}
// (a slope per value: the two values are two CHANNELS of one pixel -- accumulators computed with the weights as the MFMA's A operand)
__device__ __forceinline__ f32p belu2_fused2(f32p v, f32p beta) {
    const f32p one = {1.0f, 1.0f};
    const f32p t = p_fma(v, v, one);
    const f32p s = {__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
    return p_fma(beta, v, s) - one;
}

// BeLU(acc + bias) of one 32x32 accumulator tile -> NHWC rows at base + row*32 floats
// (row pairs r, r+1 are adjacent pixels); immediate-offset stores only.
__device__ __forceinline__ void store_belu_tile(float* base, const f32x16& acc, float bias, float beta) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 bb = {bias, bias};
        const f32x2 o = belu2(f32x2{acc[r], acc[r + 1]} + bb, beta);
        const int row = (r & 3) + 8 * (r >> 2);
        base[row * 32] = o.x;
        base[(row + 1) * 32] = o.y;
    }
}

// The same tile row through LDS (pipe form, round 6): the 32 store instructions of a tile and wave are what the epilogue costs (1.6 % of a stage
// in issue alone, profiles/r6_ab_store_shadow.txt), and a lane cannot hold four consecutive channels of a pixel without paying for it elsewhere.
// So the wave writes the row's 32 pixels x 128 bytes into a 4 KB scratch of its own in LDS as it produces them (lane = channel: sixteen
// conflict-free ds_write_b32), reads them back linearly (four ds_read_b128) and leaves with FOUR global_store_dwordx4, each 1 KB contiguous --
// a tile row of a pixel-major map is contiguous in memory.  `stage`: this wave's scratch; `grow`: the row's first pixel in the map.
__device__ __forceinline__ void store_belu_tile_lds(float* grow, float* stage, const f32x16& acc, float bias, float beta, int lane) {
    float* sp = stage + (4 * (lane >> 5)) * 32 + (lane & 31);  // pixel 4 h + row, channel i
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 bb = {bias, bias};
        const f32x2 o = belu2(f32x2{acc[r], acc[r + 1]} + bb, beta);
        const int row = (r & 3) + 8 * (r >> 2);
        sp[row * 32] = o.x;
        sp[(row + 1) * 32] = o.y;
    }
    const f32x4* lp = (const f32x4*)stage + lane;
    const f32x4 v0 = lp[0], v1 = lp[64], v2 = lp[128], v3 = lp[192];
    f32x4* gp = (f32x4*)grow + lane;
    gp[0] = v0; gp[64] = v1; gp[128] = v2; gp[192] = v3;
}

// XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each
// XCD one contiguous run of tiles so neighbouring tiles (which share halo rows
// and columns) hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ int tile_div(int t, TileDiv d) {  // see make_tile_div
    return (int)((__umulhi((uint32_t)t, d.m) + (uint32_t)t) >> d.s);
}

// Tile id within a tile class -> (image, tile column, tile row): see TileGrid.  Scalar ALU only.
__device__ __forceinline__ void tile_coords(const TileGrid& g, int tiles_x, int t, int& n, int& tx, int& ty) {
    const int tpi = tiles_x * g.tiles_y;
    n = tile_div(t, g.div_tpi);
    const int r = t - n * tpi;
    if (g.bw == 0) {
        ty = tile_div(r, g.div_tx);
        tx = r - ty * tiles_x;
    } else {
        int b = tile_div(r, g.div_blk);
        if (b > g.nfull) b = g.nfull;
        const int rr = r - b * g.bw * g.tiles_y;
        const bool full = b < g.nfull;
        const int w = full ? g.bw : tiles_x - g.nfull * g.bw;
        ty = full ? tile_div(rr, g.div_bw) : tile_div(rr, g.div_rem);
        tx = b * g.bw + (rr - ty * w);
    }
}

__device__ __forceinline__ float load_img(const void* img, int img_ch, bool u8, size_t px, int c) {
    // img_to_data (reference main.rs:170): u8 / 255 (true division), alpha dropped
    if (u8) return __fdiv_rn((float)((const uint8_t*)img)[px * img_ch + c], 255.0f);
    return ((const float*)img)[px * 3 + c];
}

// ---- split-half ("f16x3") representation -----------------------------------
// A value v is carried as two halves: hi = half(v) (round-toward-zero is as good as
// any rounding here: lo absorbs the residual exactly) and lo = half((v - hi) * 2048).
// v = hi + lo/2048 to ~2^-22 relative.  A product v*w is then
// hi_v*hi_w + (hi_v*lo_w + lo_v*hi_w)/2048 (the lo*lo term is 2^-24): three
// v_mfma_f32_32x32x16_f16 on the real matrix cores with f32 accumulation, instead
// of 8 f32 MFMAs on the vector ALU.  Subnormal halves need no special care: the f16
// MFMA does not flush them on gfx950 in the default kernel mode
// (scripts/probe_f16_denorm.hip), and lo is pre-scaled out of that range anyway.
constexpr float kLoScale = 2048.0f;
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));  // what v_cvt_pkrtz_f16_f32 returns

// Split two f32 values: packed hi halves / packed lo halves (2 x v_cvt_pkrtz, 2 x v_cvt_f32_f16,
// two subtractions, two multiplications).
__device__ __forceinline__ void split_half2(f32p v, uint32_t& hi2, uint32_t& lo2) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const f32p hf = {(float)h.x, (float)h.y};
    const f32p r = (v - hf) * f32p{kLoScale, kLoScale};
    hi2 = __builtin_bit_cast(uint32_t, h);
    lo2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r.x, r.y));
}

// Domain of the split-half mode: a value is carried as hi + lo / 2048 with hi a HALF, so it must be finite and below 65504 in
// magnitude -- v_cvt_pkrtz saturates there, silently.  Every producer of split values therefore keeps the largest hi bit pattern it has
// made (sign stripped; one v_and + one v_pk_max_u16 per PAIR of values: infinities and NaNs are the largest patterns of all) and, if a
// half reached 0x7bff (65504), raises the context's domain flag -- a word of mapped host memory (sr_api.cpp: host-pointer calls then
// recompute in exact f32, device-pointer calls leave it to sr_check_domain).  Natural images stay below 100.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void domain_track(uint32_t& dom, uint32_t hi2) {
    dom = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, dom), __builtin_bit_cast(u16x2, hi2 & 0x7fff7fffu)));
}
__device__ __forceinline__ void domain_report(uint32_t dom, int* flag) {
    if (((dom & 0xffffu) >= 0x7bffu) | ((dom >> 16) >= 0x7bffu)) *(volatile int*)flag = 1;
}

// BeLU(acc) of one 32x32 tile -> split-half pairs in the row-planar map (kPlanar).  The tile is computed with the WEIGHT fragment as the
// MFMA's A operand (the transposed product: the two fragments have the same register shape): lane (i, h) holds, of pixel i of the tile row, output channels 8 j + 4 h + (0..3) in registers 4 j + (0..3).  Two channels of a
// pixel pack into a dword with no lane exchange, four are the 8 bytes 8 h .. 8 h + 7 of the pixel's 16-byte cell in channel group j: one
// 8-byte store of hi halves and one of lo halves per group, lanes i and i + 32 completing the cell, a wave writing 512 contiguous bytes.
// `base`: the lane's pixel in the row of channel group 0, + 8 h; groups are `group_stride` bytes apart, the lo groups `lo_off` further on.
__device__ __forceinline__ void store_belu_tile_split_cr(char* base, const f32x16& accm, const f32x16& accx, const f32x4 (&beta)[4], bool write,
                                                         long group_stride, long lo_off, uint32_t& dom) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * j;
        const f32p v01 = belu2_fused2(split_value(f32p{accm[r], accm[r + 1]}, f32p{accx[r], accx[r + 1]}), f32p{beta[j][0], beta[j][1]});  // (the bias is in accm)
        const f32p v23 = belu2_fused2(split_value(f32p{accm[r + 2], accm[r + 3]}, f32p{accx[r + 2], accx[r + 3]}), f32p{beta[j][2], beta[j][3]});
        uint32_t h01, l01, h23, l23;
        split_half2(v01, h01, l01);
        split_half2(v23, h23, l23);
        domain_track(dom, h01);
        domain_track(dom, h23);
        if (write) {
            store_map8(base + j * group_stride, h01, h23);
            store_map8(base + j * group_stride + lo_off, l01, l23);
        }
    }
}

// Store the 16 accumulator rows of one 32x32 MFMA tile at `base + row*stride`
// (row = (r&3) + 8*(r>>2); the lane's +4*h is already in `base`): compile-time
// offsets, so each store is one instruction with an immediate.
template <typename F>
__device__ __forceinline__ void for_each_acc_row(F&& f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) f(r, (r & 3) + 8 * (r >> 2));
}

}  // namespace

// An address the hardware takes from SGPRs (LDS-DMA base, the origin of a tile).  Every caller passes a wave-uniform pointer; this spells it out for the
// cases where the compiler cannot prove it (folds away where it can).
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// Stage 0: conv0 5x5 3->32 + bias + BeLU.  The x tile (with halo) sits in LDS as [pixel][3 channels]; along a
// kernel row the 5 taps x 3 channels are 15 CONSECUTIVE floats from pixel i on, so K is packed per kernel row:
// 15 -> 16 slots = 8 MFMAs (K = 2 each) instead of 5 taps x (3 -> 4 channels) = 10.  40 MFMAs per tile row.
// ---------------------------------------------------------------------------
template <int TH, bool IMG_U8>
__global__ __launch_bounds__(kThreads, 2) void conv0_kernel(Conv0Args a) {
    constexpr int T = TH / 4;
    constexpr int TWH = kTW + 4, THH = TH + 4, NPIX = THH * TWH;
    __shared__ __attribute__((aligned(16))) float s_x[NPIX * 3 + 16];  // + the zero-weight slot 15 of the last pixel
    __shared__ __attribute__((aligned(16))) float s_w[5 * 8 * 64];     // [ky][j/2][h][cout][2]: B[k = 2 j + h][cout]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // conv0's 12.8 KB of weights are loaded once per workgroup, which then walks its share of the tiles
    // (grid = a few workgroups per CU, sr_launch_conv0); the image is small and L2-resident on every XCD, so the
    // tile -> XCD mapping does not matter here (measured: no change in kernel time either way)
    for (int k = tid; k < 5 * 8 * 64; k += kThreads) s_w[k] = a.wpack[k];
    if (tid < 16) s_x[NPIX * 3 + tid] = 0.f;
    // the tile-queue heads of this call's four stage kernels: entry b / 8 of queue b % 8 belongs to workgroup b of that launch
    // without asking (queue_first), so a head starts at the number of such workgroups; visible to the later launches by stream order
    if (blockIdx.x == 0 && tid < 40) a.queue_reset[tid] = (a.queue_grid[tid >> 3] - (tid & 7) + 7) >> 3;
    // img_to_data (main.rs:170) is u8 / 255 with a true division; one table entry per byte value replaces
    // ~10 VALU instructions per sample (the f32 MFMA shares the vector ALU)
    __shared__ float s_lut[256];
    if constexpr (IMG_U8) s_lut[tid] = __fdiv_rn((float)tid, 255.0f);
    const float bias = a.bias[i], beta = a.beta[i];
  for (int bid = blockIdx.x; bid < a.n_tiles; bid += gridDim.x) {
    const int n = tile_div(bid, a.div_tpi), t = bid - n * tiles_per_img;
    const int ty = tile_div(t, a.div_tx), tx = t - ty * a.tiles_x;
    const int x0 = tx * kTW, y0 = a.y_begin + ty * TH;
    const size_t img_px0 = (size_t)n * a.H * a.W;

    __syncthreads();  // the previous tile's reads of s_x are done
    // (Round 4 tried a cheaper path for interior tiles -- per-thread byte offsets computed once per workgroup, one unaligned dword
    // load per pixel, no bounds tests: 0.114 ms against 0.108 for this loop, and 0.127 with byte loads: profiles/r4_ab_variants_f32.txt.
    // A second attempt kept the kernel at its 64 VGPRs = 8 waves per SIMD (the first had silently dropped to 7: 66-68 VGPRs) with the
    // offsets parked in LDS: 0.1070 against 0.1073 ms (profiles/r4_ab_conv0_second_attempt.txt).  conv0's time is not in its staging
    // arithmetic.  Round 5: nor in the staging loads' latency -- the next tile's pixels requested under this tile's matrix work and
    // consumed before its stores (what bought the u8 parameter-free graphs 20 %, sr_aux.hip) costs this kernel its 8th wave per SIMD
    // (74 VGPRs; held to 64 it spills): 0.1093 -> 0.1200 ms, profiles/r5_ab_conv0_prefetch.txt.)
    for (int p = tid; p < NPIX; p += kThreads) {
        const int py = p / TWH, px = p - py * TWH;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const size_t gp = img_px0 + (size_t)gy * a.W + gx;
            if constexpr (IMG_U8) {
                const uint8_t* q = (const uint8_t*)a.img + gp * a.img_ch;
                v.x = s_lut[q[0]]; v.y = s_lut[q[1]]; v.z = s_lut[q[2]];
            } else {
                v.x = load_img(a.img, a.img_ch, false, gp, 0);
                v.y = load_img(a.img, a.img_ch, false, gp, 1);
                v.z = load_img(a.img, a.img_ch, false, gp, 2);
            }
        }
        s_x[p * 3] = v.x; s_x[p * 3 + 1] = v.y; s_x[p * 3 + 2] = v.z;
    }
    __syncthreads();

    f32x16 acc[T];
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const float* xa = s_x + ((wave * T) * TWH + i) * 3 + h;   // slot k = 2 j + h of pixel i's kernel row
    const float* wb = s_w + (h * 32 + i) * 2;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x2 b = *(const f32x2*)(wb + (ky * 4 + jj) * 128);
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const float* row = xa + (m + ky) * TWH * 3;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[4 * jj], b.x, acc[m], 0, 0, 0);
                // (slot 15 -- h = 1 of the row's last MFMA -- is the NEXT pixel's first channel under a zero weight: a true zero instead, or an
                // Inf / NaN pixel outside the 5x5 footprint would reach this output as 0 x Inf = NaN where the reference has none)
                const float a2 = (jj == 3 && h) ? 0.0f : row[4 * jj + 2];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b.y, acc[m], 0, 0, 0);
            }
        }
    }

    const bool full_x = x0 + kTW <= a.W;
#pragma unroll
    for (int m = 0; m < T; ++m) {
        const int y = y0 + wave * T + m;
        if (y >= a.y_end) continue;
        float* base = a.dst + ((size_t)n * a.img_stride + (long)y * a.pitch + x0 + 4 * h) * 32 + i;
        if (full_x) {
            store_belu_tile(base, acc[m], bias, beta);
        } else {
            for_each_acc_row([&](int r, int row) {
                if (x0 + 4 * h + row < a.W) base[row * 32] = belu(__fadd_rn(acc[m][r], bias), beta);
            });
        }
    }
  }
}

// ---------------------------------------------------------------------------
// Stage 0 of the split-half mode, on the f16 matrix cores.  Until round 5 this mode ran conv0_kernel too: exact f32 products on the
// vector ALU's MFMA (80 per tile and wave = 5 120 cycles), then BeLU + the hi / lo split of 32 outputs per input pixel on the same pipe:
// 0.144 ms at 1080p, 9 % of the split-half frame (0.107 ms in the exact mode, whose epilogue is shorter).  Here the image tile is
// split into halves once while it is staged ([pixel][R G B 0] as hi halves and as lo halves; a byte goes through a table of
// split(byte / 255)), K runs over tap * 4 + channel (100 slots -> 7 K-blocks of 16; the 25 slots of the padding channel and the last
// 12 carry zero weights and read pixels inside the 5x5 footprint), and a tile row is 7 x 3 v_mfma_f32_32x32x16_f16 = 672 matrix-pipe
// cycles that run beside the epilogue's vector work instead of in front of it.  The weights (7 x (hi, lo) fragments, 14 KB) sit in LDS
// for every tile the workgroup walks.  Same products as every other stage of this mode: hi.hi + (hi.lo + lo.hi) / 2048.
// ---------------------------------------------------------------------------
template <int TH, bool IMG_U8>
__global__ __launch_bounds__(kThreads, 3) void conv0_split_kernel(Conv0Args a) {
    constexpr int T = TH / 4;
    constexpr int TWH = kTW + 4, THH = TH + 4, NPIX = THH * TWH;
    constexpr int NB = 7;  // K-blocks of 16 slots: slot s = 4 tap + channel, tap = 5 ky + kx
    __shared__ __attribute__((aligned(16))) uint32_t s_px[2][NPIX * 2];  // [hi | lo][pixel][R G | B 0] halves
    __shared__ uint32_t s_lut[256];                                      // byte -> hi half | lo half << 16 of byte / 255 (img_to_data, main.rs:170)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // the tile-queue heads of this call's four stage kernels (see conv0_kernel)
    if (blockIdx.x == 0 && tid < 40) a.queue_reset[tid] = (a.queue_grid[tid >> 3] - (tid & 7) + 7) >> 3;
    if constexpr (IMG_U8) {
        uint32_t hi2, lo2;
        split_half2(f32p{__fdiv_rn((float)tid, 255.0f), 0.0f}, hi2, lo2);
        s_lut[tid] = (hi2 & 0xffffu) | (lo2 << 16);
    }
    // B fragments: lane (output channel i, K half h) of block b holds slots 16 b + 8 h + (0..7), hi halves and lo halves (sr_api.cpp
    // pack_conv0_split); 14 KB, copied into LDS once per workgroup (in registers they would cost 56 VGPRs and a third of the occupancy)
    __shared__ __attribute__((aligned(16))) f16x8 s_w[NB * 2 * 64];
    for (int k = tid; k < NB * 2 * 64; k += kThreads) s_w[k] = ((const f16x8*)a.wpack_split)[k];
    const f16x8* wl = s_w + h * 32 + i;
    // A fragments: the lane's two taps of block b (4 b + 2 h and the next; beyond the 25th: the 25th again, its weights are zero), as byte
    // offsets of their pixels from the lane's own pixel (row 0, column i) in the staged tile
    uint32_t off_a[NB], off_b[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ta = min(4 * b + 2 * h, 24), tb = min(4 * b + 2 * h + 1, 24);
        off_a[b] = (uint32_t)(((ta / 5) * TWH + ta % 5) * 8);
        off_b[b] = (uint32_t)(((tb / 5) * TWH + tb % 5) * 8);
    }
    const char* abase = (const char*)&s_px[0][0] + ((wave * T) * TWH + i) * 8;
    constexpr uint32_t LO = NPIX * 2 * 4;  // bytes from the hi array to the lo array
    // The weights are the MFMA's A operand, the pixels its B: lane (i, h) holds output channels 8 j + 4 h + (0..3) of pixel i (store_belu_tile_split_cr)
    f32x4 bias4[4], beta4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias4[j] = *(const f32x4*)(a.bias + 8 * j + 4 * h);
        beta4[j] = *(const f32x4*)(a.beta + 8 * j + 4 * h);
    }
    uint32_t dom = 0;  // the largest hi half this thread has produced, inputs included (domain_track)
    for (int bid = blockIdx.x; bid < a.n_tiles; bid += gridDim.x) {
        const int n = tile_div(bid, a.div_tpi), t = bid - n * tiles_per_img;
        const int ty = tile_div(t, a.div_tx), tx = t - ty * a.tiles_x;
        const int x0 = tx * kTW, y0 = a.y_begin + ty * TH;
        const size_t img_px0 = (size_t)n * a.H * a.W;
        __syncthreads();  // the previous tile's reads of s_px are done (first tile: the table is written)
        for (int p = tid; p < NPIX; p += kThreads) {
            const int py = p / TWH, px = p - py * TWH;
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            uint32_t h0 = 0, h1 = 0, l0 = 0, l1 = 0;  // (R G), (B 0) as hi halves / lo halves; zero padding outside the image (Padding::Same)
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t gp = img_px0 + (size_t)gy * a.W + gx;
                if constexpr (IMG_U8) {
                    const uint8_t* q = (const uint8_t*)a.img + gp * a.img_ch;
                    const uint32_t wr = s_lut[q[0]], wg = s_lut[q[1]], wb = s_lut[q[2]];
                    h0 = __builtin_amdgcn_perm(wg, wr, 0x05040100u); l0 = __builtin_amdgcn_perm(wg, wr, 0x07060302u);
                    h1 = wb & 0xffffu; l1 = wb >> 16;
                } else {
                    const float* q = (const float*)a.img + gp * 3;
                    split_half2(f32p{q[0], q[1]}, h0, l0);
                    split_half2(f32p{q[2], 0.0f}, h1, l1);
                    domain_track(dom, h0);
                    domain_track(dom, h1);
                }
            }
            *(uint2*)&s_px[0][p * 2] = make_uint2(h0, h1);
            *(uint2*)&s_px[1][p * 2] = make_uint2(l0, l1);
        }
        __syncthreads();

        f32x16 accm[T], accx[T];
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[m][r] = bias4[r >> 2][r & 3]; accx[m][r] = 0.f; }  // (the bias rides in the accumulator, see split_value)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f16x8 bh = wl[(b * 2 + 0) * 64], bl = wl[(b * 2 + 1) * 64];
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const char* ra = abase + off_a[b] + m * TWH * 8;
                const char* rb = abase + off_b[b] + m * TWH * 8;
                const uint2 ha = *(const uint2*)ra, hb2 = *(const uint2*)rb;
                const uint2 la = *(const uint2*)(ra + LO), lb2 = *(const uint2*)(rb + LO);
                const uint32_t ahw[4] = {ha.x, ha.y, hb2.x, hb2.y}, alw[4] = {la.x, la.y, lb2.x, lb2.y};
                const f16x8 ah = __builtin_bit_cast(f16x8, ahw), al = __builtin_bit_cast(f16x8, alw);
                accm[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, accm[m], 0, 0, 0);
                accx[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, accx[m], 0, 0, 0);
                accx[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, accx[m], 0, 0, 0);
            }
        }

        const bool full_x = x0 + kTW <= a.W;
#pragma unroll
        for (int m = 0; m < T; ++m) {
            const int y = y0 + wave * T + m;
            if (y >= a.y_end) continue;
            char* base = (char*)(a.dst + ((size_t)n * a.img_stride + (long)y * a.pitch) * 32) + (size_t)(x0 + i) * 16 + 8 * h;
            store_belu_tile_split_cr(base, accm[m], accx[m], beta4, full_x || x0 + i < a.W, (long)a.pitch * 16, (long)a.pitch * 64, dom);
        }
    }
    domain_report(dom, a.domain);
}

// ---------------------------------------------------------------------------
// Stages 1-4: sum of up to three 32-channel convolutions (first KS0 x KS0,
// the others 3x3) + bias, then BeLU -> NHWC feature map, or (FINAL) + the
// bilinear residual as a fourth K-segment, depth-to-space -> output image.
// ---------------------------------------------------------------------------
template <int TH, int KS>
struct TileGeom {
    static constexpr int R = KS / 2;
    static constexpr int TWH = kTW + 2 * R;
    static constexpr int THH = TH + 2 * R;
    static constexpr int NPIX = THH * TWH;
    static constexpr int NG = (NPIX + 63) / 64;    // LDS-DMA groups of 64 tile pixels
    static constexpr int PLANE = NG * 64 * 16;     // bytes per channel-group plane
};

// Stage one source tile (+halo) into LDS, planar [cin/4][pixel][16 B], by LDS-DMA
// (no VGPR round trip, no VALU).  One instruction moves 16 bytes of 64 consecutive
// tile pixels: lane l gathers channel group c of tile pixel P = 64 g + l from
// `origin + off[P] + 16 c`, and lands at plane c, slot P.  off (TileOffsets below) is the
// only per-lane operand; the origin is wave-uniform and 16 c is an instruction immediate.
// The maps carry a zero border in HBM, so the reference's zero padding (Padding::Same)
// needs no bounds test here.
// One LDS-DMA instruction: 16 bytes per lane from `base + voff + IMM` (base wave-uniform, voff a 32-bit lane
// offset) to LDS address `lds + 16 * lane + IMM` (the immediate applies to both sides).  Issued as inline
// assembly on purpose: while the compiler sees LDS-DMA in a function it treats the LGKM counter as
// unordered and puts s_waitcnt lgkmcnt(0) in front of every use of a ds_read result, which made
// software-pipelined operand reads impossible.  vmcnt bookkeeping for these is done by hand (ring_barrier).
template <int IMM>
__device__ __forceinline__ void lds_dma16(const void* base, uint32_t voff, uint32_t lds) {
    // (s_nop 3: the base may have been written by a VALU instruction just before the statement -- v_readlane of a spilled SGPR,
    // v_readfirstlane -- and a VMEM instruction may read such an SGPR only 5 wait states later; the compiler pads nothing inside)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                 :: "s"(lds), "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// The same with `base + byte_off` formed inside the statement, on the scalar ALU (vcc as the address pair).  Given base + constant
// in C the compiler hoists every step's 64-bit address out of the tile loop as a loop invariant -- 46 pointer pairs for the weight
// chunks of stage 3 -- and, 102 SGPRs being all there are, spills them into VGPR lanes: two v_readlane per step in the matrix
// stream, each worth ~10 cycles of f32-MFMA time.  A constant offset is rematerialised instead (one s_mov).
__device__ __forceinline__ void lds_dma16_at(const void* base, uint32_t byte_off, uint32_t voff, uint32_t lds) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    // (byte_off and lds are wave-uniform at every call site, but under register pressure the compiler may carry such a value -- the ring slot,
    // say -- in a vector register, and an "s" operand it cannot satisfy is an assembler error: readfirstlane folds away wherever the value
    // already sits in a scalar register)
    asm volatile("s_add_u32 vcc_lo, %0, %2\n\ts_addc_u32 vcc_hi, %1, 0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, vcc offset:0"
                 :: "s"((uint32_t)b), "s"((uint32_t)(b >> 32)), "s"(__builtin_amdgcn_readfirstlane(byte_off)), "s"(__builtin_amdgcn_readfirstlane(lds)), "v"(voff)
                 : "vcc", "scc", "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// Per-lane gather offsets of a tile: entry gi = byte offset of tile pixel P = 64 (wave + NW gi) + lane
// from the tile origin, ((P / TWH) * pitch + P % TWH) * 128.  Computed once per workgroup (a dozen
// VALU ops) and kept in registers: every later DMA then needs no vector ALU work at all, which
// matters because a staging wave gets about one VALU issue slot per MFMA of its SIMD neighbour.
// Padding slots (P >= NPIX) re-read the last pixel and land in the unused tail of the plane.
template <int TH, int KS, int NW = 4>
struct TileOffsets {
    using G = TileGeom<TH, KS>;
    static constexpr int N = (G::NG + NW - 1) / NW;
    uint32_t v[N];
    template <int PREC>
    __device__ __forceinline__ void init(int pitch, int wave, int lane) {
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            const int P = min((wave + NW * gi) * 64 + lane, G::NPIX - 1);
            const int prow = P / G::TWH, pcol = P - prow * G::TWH;
            v[gi] = kPlanar<PREC> ? (uint32_t)(prow * pitch) * 128u + (uint32_t)pcol * 16u : (uint32_t)(prow * pitch + pcol) * 128u;
        }
    }
};

template <int TH, int KS, int PREC, int NW = 4>
__device__ __forceinline__ void stage_tile(char* tile, const float* __restrict__ src, const TileOffsets<TH, KS, NW>& off,
                                           long img_stride, int pitch, int n, int y0, int x0, int wave, int /*lane*/) {
    using G = TileGeom<TH, KS>;
    const char* origin = kPlanar<PREC> ? uniform_ptr((const char*)(src + ((size_t)n * img_stride + (long)(y0 - G::R) * pitch) * 32) + (long)(x0 - G::R) * 16)
                                       : uniform_ptr(src + ((size_t)n * img_stride + (long)(y0 - G::R) * pitch + (x0 - G::R)) * 32);
#pragma unroll
    for (int gi = 0; gi < TileOffsets<TH, KS, NW>::N; ++gi) {
        const int g = wave + NW * gi;  // wave-uniform
        if (g < G::NG) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr(tile) + g * 1024);
            if constexpr (kPlanar<PREC>) {  // channel group c: its own row of the map, pitch x 16 bytes further on
#pragma unroll
                for (int c = 0; c < 8; ++c) lds_dma16_at(origin, (uint32_t)(c * pitch * 16), off.v[gi], dst + c * G::PLANE);
            } else {
                // channel group c lands in plane c; the immediate (16 c) applies to the LDS side too, hence the - 16 c
#define SR_DMA16(c) lds_dma16<(c) * 16>(origin, off.v[gi], dst + (c) * (G::PLANE - 16))
                SR_DMA16(0); SR_DMA16(1); SR_DMA16(2); SR_DMA16(3); SR_DMA16(4); SR_DMA16(5); SR_DMA16(6); SR_DMA16(7);
#undef SR_DMA16
            }
        }
    }
}

// Asynchronous 4 KB weight-chunk copy global -> LDS ring slot (LDS-DMA, no VGPR
// round trip): each wave moves 1 KB, lane l lands at base + 16*l, which is
// exactly the packed chunk order.
__device__ __forceinline__ void weight_chunk_async(char* ring_slot, const float* __restrict__ chunk,
                                                   int wave, int lane) {
    lds_dma16<0>(uniform_ptr((const char*)chunk + wave * 1024), (uint32_t)(lane * 16),
                 __builtin_amdgcn_readfirstlane(lds_addr(ring_slot) + wave * 1024));
}

// Workgroup barrier that lets the newest `PENDING` LDS-DMA chunk loads stay in
// flight across it (a plain __syncthreads() would drain them: vmcnt(0)).  All of
// this wave's LDS reads are retired first, so the slot the next DMA overwrites
// is no longer being read by anybody once every wave has passed.
template <int PENDING>
__device__ __forceinline__ void ring_barrier() {
    static_assert(PENDING >= 0 && PENDING <= 3, "");
    if constexpr (PENDING == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (PENDING == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (PENDING == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Ring bookkeeping shared by both tap loops: at tap `gtap` (its chunk sits in
// `slot`) request chunk gtap + kRingAhead into the slot freed by the previous
// barrier; after the tap's MFMAs, wait until chunk gtap + 1 has landed, i.e. until
// at most `chunks requested after it` DMAs remain in flight.
__device__ __forceinline__ void ring_request(char* ring, const float* __restrict__ wpack, int gtap, int slot,
                                             int ntaps_total, int wave, int lane) {
    if (gtap + kRingAhead < ntaps_total) {
        int s2 = slot + kRingAhead;
        if (s2 >= kRingSlots) s2 -= kRingSlots;
        weight_chunk_async(ring + s2 * 4096, wpack + (size_t)(gtap + kRingAhead) * kChunkFloats, wave, lane);
    }
}
// EXTRA = 0: when this returns, chunk gtap (the new gtap) has landed in every wave's view; EXTRA = 1: chunk
// gtap + 1 as well (the split-half loop reads its operands one step ahead).
template <int EXTRA = 0>
__device__ __forceinline__ void ring_advance(int& gtap, int& slot, int ntaps_total) {
    // chunks still allowed in flight: those requested after the newest one that must be complete
    const int pending = max(0, min(kRingAhead - 1 - EXTRA, ntaps_total - 2 - EXTRA - gtap));
    ++gtap;
    slot = slot == kRingSlots - 1 ? 0 : slot + 1;
    if (pending >= 3) ring_barrier<3>();
    else if (pending == 2) ring_barrier<2>();
    else if (pending == 1) ring_barrier<1>();
    else ring_barrier<0>();
}

// ---------------------------------------------------------------------------
// The matrix work of one HALF of a source tile (16 of its 32 input channels), shared by both kernel forms.
// A step = two taps x one half = one 4 KB weight chunk (layout: sr_api.cpp pack_pipe); steps walk the tap
// pairs (0,1), (2,3), ...; a lone last tap is half a step.  Both forms sum in this same order, so their
// results are bit-identical, whatever the tile height or the staging scheme.
//   hb     : LDS planes of this half, plane stride PS: f32 4 planes of 4 channels; split 2 hi planes of 8
//            channels, then (LO planes further on) their 2 lo planes
//   Stream : where the chunks come from and what else to request meanwhile
//              begin_step<P>()        issue this step's DMA requests (next weight chunk, a piece of a tile)
//              slot()                 ring slot of the current chunk
//              end_step<EXTRA>(last)  wait until chunk (current + 1 + EXTRA) has landed, barrier, advance
// Software-pipelined by hand: f32 requests operand group g+1 before the 8 MFMAs of group g and the first group
// of the next step right after the barrier, under the last group; split (12 MFMAs = ~400 cycles per step, one
// LDS round trip) requests the whole next step's operands before this step's MFMAs.
// ---------------------------------------------------------------------------
//
// QUAD (round 6; the last stage of the exact mode, 8-row tiles): the same steps on v_mfma_f32_4x4x1_16B_f32.  The last stage has 27
// output channels; on 32x32x2 its N is 32 and 5 columns of every instruction are padding.  The 4x4x1 instruction is sixteen 4x4 outer
// products (2 passes, the same 64 FLOP per cycle and SIMD) and with CBSZ = 4 the A operand of block ABID is broadcast to all sixteen:
//     D[lane 4 b + j][register i] += A[lane 4 ABID + i] * B[lane 4 b + j]
// A = the weight fragment exactly as the 32x32x2 loop reads it (lane (h, j) = output slot j, four channels of k-quad h -- so ONE
// register serves every slot group: ABID = 8 h + g picks slots 4 g .. 4 g + 3 of k-quad h), B = pixels (lane = pixel: lanes 0-31 the
// wave's first tile row, 32-63 its second; one ds_read_b128 per k-quad), and a k-step of 64 pixels x 28 slots is 7 instructions instead
// of the 8 instruction-equivalents of N = 32.  Same weight chunks, same LDS traffic (one weight read + two pixel reads per 8 k), and a
// LANE ends up holding every channel of ITS pixel: depth-to-space and the RGBA packing are lane-local (stage_epilogue_quad).
// The instruction chain per output is the very fmaf chain of the 32x32x2 form -- k-quad 0 then k-quad 1 of each e, as that
// instruction sums its k = h pair -- so the two forms are bit-identical (scripts/experiments/ubench_mfma4x4.hip checks the chain
// against the host's fmaf; every pipe == first form == band test checks the kernels): 4-row tiles and the first kernel form keep 32x32x2.
// Measured in the step loop's structure (profiles/r6_ubench_mfma4x4.txt): 139.6 TFLOP/s issued against 144.8 for 32x32x2, on 7 / 8 of
// the work.
template <int ABID>
__device__ __forceinline__ f32x4 mfma_quad(float w, float p, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(w, p, c, 4, ABID, 0); }
// slot groups of N-tile nt that carry expand channels (slot of a channel: sr_api.cpp expand_channel -- triples never straddle a 16-slot row)
template <int FACTOR>
constexpr int quad_groups(int nt) {
    const int ntr = FACTOR * FACTOR - 10 * nt < 10 ? FACTOR * FACTOR - 10 * nt : 10, tl = ntr - 1;
    return (16 * (tl / 5) + 3 * (tl % 5) + 2) / 4 + 1;
}
typedef f32x4 QuadAcc[8];  // one N-tile: register i of group g = slot 4 g + i of the lane's pixel
// one k of k-quad H onto the first NG slot groups
template <int H, int NG>
__device__ __forceinline__ void quad_rank1(QuadAcc& acc, float w, float p) {
    if constexpr (NG > 0) acc[0] = mfma_quad<8 * H + 0>(w, p, acc[0]);
    if constexpr (NG > 1) acc[1] = mfma_quad<8 * H + 1>(w, p, acc[1]);
    if constexpr (NG > 2) acc[2] = mfma_quad<8 * H + 2>(w, p, acc[2]);
    if constexpr (NG > 3) acc[3] = mfma_quad<8 * H + 3>(w, p, acc[3]);
    if constexpr (NG > 4) acc[4] = mfma_quad<8 * H + 4>(w, p, acc[4]);
    if constexpr (NG > 5) acc[5] = mfma_quad<8 * H + 5>(w, p, acc[5]);
    if constexpr (NG > 6) acc[6] = mfma_quad<8 * H + 6>(w, p, acc[6]);
    if constexpr (NG > 7) acc[7] = mfma_quad<8 * H + 7>(w, p, acc[7]);
}
template <int FACTOR, int H>
__device__ __forceinline__ void quad_rank1_tile(QuadAcc& acc, int nt, float w, float p) {  // nt: constant after unrolling
    if (nt == 0) quad_rank1<H, quad_groups<FACTOR>(0)>(acc, w, p);
    else quad_rank1<H, quad_groups<FACTOR>(1)>(acc, w, p);
}

template <int TWH, int PS, int KS, int T, int NTN, bool QUAD = false, int FACTOR = 3, typename Acc, typename Stream>
__device__ __forceinline__ void half_steps_f32(Acc& acc, const char* hb, const char* ring, Stream& sm, int wave, int lane) {
    constexpr int NT = KS * KS, NP = (NT + 1) / 2;
    static_assert(!QUAD || T == 2, "the quad form: 64 pixels = two tile rows per wave");
    const int i = lane & 31, h = lane >> 5;
    const int wlane = (h * 32 + i) * 16;
    // 32x32x2: lane (h, i) = pixel i of the wave's tile rows, k = h: plane h of the pair.  Quad: lane (h, i) = pixel i of tile row h, both planes.
    const char* abase = QUAD ? hb + ((wave * T + h) * TWH + i) * 16 : hb + h * PS + ((wave * T) * TWH + i) * 16;
    struct Ops { f32x4 a[T]; f32x4 b; };
    // operand group q of pair p: q = 2 * tapslot + rr  (rr: which 8 of the half's 16 channels)
    auto load = [&](Ops& o, int p, int q, int sl) {
        const int t = 2 * p + (q >> 1), ky = t / KS, kx = t - ky * KS;
        const char* ab = abase + (q & 1) * 2 * PS + (ky * TWH + kx) * 16;
        o.b = *(const f32x4*)(ring + sl * 4096 + wlane + q * 1024);
#pragma unroll
        for (int m = 0; m < T; ++m) o.a[m] = *(const f32x4*)(ab + (QUAD ? m * PS : m * TWH * 16));  // (quad: a[m] = k-quad m)
    };
    auto mfma = [&](const Ops& o, int nt) {
        if constexpr (QUAD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                quad_rank1_tile<FACTOR, 0>(acc[nt], nt, o.b[e], o.a[0][e]);
                quad_rank1_tile<FACTOR, 1>(acc[nt], nt, o.b[e], o.a[1][e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < T; ++m)
                    acc[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[m][e], o.b[e], acc[nt * T + m], 0, 0, 0);
        }
    };
    Ops cur, nxt;
    load(cur, 0, 0, sm.slot());
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        const int ngroups = (2 * p + 1 < NT) ? 4 : 2;
        sm.begin_step();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q + 1 < ngroups) {
                load(nxt, p, q + 1, sm.slot());
                if (((p * NTN + nt) * 4 + q) % NTN == 0) sm.piece(((p * NTN + nt) * 4 + q) / NTN);  // one gather instruction per NTN operand groups
                __builtin_amdgcn_sched_barrier(0);
                mfma(cur, nt);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        const bool last = p == NP - 1 && nt == NTN - 1;
        sm.template end_step<0>(last);
        if (!last) load(nxt, nt + 1 < NTN ? p : p + 1, 0, sm.slot());
        if (((p * NTN + nt) * 4 + 3) % NTN == 0) sm.piece(((p * NTN + nt) * 4 + 3) / NTN);
        __builtin_amdgcn_sched_barrier(0);
        mfma(cur, nt);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    }
}

// (Round 6, as in half_steps_h16: the WEIGHT fragment is the instruction's A operand and the pixel fragment its B -- same register shapes -- so
// the accumulators hold the transposed tile: lane (i, h) has, of pixel i of the tile row, output slots 8 j + 4 h + (0..3) in registers
// 4 j + (0..3).  This loop only serves the split-half mode's LAST stage, whose epilogue is then lane-local: stage_epilogue_final_t.)
template <int TWH, int PS, int LO, int KS, int T, int NTN, typename Stream>
__device__ __forceinline__ void half_steps_h(f32x16 (&accm)[NTN * T], f32x16 (&accx)[NTN * T], const char* hb, const char* ring,
                                             Stream& sm, int wave, int lane) {
    constexpr int NT = KS * KS, NP = (NT + 1) / 2;
    const int i = lane & 31, h = lane >> 5;
    const int wlane = (h * 32 + i) * 16;
    const char* abase = hb + h * PS + ((wave * T) * TWH + i) * 16;
    struct Ops { f16x8 bh[2], bl[2], ah[2][T], al[2][T]; };  // chunk = [hi tap0 | hi tap1 | lo tap0 | lo tap1]
    // The split loop walks the taps COLUMN by column (tap t = kernel column t / KS, kernel row t % KS): row m of
    // tap (ky, kx) is tile row ky + m of column kx, so vertically adjacent taps share tile rows.  A row that the
    // current or the previous step already holds is copied between registers instead of read again: 12 LDS reads
    // per 5-tap column and half instead of 20 (VALU copies cost the f16 matrix pipe nothing).  Round 4 measured what the reads
    // themselves cost: with EVERY operand read of the loop removed the 1080p frame is 3.7 % faster, without the A reads 1.2 %
    // (profiles/r4_split_operand_reads_removed.txt); the LDS array is 25 % busy and conflict-free (profiles/r4_pmc_split_binders.json).
    // Operand traffic is not what holds this mode at mfma_util 0.66.
    auto row_id = [](int p, int ts, int m) {  // which (column, tile row) operand slot (ts, m) of step p holds; < 0: none
        const int t = 2 * p + ts;
        return t < NT ? (t / KS) * 16 + (t % KS) + m : -1;
    };
    auto load = [&](Ops& o, const Ops& prev, int pp, int p, int sl) {  // pp: the step `prev` holds, or -1
        const char* wb = ring + sl * 4096 + wlane;
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            const int t = 2 * p + ts;
            if (t >= NT) continue;
            o.bh[ts] = *(const f16x8*)(wb + ts * 1024);
            o.bl[ts] = *(const f16x8*)(wb + 2048 + ts * 1024);
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const int id = row_id(p, ts, m);
                bool have = false;
#pragma unroll
                for (int ts2 = 0; ts2 < 2; ++ts2)
#pragma unroll
                    for (int m2 = 0; m2 < T; ++m2) {
                        if (!have && (ts2 * T + m2) < (ts * T + m) && row_id(p, ts2, m2) == id) {
                            o.ah[ts][m] = o.ah[ts2][m2]; o.al[ts][m] = o.al[ts2][m2]; have = true;
                        }
                    }
#pragma unroll
                for (int ts2 = 0; ts2 < 2; ++ts2)
#pragma unroll
                    for (int m2 = 0; m2 < T; ++m2) {
                        if (!have && pp >= 0 && row_id(pp, ts2, m2) == id) {
                            o.ah[ts][m] = prev.ah[ts2][m2]; o.al[ts][m] = prev.al[ts2][m2]; have = true;
                        }
                    }
                if (!have) {
                    const int kx = t / KS, ky = t - kx * KS;
                    const char* ab = abase + ((ky + m) * TWH + kx) * 16;
                    o.ah[ts][m] = *(const f16x8*)ab;
                    o.al[ts][m] = *(const f16x8*)(ab + LO * PS);
                }
            }
        }
    };
    // One operand of step p, item k = ts * (2 + 2 T) + j: j = 0 / 1 the hi / lo weight fragment of tap slot ts, j = 2 + 2 m / 3 + 2 m
    // the hi / lo pixels of tile row m -- a register copy when the row is already held (see above), an LDS read otherwise.
    constexpr int NI = 2 * (2 + 2 * T), NM = 2 * 3 * T;  // operand items / MFMAs of a full step
    auto load_item = [&](Ops& o, const Ops& prev, int pp, int p, int sl, int k) {
        const int ts = k / (2 + 2 * T), j = k % (2 + 2 * T), t = 2 * p + ts;
        if (t >= NT) return;
        const char* wb = ring + sl * 4096 + wlane;
        if (j == 0) { o.bh[ts] = *(const f16x8*)(wb + ts * 1024); return; }
        if (j == 1) { o.bl[ts] = *(const f16x8*)(wb + 2048 + ts * 1024); return; }
        const int m = (j - 2) >> 1;
        const bool lo = (j - 2) & 1;
        const int id = row_id(p, ts, m);
#pragma unroll
        for (int ts2 = 0; ts2 < 2; ++ts2)
#pragma unroll
            for (int m2 = 0; m2 < T; ++m2)
                if ((ts2 * T + m2) < (ts * T + m) && row_id(p, ts2, m2) == id) {
                    if (lo) o.al[ts][m] = o.al[ts2][m2]; else o.ah[ts][m] = o.ah[ts2][m2];
                    return;
                }
#pragma unroll
        for (int ts2 = 0; ts2 < 2; ++ts2)
#pragma unroll
            for (int m2 = 0; m2 < T; ++m2)
                if (pp >= 0 && row_id(pp, ts2, m2) == id) {
                    if (lo) o.al[ts][m] = prev.al[ts2][m2]; else o.ah[ts][m] = prev.ah[ts2][m2];
                    return;
                }
        const int kx = t / KS, ky = t - kx * KS;
        const char* ab = abase + ((ky + m) * TWH + kx) * 16;
        if (lo) o.al[ts][m] = *(const f16x8*)(ab + LO * PS); else o.ah[ts][m] = *(const f16x8*)ab;
    };
    Ops cur, nxt;
    load(cur, cur, -1, 0, sm.slot());
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        const bool last = p == NP - 1 && nt == NTN - 1;
        // Two waves share a SIMD's matrix pipe.  With all of a step's operand reads and DMA requests in front of its
        // 12 MFMAs the two fall into lockstep (both read, then both multiply) and the pipe idles through every read
        // phase.  So each wave keeps the pipe busy on its own: one operand read of the NEXT step (or one DMA request)
        // in the shadow of each 32-cycle MFMA of this one, pinned in that order.
        const int pn = nt + 1 < NTN ? p : p + 1;
        const int sln = sm.slot() == kRingSlots - 1 ? 0 : sm.slot() + 1;
        int q = 0;  // MFMAs issued so far in this step
        auto after_mfma = [&]() {
            if (q == 0) sm.begin_step();
#pragma unroll
            for (int k = 0; k < NI; ++k)
                if (!last && k >= q * NI / NM && k < (q + 1) * NI / NM) load_item(nxt, cur, p, pn, sln, k);
            if (nt == 0 && q == NM / 2 - 1) sm.piece(2 * p);
            if (nt == 0 && q == NM - 2) sm.piece(2 * p + 1);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            if (2 * p + ts < NT) {
#pragma unroll
                for (int m = 0; m < T; ++m) { accm[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.bh[ts], cur.ah[ts][m], accm[nt * T + m], 0, 0, 0); after_mfma(); }
#pragma unroll
                for (int m = 0; m < T; ++m) { accx[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.bl[ts], cur.ah[ts][m], accx[nt * T + m], 0, 0, 0); after_mfma(); }
#pragma unroll
                for (int m = 0; m < T; ++m) { accx[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.bh[ts], cur.al[ts][m], accx[nt * T + m], 0, 0, 0); after_mfma(); }
            } else {  // the lone last tap of an odd kernel: the second tap slot is empty, its items still have to be issued
#pragma unroll
                for (int m = 0; m < 3 * T; ++m) after_mfma();
            }
        }
        sm.template end_step<1>(last);
        cur = nxt;
      }
    }
}

// The split-half step loop on v_mfma_f32_16x16x32_f16 (kH16).  M = 16 pixels, N = 16 output channels, K = 32 = TWO TAPS x the half's 16
// input channels: lanes 0-31 hold the step's first tap (lanes 0-15 channels 0-7 = hi plane 0, lanes 16-31 channels 8-15 = plane 1),
// lanes 32-63 its second tap -- the per-lane LDS base carries the distance between the two taps' pixels, every read is still one
// ds_read_b128 at an immediate offset.  A wave's 32 x 32 output tile row is 2 pixel halves x 2 channel halves, each product 3 MFMAs of
// 16 cycles: 24 per step and tile row pair, the matrix cycles of the 12 it replaces; 12 operand reads per step as before.
// Round 6: the WEIGHT fragment is passed as the instruction's A operand and the pixel fragment as its B (the two fragments have the same
// register shape, so this costs nothing): the result is the transposed tile -- lane l holds, for pixel (l & 15) of its pixel half, the four
// CONSECUTIVE output channels 4 (l >> 4) .. + 3 of its channel half.  Two channels of one pixel pack into a dword without a lane exchange,
// four into the 8 bytes that are contiguous in the row-planar map: the epilogue loses its two DPP moves and two v_perm per value pair and
// stores 8 bytes per lane in 256-byte runs instead of 4-byte pieces (stage_epilogue_h16).
// A half has an ODD number of taps (25, 9).  Leaving one K half of the last MFMA group empty would waste 4-10 % of the matrix work,
// so the lone last tap of BOTH halves of a source shares one step: the second half BEGINS with it, lanes 0-31 reading the first half's
// buffer (XD bytes from its own), lanes 32-63 its own.  The first half is pairs only.  Steps per source: 12 + 13 (5x5), 4 + 5 (3x3)
// -- 43 per stage-3 tile instead of 46 with three half steps gone.  The weight chunks are packed in this step order (sr_api.cpp
// pack_steps_h16): [hi | lo][channel half 2][lane 64][8 halves], still 4 KB per step.
// The first half's buffer must therefore survive the second half's first step: the gather pieces of the NEXT half tile (which land
// there) are requested from the second step on, three per step in a short (3x3) half so that they still have two steps to land.
template <int TWH, int PS, int LO, int KS, int T, bool SECOND, int XD, typename Stream>
__device__ __forceinline__ void half_steps_h16(f32x4 (&accm)[T][2][2], f32x4 (&accx)[T][2][2], const char* hb, const char* ring, Stream& sm,
                                               int wave, int lane) {
    constexpr int NT = KS * KS, NP = (NT - 1) / 2;  // tap pairs of a half
    constexpr int NS = NP + (SECOND ? 1 : 0);       // its steps
    const int p16 = lane & 15, g = lane >> 4, tsel = g >> 1;
    const char* base_s = hb + (g & 1) * PS + ((wave * T) * TWH + p16) * 16;
    const char* base_v = base_s + tsel * (TWH * 16);                       // the pair's second tap is the next kernel row of the same column,
    const char* base_c = base_s + tsel * (16 - (KS - 1) * TWH * 16);       // ... or the first row of the next column,
    const char* base_x = base_s + (tsel ? 0 : XD);                         // ... or (shared step) the same tap of the other half's buffer
    const char* wl = ring + lane * 16;
    struct Ops { f16x8 bh[2], bl[2], ah[T][2], al[T][2]; };
    constexpr int NI = 4 + 4 * T, NM = 12 * T;  // operand reads / MFMAs of a step
    auto first_tap = [](int st) { return SECOND ? (st == 0 ? NT - 1 : 2 * (st - 1)) : 2 * st; };
    auto load_item = [&](Ops& o, int st, int sl, int k) {
        if (k < 4) {
            const char* w = wl + sl * 4096 + (k >> 1) * 1024;
            if (k & 1) o.bl[k >> 1] = *(const f16x8*)(w + 2048); else o.bh[k >> 1] = *(const f16x8*)w;
            return;
        }
        const int kk = k - 4, m = kk >> 2, ph = (kk >> 1) & 1;
        const bool lo = kk & 1, shared = SECOND && st == 0;
        const int ta = first_tap(st), kx = ta / KS, ky = ta - kx * KS;
        const char* b = shared ? base_x : (ky == KS - 1 ? base_c : base_v);
        const char* ab = b + ((ky + m) * TWH + kx + 16 * ph) * 16 + (lo ? LO * PS : 0);
        if (lo) o.al[m][ph] = *(const f16x8*)ab; else o.ah[m][ph] = *(const f16x8*)ab;
    };
    Ops cur, nxt;
#pragma unroll
    for (int k = 0; k < NI; ++k) load_item(cur, 0, sm.slot(), k);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const bool last = st == NS - 1;
        const int sln = sm.slot() == kRingSlots - 1 ? 0 : sm.slot() + 1;
        // gather pieces of the half tile requested meanwhile: two per step (three in a 3x3 half), none in the shared step
        constexpr int PPS = KS == 3 ? 3 : 2;
        const int ps = SECOND ? st - 1 : st;  // piece slot of this step (< 0: none)
        int q = 0;  // MFMAs issued so far in this step
        auto after_mfma = [&]() {
            if (q == 0) sm.begin_step();
#pragma unroll
            for (int k = 0; k < NI; ++k)
                if (!last && k >= q * NI / NM && k < (q + 1) * NI / NM) load_item(nxt, st + 1, sln, k);
            if (ps >= 0) {
#pragma unroll
                for (int pc = 0; pc < PPS; ++pc)
                    if (q == (pc + 1) * NM / (PPS + 1) - 1) sm.piece(PPS * ps + pc);
            }
            __builtin_amdgcn_sched_barrier(0);
            ++q;
        };
        __builtin_amdgcn_sched_barrier(0);
        // every accumulator is touched once per product round: 4 T independent MFMAs between two on the same registers
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) { accm[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.bh[ch], cur.ah[m][ph], accm[m][ph][ch], 0, 0, 0); after_mfma(); }
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) { accx[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.bl[ch], cur.ah[m][ph], accx[m][ph][ch], 0, 0, 0); after_mfma(); }
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) { accx[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.bh[ch], cur.al[m][ph], accx[m][ph][ch], 0, 0, 0); after_mfma(); }
        sm.template end_step<1>(last);
        cur = nxt;
    }
}

// Stream of the first kernel form: the whole tile is resident, chunks come through the ring in step order.
struct RingStream {
    char* ring;
    const float* __restrict__ wpack;
    int& gtap;
    int& slot_;
    int ntotal, wave, lane;
    __device__ __forceinline__ void begin_step() { ring_request(ring, wpack, gtap, slot_, ntotal, wave, lane); }
    __device__ __forceinline__ void piece(int) {}
    __device__ __forceinline__ int slot() const { return slot_; }
    template <int EXTRA> __device__ __forceinline__ void end_step(bool) { ring_advance<EXTRA>(gtap, slot_, ntotal); }
};

// Both halves of one resident source tile (first kernel form).  f32 planes: 8 channel groups of 4; split planes:
// 4 hi groups of 8 channels, then their 4 lo groups.
template <int TH, int KS, int T, int NTN, int PREC>
__device__ __forceinline__ void source_steps(f32x16 (&acc)[NTN * T], f32x16 (&accx)[PREC == 1 ? NTN * T : 1], const char* tile, char* ring,
                                             const float* __restrict__ wpack, int& gtap, int& slot, int ntotal, int wave, int lane) {
    using G = TileGeom<TH, KS>;
    RingStream sm{ring, wpack, gtap, slot, ntotal, wave, lane};
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if constexpr (PREC == 0) half_steps_f32<G::TWH, G::PLANE, KS, T, NTN>(acc, tile + half * 4 * G::PLANE, ring, sm, wave, lane);
        else half_steps_h<G::TWH, G::PLANE, 4, KS, T, NTN>(acc, accx, tile + half * 2 * G::PLANE, ring, sm, wave, lane);
    }
}

// ... and on 16x16 accumulators (kH16): first half pairs only, second half led by the step the two halves share
template <int TH, int KS, int T>
__device__ __forceinline__ void source_steps_h16(f32x4 (&accm)[T][2][2], f32x4 (&accx)[T][2][2], const char* tile, char* ring,
                                                 const float* __restrict__ wpack, int& gtap, int& slot, int ntotal, int wave, int lane) {
    using G = TileGeom<TH, KS>;
    RingStream sm{ring, wpack, gtap, slot, ntotal, wave, lane};
    half_steps_h16<G::TWH, G::PLANE, 4, KS, T, false, 0>(accm, accx, tile, ring, sm, wave, lane);
    half_steps_h16<G::TWH, G::PLANE, 4, KS, T, true, -2 * G::PLANE>(accm, accx, tile + 2 * G::PLANE, ring, sm, wave, lane);
}

// The nine fixed-weight taps of the bilinear residual on a staged image tile s_x ([pixel][4] f32, edge-replicated)
// with the weights s_w (9 x NTN x 128 floats) -- shared by both kernel forms.
template <int TH, int T, int NTN>
__device__ __forceinline__ void lin_mfma(f32x16 (&acc)[NTN * T], const float* s_x, const float* s_w, int wave, int lane) {
    constexpr int TWH = kTW + 2;
    const int i = lane & 31, h = lane >> 5;
    const float* xa = s_x + ((wave * T) * TWH + i) * 4 + h * 2;
    const float* wb = s_w + (h * 32 + i) * 2;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const f32x2 b = *(const f32x2*)(wb + ((ky * 3 + kx) * NTN + nt) * 128);
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    const f32x2 av = *(const f32x2*)(xa + ((m + ky) * TWH + kx) * 4);
                    acc[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[nt * T + m], 0, 0, 0);
                    acc[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[nt * T + m], 0, 0, 0);
                }
            }
        }
    }
}

// ... and in the quad form (half_steps_f32 QUAD): the lane's own pixel [R G B 0] in one ds_read_b128, the weight pair as lin_mfma reads it
// (lane (h, j): colours 2 h, 2 h + 1 of slot j).  lin_mfma's chain per output and tap is colour 0, 2 (its first instruction, k = h), 1, 3;
// colour 3 is the zero channel under a zero weight -- fma(0, 0, acc) = acc for every acc but -0, which a chain that starts at +0 never
// holds -- and is skipped here.
template <int NTN, int FACTOR>
__device__ __forceinline__ void lin_mfma_quad(QuadAcc (&acc)[NTN], const float* s_x, const float* s_w, int wave, int lane) {
    constexpr int TWH = kTW + 2;
    const int i = lane & 31, h = lane >> 5;
    const float* xa = s_x + ((wave * 2 + h) * TWH + i) * 4;
    const float* wb = s_w + lane * 2;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 px = *(const f32x4*)(xa + (ky * TWH + kx) * 4);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const f32x2 b = *(const f32x2*)(wb + ((ky * 3 + kx) * NTN + nt) * 128);
                quad_rank1_tile<FACTOR, 0>(acc[nt], nt, b.x, px.x);
                quad_rank1_tile<FACTOR, 1>(acc[nt], nt, b.x, px.z);
                quad_rank1_tile<FACTOR, 0>(acc[nt], nt, b.y, px.y);
            }
        }
    }
}

// The same residual in the split-half mode, on the f16 matrix cores (round 5: on the f32 MFMA of the vector ALU the nine taps were
// 9.7 % of this mode's last stage, profiles/r5_ab_nolin.txt).  The phase weights of LinearInterp x f are products of multiples of
// 1 / (2 f), so (2 f)^2 times a weight is a small INTEGER, exact in a half: the staged pixels are divided by (2 f)^2 once (and split
// into hi / lo halves, [pixel][R G B 0] each), the weights need no lo part, and a K-block is two MFMAs -- x_hi . k into the main
// accumulators, x_lo . k into the cross accumulators (both already carry the conv taps in these scales).  K slot = 4 tap + channel:
// 36 slots = 3 K-blocks of 16 (the 12 padding slots under zero weights read the ninth tap's pixel: inside the footprint).
template <int TH, int T, int NTN>
__device__ __forceinline__ void lin_mfma_h(f32x16 (&accm)[NTN * T], f32x16 (&accx)[NTN * T], const char* s_hi, int lo_off, const f16x8* s_w, int wave, int lane) {
    constexpr int TWH = kTW + 2;
    const int i = lane & 31, h = lane >> 5;
    const char* xa = s_hi + ((wave * T) * TWH + i) * 8;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int ta = min(4 * b + 2 * h, 8), tb = min(4 * b + 2 * h + 1, 8);
        const int oa = ((ta / 3) * TWH + ta % 3) * 8, ob = ((tb / 3) * TWH + tb % 3) * 8;
        f16x8 ah[T], al[T];
#pragma unroll
        for (int m = 0; m < T; ++m) {
            const char* ra = xa + oa + m * TWH * 8;
            const char* rb = xa + ob + m * TWH * 8;
            const uint2 ha = *(const uint2*)ra, hb2 = *(const uint2*)rb, la = *(const uint2*)(ra + lo_off), lb2 = *(const uint2*)(rb + lo_off);
            const uint32_t ahw[4] = {ha.x, ha.y, hb2.x, hb2.y}, alw[4] = {la.x, la.y, lb2.x, lb2.y};
            ah[m] = __builtin_bit_cast(f16x8, ahw); al[m] = __builtin_bit_cast(f16x8, alw);
        }
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const f16x8 bw = s_w[((b * NTN + nt) * 2 + h) * 32 + i];
#pragma unroll
            for (int m = 0; m < T; ++m) {
                accm[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, ah[m], accm[nt * T + m], 0, 0, 0);  // (weights as A: see half_steps_h)
                accx[nt * T + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, al[m], accx[nt * T + m], 0, 0, 0);
            }
        }
    }
}
// one staged pixel of that tile: (R, G, B) / (2 f)^2 as hi halves and lo halves.  u8: through the table lin_split_table_entry fills.
__device__ __forceinline__ uint32_t lin_split_table_entry(int byte, float scale) {
    uint32_t hi2, lo2;
    split_half2(f32p{__fdiv_rn(__fdiv_rn((float)byte, 255.0f), scale), 0.0f}, hi2, lo2);
    return (hi2 & 0xffffu) | (lo2 << 16);
}
__device__ __forceinline__ void lin_split_store_u8(char* s_hi, int lo_off, int p, uint32_t wr, uint32_t wg, uint32_t wb) {
    *(uint2*)(s_hi + p * 8) = make_uint2(__builtin_amdgcn_perm(wg, wr, 0x05040100u), wb & 0xffffu);
    *(uint2*)(s_hi + lo_off + p * 8) = make_uint2(__builtin_amdgcn_perm(wg, wr, 0x07060302u), wb >> 16);
}
__device__ __forceinline__ void lin_split_store_f32(char* s_hi, int lo_off, int p, float r, float g, float b, float scale) {
    uint32_t h0, l0, h1, l1;
    split_half2(f32p{__fdiv_rn(r, scale), __fdiv_rn(g, scale)}, h0, l0);
    split_half2(f32p{__fdiv_rn(b, scale), 0.0f}, h1, l1);
    *(uint2*)(s_hi + p * 8) = make_uint2(h0, h1);
    *(uint2*)(s_hi + lo_off + p * 8) = make_uint2(l0, l1);
}

// Final stage only: the bilinear x3 residual (LinearInterp, network.rs:27) as nine
// more taps of a 4-channel (RGB + zero) source.  The image tile is staged with
// edge-REPLICATED coordinates (the interp clamps indices, it does not zero-pad),
// the fixed weights (phase products {1/3, 2/3, 1}^2, built on the host) sit in the
// weight pack behind the conv chunks: 9 x [cin/2][cout 32][2] floats.
template <int TH, int T, bool IMG_U8, int NTHREADS, int NTN, int PREC, int FACTOR>
__device__ __forceinline__ void lin_taps(f32x16 (&acc)[NTN * T], f32x16 (&accx)[PREC == 1 ? NTN * T : 1], char* tile, char* ring, const StageArgs& a,
                                         const float* __restrict__ wlin, int n, int y0, int x0, int wave,
                                         int lane, int tid) {
    constexpr int TWH = kTW + 2, THH = TH + 2, NPIX = THH * TWH;
    float* s_x = (float*)tile;   // [pixel][4]
    float* s_w = (float*)ring;   // 9 x NTN x 128 floats
    const size_t img_px0 = (size_t)n * a.H * a.W;
    if constexpr (PREC == 1) {  // split-half mode: the taps on the f16 pipe (lin_mfma_h), weights 3 x NTN x 256 floats' worth of halves
        constexpr float kScale = 4.0f * FACTOR * FACTOR;
        constexpr int LO = NPIX * 8;
        for (int k = tid; k < 3 * NTN * 256; k += NTHREADS) s_w[k] = wlin[k];
        for (int p = tid; p < NPIX; p += NTHREADS) {
            const int py = p / TWH, px = p - py * TWH;
            const int gy = min(max(y0 - 1 + py, 0), a.H - 1), gx = min(max(x0 - 1 + px, 0), a.W - 1);
            const size_t gp = img_px0 + (size_t)gy * a.W + gx;
            if constexpr (IMG_U8) {
                const uint8_t* q = (const uint8_t*)a.img + gp * a.img_ch;
                lin_split_store_u8(tile, LO, p, lin_split_table_entry(q[0], kScale), lin_split_table_entry(q[1], kScale), lin_split_table_entry(q[2], kScale));
            } else {
                const float* q = (const float*)a.img + gp * 3;
                lin_split_store_f32(tile, LO, p, q[0], q[1], q[2], kScale);
            }
        }
        __syncthreads();
        lin_mfma_h<TH, T, NTN>(acc, accx, tile, LO, (const f16x8*)s_w, wave, lane);
        return;
    }
    for (int k = tid; k < 9 * NTN * 128; k += NTHREADS) s_w[k] = wlin[k];
    for (int p = tid; p < NPIX; p += NTHREADS) {
        const int py = p / TWH, px = p - py * TWH;
        const int gy = min(max(y0 - 1 + py, 0), a.H - 1), gx = min(max(x0 - 1 + px, 0), a.W - 1);
        const size_t gp = img_px0 + (size_t)gy * a.W + gx;
        f32x4 v;
        v.x = load_img(a.img, a.img_ch, IMG_U8, gp, 0);
        v.y = load_img(a.img, a.img_ch, IMG_U8, gp, 1);
        v.z = load_img(a.img, a.img_ch, IMG_U8, gp, 2);
        v.w = 0.f;
        *(f32x4*)&s_x[p * 4] = v;
    }
    __syncthreads();
    lin_mfma<TH, T, NTN>(acc, s_x, s_w, wave, lane);
}

// What happens to a finished tile of the EXACT mode (32x32 accumulators: lane = output channel, registers = pixels): bias + BeLU -> the
// node's feature map, or, for the final stage, + expand_bias, depth-to-space (Expand, network.rs:39) and optionally the u8 quantiser.
// (The split-half mode's tiles: stage_epilogue_h16, stage_epilogue_final_t; the exact mode's last stage on 8-row tiles: stage_epilogue_quad.)
// Every epilogue returns how many store instructions this wave has CERTAINLY issued (wave-uniform conditions only; fewer is safe, more is not):
// tile_body counts them among the wave's memory operations, see step_advance.
template <int TH, int T, int NTN, bool FINAL, bool OUT_U8, int FACTOR>
__device__ __forceinline__ int stage_epilogue(const StageArgs& a, f32x16 (&acc)[NTN * T], const float (&bias)[NTN], float beta, int n, int x0, int y0,
                                              int wave, int lane, float* stage_lds = nullptr) {
    const int i = lane & 31, h = lane >> 5;
    const bool full_x = x0 + kTW <= a.W;
    int stores = 0;
    if constexpr (!FINAL) {
#pragma unroll
        for (int m = 0; m < T; ++m) {
            const int y = y0 + wave * T + m;
            if (y >= a.y_end) continue;
            float* base = a.dst + ((size_t)n * a.img_stride + (long)y * a.pitch + x0 + 4 * h) * 32 + i;
            if (full_x && stage_lds) {  // (pipe form: the row leaves through the wave's LDS scratch as four 1 KB stores, store_belu_tile_lds)
                store_belu_tile_lds(a.dst + ((size_t)n * a.img_stride + (long)y * a.pitch + x0) * 32, stage_lds, acc[m], bias[0], beta, lane);
                stores += 4;
            } else if (full_x) {
                store_belu_tile(base, acc[m], bias[0], beta);
                stores += 16;
            } else {
                for_each_acc_row([&](int r, int row) {
                    if (x0 + 4 * h + row < a.W) base[row * 32] = belu(__fadd_rn(acc[m][r], bias[0]), beta);
                });
            }
        }
    } else {
        // Expand (network.rs:39): out[f y+dy][f x+dx][c] = (bilinear + convs, all in acc) + expand_bias.
        // Lane i of N-tile nt (15 of every 16 lanes) owns colour c = (i % 16) % 3 of sub-pixel triple
        // tr = 10 nt + 5 (i / 16) + (i % 16) / 3 (dy = tr / f, dx = tr % f); the host packs the weights in that order
        // (sr_api.cpp expand_channel): a triple never straddles a 16-lane DPP row.
        const int OW = a.W * FACTOR;
        const int h_band = a.y_end - a.y_begin;
        const int jj = i & 15, tl = 5 * (i >> 4) + jj / 3, c = jj % 3;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            const int tr = nt * 10 + tl;
            const bool valid = jj < 15 && tr < FACTOR * FACTOR;
            const int trc = valid ? tr : 0;
            const int dy = trc / FACTOR, dx = trc - dy * FACTOR;
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const int y = y0 + wave * T + m;
                if (y >= a.y_end) continue;
                const f32x16& av = acc[nt * T + m];
                const size_t opx = ((size_t)n * h_band * FACTOR + (size_t)(y - a.y_begin) * FACTOR + dy) * OW +
                                   FACTOR * (x0 + 4 * h) + dx;
                // Every EXEC change in here costs matrix-pipe time (the pipe drains first), so a full-width tile computes all
                // sixteen values on every lane and stores them inside ONE lane-masked region; only the image's last tile column
                // tests each pixel.
                if constexpr (!OUT_U8) {
                    float* base = (float*)a.out + opx * 3 + c;
                    if (full_x) {
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const f32x2 s = f32x2{av[r], av[r + 1]} + f32x2{bias[nt], bias[nt]};
                            v[r] = s.x; v[r + 1] = s.y;
                        }
                        if (valid) for_each_acc_row([&](int r, int row) { base[row * FACTOR * 3] = v[r]; });
                    } else {
                        for_each_acc_row([&](int r, int row) {
                            if (valid && x0 + 4 * h + row < a.W) base[row * FACTOR * 3] = __fadd_rn(av[r], bias[nt]);
                        });
                    }
                } else {
                    // data_to_img (main.rs:175): clamp(floor(255 v + 0.5), 0, 255), alpha 255.  v_cvt_pk_u8_f32 saturates to
                    // [0, 255] and drops the byte into place c of the pixel in one instruction, but it rounds to nearest even
                    // (scripts/experiments/probe_cvt_pk_u8.hip), so it is fed the floor, an integer: clamp, convert and shift
                    // are then one instruction instead of three.  The lane holding R carries alpha in its `old` operand and ORs
                    // in the two lanes above it with row_shl DPP moves -- pure VALU (a __shfl_down is a ds_bpermute: LDS
                    // crossbar + an lgkmcnt wait per row).
                    uint32_t* base = (uint32_t*)a.out + opx;
                    const bool writer = valid && c == 0;
                    const uint32_t old = c == 0 ? 0xff000000u : 0u;
                    auto quant2 = [&](int r, uint32_t& p0, uint32_t& p1) {
                        const f32x2 s = (f32x2{av[r], av[r + 1]} + f32x2{bias[nt], bias[nt]}) * f32x2{255.0f, 255.0f} + f32x2{0.5f, 0.5f};
                        const uint32_t q0 = __builtin_amdgcn_cvt_pk_u8_f32(floorf(s.x), (uint32_t)c, old);
                        const uint32_t q1 = __builtin_amdgcn_cvt_pk_u8_f32(floorf(s.y), (uint32_t)c, old);
                        // (two 2-input ORs, kept apart by an empty asm: each then folds its DPP move into a v_or_b32_dpp -- two instructions
                        // per value; written as one expression the compiler forms a v_or3_b32, which cannot carry DPP: two moves + the OR)
                        p0 = q0 | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q0, 0x101, 0xF, 0xF, true);    // row_shl:1 (G)
                        p1 = q1 | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q1, 0x101, 0xF, 0xF, true);
                        asm volatile("" : "+v"(p0), "+v"(p1));
                        p0 |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q0, 0x102, 0xF, 0xF, true);        // row_shl:2 (B)
                        p1 |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q1, 0x102, 0xF, 0xF, true);
                    };
                    if (full_x) {
                        uint32_t px[16];
#pragma unroll
                        for (int r = 0; r < 16; r += 2) quant2(r, px[r], px[r + 1]);
                        if (writer) for_each_acc_row([&](int r, int row) { base[row * FACTOR] = px[r]; });
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            uint32_t p0, p1;
                            quant2(r, p0, p1);
                            const int row0 = (r & 3) + 8 * (r >> 2);
                            if (writer && x0 + 4 * h + row0 < a.W) base[row0 * FACTOR] = p0;
                            if (writer && x0 + 4 * h + row0 + 1 < a.W) base[(row0 + 1) * FACTOR] = p1;
                        }
                    }
                }
            }
        }
    }
    return stores;
}

// n consecutive dwords (n a constant after unrolling) at a 4-byte-aligned address, as the widest stores there are
template <int N>
struct __attribute__((packed, aligned(4))) DwordRun { uint32_t w[N]; };
template <int N>
__device__ __forceinline__ void store_run_n(char* o, const uint32_t* w) {
    DwordRun<N> r;
#pragma unroll
    for (int k = 0; k < N; ++k) r.w[k] = w[k];
    *(DwordRun<N>*)o = r;
}
__device__ __forceinline__ void store_run_dwords(char* o, const uint32_t* w, int n) {
    if (n == 1) store_run_n<1>(o, w);
    else if (n == 2) store_run_n<2>(o, w);
    else if (n == 3) store_run_n<3>(o, w);
    else if (n == 4) store_run_n<4>(o, w);
    else if (n == 6) store_run_n<6>(o, w);
    else if (n == 9) store_run_n<9>(o, w);
    else if (n == 12) store_run_n<12>(o, w);
    else __builtin_trap();  // (no other run length exists for factors 2-4; n is a constant at every call, so this leg compiles away)
}
// ... of the split-half mode's last stage (transposed accumulators, see half_steps_h): lane (i, h) holds, of pixel x0 + i of tile row m, the
// sixteen output slots 8 (r >> 2) + 4 h + (r & 3), r = 0..15, of every N-tile -- and the host packs the expand channels so that those are
// the five WHOLE RGB triples tl = 5 h + r / 3, colour r % 3, of sub-pixel tr = 10 nt + tl (sr_api.cpp expand_channel_t; r = 15 idle): the
// triples of "16-lane row" h of the exact mode's layout, whose biases a.bias[32 nt + 16 h + r] are therefore sixteen consecutive floats.
// Depth-to-space and the RGBA packing are lane-local (no DPP), a lane's sub-pixels of one output row are contiguous and stored together,
// consecutive lanes continue the row.  Same arithmetic per value as stage_epilogue.
template <int T, int NTN, bool OUT_U8, int FACTOR>
__device__ __forceinline__ int stage_epilogue_final_t(const StageArgs& a, f32x16 (&accm)[NTN * T], f32x16 (&accx)[NTN * T], const f32x4 (&fbias)[NTN][4],
                                                      int n, int x0, int y0, int wave, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const int OW = a.W * FACTOR, h_band = a.y_end - a.y_begin;
    constexpr int OPX = OUT_U8 ? 4 : 12;  // bytes per output pixel
    // (certain stores: a full-width tile's rows inside the band issue at least two run stores per N-tile over the two lane halves)
    int stores = 0;
    if (x0 + kTW <= a.W) {
#pragma unroll
        for (int m = 0; m < T; ++m) stores += (y0 + wave * T + m < a.y_end) ? 2 * NTN : 0;
    }
    if (x0 + i >= a.W) return stores;
    // output row of the wave's first tile row, sub-row 0, at the lane's pixel
    char* lbase = (char*)a.out + (((size_t)n * h_band + (size_t)(y0 + wave * T - a.y_begin)) * FACTOR * OW + (size_t)(x0 + i) * FACTOR) * OPX;
    auto body = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
#pragma unroll
        for (int m = 0; m < T; ++m) {
            if (y0 + wave * T + m >= a.y_end) continue;
            char* mbase = lbase + (size_t)m * FACTOR * OW * OPX;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const f32x16& am = accm[nt * T + m];
                const f32x16& ax = accx[nt * T + m];
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32p s = (f32p{am[r], am[r + 1]} + f32p{ax[r], ax[r + 1]} * f32p{1.0f / kLoScale, 1.0f / kLoScale}) +
                                   f32p{fbias[nt][r >> 2][r & 3], fbias[nt][(r + 1) >> 2][(r + 1) & 3]};
                    if constexpr (OUT_U8) {
                        // data_to_img (main.rs:175): clamp(floor(255 v + 0.5), 0, 255), alpha 255 -- v_cvt_pk_u8_f32 of the floor (see stage_epilogue)
                        const f32p q = s * f32p{255.0f, 255.0f} + f32p{0.5f, 0.5f};
                        v[r] = floorf(q.x); v[r + 1] = floorf(q.y);
                    } else {
                        v[r] = s.x; v[r + 1] = s.y;
                    }
                }
                // the lane's triples tl = 0..4 are sub-pixels tr = 10 nt + 5 H + tl (below f^2): a run of consecutive dx within one output row dy
                // starts at the lane's first triple and wherever dx = 0 (everything here is a constant after unrolling)
#pragma unroll
                for (int tl = 0; tl < 5; ++tl) {
                    constexpr int F2 = FACTOR * FACTOR;
                    const int tr = 10 * nt + 5 * H + tl;
                    if (tr >= F2) continue;
                    const int dy = tr / FACTOR, dx = tr % FACTOR;
                    if (tl != 0 && dx != 0) continue;
                    int run = FACTOR - dx;                    // to the end of the output row ...
                    if (run > 5 - tl) run = 5 - tl;           // ... of the lane's triples ...
                    if (run > F2 - tr) run = F2 - tr;         // ... of the sub-pixels
                    char* o = mbase + ((size_t)dy * OW + dx) * OPX;
                    uint32_t w[12];
                    int ndw;
                    if constexpr (OUT_U8) {
                        ndw = run;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (k >= run) continue;
                            uint32_t px = 0xff000000u;
                            px = __builtin_amdgcn_cvt_pk_u8_f32(v[3 * (tl + k) + 0], 0u, px);
                            px = __builtin_amdgcn_cvt_pk_u8_f32(v[3 * (tl + k) + 1], 1u, px);
                            px = __builtin_amdgcn_cvt_pk_u8_f32(v[3 * (tl + k) + 2], 2u, px);
                            w[k] = px;
                        }
                    } else {
                        ndw = 3 * run;
#pragma unroll
                        for (int k = 0; k < 12; ++k)
                            if (k < 3 * run) w[k] = __float_as_uint(v[3 * tl + k]);
                    }
                    store_run_dwords(o, w, ndw);
                }
            }
        }
    };
    if (h == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    return stores;
}

// ... of a tile computed in the quad form: the lane holds every expand channel of ITS pixel (x0 + lane % 32, tile row 2 wave + lane / 32),
// channel (triple tr = dy f + dx, colour c) in slot 16 (tl / 5) + 3 (tl % 5) + c of N-tile tr / 10, tl = tr % 10.  Same arithmetic per
// value as stage_epilogue (so the same bits); the f sub-pixels of an output row are contiguous in the lane (RGBA: f dwords, f32: 3 f
// floats) and consecutive lanes continue the row, every store inside one lane-masked region.
template <int NTN, bool OUT_U8, int FACTOR>
__device__ __forceinline__ int stage_epilogue_quad(const StageArgs& a, QuadAcc (&acc)[NTN], const float (&bias)[3 * FACTOR * FACTOR], int n, int x0, int y0,
                                                   int wave, int lane) {
    const int x = x0 + (lane & 31), y = y0 + wave * 2 + (lane >> 5);
    // (certain stores: with every lane's pixel inside the image, one store per output sub-row at least)
    const int stores = (x0 + kTW <= a.W && y0 + wave * 2 + 1 < a.y_end) ? FACTOR : 0;
    const int OW = a.W * FACTOR, h_band = a.y_end - a.y_begin;
    struct __attribute__((packed, aligned(4))) RowU8 { uint32_t px[FACTOR]; };
    struct __attribute__((packed, aligned(4))) RowF32 { float v[3 * FACTOR]; };
    constexpr int OPX = OUT_U8 ? 4 : 12;  // bytes per output pixel
    // output row of the wave's first tile row, sub-row 0, at the tile's first column: wave-uniform (scalar ALU); the lane adds its tile row and column
    char* wbase = (char*)a.out + (((size_t)n * h_band + (size_t)(y0 + wave * 2 - a.y_begin)) * FACTOR * OW + (size_t)x0 * FACTOR) * OPX;
    const uint32_t loff = ((uint32_t)(lane >> 5) * FACTOR * OW + (uint32_t)(lane & 31) * FACTOR) * OPX;
    if (x < a.W && y < a.y_end) {
#pragma unroll
        for (int dy = 0; dy < FACTOR; ++dy) {
            char* obase = wbase + (size_t)dy * OW * OPX;
            float v[3 * FACTOR + 1];
#pragma unroll
            for (int k = 0; k < 3 * FACTOR; ++k) {
                const int tr = dy * FACTOR + k / 3, nt = tr / 10, tl = tr % 10, slot = 16 * (tl / 5) + 3 * (tl % 5) + k % 3;
                v[k] = acc[nt][slot >> 2][slot & 3];
            }
            v[3 * FACTOR] = 0.f;
            if constexpr (!OUT_U8) {
                RowF32 o;
#pragma unroll
                for (int k = 0; k < 3 * FACTOR; k += 2) {
                    const f32x2 b2 = {bias[dy * 3 * FACTOR + k], k + 1 < 3 * FACTOR ? bias[dy * 3 * FACTOR + k + 1] : 0.f};
                    const f32x2 s = f32x2{v[k], v[k + 1]} + b2;
                    o.v[k] = s.x;
                    if (k + 1 < 3 * FACTOR) o.v[k + 1] = s.y;
                }
                *(RowF32*)(obase + loff) = o;
            } else {
                // data_to_img (main.rs:175): clamp(floor(255 v + 0.5), 0, 255), alpha 255 -- v_cvt_pk_u8_f32 of the floor (see stage_epilogue)
                RowU8 o;
                float q[3 * FACTOR + 1];
#pragma unroll
                for (int k = 0; k < 3 * FACTOR; k += 2) {
                    const f32x2 b2 = {bias[dy * 3 * FACTOR + k], k + 1 < 3 * FACTOR ? bias[dy * 3 * FACTOR + k + 1] : 0.f};
                    const f32x2 s = (f32x2{v[k], v[k + 1]} + b2) * f32x2{255.0f, 255.0f} + f32x2{0.5f, 0.5f};
                    q[k] = floorf(s.x); q[k + 1] = floorf(s.y);
                }
#pragma unroll
                for (int dx = 0; dx < FACTOR; ++dx) {
                    uint32_t w = 0xff000000u;
                    w = __builtin_amdgcn_cvt_pk_u8_f32(q[3 * dx + 0], 0u, w);
                    w = __builtin_amdgcn_cvt_pk_u8_f32(q[3 * dx + 1], 1u, w);
                    w = __builtin_amdgcn_cvt_pk_u8_f32(q[3 * dx + 2], 2u, w);
                    o.px[dx] = w;
                }
                *(RowU8*)(obase + loff) = o;
            }
        }
    }
    return stores;
}

// ... of a tile computed on 16x16 accumulators (kH16; row-planar split-half map; channels in registers, see half_steps_h16): lane l holds
// output channels 16 ch + 4 (l >> 4) + (0..3) of pixel 16 ph + (l & 15) of tile row m.  BeLU, the hi / lo split, then one 8-byte store of the
// four hi halves and one of the four lo halves: half a 16-byte cell of channel group 2 ch + (l >> 5) each (lanes l and l ^ 16 complete it).
template <int T>
__device__ __forceinline__ int stage_epilogue_h16(const StageArgs& a, f32x4 (&accm)[T][2][2], f32x4 (&accx)[T][2][2],
                                                  const f32x4 (&beta)[2], int n, int x0, int y0, int wave, int lane, uint32_t& dom) {
    int stores = 0;
    const int p16 = lane & 15, g = lane >> 4;
    const bool full_x = x0 + kTW <= a.W;
    const long lo_off = (long)a.pitch * 64;  // the lo group's row: four channel groups further on
#pragma unroll
    for (int m = 0; m < T; ++m) {
        const int y = y0 + wave * T + m;
        if (y >= a.y_end) continue;
        if (full_x) stores += 8;  // (2 channel halves x 2 pixel halves x hi / lo)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            // channel group 2 ch + (g >> 1), bytes 8 (g & 1) .. + 7 of the pixel's 16-byte cell
            char* row = (char*)(a.dst + ((size_t)n * a.img_stride + (long)y * a.pitch) * 32) + (size_t)(2 * ch + (g >> 1)) * a.pitch * 16 + (g & 1) * 8;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int x = x0 + 16 * ph + p16;
                const f32x4 vm = accm[m][ph][ch], vx = accx[m][ph][ch];  // (the bias is in accm: see split_value)
                const f32p v01 = belu2_fused2(split_value(f32p{vm[0], vm[1]}, f32p{vx[0], vx[1]}), f32p{beta[ch][0], beta[ch][1]});
                const f32p v23 = belu2_fused2(split_value(f32p{vm[2], vm[3]}, f32p{vx[2], vx[3]}), f32p{beta[ch][2], beta[ch][3]});
                uint32_t h01, l01, h23, l23;
                split_half2(v01, h01, l01);
                split_half2(v23, h23, l23);
                domain_track(dom, h01);
                domain_track(dom, h23);
                if (full_x || x < a.W) {
                    store_map8(row + (long)x * 16, h01, h23);
                    store_map8(row + (long)x * 16 + lo_off, l01, l23);
                }
            }
        }
    }
    return stores;
}

// Dynamic tile queue of the persistent forms.  A launch has `nbig` 8-row tiles and `nsmall` 4-row tiles (either may be
// 0).  Each class is cut into 8 contiguous runs, one per XCD (the dispatcher places block b on XCD b % 8: neighbouring
// tiles share halo rows in that XCD's L2); an XCD's queue is its run of big tiles followed by its run of small ones, with
// one head counter in HBM (zeroed before the launch).  A workgroup pulls from its own XCD's queue and, once that is
// exhausted, steals from the others.  Returned: a big tile's id t as t, a small tile's as nbig + t, -1 when all is gone.
__device__ __forceinline__ void run_of_xcd(int x, int ntiles, int& start, int& count) {
    const int q = ntiles >> 3, r = ntiles & 7;
    count = q + (x < r ? 1 : 0);
    start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
__device__ __forceinline__ int queue_slot(int x, int nbig, int nsmall, int idx) {
    int start, count;
    run_of_xcd(x, nbig, start, count);
    if (idx < count) return start + idx;
    idx -= count;
    run_of_xcd(x, nsmall, start, count);
    return idx < count ? nbig + start + idx : -1;
}
__device__ __forceinline__ int queue_total(int x, int nbig, int nsmall) {
    int start, cb, cs;
    run_of_xcd(x, nbig, start, cb);
    run_of_xcd(x, nsmall, start, cs);
    return cb + cs;
}
// Workgroup b's FIRST tile needs no atomic: it is entry b / 8 of queue b % 8, and the heads start at the number of
// workgroups that own such an entry (conv0, the call's first launch, writes them: Conv0Args::queue_grid).
__device__ __forceinline__ int queue_first(int block, int nbig, int nsmall) { return queue_slot(block & 7, nbig, nsmall, block >> 3); }
// First form of the stage kernels: ONE workgroup per tile (grid = tiles of one class, XCD-remapped), the whole source
// tile resident in LDS, sources staged one after the other.  It is what small launches of the exact-f32 mode run (no
// queue, nothing to amortise: 256x256 is one round of 4-row tiles) and the in-library cross-check of the pipe form
// (sr_set_experiment "pipe" = "none"): same matrix loops, same step order, same weight chunks, bit-identical results.
template <int TH, int NSRC, int KS0, bool FINAL, bool IMG_U8, bool OUT_U8, int PREC, int FACTOR = 3>
__global__ __launch_bounds__(256, 2) void conv_stage_kernel(StageArgs a) {
    // Two workgroups share each SIMD.  A wave streaming MFMAs is the older one and wins every
    // arbitration, leaving the other workgroup's prologue / staging / epilogue code roughly one
    // issue slot per MFMA.  The matrix stream only needs one slot per 64 cycles, so everything
    // that is not the tap loop runs at raised priority -- from the first instruction on.
    __builtin_amdgcn_s_setprio(3);
    constexpr int NW = 4, T = TH / NW;  // tile rows per wave
    // The final stage has 3 f^2 expand channels (network.rs:37).  They are laid out in whole RGB
    // triples, 10 per 32-lane N-tile (so the u8 packing never straddles tiles): f = 2, 3 -> 1 tile,
    // f = 4 -> 16 triples -> 2 tiles.
    constexpr int NTN = FINAL ? (FACTOR * FACTOR + 9) / 10 : 1;
    using G0 = TileGeom<TH, KS0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;
    char* ring = smem + 8 * G0::PLANE;  // KS0 >= 3: the first source has the largest tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    constexpr bool H16 = kH16<PREC, FINAL>;  // stages 1-3 of the split-half mode: 16x16x32 MFMAs, no half steps (half_steps_h16)
    constexpr int NTAPS = H16 ? KS0 * KS0 + (NSRC - 1) * 9
                              : 2 * ((KS0 * KS0 + 1) / 2 + (NSRC - 1) * 5) * NTN;  // ring chunks: one per (step, N-tile), see half_steps_*
    const TileGrid& grid = a.grid[TH == 8 ? 0 : 1];  // this form runs one tile class per launch
    float bias[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) bias[nt] = a.bias[nt * 32 + i];
    const float beta = FINAL ? 0.f : a.beta[i];

    TileOffsets<TH, KS0, NW> off0;
    TileOffsets<TH, 3, NW> off3;
    off0.template init<PREC>(a.pitch, wave, lane);
    if constexpr (NSRC >= 2) off3.template init<PREC>(a.pitch, wave, lane);
    // everything the first phase needs: weight chunks 0..3 and the first source tile
    int n, tx, ty;
    tile_coords(grid, a.tiles_x, xcd_remap(blockIdx.x, gridDim.x), n, tx, ty);
    const int x0 = tx * kTW, y0 = grid.y0 + ty * TH;
#pragma unroll
    for (int k = 0; k < kRingAhead; ++k) weight_chunk_async(ring + k * 4096, a.wpack + k * kChunkFloats, wave, lane);
    stage_tile<TH, KS0, PREC, NW>(tile, a.src[0], off0, a.img_stride, a.pitch, n, y0, x0, wave, lane);

    f32x16 acc[NTN * T], accx[PREC == 1 ? NTN * T : 1];  // accx: the cross products of the split-half mode, x2048
#pragma unroll
    for (int m = 0; m < NTN * T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[m][r] = (PREC == 1 && !FINAL) ? bias[0] : 0.f;  // (split-half producers: the bias rides in the accumulator, see split_value)
            if constexpr (PREC == 1) accx[m][r] = 0.f;
        }
    f32x4 qm[H16 ? T : 1][2][2], qx[H16 ? T : 1][2][2];  // ... or, kH16, the same tile as 16x16 accumulators
    // (kH16: lane l holds output channels 16 ch + 4 (l >> 4) + (0..3) of its pixels)
    f32x4 bias2[2], beta2[2];
    if constexpr (H16) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            bias2[ch] = *(const f32x4*)(a.bias + 16 * ch + 4 * (lane >> 4));
            beta2[ch] = *(const f32x4*)(a.beta + 16 * ch + 4 * (lane >> 4));
        }
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) { qm[m][k >> 1][k & 1] = bias2[k & 1]; qx[m][k >> 1][k & 1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    int gtap = 0, slot = 0;
    auto taps = [&](auto ks_tag) {
        constexpr int KS = decltype(ks_tag)::value;
        if constexpr (H16) source_steps_h16<TH, KS, T>(qm, qx, tile, ring, a.wpack, gtap, slot, NTAPS, wave, lane);
        else source_steps<TH, KS, T, NTN, PREC>(acc, accx, tile, ring, a.wpack, gtap, slot, NTAPS, wave, lane);
    };
    ring_barrier<0>();  // every wave's tile + weight DMAs have landed
    __builtin_amdgcn_s_setprio(0);
    taps(std::integral_constant<int, KS0>{});
    if constexpr (NSRC >= 2) {
        __builtin_amdgcn_s_setprio(3);
        stage_tile<TH, 3, PREC, NW>(tile, a.src[1], off3, a.img_stride, a.pitch, n, y0, x0, wave, lane);
        ring_barrier<0>();
        __builtin_amdgcn_s_setprio(0);
        taps(std::integral_constant<int, 3>{});
    }
    if constexpr (NSRC >= 3) {
        __builtin_amdgcn_s_setprio(3);
        stage_tile<TH, 3, PREC, NW>(tile, a.src[2], off3, a.img_stride, a.pitch, n, y0, x0, wave, lane);
        ring_barrier<0>();
        __builtin_amdgcn_s_setprio(0);
        taps(std::integral_constant<int, 3>{});
    }
    __builtin_amdgcn_s_setprio(3);
    if constexpr (FINAL)
        lin_taps<TH, T, IMG_U8, NW * 64, NTN, PREC, FACTOR>(acc, accx, tile, ring, a, a.wpack + (size_t)NTAPS * kChunkFloats, n, y0, x0, wave, lane, tid);
    uint32_t dom = 0;
    if constexpr (H16) {
        stage_epilogue_h16<T>(a, qm, qx, beta2, n, x0, y0, wave, lane, dom);
    } else if constexpr (FINAL && PREC == 1) {
        f32x4 fbias[NTN][4];  // (transposed accumulators: lane (i, h) holds the slots whose biases are a.bias[32 nt + 16 h + (0..15)])
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) fbias[nt][k] = *(const f32x4*)(a.bias + 32 * nt + 16 * (lane >> 5) + 4 * k);
        stage_epilogue_final_t<T, NTN, OUT_U8, FACTOR>(a, acc, accx, fbias, n, x0, y0, wave, lane);
    } else {
        static_assert(PREC == 0, "the split-half mode's tiles have epilogues of their own");
        stage_epilogue<TH, T, NTN, FINAL, OUT_U8, FACTOR>(a, acc, bias, beta, n, x0, y0, wave, lane);
    }
    if constexpr (PREC == 1 && !FINAL) domain_report(dom, a.domain);
}

// ---------------------------------------------------------------------------
// Stage kernel, second form ("pipe"): every source tile is staged in two HALVES of 16 input channels
// (4 LDS planes, 28 KB for a 5x5 halo) into two buffers that alternate, so the gather DMA of half j+1
// -- and, at the end of a tile, of the NEXT tile's first half -- runs under the taps of half j.  Nothing
// but the epilogue is left outside the matrix stream: no staging waits between sources, no prologue per
// tile (persistent workgroups, tiles from the per-XCD queue).  A step = two taps of one half = one 4 KB
// weight chunk = 32 f32 / 12 f16 MFMAs per wave, same size as a whole tap of the first form.
//
// One launch runs BOTH tile classes of StageArgs: 8-row tiles (two tile rows per wave) and, after them in every
// XCD's queue, 4-row tiles (one row per wave).  What a launch loses at its end is the spread of its workgroups'
// finishing times -- measured on MI355X (profiles/r3_band_profile_baseline_*.jsonl): half a tile time per launch,
// 35-50 us for the 5x5 stages whatever the image size -- so the last tiles handed out are half-size.  A small tile
// lives in the first rows of the same LDS buffers (same plane stride, same row pitch TWH: its pixel p sits where the
// big tile's pixel p does), so the two tile bodies differ in nothing but T, the rows per wave, and the number of
// gather groups requested; both sum in the same order as the first form, so all forms stay bit-identical.
// ---------------------------------------------------------------------------
// s_waitcnt vmcnt(pending) [lgkmcnt(0)]; s_barrier.  LGKM = false leaves this wave's LDS reads in flight across the
// barrier.  That is safe at the end of a step: what the next step's DMAs overwrite -- the ring slot of the chunk consumed
// one step ago, the half-tile buffer the previous half has left -- was read into registers at least one step earlier, and
// those reads had to return before the MFMAs that consumed them could issue; the reads still in flight are the NEXT
// step's operands (another ring slot, the current half's buffer).  It matters in the split-half mode, whose step is only
// 12 x 32 cycles of matrix work: with lgkmcnt(0) a wave sat out one LDS round trip per step in front of the barrier.
// (A plain LDS WRITE that another wave reads behind the barrier -- the tile queue's mailbox -- waits for itself.)
// The immediate of s_waitcnt must be a constant, `pending` is a run-time number (how many of the newest DMAs may stay in flight:
// StepStream).  A C switch over it compiles to a chain of ~50 scalar instructions and ~10 branches per wait (and, where the
// compiler does not see that the number is wave-uniform, to a tree of EXEC-masked branches with v_cmp / s_and_saveexec, each EXEC
// write draining the matrix pipe) -- once per step, i.e. per 12 MFMAs of 32 cycles in the split-half mode.  Here: one indexed
// jump into a table of sixteen `s_waitcnt vmcnt(k); s_branch end` pairs (8 bytes each), eight scalar instructions and two taken
// branches whatever the number.  Fewer than `pending` in flight is always safe, so the index is min(pending, 15), and a negative
// number (never produced) waits for everything.
#define SR_WAIT_ROW(k, extra) "s_waitcnt vmcnt(" #k ")" extra "\n\ts_branch .Lsrwait%=\n\t"
#define SR_WAIT_ROWS10(b, extra)                                                                                              \
    SR_WAIT_ROW(b##0, extra) SR_WAIT_ROW(b##1, extra) SR_WAIT_ROW(b##2, extra) SR_WAIT_ROW(b##3, extra) SR_WAIT_ROW(b##4, extra) \
    SR_WAIT_ROW(b##5, extra) SR_WAIT_ROW(b##6, extra) SR_WAIT_ROW(b##7, extra) SR_WAIT_ROW(b##8, extra) SR_WAIT_ROW(b##9, extra)
#define SR_WAIT_JUMP(maxn)                                                                                                   \
    "s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, " #maxn "\n\ts_lshl_b32 %0, %0, 3\n\ts_add_u32 %0, %0, 12\n\t"                 \
    "s_getpc_b64 vcc\n\ts_add_u32 vcc_lo, vcc_lo, %0\n\ts_addc_u32 vcc_hi, vcc_hi, 0\n\ts_setpc_b64 vcc\n\t"
#define SR_WAIT_TABLE(extra)                                                                                                \
    SR_WAIT_JUMP(15) SR_WAIT_ROWS10(, extra) SR_WAIT_ROW(10, extra) SR_WAIT_ROW(11, extra) SR_WAIT_ROW(12, extra)               \
    SR_WAIT_ROW(13, extra) SR_WAIT_ROW(14, extra) SR_WAIT_ROW(15, extra) ".Lsrwait%=:"
// ... and the same with all 64 values of the counter (512 bytes): only for the waits that follow an epilogue, see step_advance
#define SR_WAIT_TABLE64(extra)                                                                                              \
    SR_WAIT_JUMP(63) SR_WAIT_ROWS10(, extra) SR_WAIT_ROWS10(1, extra) SR_WAIT_ROWS10(2, extra) SR_WAIT_ROWS10(3, extra)         \
    SR_WAIT_ROWS10(4, extra) SR_WAIT_ROWS10(5, extra) SR_WAIT_ROW(60, extra) SR_WAIT_ROW(61, extra) SR_WAIT_ROW(62, extra)      \
    SR_WAIT_ROW(63, extra) ".Lsrwait%=:"
template <bool LGKM, bool WIDE = false>
__device__ __forceinline__ void wait_vm_n(int pending) {  // s_waitcnt vmcnt(min(pending, 15 or 63)) [lgkmcnt(0)]
    if (__builtin_constant_p(pending)) {  // (resolved after inlining and unrolling: many steps know their number at compile time)
        switch (pending < 0 ? 0 : pending) {
#define SR_CASE(k) case k: if constexpr (LGKM) asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            SR_CASE(0) SR_CASE(1) SR_CASE(2) SR_CASE(3) SR_CASE(4) SR_CASE(5) SR_CASE(6) SR_CASE(7) SR_CASE(8) SR_CASE(9) SR_CASE(10)
            SR_CASE(11) SR_CASE(12) SR_CASE(13) SR_CASE(14)
#undef SR_CASE
            default: if constexpr (LGKM) asm volatile("s_waitcnt vmcnt(15) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        }
        return;
    }
    const int p = __builtin_amdgcn_readfirstlane(pending);
    int t;
    if constexpr (WIDE) {
        if constexpr (LGKM) asm volatile(SR_WAIT_TABLE64(" lgkmcnt(0)") : "=&s"(t) : "s"(p) : "vcc", "scc", "memory");
        else asm volatile(SR_WAIT_TABLE64("") : "=&s"(t) : "s"(p) : "vcc", "scc", "memory");
    } else {
        if constexpr (LGKM) asm volatile(SR_WAIT_TABLE(" lgkmcnt(0)") : "=&s"(t) : "s"(p) : "vcc", "scc", "memory");
        else asm volatile(SR_WAIT_TABLE("") : "=&s"(t) : "s"(p) : "vcc", "scc", "memory");
    }
}
template <bool LGKM, bool WIDE = false>
__device__ __forceinline__ void wait_vm_barrier(int pending) {
    wait_vm_n<LGKM, WIDE>(pending);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int KS>
struct HalfTile {
    using G = TileGeom<8, KS>;                           // LDS geometry: always that of the 8-row tile
    static constexpr int NG_SMALL = TileGeom<4, KS>::NG; // gather groups that cover a 4-row tile (+halo)
    static constexpr int BYTES = 4 * G::PLANE;
    static constexpr int STEPS = (KS * KS + 1) / 2;
    static_assert(TileGeom<4, KS>::TWH == G::TWH, "a small tile is the top of a big one");
    uint32_t off[G::NG];  // gather offset of tile pixel 64 g + lane (same for every wave: wave w moves plane w)
    template <int PREC>
    __device__ __forceinline__ void init(int pitch, int lane) {
#pragma unroll
        for (int g = 0; g < G::NG; ++g) {
            const int P = min(g * 64 + lane, G::NPIX - 1);
            const int prow = P / G::TWH, pcol = P - prow * G::TWH;
            off[g] = kPlanar<PREC> ? (uint32_t)(prow * pitch) * 128u + (uint32_t)pcol * 16u : (uint32_t)(prow * pitch + pcol) * 128u;
        }
    }
    // 16-byte channel group wave `wave` moves for half `khalf`: f32 map = 8 groups of 4 channels; split map = 4 groups of
    // hi halves then 4 of lo halves (8 channels each)
    template <int PREC>
    static __device__ __forceinline__ int chunk_of(int khalf, int wave) {
        return PREC == 0 ? 4 * khalf + wave : (wave < 2 ? 2 * khalf + wave : 4 + 2 * khalf + (wave - 2));
    }
    // request half `khalf` of the tile at (n, y0, x0) of `src` into the LDS buffer `buf`: G::NG DMAs per wave
    template <int PREC>
    __device__ __forceinline__ void stage(uint32_t buf, const float* __restrict__ src, int khalf, long img_stride, int pitch,
                                          int n, int y0, int x0, int wave) const {
        const char* origin = origin_of<PREC>(src, khalf, img_stride, pitch, n, y0, x0, wave);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(buf + wave * G::PLANE);
#pragma unroll
        for (int g = 0; g < G::NG; ++g) lds_dma16<0>(origin, off[g], dst + g * 1024);
    }
    // the same request in pieces: where this wave's channel group of the tile starts in HBM / its plane in LDS (once per half) ...
    template <int PREC>
    static __device__ __forceinline__ const char* origin_of(const float* __restrict__ src, int khalf, long img_stride, int pitch, int n,
                                                            int y0, int x0, int wave) {
        if constexpr (kPlanar<PREC>)  // row-planar map: the channel group's own row, pitch x 16 bytes per group
            return uniform_ptr((const char*)(src + ((size_t)n * img_stride + (long)(y0 - G::R) * pitch) * 32) + (long)(x0 - G::R) * 16 +
                               (size_t)chunk_of<PREC>(khalf, wave) * pitch * 16);
        else
            return uniform_ptr((const char*)(src + ((size_t)n * img_stride + (long)(y0 - G::R) * pitch + (x0 - G::R)) * 32) +
                               chunk_of<PREC>(khalf, wave) * 16);
    }
    static __device__ __forceinline__ uint32_t plane_of(uint32_t buf, int wave) { return __builtin_amdgcn_readfirstlane(buf + wave * G::PLANE); }
    // ... and one gather instruction of it: pixel group g (64 tile pixels)
    __device__ __forceinline__ void stage_one(int g, const char* origin, uint32_t dst) const { lds_dma16<0>(origin, off[g], dst + g * 1024); }
};

// Bookkeeping of the pipe form's DMA traffic.  Every LDS-DMA instruction a wave issues gets a sequence number;
// loads complete in issue order, so "everything up to number q has landed" is s_waitcnt vmcnt(issued - q).
// (Stores and plain loads issued in between and not counted only make that wait stricter, never weaker.)
// Weight chunks: step gs of a tile uses chunk gs; it is requested kRingAhead steps earlier, wrapping into the
// next tile.  Half tiles: requested piecemeal over the first steps of the previous half (a burst of gathers
// stalls the issuing wave for ~250 cycles per instruction), needed at its end.
struct StepStream {
    int slot;            // ring slot of the current step's chunk
    int nsteps;          // steps per tile
    bool have_next;      // another tile follows (its chunks are requested by the last steps of this one)
    int issued;          // DMA instructions issued so far by this wave (+ the epilogues' stores, as far as they are certain: tile_body)
    int q[kRingAhead];   // sequence numbers of the requests of chunks gs + 1 .. gs + kRingAhead
    int tile_seq;        // sequence number of the newest half-tile DMA
};

// `gs`: this step's number within the tile -- a constant at every call site once the half's loops are unrolled (PipeStream counts
// it), so that the chunk's byte offset is an immediate; `wbase`: the weight pack + this wave's quarter of a chunk.
__device__ __forceinline__ void step_request(StepStream& st, int gs, uint32_t ring_lds, const char* wbase, int lane) {
    int req = gs + kRingAhead;
    bool go = true;
    if (req >= st.nsteps) {
        go = st.have_next;
        req -= st.nsteps;
    }
#pragma unroll
    for (int k = 0; k + 1 < kRingAhead; ++k) st.q[k] = st.q[k + 1];
    if (go) {
        int s2 = st.slot + kRingAhead;
        if (s2 >= kRingSlots) s2 -= kRingSlots;
        lds_dma16_at(wbase, (uint32_t)req * 4096u, (uint32_t)(lane * 16), ring_lds + s2 * 4096);
        ++st.issued;
    }
    st.q[kRingAhead - 1] = st.issued;  // nothing requested: nothing newer to wait for either
}
// End of a step.  On return chunk gs + 1 + EXTRA has landed; with `tile` also every half-tile DMA issued so far.
// `wide` (a constant after unrolling): one of a tile's first steps.  The chunk such a step waits for was requested BEFORE the previous tile's
// epilogue, whose stores sit behind it in the wave's one in-order memory counter (gfx9 has no separate store counter): they are counted
// (StepStream::issued, see tile_body) so that the wait lets them stay in flight -- which takes a number beyond 15, the 64-row table.
// Uncounted, the first barriers of every split-half tile sat out the write acknowledgement of the whole epilogue (its steps are a few
// hundred cycles): 0.8 % of a frame, profiles/r6_ab_store_shadow.txt.
template <int EXTRA>
__device__ __forceinline__ void step_advance(StepStream& st, bool tile, bool wide = false) {
    // after this step's step_request, q[k] is the request of chunk gs + 1 + k
    int need = st.q[EXTRA];
    if (tile && st.tile_seq > need) need = st.tile_seq;
    // (EXTRA = 1: the split-half loop, which reads a step ahead)
    if (wide) wait_vm_barrier<EXTRA == 0, true>(st.issued - need);
    else wait_vm_barrier<EXTRA == 0>(st.issued - need);
    st.slot = st.slot == kRingSlots - 1 ? 0 : st.slot + 1;
}

// What a half's steps should request on the side: the first `ng` gather groups of half `khalf` of the tile at
// (n, y0, x0) of `src` into `buf`.  ng = 0: nothing (last half of the last tile).
struct HalfRequest {
    int ng;
    const char* origin;  // this wave's 16-byte channel group of the tile's first (halo) pixel: formed ONCE per half, in SGPRs
    uint32_t dst;        // this wave's plane of the LDS buffer
};

__device__ __forceinline__ void wait_vm(int pending) { wait_vm_n<false>(pending); }  // s_waitcnt vmcnt(pending), any pending (> 15: 15)

// Final stage of the pipe form: the input pixels the bilinear taps need (a (TH+2) x (32+2) tile, edge-replicated
// coordinates: LinearInterp clamps indices) are requested at the start of the tile's LAST half and parked in
// registers, so that their latency runs under that half's matrix work.  Inline-asm loads: the compiler would
// otherwise wait for them with vmcnt(0), i.e. for every DMA of the next tile too; here the wait is numbered like
// all the others (StepStream).  A thread loads up to two pixels (an 8-row tile has 340, a 4-row tile 204).
template <bool IMG_U8, int TH>
struct LinPrefetch {
    static constexpr int TWH = kTW + 2, THH = TH + 2, NPIX = THH * TWH, PER = (NPIX + 255) / 256;
    uint32_t raw[PER][3];
    int seq;
    __device__ __forceinline__ void issue(const StageArgs& a, int n, int y0, int x0, int tid, StepStream& st) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = min(tid + 256 * k, NPIX - 1);
            const int py = p / TWH, px = p - py * TWH;
            const int gy = min(max(y0 - 1 + py, 0), a.H - 1), gx = min(max(x0 - 1 + px, 0), a.W - 1);
            const size_t gp = (size_t)n * a.H * a.W + (size_t)gy * a.W + gx;
            if constexpr (IMG_U8) {
                const uint8_t* q = (const uint8_t*)a.img + gp * a.img_ch;
                asm volatile("global_load_ubyte %0, %1, off" : "=v"(raw[k][0]) : "v"(q) : "memory");
                asm volatile("global_load_ubyte %0, %1, off offset:1" : "=v"(raw[k][1]) : "v"(q) : "memory");
                asm volatile("global_load_ubyte %0, %1, off offset:2" : "=v"(raw[k][2]) : "v"(q) : "memory");
            } else {
                const float* q = (const float*)a.img + gp * 3;
                asm volatile("global_load_dword %0, %1, off" : "=v"(raw[k][0]) : "v"(q) : "memory");
                asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(raw[k][1]) : "v"(q) : "memory");
                asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(raw[k][2]) : "v"(q) : "memory");
            }
        }
        st.issued += 3 * PER;
        seq = st.issued;
    }
    // split-half mode: the same pixels divided by `scale` = (2 f)^2 and split into hi / lo halves (lin_mfma_h); s_lut2: lin_split_table_entry per byte
    __device__ __forceinline__ void store_split(char* s_hi, int lo_off, const uint32_t* s_lut2, float scale, int tid, const StepStream& st) {
        wait_vm(st.issued - seq);
#pragma unroll
        for (int k = 0; k < PER; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) asm volatile("" : "+v"(raw[k][ch]));  // (see store: no use before the wait)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = tid + 256 * k;
            if (p < NPIX) {
                if constexpr (IMG_U8) lin_split_store_u8(s_hi, lo_off, p, s_lut2[raw[k][0]], s_lut2[raw[k][1]], s_lut2[raw[k][2]]);
                else lin_split_store_f32(s_hi, lo_off, p, __uint_as_float(raw[k][0]), __uint_as_float(raw[k][1]), __uint_as_float(raw[k][2]), scale);
            }
        }
    }
    // wait for the pixels, convert (img_to_data: u8 / 255, true division) and write the [pixel][4] tile
    // (u8: through the table of byte / 255 -- the division itself is ~10 vector-ALU instructions per sample, in the matrix stream)
    __device__ __forceinline__ void store(float* s_x, const float* s_lut, int tid, const StepStream& st) {
        wait_vm(st.issued - seq);
        // (the loads above returned into raw[] behind the compiler's back: tie every use to this point, after the wait -- without the
        // dependence a pure use, the conversion below, may be scheduled above the wait; see also scripts/check_async_regs.py)
#pragma unroll
        for (int k = 0; k < PER; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) asm volatile("" : "+v"(raw[k][ch]));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int p = tid + 256 * k;
            if (p < NPIX) {
                f32x4 v;
                if constexpr (IMG_U8) {
                    v.x = s_lut[raw[k][0]]; v.y = s_lut[raw[k][1]]; v.z = s_lut[raw[k][2]];
                } else {
                    v.x = __uint_as_float(raw[k][0]); v.y = __uint_as_float(raw[k][1]); v.z = __uint_as_float(raw[k][2]);
                }
                v.w = 0.f;
                *(f32x4*)&s_x[p * 4] = v;
            }
        }
    }
};

// A workgroup's dealings with the tile queue, kept out of the matrix stream's way.  HIP's atomicAdd() is followed at once
// by s_waitcnt vmcnt(0) (the compiler needs the value for its cross-lane combining) -- at the start of every tile that
// drained the weight ring and parked wave 0 for a memory round trip while the other waves waited at the first barrier.
// Here the next tile's index is requested by ONE inline-asm returning atomic at the tile's start, numbered like every
// other request (StepStream), and looked at three steps before half 0 ends, when it has long returned.  Only when this
// XCD's queue has run dry does wave 0 read all eight heads (one load, lanes 0-7) and, at the end of the half, try those
// queues that still showed tiles: the tail of a launch costs no round of seven failing atomics per workgroup.
struct QueueState {      // meaningful in wave 0
    uint32_t pulled;     // lane 0: what the atomic on this XCD's head returned
    uint32_t heads;      // lanes 0-7: snapshot of the eight heads
    int seq_pull, seq_snap;
    int own;             // the tile `pulled` stands for, -1: this XCD's queue is exhausted
};
__device__ __forceinline__ void queue_pull_async(const StageArgs& a, int xcd, int wave, int lane, StepStream& st, QueueState& qs) {
    if (wave != 0) return;
    if (lane == 0) {
        const int* p = a.queue + xcd;
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(qs.pulled) : "v"(p), "v"(1u) : "memory");
    }
    qs.seq_pull = ++st.issued;
}
__device__ __forceinline__ void queue_presolve(const StageArgs& a, int xcd, int wave, int lane, StepStream& st, QueueState& qs) {
    if (wave != 0) return;
    wait_vm(st.issued - qs.seq_pull);
    asm volatile("" : "+v"(qs.pulled));  // the atomic's result: no use of it may move above the wait
    const int idx = __builtin_amdgcn_readfirstlane((int)qs.pulled);
    qs.own = queue_slot(xcd, a.grid[0].ntiles, a.grid[1].ntiles, idx);
    if (qs.own < 0) {
        const int* p = a.queue + (lane & 7);
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(qs.heads) : "v"(p) : "memory");
        qs.seq_snap = ++st.issued;
    }
}
// the next tile's number (or -1) into the mailbox; called before the barrier that ends half 0
__device__ __forceinline__ void queue_publish(const StageArgs& a, int xcd, int wave, int lane, StepStream& st, QueueState& qs, volatile int* mailbox) {
    if (wave != 0) return;
    int t = qs.own;
    if (t < 0) {
        const int nbig = a.grid[0].ntiles, nsmall = a.grid[1].ntiles;
        wait_vm(st.issued - qs.seq_snap);
        asm volatile("" : "+v"(qs.heads));
        for (int k = 1; k < 8 && t < 0; ++k) {
            const int x = (xcd + k) & 7;
            if ((int)__builtin_amdgcn_readlane((int)qs.heads, x) >= queue_total(x, nbig, nsmall)) continue;  // heads only grow: nothing there
            int j = 0;
            if (lane == 0) j = atomicAdd(&a.queue[x], 1);
            t = queue_slot(x, nbig, nsmall, __builtin_amdgcn_readfirstlane(j));
        }
    }
    if (lane == 0) *mailbox = t;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the write is in LDS before this wave reaches the barrier its readers wait behind
}

// Stream of the pipe form: chunks through the ring with sequence-numbered waits, plus a piece of the next half
// tile in each of the first steps; the first half of a tile also publishes the next tile's number.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int PREC, int KSN, typename Hook = NoHook>
struct PipeStream {
    StepStream& st;
    const HalfRequest& rq;
    const HalfTile<KSN>& htn;
    const StageArgs& a;
    uint32_t ring_lds;      // LDS address of ring slot 0 + this wave's quarter of a chunk
    const char* wbase;      // the stage's weight pack + this wave's quarter of a chunk
    int gs0;                // number of this half's first step within the tile
    int wave, lane;
    volatile int* mailbox;  // non-null (half 0): the next tile's number goes there before the last step's barrier
    int xcd;
    QueueState& qs;
    int snap_step;          // half 0: the step at which the queue's answer is looked at
    int step_no;
    Hook last_step_hook;    // runs in front of the barrier that ends the half's last step (final stage: the bilinear taps' image tile goes to LDS)
    __device__ __forceinline__ void begin_step() {
        step_request(st, gs0 + step_no, ring_lds, wbase, lane);
        if (mailbox && step_no == snap_step) queue_presolve(a, xcd, wave, lane, st, qs);
        ++step_no;
    }
    // gather instruction number g of the half tile being requested (g compile-time after unrolling)
    __device__ __forceinline__ void piece(int g) {
        if (g < HalfTile<KSN>::G::NG && g < rq.ng) {
            htn.stage_one(g, rq.origin, rq.dst);
            st.tile_seq = ++st.issued;
        }
    }
    __device__ __forceinline__ int slot() const { return st.slot; }
    template <int EXTRA> __device__ __forceinline__ void end_step(bool last) {
        if (last) last_step_hook();
        if (last && mailbox) queue_publish(a, xcd, wave, lane, st, qs, mailbox);
        step_advance<EXTRA>(st, last, PREC == 1 && gs0 + step_no <= kRingAhead);  // (step_no counts this step already: the tile's steps 0 .. kRingAhead - 1)
    }
};

template <int NSRC, int KS0, bool FINAL, bool IMG_U8, bool OUT_U8, int PREC, int FACTOR = 3>
__global__ __launch_bounds__(256, 2) void conv_stage_pipe_kernel(StageArgs a) {
    __builtin_amdgcn_s_setprio(3);
    constexpr int NTN = FINAL ? (FACTOR * FACTOR + 9) / 10 : 1;  // N-tiles of the node (expand at factor 4: 48 channels = 2)
    using H0 = HalfTile<KS0>;
    using H3 = HalfTile<3>;
    constexpr int HB = H0::BYTES;  // KS0 >= 3: the first source has the largest half tile
    constexpr int NH = 2 * NSRC;
    constexpr bool H16 = kH16<PREC, FINAL>;  // stages 1-3 of the split-half mode: 16x16x32 MFMAs, a source's halves share their odd tap's step
    constexpr int NSTEPS = H16 ? KS0 * KS0 + (NSRC - 1) * 9
                               : 2 * (H0::STEPS + (NSRC - 1) * H3::STEPS) * NTN;  // weight chunks per tile: one per (step, N-tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem + 2 * HB;
    volatile int* s_next = (volatile int*)(ring + kRingBytes);
    float* s_wlin = (float*)(ring + kRingBytes + 16);  // final stage: the 9 x 128 fixed weights of the bilinear taps per N-tile, loaded once
    float* s_lut = s_wlin + 9 * NTN * 128;             // final stage, u8 input: byte / 255 (img_to_data, a true division) for every byte value
    // final stage, one N-tile: the image tile of the bilinear taps has LDS of its own (round 6).  It used to go into the half-tile buffer the
    // last half had just left -- a barrier in front of the taps (the buffer's last readers) and one behind them (the next tile's second
    // half lands there).  With a region nothing else uses it is written in the tile's LAST step, in front of that step's own barrier, and
    // read behind it: no barrier of its own, and the next tile's write is a whole tile of barriers away from this tile's reads.
    constexpr bool kLinOwn = FINAL && NTN == 1;
    float* s_xown = s_lut + 256;
    const uint32_t lds0 = lds_addr(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int nbig = a.grid[0].ntiles, nsmall = a.grid[1].ntiles;
    const int xcd = blockIdx.x & 7;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring) + wave * 1024);
    const char* wbase = uniform_ptr((const char*)a.wpack + wave * 1024);
    float bias[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) bias[nt] = a.bias[nt * 32 + i];
    const float beta = FINAL ? 0.f : a.beta[i];
    // (kH16: lane l holds output channels 16 ch + 4 (l >> 4) + (0..3) of its pixels)
    f32x4 bias2[2], beta2[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        bias2[ch] = H16 ? *(const f32x4*)(a.bias + 16 * ch + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        beta2[ch] = H16 ? *(const f32x4*)(a.beta + 16 * ch + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (quad form of the exact mode's last stage: a lane holds every channel of its pixel, so it needs every channel's bias -- wave-uniform
    // values kept in vector registers for the whole launch; loaded per tile they came through the vector memory path, whose waits sat out
    // the next tile's DMAs)
    // (split-half mode's last stage, transposed accumulators: lane (i, h) holds the slots whose biases are a.bias[32 nt + 16 h + (0..15)], stage_epilogue_final_t)
    constexpr bool kFinalT = FINAL && PREC == 1;
    f32x4 fbias[kFinalT ? NTN : 1][4];
    if constexpr (kFinalT) {
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) fbias[nt][k] = *(const f32x4*)(a.bias + 32 * nt + 16 * (lane >> 5) + 4 * k);
    }
    constexpr bool kQuad = FINAL && PREC == 0;
    float qbias[kQuad ? 3 * FACTOR * FACTOR : 1];
    if constexpr (kQuad) {
#pragma unroll
        for (int ch = 0; ch < 3 * FACTOR * FACTOR; ++ch) {
            const int tr = ch / 3, tl = tr % 10;
            qbias[ch] = a.bias[(tr / 10) * 32 + 16 * (tl / 5) + 3 * (tl % 5) + ch % 3];
        }
    }
    H0 h0;
    H3 h3;
    h0.template init<PREC>(a.pitch, lane);
    if constexpr (NSRC >= 2) h3.template init<PREC>(a.pitch, lane);

    // combined tile id (queue_resolve) -> image, tile origin, tile class.  The 16 dwords of the tile class's TileGrid (its divisors)
    // are needed once per tile; held in SGPRs across the tile they are spilled to VGPR lanes (102 SGPRs are all there are) and come
    // back as ~60 v_readlane per tile -- vector-ALU instructions in the matrix stream.  They are therefore re-read from the kernel
    // argument segment with one scalar load (the pointer is laundered through an empty asm so that the compiler cannot merge the
    // load with an earlier one and keep the values live again).
    typedef const uint32_t __attribute__((address_space(4)))* KernWords;
    static_assert(sizeof(TileGrid) == 64 && offsetof(StageArgs, grid) % 4 == 0, "TileGrid is read as 16 dwords");
    auto coords = [&](int t, int& n, int& x0, int& y0, bool& small) {
        KernWords kw = (KernWords)__builtin_amdgcn_kernarg_segment_ptr();  // StageArgs is the kernel's only explicit argument: offset 0
        asm volatile("" : "+s"(kw));
        int tx, ty;
        small = t >= nbig;
        uint32_t words[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) words[k] = kw[offsetof(StageArgs, grid) / 4 + (small ? 16 : 0) + k];
        TileGrid g;
        __builtin_memcpy(&g, words, sizeof(g));
        tile_coords(g, a.tiles_x, small ? t - nbig : t, n, tx, ty);
        y0 = g.y0 + ty * (small ? 4 : 8);
        x0 = tx * kTW;
    };

    if constexpr (FINAL) {  // (read after the tile's steps, many barriers later)
        const float* wlin = a.wpack + (size_t)NSTEPS * kChunkFloats;
        // (split-half mode: the residual's integer weights as halves, 3 K-blocks of 1 KB per N-tile, and byte -> split(byte / 255 / (2 f)^2))
        for (int k = tid; k < (PREC == 1 ? 3 * NTN * 256 : 9 * NTN * 128); k += 256) s_wlin[k] = wlin[k];
        if constexpr (IMG_U8) s_lut[tid] = PREC == 1 ? __uint_as_float(lin_split_table_entry(tid, 4.0f * FACTOR * FACTOR)) : __fdiv_rn((float)tid, 255.0f);
    }
    const int first = queue_first(blockIdx.x, nbig, nsmall);
    if (first < 0) return;
    // a launch with a workgroup per tile (small images) has no queue to ask: its tile is the workgroup's only one
    const bool single = (int)gridDim.x >= nbig + nsmall;
    int n, x0, y0;
    bool small;
    coords(first, n, x0, y0, small);
    // first tile only: its first half and the first weight chunks are requested here; every later tile finds
    // them already on the way (requested by the last half / the last steps of the tile before)
    h0.template stage<PREC>(lds0, a.src[0], 0, a.img_stride, a.pitch, n, y0, x0, wave);
#pragma unroll
    for (int k = 0; k < kRingAhead; ++k) weight_chunk_async(ring + k * 4096, a.wpack + k * kChunkFloats, wave, lane);
    constexpr int NG0 = H0::G::NG;
    // sequence numbers so far: half tile 1..NG0, chunks 0..3 = NG0+1..NG0+4; q[] is shifted before use (step_request)
    StepStream st{0, NSTEPS, false, NG0 + kRingAhead, {0, NG0 + 2, NG0 + 3, NG0 + 4}, NG0};
    static_assert(kRingAhead == 4, "q[] initialiser");
    // half 0 and chunk 0 (split: and 1) are in; the later chunks may still be in flight
    wait_vm_barrier<true>(st.issued - (NG0 + 1 + (PREC == 1 ? 1 : 0)));

    int nn = 0, nx0 = 0, ny0 = 0;  // the tile after this one (known from half 1 on)
    bool nsmall_tile = false;
    uint32_t dom = 0;  // split-half mode: the largest hi half this thread has stored (domain_track)

    // One tile: T tile rows per wave (2: an 8-row tile, 1: a 4-row tile in the first rows of the same buffers).
    auto tile_body = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr int TH = 4 * T;
        f32x16 acc[NTN * T], accx[PREC == 1 ? NTN * T : 1];
#pragma unroll
        for (int m = 0; m < NTN * T; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][r] = (PREC == 1 && !FINAL) ? bias[0] : 0.f;  // (split-half producers: the bias rides in the accumulator, see split_value)
                if constexpr (PREC == 1) accx[m][r] = 0.f;
            }
        f32x4 qm[H16 ? T : 1][2][2], qx[H16 ? T : 1][2][2];  // kH16: the same tile as 16x16 accumulators
        if constexpr (H16) {
#pragma unroll
            for (int m = 0; m < T; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) { qm[m][k >> 1][k & 1] = bias2[k & 1]; qx[m][k >> 1][k & 1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // the exact mode's last stage, 8-row tiles: 4x4x1 MFMAs, the lane's own pixel in 4-slot groups (half_steps_f32 QUAD)
        constexpr bool QUAD = FINAL && PREC == 0 && T == 2;
        QuadAcc qa[QUAD ? NTN : 1];
        if constexpr (QUAD) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int g = 0; g < 8; ++g) qa[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // (local to the tile ON PURPOSE: the registers the asynchronous atomic / load return into must not be live across the
        // tile loop's merge of the two tile bodies -- the compiler then copies them right after the asm statement, i.e. before
        // the data has arrived; it cannot know these asm outputs land later)
        QueueState qs{0u, 0u, 0, 0, -1};
        if (!single) queue_pull_async(a, xcd, wave, lane, st, qs);  // the answer is looked at towards the end of half 0
        st.have_next = false;  // not known yet: nothing of the next tile is requested before half 1
        LinPrefetch<IMG_U8, TH> linpx;
        auto do_half = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int src = j >> 1;
            constexpr int KSJ = src == 0 ? KS0 : 3;                               // this half's kernel size
            constexpr int KSN = (j + 1 < NH) ? (((j + 1) >> 1) == 0 ? KS0 : 3) : KS0;  // the half requested meanwhile
            const uint32_t other = lds0 + ((j + 1) & 1) * HB;
            const char* hb = smem + (j & 1) * HB;
            if constexpr (j == 1) {  // the mailbox was published before the barrier that ended half 0
                const int next = single ? -1 : __builtin_amdgcn_readfirstlane(*s_next);
                st.have_next = next >= 0;
                if (st.have_next) coords(next, nn, nx0, ny0, nsmall_tile);
            }
            // half j+1 of this tile, or the next tile's half 0, goes into the buffer half j-1 has just left; a 4-row tile
            // needs only the first NG_SMALL gather groups of it
            HalfRequest rq;
            if constexpr (j + 1 < NH)
                rq = HalfRequest{T == 2 ? HalfTile<KSN>::G::NG : HalfTile<KSN>::NG_SMALL,
                                 HalfTile<KSN>::template origin_of<PREC>(a.src[(j + 1) >> 1], (j + 1) & 1, a.img_stride, a.pitch, n, y0, x0, wave),
                                 HalfTile<KSN>::plane_of(other, wave)};
            else
                rq = HalfRequest{!st.have_next ? 0 : nsmall_tile ? HalfTile<KSN>::NG_SMALL : HalfTile<KSN>::G::NG,
                                 HalfTile<KSN>::template origin_of<PREC>(a.src[0], 0, a.img_stride, a.pitch, nn, ny0, nx0, wave),
                                 HalfTile<KSN>::plane_of(other, wave)};
            // (kLinOwn: requested and written one half EARLIER than they are read, so that the tile's last half carries only the next
            // tile's first requests -- profiles/r6_tile_timeline.txt: the last half was the longest of the six)
            if constexpr (FINAL && j == (kLinOwn ? NH - 2 : NH - 1)) linpx.issue(a, n, y0, x0, tid, st);
            const HalfTile<KSN>* htn;
            if constexpr (KSN == KS0) htn = (const HalfTile<KSN>*)&h0; else htn = (const HalfTile<KSN>*)&h3;
            using GJ = TileGeom<8, KSJ>;
            // steps of this half / of the halves before it (kH16: a source's first half is its tap pairs, the second one step more)
            constexpr int PAIRS_J = (KSJ * KSJ - 1) / 2;
            constexpr int STEPS_J = H16 ? PAIRS_J + (j & 1) : HalfTile<KSJ>::STEPS * NTN;
            constexpr int GS0 = H16 ? (src == 0 ? 0 : KS0 * KS0 + (src - 1) * 9) + (j & 1) * PAIRS_J
                                    : (j == 0 ? 0 : j == 1 ? H0::STEPS : 2 * H0::STEPS + (j - 2) * H3::STEPS) * NTN;
            constexpr int LIN_LO_J = LinPrefetch<IMG_U8, TH>::NPIX * 8;  // split-half mode: bytes from the hi halves of the image tile to its lo halves
            auto lin_store = [&]() {  // (kLinOwn: the pixels requested at this half's start are long in; see s_xown)
                if constexpr (kLinOwn && j == NH - 2) {
                    if constexpr (PREC == 1) linpx.store_split((char*)s_xown, LIN_LO_J, (const uint32_t*)s_lut, 4.0f * FACTOR * FACTOR, tid, st);
                    else linpx.store(s_xown, s_lut, tid, st);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the writes are in LDS before this wave reaches the step's barrier
                }
            };
            PipeStream<PREC, KSN, decltype(lin_store)> sm{st, rq, *htn, a, ring_lds, wbase, GS0, wave, lane, (j == 0 && !single) ? s_next : nullptr, xcd, qs, STEPS_J > 3 ? STEPS_J - 3 : 0, 0, lin_store};
            if constexpr (QUAD) half_steps_f32<GJ::TWH, GJ::PLANE, KSJ, T, NTN, true, FACTOR>(qa, hb, ring, sm, wave, lane);
            else if constexpr (PREC == 0) half_steps_f32<GJ::TWH, GJ::PLANE, KSJ, T, NTN>(acc, hb, ring, sm, wave, lane);
            else if constexpr (H16) half_steps_h16<GJ::TWH, GJ::PLANE, 2, KSJ, T, (j & 1) != 0, -HB>(qm, qx, hb, ring, sm, wave, lane);
            else half_steps_h<GJ::TWH, GJ::PLANE, 2, KSJ, T, NTN>(acc, accx, hb, ring, sm, wave, lane);
        };
        do_half(std::integral_constant<int, 0>{});
        do_half(std::integral_constant<int, 1>{});
        if constexpr (NH > 2) { do_half(std::integral_constant<int, 2>{}); do_half(std::integral_constant<int, 3>{}); }
        if constexpr (NH > 4) { do_half(std::integral_constant<int, 4>{}); do_half(std::integral_constant<int, 5>{}); }
        if constexpr (FINAL) {
            constexpr int LIN_LO = LinPrefetch<IMG_U8, TH>::NPIX * 8;  // split-half mode: bytes from the hi halves of the tile to its lo halves
            float* s_x = kLinOwn ? s_xown : (float*)(smem + HB);
            if constexpr (!kLinOwn) {
                // two N-tiles (factor 4): no LDS to spare -- the image tile goes into the buffer the last half has just left (buffer 1)
                if constexpr (PREC == 1) linpx.store_split((char*)s_x, LIN_LO, (const uint32_t*)s_lut, 4.0f * FACTOR * FACTOR, tid, st);
                else linpx.store(s_x, s_lut, tid, st);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (PREC == 1) lin_mfma_h<TH, T, NTN>(acc, accx, (const char*)s_x, LIN_LO, (const f16x8*)s_wlin, wave, lane);
            else if constexpr (QUAD) lin_mfma_quad<NTN, FACTOR>(qa, s_x, s_wlin, wave, lane);
            else lin_mfma<TH, T, NTN>(acc, s_x, s_wlin, wave, lane);
            if constexpr (!kLinOwn) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // everybody is done with buffer 1 before the next tile's second half lands there
                asm volatile("" ::: "memory");
            }
        }
        // The epilogue's stores are memory operations of this wave like its DMAs, returning in issue order with them: counted (as far as they
        // are certain), the next tile's first waits -- for chunks requested before them -- let them stay in flight (step_advance).
        int stores;
        if constexpr (H16) stores = stage_epilogue_h16<T>(a, qm, qx, beta2, n, x0, y0, wave, lane, dom);
        else if constexpr (QUAD) stores = stage_epilogue_quad<NTN, OUT_U8, FACTOR>(a, qa, qbias, n, x0, y0, wave, lane);
        else if constexpr (kFinalT) stores = stage_epilogue_final_t<T, NTN, OUT_U8, FACTOR>(a, acc, accx, fbias, n, x0, y0, wave, lane);
        else {
            // (stages 1-3: the buffer the tile's LAST half has just left -- every wave is past that half's last barrier, the next tile's second
            // half is requested into it by the steps to come -- lends each wave its own plane as the epilogue's scratch)
            float* scratch = FINAL ? nullptr : (float*)(smem + ((NH - 1) & 1) * HB + wave * H0::G::PLANE);
            stores = stage_epilogue<TH, T, NTN, FINAL, OUT_U8, FACTOR>(a, acc, bias, beta, n, x0, y0, wave, lane, scratch);
        }
        // (Split-half mode only: its steps are shorter than a write acknowledgement takes, so the uncounted stores stalled each tile's first
        // barriers -- 0.8 % of a frame.  An exact-mode step is ten times longer and never saw it: measured, no change, its code is left as it was.)
        if constexpr (PREC == 1) st.issued += stores;
    };

    // The 48 expand channels of factor 4 are two N-tiles: in the split-half mode an 8-row tile body would hold 128 accumulator registers
    // beside 96 of operands and spill (256 VGPRs + 124 B of scratch in round 4).  That one instantiation therefore has the 4-row body
    // only; sr_api.cpp plans its launches with 4-row tiles (StackJob::prepare).
    constexpr bool kBigTiles = !(FINAL && FACTOR == 4 && PREC == 1);
    int budget = nbig + nsmall;  // no workgroup can be handed more tiles than the launch has: a bound on the loop, whatever happens
    while (true) {
        if constexpr (kBigTiles) {
            if (small) tile_body(std::integral_constant<int, 1>{});
            else tile_body(std::integral_constant<int, 2>{});
        } else {
            if (!small) __builtin_trap();  // (never planned, see above)
            tile_body(std::integral_constant<int, 1>{});
        }
        if (!st.have_next) break;
        if (--budget <= 0) {
            // Unreachable while the queue is consistent.  A successor was announced, so its first half tile and weight chunks are on
            // their way into LDS and a tile has been taken from the queue: leaving quietly would drop that tile.  Let the DMAs land, then
            // fail the launch loudly (the host sees a HIP error on this stream) rather than return an image with a hole in it.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_trap();
        }
        n = nn; x0 = nx0; y0 = ny0; small = nsmall_tile;
    }
    if constexpr (PREC == 1 && !FINAL) domain_report(dom, a.domain);
    // (nothing is in flight towards LDS here: the last tile requested no successor; s_endpgm waits for the stores)
}

// (the two parameter-free graphs of the reference, bilinear_net and downsample_net, live in sr_aux.hip)

// ---------------------------------------------------------------------------
// zero border of the feature maps for a new geometry: one workgroup per buffer row (and per map)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clear_borders_kernel(ClearArgs a) {
    float4* m = reinterpret_cast<float4*>(a.map[blockIdx.y]);
    const long rows_per = a.img_stride / a.pitch;
    const long row = blockIdx.x;
    const long base = row * a.pitch;            // first pixel of the row
    long end = base + a.pitch;
    if (end > a.total_px) end = a.total_px;     // the slack after the last row is a partial row
    const long img = row / rows_per, local = row - img * rows_per;
    const bool interior = img < a.n && local >= kFeatPad && local < kFeatPad + a.H;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!interior) {
        for (long q = base * 8 + threadIdx.x; q < end * 8; q += 256) m[q] = z;   // 8 float4 per 32-channel pixel
        return;
    }
    if (a.planar) {  // row-planar map: the row holds eight runs of `pitch` 16-byte groups, each with its own left and right border
        const int edge = kFeatPad + (a.pitch - kFeatPad - a.W);  // border cells per run
        for (int k = threadIdx.x; k < 8 * edge; k += 256) {
            const int c = k / edge, e = k - c * edge;
            m[base * 8 + (long)c * a.pitch + (e < kFeatPad ? e : a.W + e)] = z;
        }
        return;
    }
    for (long q = base * 8 + threadIdx.x; q < (base + kFeatPad) * 8; q += 256) m[q] = z;
    for (long q = (base + kFeatPad + a.W) * 8 + threadIdx.x; q < end * 8; q += 256) m[q] = z;
}

hipError_t sr_launch_clear_borders(const ClearArgs& a, hipStream_t s) {
    const long rows = (a.total_px + a.pitch - 1) / a.pitch;
    hipLaunchKernelGGL(clear_borders_kernel, dim3((unsigned)rows, 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host-side launchers (called from sr_api.cpp through sr_kernels.h)
// ---------------------------------------------------------------------------
template <int TH, int KS0>
static constexpr size_t stage_lds_bytes() {
    return 8 * (size_t)TileGeom<TH, KS0>::PLANE + kRingBytes;
}

template <int TH>
static hipError_t launch_conv0_t(const Conv0Args& a, int nblk, bool img_u8, hipStream_t s) {
    if (img_u8)
        hipLaunchKernelGGL((conv0_kernel<TH, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((conv0_kernel<TH, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

template <int TH>
static hipError_t launch_conv0_split_t(const Conv0Args& a, int nblk, bool img_u8, hipStream_t s) {
    if (img_u8)
        hipLaunchKernelGGL((conv0_split_kernel<TH, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((conv0_split_kernel<TH, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t sr_launch_conv0(const Conv0Args& a, int th, int prec, int nblk, bool img_u8, hipStream_t s) {
    if (prec == 0) return th == 8 ? launch_conv0_t<8>(a, nblk, img_u8, s) : launch_conv0_t<4>(a, nblk, img_u8, s);
    return th == 8 ? launch_conv0_split_t<8>(a, nblk, img_u8, s) : launch_conv0_split_t<4>(a, nblk, img_u8, s);
}

// The > 64 KB dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is a property of a (kernel, device)
// pair: every stage kernel instantiation has the same C++ type, and a process may drive several GPUs from several
// host threads (sr_upscale_*_multi, sr_upscale_sharded_*_all), so the "already configured" set is keyed on both
// and guarded by a mutex.
template <typename K>
static hipError_t launch_with_lds(K kern, const StageArgs& a, int nblk, size_t lds, hipStream_t s) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> configured;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_pair((const void*)kern, dev);
        if (!configured.count(key)) {
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured.insert(key);
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}

template <int TH, int PREC>
static hipError_t launch_stage_t(int stage, int factor, const StageArgs& a, int nblk, bool img_u8, bool out_u8,
                                 hipStream_t s) {
    switch (stage) {
        case 1: return launch_with_lds(conv_stage_kernel<TH, 1, 5, false, false, false, PREC>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 2: return launch_with_lds(conv_stage_kernel<TH, 2, 5, false, false, false, PREC>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 3: return launch_with_lds(conv_stage_kernel<TH, 3, 5, false, false, false, PREC>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 4:
#define SR_FINAL(F)                                                                                                          \
            if (img_u8 && out_u8) return launch_with_lds(conv_stage_kernel<TH, 3, 3, true, true, true, PREC, F>, a, nblk, stage_lds_bytes<TH, 3>(), s); \
            if (!img_u8 && !out_u8) return launch_with_lds(conv_stage_kernel<TH, 3, 3, true, false, false, PREC, F>, a, nblk, stage_lds_bytes<TH, 3>(), s); \
            return hipErrorInvalidValue;
            if (factor == 3) { SR_FINAL(3) }
            if (factor == 2) { SR_FINAL(2) }
            if (factor == 4) { SR_FINAL(4) }
#undef SR_FINAL
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// Pipe form (both tile classes of the launch): grid = co-resident workgroups, LDS = two half tiles + ring + mailbox.
template <int PREC>
static hipError_t launch_stage_pipe_t(int stage, int factor, const StageArgs& a, int grid, bool img_u8, bool out_u8, hipStream_t s) {
    constexpr size_t lds5 = 2 * (size_t)HalfTile<5>::BYTES + kRingBytes + 16;
    // + bilinear weights per N-tile, byte / 255 table, and (one N-tile: factor 2, 3) the bilinear taps' own image tile of (8 + 2) x (32 + 2) pixels
    constexpr size_t lds3_1 = 2 * (size_t)HalfTile<3>::BYTES + kRingBytes + 16 + 9 * 128 * sizeof(float) + 256 * sizeof(float) + 10 * 34 * 16;
    constexpr size_t lds3_2 = 2 * (size_t)HalfTile<3>::BYTES + kRingBytes + 16 + 9 * 2 * 128 * sizeof(float) + 256 * sizeof(float);
    static_assert(lds3_1 <= 80 * 1024 && lds3_2 <= 80 * 1024, "two workgroups per CU");
    const size_t lds3 = factor == 4 ? lds3_2 : lds3_1;
    switch (stage) {
        case 1: return launch_with_lds(conv_stage_pipe_kernel<1, 5, false, false, false, PREC>, a, grid, lds5, s);
        case 2: return launch_with_lds(conv_stage_pipe_kernel<2, 5, false, false, false, PREC>, a, grid, lds5, s);
        case 3: return launch_with_lds(conv_stage_pipe_kernel<3, 5, false, false, false, PREC>, a, grid, lds5, s);
        case 4:
#define SR_FINAL(F)                                                                                                          \
            if (img_u8 && out_u8) return launch_with_lds(conv_stage_pipe_kernel<3, 3, true, true, true, PREC, F>, a, grid, lds3, s); \
            if (!img_u8 && !out_u8) return launch_with_lds(conv_stage_pipe_kernel<3, 3, true, false, false, PREC, F>, a, grid, lds3, s); \
            return hipErrorInvalidValue;
            if (factor == 3) { SR_FINAL(3) }
            if (factor == 2) { SR_FINAL(2) }
            if (factor == 4) { SR_FINAL(4) }
#undef SR_FINAL
            return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
hipError_t sr_launch_stage_pipe(int stage, int factor, const StageArgs& a, int prec, int grid, bool img_u8, bool out_u8, hipStream_t s) {
    return prec == 0 ? launch_stage_pipe_t<0>(stage, factor, a, grid, img_u8, out_u8, s)
                     : launch_stage_pipe_t<1>(stage, factor, a, grid, img_u8, out_u8, s);
}

hipError_t sr_launch_stage(int stage, int factor, const StageArgs& a, int th, int prec, int nblk, bool img_u8,
                           bool out_u8, hipStream_t s) {
    if (prec == 0)
        return th == 8 ? launch_stage_t<8, 0>(stage, factor, a, nblk, img_u8, out_u8, s)
                       : launch_stage_t<4, 0>(stage, factor, a, nblk, img_u8, out_u8, s);
    return th == 8 ? launch_stage_t<8, 1>(stage, factor, a, nblk, img_u8, out_u8, s)
                   : launch_stage_t<4, 1>(stage, factor, a, nblk, img_u8, out_u8, s);
}
