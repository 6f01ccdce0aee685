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
//    K-concatenation; accumulators never leave registers);
//  * v_mfma_f32_32x32x2_f32 -- exact f32 (bitwise an fmaf chain), so parity with
//    the f32 CPU path holds to rounding-order noise (~1e-6), far inside 1e-4;
//  * a workgroup (4 waves) owns a TH x 32 pixel tile; the source tile + halo is
//    staged in LDS in a channel-group-planar layout [cin/4][pixel][4 f32] so
//    that every A-operand fetch is one conflict-free ds_read_b128 at a
//    compile-time offset from a per-lane base (one lane = one pixel, 16 lanes of
//    a service group = 16 consecutive 16-B slots);
//  * weights are pre-packed on the host into 4 KB per-tap chunks in the exact
//    order the B-operand ds_read_b128 wants ([cin/4][cout][4]) and streamed
//    through a 2-slot LDS ring, one chunk per tap, one barrier per tap;
//  * LDS per workgroup <= 64 KB -> 2 workgroups per CU: one stages its tile
//    while the other keeps the matrix pipe busy;
//  * bias + BeLU (or bias + bilinear residual + depth-to-space [+ u8 RGBA
//    quantisation]) are fused into the epilogue: no elementwise kernel exists.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sr_kernels.h"

// The reference CPU path (rustc/LLVM) never fuses a*b+c; neither may we, or the two
// instantiations of one kernel can differ in the last bit (packed-math vs fma selection).
#pragma clang fp contract(off)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 256;
constexpr int kTW = 32;            // tile width = one MFMA M-tile of pixels
constexpr int kChunkFloats = 1024; // one tap: 32 cin x 32 cout

__device__ __forceinline__ float belu(float v, float beta) {
    // alumina BeLU (network.rs:35,54-56): beta*x + sqrt(x*x+1) - 1, kept in the
    // same naive form as the reference; no contraction.
    return __fadd_rn(__fadd_rn(__fmul_rn(beta, v), __fsqrt_rn(__fadd_rn(__fmul_rn(v, v), 1.0f))), -1.0f);
}

// XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each
// XCD one contiguous run of tiles so neighbouring tiles (which share halo rows
// and columns) hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float load_img(const void* img, int img_ch, bool u8, size_t px, int c) {
    // img_to_data (reference main.rs:170): u8 / 255 (true division), alpha dropped
    if (u8) return __fdiv_rn((float)((const uint8_t*)img)[px * img_ch + c], 255.0f);
    return ((const float*)img)[px * 3 + c];
}

}  // namespace

// ---------------------------------------------------------------------------
// Stage 0: conv0 5x5 3->32 + bias + BeLU.  K = 25 taps x 4 (3 ch + zero pad),
// two MFMAs per tap; x tile (with halo) and all of conv0's weights sit in LDS.
// ---------------------------------------------------------------------------
template <int TH, bool IMG_U8>
__global__ __launch_bounds__(kThreads, 2) void conv0_kernel(Conv0Args a) {
    constexpr int T = TH / 4;
    constexpr int TWH = kTW + 4, THH = TH + 4, NPIX = THH * TWH;
    __shared__ __attribute__((aligned(16))) float s_x[NPIX * 4];
    __shared__ __attribute__((aligned(16))) float s_w[25 * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / tiles_per_img, t = bid - n * tiles_per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int x0 = tx * kTW, y0 = a.y_begin + ty * TH;
    const size_t img_px0 = (size_t)n * a.H * a.W;

    for (int k = tid; k < 25 * 128; k += kThreads) s_w[k] = a.wpack[k];
    for (int p = tid; p < NPIX; p += kThreads) {
        const int py = p / TWH, px = p - py * TWH;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const size_t gp = img_px0 + (size_t)gy * a.W + gx;
            v.x = load_img(a.img, a.img_ch, IMG_U8, gp, 0);
            v.y = load_img(a.img, a.img_ch, IMG_U8, gp, 1);
            v.z = load_img(a.img, a.img_ch, IMG_U8, gp, 2);
        }
        *(f32x4*)&s_x[p * 4] = v;
    }
    __syncthreads();

    f32x16 acc[T];
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const float* xa = s_x + ((wave * T) * TWH + i) * 4 + h * 2;
    const float* wb = s_w + (h * 32 + i) * 2;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const f32x2 b = *(const f32x2*)(wb + (ky * 5 + kx) * 128);
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const f32x2 av = *(const f32x2*)(xa + ((m + ky) * TWH + kx) * 4);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[m], 0, 0, 0);
            }
        }
    }

    const float bias = a.bias[i], beta = a.beta[i];
#pragma unroll
    for (int m = 0; m < T; ++m) {
        const int y = y0 + wave * T + m;
        if (y >= a.y_end) continue;
        float* drow = a.dst + (((size_t)n * a.H + y) * a.W) * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (x < a.W) drow[(size_t)x * 32] = belu(__fadd_rn(acc[m][r], bias), beta);
        }
    }
}

// ---------------------------------------------------------------------------
// Stages 1-4: sum of up to three 32-channel convolutions (first KS0 x KS0,
// the others 3x3) + bias, then BeLU -> NHWC feature map, or (FINAL) bilinear
// residual + depth-to-space -> output image.
// ---------------------------------------------------------------------------
template <int TH, int KS>
struct TileGeom {
    static constexpr int R = KS / 2;
    static constexpr int TWH = kTW + 2 * R;
    static constexpr int THH = TH + 2 * R;
    static constexpr int NPIX = THH * TWH;
    static constexpr int PLANE = (NPIX | 1) * 16;  // bytes; odd pixel count -> conflict-free staging writes
};

template <int TH, int KS>
__device__ __forceinline__ void stage_tile(char* tile, const float* __restrict__ src, int n, int H,
                                           int W, int y0, int x0, int tid) {
    using G = TileGeom<TH, KS>;
    constexpr int ITEMS = G::NPIX * 8;  // 16-byte items: (pixel, cin/4)
    constexpr int ROUNDS = (ITEMS + kThreads - 1) / kThreads;
    f32x4 v[ROUNDS];
    // branch-free: out-of-image pixels load from a clamped (valid) address and are
    // zeroed by a select -- the reference's zero padding (Padding::Same, network.rs:33).
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int item = min(tid + k * kThreads, ITEMS - 1);
        const int p = item >> 3, c = item & 7;
        const int py = p / G::TWH, px = p - py * G::TWH;
        const int gy = y0 - G::R + py, gx = x0 - G::R + px;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        const f32x4 ld = *(const f32x4*)(src + (((size_t)n * H + cy) * W + cx) * 32 + c * 4);
        const bool inb = (gy == cy) && (gx == cx);
        v[k] = inb ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int item = tid + k * kThreads;
        const int p = item >> 3, c = item & 7;
        if (item < ITEMS) *(f32x4*)(tile + c * G::PLANE + p * 16) = v[k];
    }
}

// Asynchronous 4 KB weight-chunk copy global -> LDS ring slot (LDS-DMA, no VGPR
// round trip): each wave moves 1 KB, lane l lands at base + 16*l, which is
// exactly the packed chunk order.  Completion is covered by the vmcnt(0) the
// compiler places in front of the next __syncthreads().
__device__ __forceinline__ void weight_chunk_async(char* ring_slot, const float* __restrict__ chunk,
                                                   int wave, int lane) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(chunk + (wave * 64 + lane) * 4),
        (__attribute__((address_space(3))) void*)(ring_slot + wave * 1024), 16, 0, 0);
}

template <int TH, int KS, int T>
__device__ __forceinline__ void conv_taps(f32x16 (&acc)[T], const char* tile, char* ring,
                                          const float* __restrict__ wpack, int& gtap, int ntaps_total,
                                          int wave, int lane) {
    using G = TileGeom<TH, KS>;
    const int i = lane & 31, h = lane >> 5;
    const char* abase = tile + h * G::PLANE + ((wave * T) * G::TWH + i) * 16;
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            // next tap's 4 KB weight chunk -> the ring slot nobody reads during this tap
            if (gtap + 1 < ntaps_total)
                weight_chunk_async(ring + ((gtap + 1) & 1) * 4096, wpack + (size_t)(gtap + 1) * kChunkFloats, wave, lane);
            const char* wb = ring + (gtap & 1) * 4096 + (h * 32 + i) * 16;
            const char* ab = abase + (ky * G::TWH + kx) * 16;
            // operands of channel-group rr+1 are fetched while group rr's MFMAs run
            f32x4 b[2], av[2][T];
            b[0] = *(const f32x4*)(wb);
#pragma unroll
            for (int m = 0; m < T; ++m) av[0][m] = *(const f32x4*)(ab + m * G::TWH * 16);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int cur = rr & 1, nxt = cur ^ 1;
                if (rr < 3) {
                    b[nxt] = *(const f32x4*)(wb + (rr + 1) * 1024);
#pragma unroll
                    for (int m = 0; m < T; ++m)
                        av[nxt][m] = *(const f32x4*)(ab + (rr + 1) * 2 * G::PLANE + m * G::TWH * 16);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int m = 0; m < T; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][m][q], b[cur][q], acc[m], 0, 0, 0);
                // pin the interleave: one MFMA, then the next group's LDS reads (their
                // latency hides under the remaining 4T-1 MFMAs of 64 cycles each)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (rr < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1 + T, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * T - 1, 0);
            }
            ++gtap;
            __syncthreads();
        }
    }
}

template <int TH, int NSRC, int KS0, bool FINAL, bool IMG_U8, bool OUT_U8>
__global__ __launch_bounds__(kThreads, 2) void conv_stage_kernel(StageArgs a) {
    constexpr int T = TH / 4;
    using G0 = TileGeom<TH, KS0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;
    char* ring = smem + 8 * G0::PLANE;  // KS0 >= 3: the first source has the largest tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / tiles_per_img, t = bid - n * tiles_per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int x0 = tx * kTW, y0 = a.y_begin + ty * TH;
    constexpr int NTAPS = KS0 * KS0 + (NSRC - 1) * 9;

    f32x16 acc[T];
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // weight chunk 0 -> ring slot 0
    weight_chunk_async(ring, a.wpack, wave, lane);
    int gtap = 0;

    stage_tile<TH, KS0>(tile, a.src[0], n, a.H, a.W, y0, x0, tid);
    __syncthreads();
    conv_taps<TH, KS0, T>(acc, tile, ring, a.wpack, gtap, NTAPS, wave, lane);
    if constexpr (NSRC >= 2) {
        stage_tile<TH, 3>(tile, a.src[1], n, a.H, a.W, y0, x0, tid);
        __syncthreads();
        conv_taps<TH, 3, T>(acc, tile, ring, a.wpack, gtap, NTAPS, wave, lane);
    }
    if constexpr (NSRC >= 3) {
        stage_tile<TH, 3>(tile, a.src[2], n, a.H, a.W, y0, x0, tid);
        __syncthreads();
        conv_taps<TH, 3, T>(acc, tile, ring, a.wpack, gtap, NTAPS, wave, lane);
    }

    const float bias = a.bias[i];
    if constexpr (!FINAL) {
        const float beta = a.beta[i];
#pragma unroll
        for (int m = 0; m < T; ++m) {
            const int y = y0 + wave * T + m;
            if (y >= a.y_end) continue;
            float* drow = a.dst + (((size_t)n * a.H + y) * a.W) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (x < a.W) drow[(size_t)x * 32] = belu(__fadd_rn(acc[m][r], bias), beta);
            }
        }
    } else {
        // lane i < 27 owns expand channel i = (dy*3+dx)*3+c (network.rs:37,39).
        // LinearInterp x3 (network.rs:27): half-pixel centres, edge clamped;
        // phase 0: 1/3 in[i-1] + 2/3 in[i]; phase 1: in[i]; phase 2: 2/3 in[i] + 1/3 in[i+1].
        const int ch = i < 27 ? i : 26;
        const int dy = ch / 9, dx = (ch - dy * 9) / 3, c = ch - dy * 9 - dx * 3;
        const float tyw = dy == 0 ? (2.0f / 3.0f) : (dy == 1 ? 0.0f : (1.0f / 3.0f));
        const float txw = dx == 0 ? (2.0f / 3.0f) : (dx == 1 ? 0.0f : (1.0f / 3.0f));
        const int oy_d = dy == 0 ? -1 : 0, ox_d = dx == 0 ? -1 : 0;
        const size_t img_px0 = (size_t)n * a.H * a.W;
        const int OW = a.W * 3;
        const int h_band = a.y_end - a.y_begin;
#pragma unroll
        for (int m = 0; m < T; ++m) {
            const int y = y0 + wave * T + m;
            if (y >= a.y_end) continue;
            const int ya = min(max(y + oy_d, 0), a.H - 1), yb = min(max(y + oy_d + 1, 0), a.H - 1);
            const size_t orow = ((size_t)n * h_band * 3 + (size_t)(y - a.y_begin) * 3 + dy) * OW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int xc = min(x, a.W - 1);
                const int xa = min(max(xc + ox_d, 0), a.W - 1), xb = min(max(xc + ox_d + 1, 0), a.W - 1);
                const float v00 = load_img(a.img, a.img_ch, IMG_U8, img_px0 + (size_t)ya * a.W + xa, c);
                const float v01 = load_img(a.img, a.img_ch, IMG_U8, img_px0 + (size_t)ya * a.W + xb, c);
                const float v10 = load_img(a.img, a.img_ch, IMG_U8, img_px0 + (size_t)yb * a.W + xa, c);
                const float v11 = load_img(a.img, a.img_ch, IMG_U8, img_px0 + (size_t)yb * a.W + xb, c);
                const float ra = __fadd_rn(__fmul_rn(1.0f - txw, v00), __fmul_rn(txw, v01));
                const float rb = __fadd_rn(__fmul_rn(1.0f - txw, v10), __fmul_rn(txw, v11));
                const float lin = __fadd_rn(__fmul_rn(1.0f - tyw, ra), __fmul_rn(tyw, rb));
                const float v = __fadd_rn(lin, __fadd_rn(acc[m][r], bias));
                const size_t opx = orow + (size_t)x * 3 + dx;
                if constexpr (!OUT_U8) {
                    if (i < 27 && x < a.W) ((float*)a.out)[opx * 3 + c] = v;
                } else {
                    // data_to_img (main.rs:175): clamp(floor(255 v + 0.5), 0, 255), alpha 255
                    float q = floorf(__fadd_rn(__fmul_rn(255.0f, v), 0.5f));
                    q = fminf(fmaxf(q, 0.0f), 255.0f);
                    const uint32_t qi = (uint32_t)q;
                    const uint32_t g = __shfl_down(qi, 1), b = __shfl_down(qi, 2);
                    if (i < 27 && c == 0 && x < a.W)
                        ((uint32_t*)a.out)[opx] = qi | (g << 8) | (b << 16) | 0xff000000u;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launchers (called from sr_api.cpp through sr_kernels.h)
// ---------------------------------------------------------------------------
template <int TH, int KS0>
static constexpr size_t stage_lds_bytes() {
    return 8 * (size_t)TileGeom<TH, KS0>::PLANE + 2 * 4096;
}

template <int TH>
static hipError_t launch_conv0_t(const Conv0Args& a, int nblk, bool img_u8, hipStream_t s) {
    if (img_u8)
        hipLaunchKernelGGL((conv0_kernel<TH, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((conv0_kernel<TH, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t sr_launch_conv0(const Conv0Args& a, int th, int nblk, bool img_u8, hipStream_t s) {
    return th == 8 ? launch_conv0_t<8>(a, nblk, img_u8, s) : launch_conv0_t<4>(a, nblk, img_u8, s);
}

template <typename K>
static hipError_t launch_with_lds(K kern, const StageArgs& a, int nblk, size_t lds, hipStream_t s) {
    static bool configured = false;  // one instance per kernel template instantiation
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}

template <int TH>
static hipError_t launch_stage_t(int stage, const StageArgs& a, int nblk, bool img_u8, bool out_u8,
                                 hipStream_t s) {
    switch (stage) {
        case 1: return launch_with_lds(conv_stage_kernel<TH, 1, 5, false, false, false>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 2: return launch_with_lds(conv_stage_kernel<TH, 2, 5, false, false, false>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 3: return launch_with_lds(conv_stage_kernel<TH, 3, 5, false, false, false>, a, nblk, stage_lds_bytes<TH, 5>(), s);
        case 4:
            if (img_u8 && out_u8) return launch_with_lds(conv_stage_kernel<TH, 3, 3, true, true, true>, a, nblk, stage_lds_bytes<TH, 3>(), s);
            if (!img_u8 && !out_u8) return launch_with_lds(conv_stage_kernel<TH, 3, 3, true, false, false>, a, nblk, stage_lds_bytes<TH, 3>(), s);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

hipError_t sr_launch_stage(int stage, const StageArgs& a, int th, int nblk, bool img_u8, bool out_u8,
                           hipStream_t s) {
    return th == 8 ? launch_stage_t<8>(stage, a, nblk, img_u8, out_u8, s)
                   : launch_stage_t<4>(stage, a, nblk, img_u8, out_u8, s);
}
