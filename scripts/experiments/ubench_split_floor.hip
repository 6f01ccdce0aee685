// Structural floor of the split-half (f16 x 3) stage kernels on MI355X -- round 5.
//
// Question: would a stage kernel with more waves per SIMD, or with its LDS-DMA issue moved to a
// producer wave, lift the split-half mode from mfma_util 0.66 to >= 0.75 and stage 3 from 0.464 to <= 0.40 ms?  Before building
// that kernel family this program measures, with the SAME step shape (2 taps x 16 channels: 6 T v_mfma_f32_32x32x16_f16 per wave,
// their operand ds_read_b128s, one barrier, a 4 KB weight chunk by LDS-DMA, a tile gather in pieces, a BeLU + split epilogue per
// 50 steps):
//   part 1  the matrix pipe alone: TFLOP/s, the shader clock the part settles at and the board power, for constant / random
//           operands, a duty-cycle sweep (s_sleep between steps: what do idle matrix cycles buy back?), operand orders,
//           bf16 and the 16x16x32 shape;
//   part 2  the step loop under every structure in question: 4-wave workgroups x 2 per CU with two tile rows per wave (what ships),
//           8-wave workgroups x 2 per CU with one row per wave (4 waves per SIMD at the same LDS footprint), 4-wave workgroups x 4
//           per CU (would need half the LDS), each with / without DMA traffic and epilogue, and with the DMA issued by the matrix
//           waves themselves or by a fifth (ninth) producer wave.
// Clock = delta s_memtime / delta s_memrealtime (100 MHz) of one wave; power = hwmon power1_average sampled on a host thread.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread scripts/experiments/ubench_split_floor.hip -o rusty_sr_amd/build/ubench_split_floor
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <glob.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

struct Res {
    unsigned long long clk0, clk1, rt0, rt1;
};

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// PAT 0: zeros, 1: one constant, 2: random sign / mantissa, exponent of 1.0 (products +-[1,4): sums stay small, every input bit toggles)
template <int PAT>
__device__ __forceinline__ uint32_t pat_word(uint32_t seed) {
    if (PAT == 0) return 0u;
    if (PAT == 1) return 0x3c003c00u;
    return 0x3c003c00u ^ (mix(seed) & 0x83ff83ffu);
}
template <int PAT>
__device__ __forceinline__ f16x8 pat_vec(uint32_t seed) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = pat_word<PAT>(seed * 4u + k);
    return __builtin_bit_cast(f16x8, w);
}

__device__ __forceinline__ void stamp(Res* res, bool first) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long c = clock64(), r = wall_clock64();
        if (first) { res->clk0 = c; res->rt0 = r; } else { res->clk1 = c; res->rt1 = r; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Part 1: the matrix pipe alone.  One step = 12 MFMAs (2 taps x {hi.hi, hi.lo, lo.hi} x 2 tile rows), operands in registers.
// INST 0: v_mfma_f32_32x32x16_f16, 1: the same in bf16, 2: 24 x v_mfma_f32_16x16x32_f16 (same FLOPs)
// ORDER 0: the kernel's order (B operand shared by consecutive pairs), 1: every consecutive pair shares one operand, 2: none shared
// ---------------------------------------------------------------------------------------------------------------------------
template <int PAT, int ORDER, int SLEEP, int OCC, int INST>
__global__ __launch_bounds__(256, OCC) void stream_kernel(float* out, int iters, Res* res) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const uint32_t s0 = (blockIdx.x * 256u + tid) * 64u;
    f16x8 ah[2][2], al[2][2], bh[2], bl[2];
#pragma unroll
    for (int ts = 0; ts < 2; ++ts) {
        bh[ts] = pat_vec<PAT>(s0 + ts * 8 + 0);
        bl[ts] = pat_vec<PAT>(s0 + ts * 8 + 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[ts][m] = pat_vec<PAT>(s0 + ts * 8 + 2 + m);
            al[ts][m] = pat_vec<PAT>(s0 + ts * 8 + 4 + m);
        }
    }
    if (tid == 0) smem[0] = 0;  // (the dynamic LDS only pins the occupancy)
    f32x16 m0, m1, x0, x1;
    f32x4 q[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) { m0[r] = 0; m1[r] = 0; x0[r] = 0; x1[r] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = f32x4{0, 0, 0, 0};
    stamp(res, true);
#define MM(acc, a, b)                                                                                                 \
    {                                                                                                                 \
        if constexpr (INST == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0); \
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define MQ(k, a, b) { q[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q[k], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            if constexpr (INST == 2) {
                MQ(0, ah[ts][0], bh[ts]) MQ(1, ah[ts][1], bh[ts]) MQ(2, ah[ts][0], bl[ts]) MQ(3, ah[ts][1], bl[ts])
                MQ(4, al[ts][0], bh[ts]) MQ(5, al[ts][1], bh[ts]) MQ(6, ah[ts][0], bh[ts]) MQ(7, ah[ts][1], bh[ts])
                MQ(0, ah[ts][0], bl[ts]) MQ(1, ah[ts][1], bl[ts]) MQ(2, al[ts][0], bh[ts]) MQ(3, al[ts][1], bh[ts])
            } else if constexpr (ORDER == 0) {
                MM(m0, ah[ts][0], bh[ts]) MM(m1, ah[ts][1], bh[ts]) MM(x0, ah[ts][0], bl[ts]) MM(x1, ah[ts][1], bl[ts])
                MM(x0, al[ts][0], bh[ts]) MM(x1, al[ts][1], bh[ts])
            } else if constexpr (ORDER == 1) {
                MM(m0, ah[ts][0], bh[ts]) MM(x0, ah[ts][0], bl[ts]) MM(x1, ah[ts][1], bl[ts]) MM(m1, ah[ts][1], bh[ts])
                MM(x1, al[ts][1], bh[ts]) MM(x0, al[ts][0], bh[ts])
            } else {
                MM(m0, ah[ts][0], bh[ts]) MM(x1, ah[ts][1], bl[ts]) MM(x0, al[ts][0], bh[ts]) MM(m1, ah[ts][1], bh[ts])
                MM(x0, ah[ts][0], bl[ts]) MM(x1, al[ts][1], bh[ts])
            }
        }
        if constexpr (SLEEP > 0) { __builtin_amdgcn_s_sleep(SLEEP); __builtin_amdgcn_sched_barrier(0); }
    }
#undef MM
#undef MQ
    stamp(res, false);
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += m0[r] + m1[r] + x0[r] + x1[r];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += q[k][0] + q[k][1] + q[k][2] + q[k][3];
    out[blockIdx.x * 256 + tid] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Part 2: the step loop.  LDS: two half-tile buffers (4 planes of 7 KB: 2 hi, 2 lo; row pitch 36 px) + a 5-slot ring of 4 KB
// weight chunks, as in conv_stage_pipe_kernel<.., PREC = 1>.  NW matrix waves per workgroup, T tile rows per wave (NW * T = 8
// rows: the same tile), OCC workgroups per CU.  SMALL: planes of 3.5 KB and a 3-slot ring so that four workgroups fit a CU
// (a what-if: the shipped LDS plan does not allow it).
//   DMA  0: none (operands are whatever the LDS holds)
//        1: every matrix wave issues its share: the step's weight-chunk quarter and, on 3 steps of 4, one gather instruction
//           (64 lanes x 16 B, every lane its own 128-B line) -- with NW = 8 the two wave groups take turns
//        2: one extra PRODUCER wave issues all of it (4 chunk quarters + the gather's 4 planes per step); the matrix waves
//           issue ds_read + v_mfma + s_barrier only
//   EPI  BeLU + hi / lo split + pack + stores of the tile every 50 steps (the kernel's store_belu_tile_split)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kStepsPerTile = 50;

__device__ __forceinline__ void lds_dma16(const void* base, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 offset:0" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ f32x2 belu2(f32x2 v, float beta) {
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 t = v * v + one;
    const f32x2 s = {__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
    const f32x2 b = {beta, beta};
    return (b * v + s) - one;
}
__device__ __forceinline__ void split_half2(f32x2 v, uint32_t& hi2, uint32_t& lo2) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const f32x2 hf = {(float)h.x, (float)h.y};
    const f32x2 r = (v - hf) * f32x2{2048.0f, 2048.0f};
    hi2 = __builtin_bit_cast(uint32_t, h);
    lo2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r.x, r.y));
}
__device__ __forceinline__ void store_tile(char* base, const f32x16& accm, const f32x16& accx, float bias, float beta, bool odd) {
    const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
    const f32x2 bb = {bias, bias}, ks = {1.0f / 2048.0f, 1.0f / 2048.0f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 v = belu2(f32x2{accm[r], accm[r + 1]} + f32x2{accx[r], accx[r + 1]} * ks + bb, beta);
        uint32_t mh, ml;
        split_half2(v, mh, ml);
        const uint32_t ph = (uint32_t)__builtin_amdgcn_mov_dpp((int)mh, 0xB1, 0xF, 0xF, true), pl = (uint32_t)__builtin_amdgcn_mov_dpp((int)ml, 0xB1, 0xF, 0xF, true);
        const uint32_t oh = __builtin_amdgcn_perm(ph, mh, sel), ol = __builtin_amdgcn_perm(pl, ml, sel);
        const int row = (r & 3) + 8 * (r >> 2);
        *(uint32_t*)(base + row * 128) = oh;
        *(uint32_t*)(base + row * 128 + 64) = ol;
    }
}

template <int NW, int T, int OCC, int DMA, bool EPI, bool SMALL>
__global__ __launch_bounds__((NW + (DMA == 2 ? 1 : 0)) * 64, OCC) void step_kernel(const char* wbuf, const char* gbuf, size_t gbytes, char* obuf, size_t obytes,
                                                                                    float* out, int tiles, Res* res, int gstride) {
    constexpr int PLANE = SMALL ? 3584 : 7168;  // bytes per 8-channel plane of a half tile
    constexpr int HB = 4 * PLANE;               // one half-tile buffer: 2 hi planes, 2 lo planes
    constexpr int SLOTS = SMALL ? 3 : 5;
    constexpr int RING = 2 * HB;
    constexpr int TWH = 36;
    constexpr int NTHREADS = (NW + (DMA == 2 ? 1 : 0)) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    for (int k = tid; k < (RING + SLOTS * 4096) / 4; k += NTHREADS) ((uint32_t*)smem)[k] = pat_word<2>(blockIdx.x * 65536u + k);
    __syncthreads();
    const uint32_t lds0 = lds_addr(smem);
    const bool producer = DMA == 2 && wave == NW;
    const size_t gmask = gbytes - 1, omask = obytes - 1;
    stamp(res, true);

    if (producer) {
        // all of the workgroup's DMA traffic: per step the 4 quarters of the weight chunk four steps ahead and (3 steps of 4) the
        // 4 planes of one gather group; in flight across the barrier: the newest three steps' worth
        for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
            for (int s10 = 0; s10 < kStepsPerTile; s10 += 10) {
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int s = s10 + u;
                    const int slot = (u + 4) % SLOTS;
                    const char* wsrc = uniform_ptr(wbuf + (size_t)((s + 4) % kStepsPerTile) * 4096);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) lds_dma16(wsrc + qd * 1024, (uint32_t)(lane * 16), lds0 + RING + slot * 4096 + qd * 1024);
                    if (u % 4 != 3) {
                        const size_t goff = (((size_t)blockIdx.x * tiles + tile) * kStepsPerTile + s) * 8192;
                        const char* gsrc = uniform_ptr(gbuf + (goff & gmask));
#pragma unroll
                        for (int pl = 0; pl < 4; ++pl)
                            lds_dma16(gsrc + (gstride == 128 ? pl * 16 : pl * 2048), (uint32_t)(lane * gstride), lds0 + ((s10 / 10) & 1) * HB + pl * PLANE + (u % (SMALL ? 3 : 7)) * 1024);
                    }
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous step's eight requests at most stay in flight
                    __builtin_amdgcn_s_barrier();
                }
            }
        }
        stamp(res, false);
        return;
    }

    f32x16 accm[T], accx[T];
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[m][r] = 0; accx[m][r] = 0; }
    struct Ops { f16x8 bh[2], bl[2], ah[2][T], al[2][T]; };
    const char* abase = smem + h * PLANE + ((wave * T) * TWH + i) * 16;
    const char* wlane = smem + RING + (h * 32 + i) * 16;
    auto load = [&](Ops& o, int u, int buf) {  // operands of the step with unroll index u
        const int slot = u % SLOTS;
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            o.bh[ts] = *(const f16x8*)(wlane + slot * 4096 + ts * 1024);
            o.bl[ts] = *(const f16x8*)(wlane + slot * 4096 + 2048 + ts * 1024);
            const int kx = u % 5, ky = ts;
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const char* ab = abase + buf * HB + ((ky + m) * TWH + kx) * 16;
                o.ah[ts][m] = *(const f16x8*)ab;
                o.al[ts][m] = *(const f16x8*)(ab + 2 * PLANE);
            }
        }
    };
    Ops cur, nxt;
    load(cur, 0, 0);
    const int grp = NW == 8 ? (wave >> 2) : 0;  // NW = 8: the two wave groups take turns at the DMA issue
    const int w4 = wave & 3;
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int s10 = 0; s10 < kStepsPerTile; s10 += 10) {
            const int buf = (s10 / 10) & 1;
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int s = s10 + u;
                // next step's operands, one read in the shadow of each MFMA is what the kernel does; here: all up front, pinned
                load(nxt, (u + 1) % 10, buf);
                if constexpr (DMA == 1 || DMA == 3 || DMA == 4) {
                    if (DMA != 3 && (NW == 4 || grp == (u & 1))) {
                        const int slot = (u + 4) % SLOTS;
                        const char* wsrc = uniform_ptr(wbuf + (size_t)((s + 4) % kStepsPerTile) * 4096 + w4 * 1024);
                        lds_dma16(wsrc, (uint32_t)(lane * 16), lds0 + RING + slot * 4096 + w4 * 1024);
                    }
                    if (DMA != 4 && u % 4 != 3 && (NW == 4 || grp == ((u >> 1) & 1))) {
                        const size_t goff = (((size_t)blockIdx.x * tiles + tile) * kStepsPerTile + s) * 8192;
                        const char* gsrc = uniform_ptr(gbuf + (goff & gmask) + (gstride == 128 ? w4 * 16 : w4 * 2048));
                        lds_dma16(gsrc, (uint32_t)(lane * gstride), lds0 + (buf ^ 1) * HB + w4 * PLANE + (u % (SMALL ? 3 : 7)) * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ts = 0; ts < 2; ++ts) {
#pragma unroll
                    for (int m = 0; m < T; ++m) accm[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.ah[ts][m], cur.bh[ts], accm[m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < T; ++m) accx[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.ah[ts][m], cur.bl[ts], accx[m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < T; ++m) accx[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.al[ts][m], cur.bh[ts], accx[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DMA == 1 || DMA == 3 || DMA == 4) wait_vm<NW == 4 ? 6 : 3>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        if constexpr (EPI) {
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const size_t row = ((size_t)blockIdx.x * tiles + tile) * 8 + wave * T + m;
                char* base = obuf + ((row * 36 * 128 + (size_t)(4 * h + (i & 1)) * 128 + (i & ~1) * 2) & omask);
                store_tile(base, accm[m], accx[m], 0.01f * i, 0.5f, i & 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) { accm[m][r] = 0; accx[m][r] = 0; }
            }
        }
    }
    stamp(res, false);
    float sum = 0;
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += accm[m][r] + accx[m][r];
    out[blockIdx.x * NTHREADS + tid] = sum;
}

// The same step loop on v_mfma_f32_16x16x32_f16 (M = 16 pixels, N = 16 output channels, K = 32 = the step's two taps x 16 channels):
// per step and tile row 2 pixel halves x 2 channel halves x 3 products = 12 MFMAs of 16 cycles (the same FLOPs as 6 of 32x32x16), operand
// reads: 4 B fragments (channel half x hi / lo) + 4 T A fragments (row x pixel half x hi / lo) -- as many ds_read_b128 as the 32x32x16 form.
// A lane holds tap t0's channels (lanes 0-31: planes 0 / 1) or its right-hand neighbour's (lanes 32-63): the lane base carries the + 1 px.
__device__ __forceinline__ void store_quad(char* base, const f32x4& accm, const f32x4& accx, float bias, float beta, bool odd) {
    const uint32_t sel = odd ? 0x03020706u : 0x05040100u;
    const f32x2 bb = {bias, bias}, ks = {1.0f / 2048.0f, 1.0f / 2048.0f};
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 v = belu2(f32x2{accm[r], accm[r + 1]} + f32x2{accx[r], accx[r + 1]} * ks + bb, beta);
        uint32_t mh, ml;
        split_half2(v, mh, ml);
        const uint32_t ph = (uint32_t)__builtin_amdgcn_mov_dpp((int)mh, 0xB1, 0xF, 0xF, true), pl = (uint32_t)__builtin_amdgcn_mov_dpp((int)ml, 0xB1, 0xF, 0xF, true);
        const uint32_t oh = __builtin_amdgcn_perm(ph, mh, sel), ol = __builtin_amdgcn_perm(pl, ml, sel);
        *(uint32_t*)(base + r * 128) = oh;
        *(uint32_t*)(base + r * 128 + 64) = ol;
    }
}

template <int NW, int T, int OCC, int DMA, bool EPI>
__global__ __launch_bounds__(NW * 64, OCC) void step16_kernel(const char* wbuf, const char* gbuf, size_t gbytes, char* obuf, size_t obytes, float* out,
                                                              int tiles, Res* res, int gstride) {
    constexpr int PLANE = 7168, HB = 4 * PLANE, SLOTS = 5, RING = 2 * HB, TWH = 36, NTHREADS = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < (RING + SLOTS * 4096) / 4; k += NTHREADS) ((uint32_t*)smem)[k] = pat_word<2>(blockIdx.x * 65536u + k);
    __syncthreads();
    const uint32_t lds0 = lds_addr(smem);
    const size_t gmask = gbytes - 1, omask = obytes - 1;
    stamp(res, true);
    f32x4 accm[T][2][2], accx[T][2][2];
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { accm[m][a][b] = f32x4{0, 0, 0, 0}; accx[m][a][b] = f32x4{0, 0, 0, 0}; }
    struct Ops { f16x8 bh[2], bl[2], ah[T][2], al[T][2]; };
    const char* abase = smem + ((lane >> 4) & 1) * PLANE + ((wave * T) * TWH + (lane & 15) + (lane >> 5)) * 16;
    const char* wlane = smem + RING + lane * 16;
    auto load = [&](Ops& o, int u, int buf) {
        const int slot = u % SLOTS;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            o.bh[ch] = *(const f16x8*)(wlane + slot * 4096 + ch * 1024);
            o.bl[ch] = *(const f16x8*)(wlane + slot * 4096 + 2048 + ch * 1024);
        }
        const int kx = u % 4, ky = (u >> 2) & 1;
#pragma unroll
        for (int m = 0; m < T; ++m)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const char* ab = abase + buf * HB + ((ky + m) * TWH + kx + 16 * ph) * 16;
                o.ah[m][ph] = *(const f16x8*)ab;
                o.al[m][ph] = *(const f16x8*)(ab + 2 * PLANE);
            }
    };
    Ops cur, nxt;
    load(cur, 0, 0);
    const int grp = NW == 8 ? (wave >> 2) : 0;
    const int w4 = wave & 3;
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int s10 = 0; s10 < kStepsPerTile; s10 += 10) {
            const int buf = (s10 / 10) & 1;
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int s = s10 + u;
                load(nxt, (u + 1) % 10, buf);
                if constexpr (DMA == 1 || DMA == 3 || DMA == 4) {
                    if (DMA != 3 && (NW == 4 || grp == (u & 1))) {
                        const int slot = (u + 4) % SLOTS;
                        const char* wsrc = uniform_ptr(wbuf + (size_t)((s + 4) % kStepsPerTile) * 4096 + w4 * 1024);
                        lds_dma16(wsrc, (uint32_t)(lane * 16), lds0 + RING + slot * 4096 + w4 * 1024);
                    }
                    if (DMA != 4 && u % 4 != 3 && (NW == 4 || grp == ((u >> 1) & 1))) {
                        const size_t goff = (((size_t)blockIdx.x * tiles + tile) * kStepsPerTile + s) * 8192;
                        const char* gsrc = uniform_ptr(gbuf + (goff & gmask) + (gstride == 128 ? w4 * 16 : w4 * 2048));
                        lds_dma16(gsrc, (uint32_t)(lane * gstride), lds0 + (buf ^ 1) * HB + w4 * PLANE + (u % 7) * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < T; ++m)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) accm[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.ah[m][ph], cur.bh[ch], accm[m][ph][ch], 0, 0, 0);
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) accx[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.ah[m][ph], cur.bl[ch], accx[m][ph][ch], 0, 0, 0);
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) accx[m][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.al[m][ph], cur.bh[ch], accx[m][ph][ch], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DMA == 1 || DMA == 3 || DMA == 4) wait_vm<NW == 4 ? 6 : 3>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        if constexpr (EPI) {
            const int i = lane & 15, g = lane >> 4;
#pragma unroll
            for (int m = 0; m < T; ++m) {
                const size_t row = ((size_t)blockIdx.x * tiles + tile) * 8 + wave * T + m;
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        char* base = obuf + ((row * 36 * 128 + (size_t)(16 * ph + 4 * g + (i & 1)) * 128 + ch * 32 + (i & ~1) * 2) & omask);
                        store_quad(base, accm[m][ph][ch], accx[m][ph][ch], 0.01f * i, 0.5f, i & 1);
                        accm[m][ph][ch] = f32x4{0, 0, 0, 0}; accx[m][ph][ch] = f32x4{0, 0, 0, 0};
                    }
            }
        }
    }
    stamp(res, false);
    float sum = 0;
#pragma unroll
    for (int m = 0; m < T; ++m)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += accm[m][a][b][r] + accx[m][a][b][r];
    out[blockIdx.x * NTHREADS + tid] = sum;
}

__global__ void fill_random(uint32_t* p, size_t n, uint32_t seed) {  // what the DMAs bring into LDS must toggle like real operands do
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = pat_word<2>(seed + (uint32_t)k);
}

// ---------------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------------
struct Monitor {  // board power (hwmon, microwatts) and the driver's sclk level, sampled every 10 ms while a kernel runs
    std::string power_path, sclk_path;
    std::atomic<bool> run{false};
    std::thread th;
    std::vector<double> watts, mhz;
    Monitor() {
        glob_t g;
        for (const char* pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"}) {
            if (power_path.empty() && glob(pat, 0, nullptr, &g) == 0) {
                if (g.gl_pathc > 0) power_path = g.gl_pathv[0];
                globfree(&g);
            }
        }
        if (glob("/sys/class/drm/card*/device/pp_dpm_sclk", 0, nullptr, &g) == 0) {
            if (g.gl_pathc > 0) sclk_path = g.gl_pathv[0];
            globfree(&g);
        }
    }
    static bool slurp(const std::string& p, char* buf, size_t cap) {
        FILE* f = fopen(p.c_str(), "r");
        if (!f) return false;
        const size_t n = fread(buf, 1, cap - 1, f);
        fclose(f);
        buf[n] = 0;
        return n > 0;
    }
    void start() {
        watts.clear(); mhz.clear();
        run = true;
        th = std::thread([this] {
            char buf[4096];
            while (run) {
                if (!power_path.empty() && slurp(power_path, buf, sizeof(buf))) watts.push_back(atof(buf) * 1e-6);
                if (!sclk_path.empty() && slurp(sclk_path, buf, sizeof(buf))) {
                    for (char* ln = strtok(buf, "\n"); ln; ln = strtok(nullptr, "\n"))
                        if (strchr(ln, '*')) { const char* c = strchr(ln, ':'); if (c) mhz.push_back(atof(c + 1)); }
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        });
    }
    void stop(double& w, double& f) {
        run = false;
        th.join();
        auto tail_mean = [](const std::vector<double>& v) {  // the second half of the samples: the power manager has settled
            if (v.empty()) return 0.0;
            double s = 0; size_t n = 0;
            for (size_t k = v.size() / 2; k < v.size(); ++k) { s += v[k]; ++n; }
            return n ? s / n : 0.0;
        };
        w = tail_mean(watts); f = tail_mean(mhz);
    }
};

static Monitor* g_mon;
static float* g_out;
static Res* g_res;
static char *g_w, *g_g, *g_o;
static const size_t kGBytes = (size_t)1 << 28, kOBytes = (size_t)1 << 28;

static void report(const char* name, float ms, double mfma_per_wave, int waves_per_simd, double flops, double watts, double sclk) {
    Res r;
    CHECK(hipMemcpy(&r, g_res, sizeof(r), hipMemcpyDeviceToHost));
    const double cyc = (double)(r.clk1 - r.clk0), rt = (double)(r.rt1 - r.rt0) * 1e-8;  // s_memrealtime: 100 MHz
    const double ghz = rt > 0 ? cyc / rt * 1e-9 : 0;
    // (older waves win the matrix pipe's arbitration, so block 0 finishes long before the kernel does when several workgroups share a CU:
    // its cycle count is no measure of the pipe's occupancy -- the clock, a ratio, is.  busy = TFLOP/s over the dense f16 peak AT THAT CLOCK.)
    (void)mfma_per_wave; (void)waves_per_simd; (void)sclk;
    const double tf = flops / ms * 1e-9, util = ghz > 0 ? tf / (2500.0 * ghz / 2.4) : 0;
    printf("%-66s %8.2f ms %7.1f TF  clk %5.3f GHz  busy@clk %5.3f  %6.1f W\n", name, ms, tf, ghz, util, watts);
    fflush(stdout);
}

template <int PAT, int ORDER, int SLEEP, int OCC, int INST>
static void run_stream(const char* name, double target_ms = 250) {
    const int grid = 256 * OCC;
    const size_t lds = OCC == 1 ? 90 * 1024 : OCC == 2 ? 70 * 1024 : 36 * 1024;
    auto kern = stream_kernel<PAT, ORDER, SLEEP, OCC, INST>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int iters = 20000;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {  // a calibration pass, then ~target_ms under the power monitor
        if (pass) g_mon->start();
        CHECK(hipEventRecord(e0));
        kern<<<grid, 256, lds>>>(g_out, iters, g_res);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (!pass) iters = (int)std::min(4e7, std::max(1000.0, iters * target_ms / ms));
    }
    double w, f;
    g_mon->stop(w, f);
    report(name, ms, (double)iters * 12, OCC, (double)grid * 4 * iters * 12 * 2.0 * 32 * 32 * 16, w, f);
}

template <int NW, int T, int OCC, int DMA, bool EPI, bool SMALL>
static void run_step(const char* name, double target_ms = 250, int gstride = 128) {
    constexpr int NTH = (NW + (DMA == 2 ? 1 : 0)) * 64;
    const int grid = 256 * OCC;
    const size_t lds = SMALL ? (size_t)2 * 4 * 3584 + 3 * 4096 : (size_t)2 * 4 * 7168 + 5 * 4096;
    auto kern = step_kernel<NW, T, OCC, DMA, EPI, SMALL>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NTH, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int tiles = 40;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) g_mon->start();
        CHECK(hipEventRecord(e0));
        kern<<<grid, NTH, lds>>>(g_w, g_g, kGBytes, g_o, kOBytes, g_out, tiles, g_res, gstride);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (!pass) tiles = (int)std::min(4e6, std::max(20.0, tiles * target_ms / ms));
    }
    double w, f;
    g_mon->stop(w, f);
    char full[160];
    snprintf(full, sizeof(full), "%s [%d wg/CU]", name, nb);
    const double mfma_per_wave = (double)tiles * kStepsPerTile * 6 * T;
    report(full, ms, mfma_per_wave, OCC * NW / 4, (double)grid * NW * mfma_per_wave * 2.0 * 32 * 32 * 16, w, f);
}

template <int NW, int T, int OCC, int DMA, bool EPI>
static void run_step16(const char* name, double target_ms = 250, int gstride = 128) {
    constexpr int NTH = NW * 64;
    const int grid = 256 * OCC;
    const size_t lds = (size_t)2 * 4 * 7168 + 5 * 4096;
    auto kern = step16_kernel<NW, T, OCC, DMA, EPI>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NTH, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int tiles = 40;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) g_mon->start();
        CHECK(hipEventRecord(e0));
        kern<<<grid, NTH, lds>>>(g_w, g_g, kGBytes, g_o, kOBytes, g_out, tiles, g_res, gstride);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (!pass) tiles = (int)std::min(4e6, std::max(20.0, tiles * target_ms / ms));
    }
    double w, f;
    g_mon->stop(w, f);
    char full[160];
    snprintf(full, sizeof(full), "%s [%d wg/CU]", name, nb);
    const double mfma32_per_wave = (double)tiles * kStepsPerTile * 6 * T;  // in units of one 32x32x16
    report(full, ms, mfma32_per_wave, OCC * NW / 4, (double)grid * NW * mfma32_per_wave * 2.0 * 32 * 32 * 16, w, f);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    Monitor mon;
    g_mon = &mon;
    printf("power: %s   sclk: %s\n", mon.power_path.empty() ? "(none)" : mon.power_path.c_str(), mon.sclk_path.empty() ? "(none)" : mon.sclk_path.c_str());
    CHECK(hipMalloc(&g_out, (size_t)1024 * 1024 * 4));
    CHECK(hipMalloc(&g_res, sizeof(Res)));
    CHECK(hipMalloc(&g_w, (size_t)kStepsPerTile * 4096 + 8192));
    CHECK(hipMalloc(&g_g, kGBytes + (1 << 20)));
    CHECK(hipMalloc(&g_o, kOBytes + (1 << 20)));
    fill_random<<<256, 256>>>((uint32_t*)g_w, ((size_t)kStepsPerTile * 4096 + 8192) / 4, 1u);
    fill_random<<<4096, 256>>>((uint32_t*)g_g, (kGBytes + (1 << 20)) / 4, 77u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(g_o, 0, kOBytes + (1 << 20)));
    const double T = quick ? 40 : 250;

    printf("== part 1: the matrix pipe alone (12 MFMAs per step, operands in registers) ==\n");
    run_stream<0, 0, 0, 2, 0>("f16 32x32x16  zeros            2 waves/SIMD", T);
    run_stream<1, 0, 0, 2, 0>("f16 32x32x16  constant         2 waves/SIMD", T);
    run_stream<2, 0, 0, 2, 0>("f16 32x32x16  random           2 waves/SIMD", T);
    run_stream<2, 0, 0, 1, 0>("f16 32x32x16  random           1 wave/SIMD", T);
    run_stream<2, 0, 0, 4, 0>("f16 32x32x16  random           4 waves/SIMD", T);
    run_stream<2, 1, 0, 2, 0>("f16 32x32x16  random, operand-sharing order", T);
    run_stream<2, 2, 0, 2, 0>("f16 32x32x16  random, no operand shared", T);
    run_stream<2, 0, 0, 2, 1>("bf16 32x32x16 random           2 waves/SIMD", T);
    run_stream<2, 0, 0, 2, 2>("f16 16x16x32  random (24 per step) 2 waves/SIMD", T);
    run_stream<2, 0, 0, 1, 2>("f16 16x16x32  random (24 per step) 1 wave/SIMD", T);
    run_stream<1, 0, 0, 2, 2>("f16 16x16x32  constant         2 waves/SIMD", T);
    printf("-- duty cycle: one wave per SIMD, s_sleep k (64 k cycles) after every 12 MFMAs (384 cycles) --\n");
    run_stream<2, 0, 1, 1, 0>("f16 random, sleep 1  (nominal busy 0.86)", T);
    run_stream<2, 0, 2, 1, 0>("f16 random, sleep 2  (0.75)", T);
    run_stream<2, 0, 3, 1, 0>("f16 random, sleep 3  (0.67)", T);
    run_stream<2, 0, 4, 1, 0>("f16 random, sleep 4  (0.60)", T);
    run_stream<2, 0, 6, 1, 0>("f16 random, sleep 6  (0.50)", T);
    run_stream<2, 0, 12, 1, 0>("f16 random, sleep 12 (0.33)", T);
    run_stream<1, 0, 3, 1, 0>("f16 constant, sleep 3 (0.67)", T);
    run_stream<2, 0, 2, 1, 2>("f16 16x16x32 random, sleep 2 (0.75)", T);
    run_stream<2, 0, 4, 1, 2>("f16 16x16x32 random, sleep 4 (0.60)", T);

    printf("== part 2: the step loop (2 taps x 16 channels per step, 8 x 32 px tile per workgroup) ==\n");
    run_step<4, 2, 2, 0, false, false>("4 waves x T=2, 2 wg/CU  (as shipped), bare loop", T);
    run_step<4, 2, 2, 1, false, false>("4 waves x T=2, 2 wg/CU, + DMA by every wave", T);
    run_step<4, 2, 2, 0, true, false>("4 waves x T=2, 2 wg/CU, + epilogue", T);
    run_step<4, 2, 2, 1, true, false>("4 waves x T=2, 2 wg/CU, + DMA + epilogue  (= the kernel)", T);
    run_step<4, 2, 2, 2, true, false>("4+1 waves x T=2, 2 wg/CU, producer wave + epilogue", T);
    run_step<4, 2, 2, 2, false, false>("4+1 waves x T=2, 2 wg/CU, producer wave, no epilogue", T);
    run_step<4, 2, 1, 1, true, false>("4 waves x T=2, 1 wg/CU, + DMA + epilogue", T);
    run_step<8, 1, 2, 0, false, false>("8 waves x T=1, 2 wg/CU  (4 waves/SIMD), bare loop", T);
    run_step<8, 1, 2, 1, false, false>("8 waves x T=1, 2 wg/CU, + DMA by turns", T);
    run_step<8, 1, 2, 1, true, false>("8 waves x T=1, 2 wg/CU, + DMA + epilogue", T);
    run_step<8, 1, 2, 2, true, false>("8+1 waves x T=1, 2 wg/CU, producer wave + epilogue", T);
    run_step<4, 1, 4, 0, false, true>("4 waves x T=1, 4 wg/CU (half LDS: what-if), bare loop", T);
    run_step<4, 1, 4, 1, true, true>("4 waves x T=1, 4 wg/CU (half LDS), + DMA + epilogue", T);
    run_step<4, 2, 3, 1, true, true>("4 waves x T=2, 3 wg/CU (half LDS), + DMA + epilogue", T);
    run_step<4, 4, 1, 0, false, false>("4 waves x T=4, 1 wg/CU (16-row tile: half the B reads), bare loop", T);
    run_step<4, 4, 1, 1, true, false>("4 waves x T=4, 1 wg/CU, + DMA + epilogue", T);
    printf("-- what the DMA traffic costs, and what a line-contiguous gather (row-planar maps in HBM) would save --\n");
    run_step<4, 2, 2, 3, false, false>("4 waves x T=2, 2 wg/CU, gathers only (one 128-B line per lane)", T);
    run_step<4, 2, 2, 4, false, false>("4 waves x T=2, 2 wg/CU, weight chunks only", T);
    run_step<4, 2, 2, 3, false, false>("4 waves x T=2, 2 wg/CU, gathers only, CONTIGUOUS (16 B per lane)", T, 16);
    run_step<4, 2, 2, 1, false, false>("4 waves x T=2, 2 wg/CU, + DMA, contiguous gathers", T, 16);
    run_step<4, 2, 2, 1, true, false>("4 waves x T=2, 2 wg/CU, + DMA + epilogue, contiguous gathers", T, 16);
    run_step<8, 2, 1, 1, true, false>("8 waves x T=2, 1 wg/CU (16-row tile, one ring), + DMA + epilogue", T);
    run_step<8, 2, 1, 1, true, false>("8 waves x T=2, 1 wg/CU (16-row tile, one ring), + DMA + epi, contiguous", T, 16);
    run_step<4, 4, 1, 1, true, false>("4 waves x T=4, 1 wg/CU, + DMA + epilogue, contiguous gathers", T, 16);
    printf("== part 3: the same step loop on v_mfma_f32_16x16x32_f16 (K = the step's two taps x 16 channels) ==\n");
    run_step16<4, 2, 2, 0, false>("16x16x32: 4 waves x T=2, 2 wg/CU, bare loop", T);
    run_step16<4, 2, 2, 1, false>("16x16x32: 4 waves x T=2, 2 wg/CU, + DMA", T);
    run_step16<4, 2, 2, 0, true>("16x16x32: 4 waves x T=2, 2 wg/CU, + epilogue", T);
    run_step16<4, 2, 2, 1, true>("16x16x32: 4 waves x T=2, 2 wg/CU, + DMA + epilogue", T);
    run_step16<8, 1, 2, 1, true>("16x16x32: 8 waves x T=1, 2 wg/CU, + DMA + epilogue", T);
    run_step16<4, 2, 1, 1, true>("16x16x32: 4 waves x T=2, 1 wg/CU, + DMA + epilogue", T);
    run_step16<4, 2, 2, 1, false>("16x16x32: 4 waves x T=2, 2 wg/CU, + DMA, contiguous gathers", T, 16);
    run_step16<4, 2, 2, 1, true>("16x16x32: 4 waves x T=2, 2 wg/CU, + DMA + epilogue, contiguous gathers", T, 16);
    run_step16<8, 2, 1, 1, true>("16x16x32: 8 waves x T=2, 1 wg/CU, + DMA + epilogue, contiguous gathers", T, 16);
    run_step16<4, 4, 1, 1, true>("16x16x32: 4 waves x T=4, 1 wg/CU, + DMA + epilogue, contiguous gathers", T, 16);
    return 0;
}
