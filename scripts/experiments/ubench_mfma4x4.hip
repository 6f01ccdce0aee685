// Round 6: can v_mfma_f32_4x4x1_16B_f32 carry the last stage (27 -> 28 output channels instead of 27 -> 32)?
//
// The instruction is sixteen independent 4x4x1 outer products, 512 FLOP in 2 passes (8 cycles): the same 64 FLOP / clk / SIMD as
// v_mfma_f32_32x32x2_f32, but N moves in steps of 4.  With CBSZ = 4 the A operand of block ABID is broadcast to all sixteen blocks:
//     D[lane 4 b + j][vgpr i] += A[lane 4 ABID + i] * B[lane 4 b + j]
// so with A = weights (lane = output channel, ONE register per k holds all 28 channels: group g = lanes 4 g .. 4 g + 3) and
// B = pixels (lane = pixel) a k-step of 64 pixels x 28 channels is 7 instructions on one weight register and one pixel register, and
// a lane ends up holding all 28 channels of ITS pixel (depth-to-space becomes lane-local).
//
//   part 1 (probe): is that what CBSZ / ABID do on gfx950?
//   part 2 (rate):  the stream alone, with the stage loop's LDS operand reads (1 weight + 2 pixel ds_read_b128 per 56 MFMAs),
//                   with a counted barrier + a weight chunk by LDS-DMA per step (224 MFMAs), against the 32x32x2 loop of the
//                   shipped kernel under the same structure.
// hipcc --offload-arch=gfx950 -O3 scripts/experiments/ubench_mfma4x4.hip -o /tmp/ub4 && /tmp/ub4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ABID>
__global__ void probe(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}
template <int ABID>
static int run_probe() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 1000.0f + 3 * l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<ABID><<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) bad += hd[l * 4 + i] != ha[4 * ABID + i] * hb[l];
    hipFree(a); hipFree(b); hipFree(d);
    return bad;
}


// part 1b: is a chain of 4x4x1 instructions the same fmaf chain, bit for bit, as the 32x32x2 instructions it would replace?
// 32 pixels x 32 channels over K = 8: 32x32x2 takes k = (e, 4 + e) per instruction (lane half h = the k of the pair), e = 0..3;
// the 4x4x1 chain takes (k-quad 0, e) then (k-quad 1, e).
__global__ void chain32(const float* P, const float* W, float* D) {  // P[64][8], W[8][32] -> D[32 px][32 ch]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.125f;
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(P[i * 8 + 4 * h + e], W[(4 * h + e) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ void chain4(const float* P, const float* W, float* D) {  // -> D[64 px][32 ch]
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    f32x4 acc[8];
    for (int g = 0; g < 8; ++g) acc[g] = f32x4{0.125f, 0.125f, 0.125f, 0.125f};
    for (int e = 0; e < 4; ++e) {
        const float w = W[(4 * h + e) * 32 + j], p0 = P[lane * 8 + e], p1 = P[lane * 8 + 4 + e];
#define G(g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, p0, acc[g], 4, g, 0);
        G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7)
#undef G
#define G(g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, p1, acc[g], 4, 8 + g, 0);
        G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7)
#undef G
    }
    for (int g = 0; g < 8; ++g)
        for (int i = 0; i < 4; ++i) D[lane * 32 + 4 * g + i] = acc[g][i];
}
static void run_chain() {
    std::vector<float> hp(64 * 8), hw(8 * 32), d32(32 * 32), d4(64 * 32), ref(64 * 32);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 21); };
    for (auto& v : hp) v = rnd();
    for (auto& v : hw) v = rnd();
    float *P, *W, *D;
    hipMalloc(&P, hp.size() * 4); hipMalloc(&W, hw.size() * 4); hipMalloc(&D, d4.size() * 4);
    hipMemcpy(P, hp.data(), hp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    chain32<<<1, 64>>>(P, W, D);
    hipMemcpy(d32.data(), D, d32.size() * 4, hipMemcpyDeviceToHost);
    chain4<<<1, 64>>>(P, W, D);
    hipMemcpy(d4.data(), D, d4.size() * 4, hipMemcpyDeviceToHost);
    // the fmaf chain on the host, k order 0 4 1 5 2 6 3 7
    int bad4 = 0, bad32 = 0, differ = 0;
    for (int p = 0; p < 64; ++p)
        for (int c = 0; c < 32; ++c) {
            float a = 0.125f;
            for (int e = 0; e < 4; ++e) { a = __builtin_fmaf(hp[p * 8 + e], hw[e * 32 + c], a); a = __builtin_fmaf(hp[p * 8 + 4 + e], hw[(4 + e) * 32 + c], a); }
            ref[p * 32 + c] = a;
            bad4 += __builtin_memcmp(&a, &d4[p * 32 + c], 4) != 0;
            if (p < 32) { bad32 += __builtin_memcmp(&a, &d32[p * 32 + c], 4) != 0; differ += __builtin_memcmp(&d32[p * 32 + c], &d4[p * 32 + c], 4) != 0; }
        }
    printf("chain: 4x4x1 vs host fmaf chain: %d of 2048 differ; 32x32x2 vs host fmaf chain: %d of 1024 differ; 4x4x1 vs 32x32x2: %d of 1024 differ\n", bad4, bad32, differ);
    hipFree(P); hipFree(W); hipFree(D);
}

// MODE 0: MFMAs only.  1: + operand reads from LDS.  2: + counted barrier and one 4 KB LDS-DMA weight chunk per step.
// SHAPE 0: 4x4x1 (7 groups = 28 channels; PX sets of 64 pixels per wave).  SHAPE 1: 32x32x2 (32 channels; PX x 2 tile rows of 32).
template <int SHAPE, int MODE, int PX>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) ((float*)smem)[i] = (float)((i * 7) & 15) * 0.001f - 0.004f;
    __syncthreads();
    const char* base = smem + lane * 16;
    f32x4 acc[PX][7];
    f32x16 big[PX * 2];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int g = 0; g < 7; ++g) acc[p][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PX * 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) big[p][r] = 0.f;
    f32x4 wv = {0.5f, 0.25f, 0.125f, 1.f}, pv[PX * 2];
#pragma unroll
    for (int p = 0; p < PX * 2; ++p) pv[p] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {  // one step = 4 operand groups of 8 k = 32 k
        if (MODE >= 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + ((it & 15) * 256 + tid) * 4),
                                             (__attribute__((address_space(3))) void*)(smem + 49152 + (it & 1) * 4096 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE >= 1) {
                wv = *(const f32x4*)(base + 32768 + q * 1024);
#pragma unroll
                for (int p = 0; p < PX * 2; ++p) pv[p] = *(const f32x4*)(base + ((it + q) & 7) * 2048 + p * 1024);
            }
            if constexpr (SHAPE == 0) {
                // pv[2 p + h]: k-quad h of pixel set p (the real kernel reads plane 2 (q & 1) + h with all 64 lanes)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
#define G(g, h) acc[p][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], pv[2 * p + h][e], acc[p][g], 4, 8 * h + g, 0);
                        G(0, 0) G(1, 0) G(2, 0) G(3, 0) G(4, 0) G(5, 0) G(6, 0)
                        G(0, 1) G(1, 1) G(2, 1) G(3, 1) G(4, 1) G(5, 1) G(6, 1)
#undef G
                    }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int p = 0; p < PX * 2; ++p) big[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[p][e], wv[e], big[p], 0, 0, 0);
            }
        }
        if (MODE >= 2) { asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    float s = 0;
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int g = 0; g < 7; ++g) s += acc[p][g][0] + acc[p][g][1] + acc[p][g][2] + acc[p][g][3];
#pragma unroll
    for (int p = 0; p < PX * 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += big[p][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int MODE, int PX>
static void run(const char* name, int iters) {
    const int grid = 512;
    float *out, *w;
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&w, 16 * 1024 * 4);
    hipMemset(w, 0, 16 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<SHAPE, MODE, PX>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<SHAPE, MODE, PX><<<grid, 256, 65536>>>(out, w, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<SHAPE, MODE, PX><<<grid, 256, 65536>>>(out, w, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per step and wave: 4x4x1: 4 q x 56 x PX MFMAs of 512 FLOP; 32x32x2: 4 q x 4 x 2 PX MFMAs of 4096 FLOP
    const double per_step = SHAPE == 0 ? 4.0 * 56 * PX * 512 : 4.0 * 4 * 2 * PX * 4096;
    const double flops = (double)grid * 4 * iters * per_step;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s issued  = %5.3f of 157.3\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(out); hipFree(w);
}

int main() {
    int bad = 0;
    bad += run_probe<0>(); bad += run_probe<3>(); bad += run_probe<6>(); bad += run_probe<8>(); bad += run_probe<14>(); bad += run_probe<15>();
    printf("probe: D[4b+j][i] += A[4*ABID+i] * B[4b+j] with CBSZ=4: %s (%d mismatches over ABID 0,3,6,8,14,15)\n", bad ? "NO" : "yes", bad);
    run_chain();
    const int it = 3000;
    run<1, 0, 1>("32x32x2 stream, 64 px per wave (the shipped loop's shape)", it);
    run<1, 1, 1>("32x32x2 + LDS operand reads", it);
    run<1, 2, 1>("32x32x2 + reads + barrier + weight DMA per step", it);
    run<0, 0, 1>("4x4x1 stream, 64 px per wave, 28 channels", it);
    run<0, 1, 1>("4x4x1 + LDS operand reads", it);
    run<0, 2, 1>("4x4x1 + reads + barrier + weight DMA per step", it);
    run<0, 0, 2>("4x4x1 stream, 128 px per wave", it);
    run<0, 1, 2>("4x4x1 + LDS operand reads, 128 px per wave", it);
    run<0, 2, 2>("4x4x1 + reads + barrier + weight DMA, 128 px per wave", it);
    return 0;
}
