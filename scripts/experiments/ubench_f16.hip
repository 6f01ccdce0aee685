// Micro-benchmark: how well do ds_read_b128 operand fetches overlap v_mfma_f32_32x32x16_f16
// on MI355X?  Per iteration each wave issues NREADS independent ds_read_b128 (into registers the
// MFMAs of the NEXT iteration consume) and 12 MFMAs (2 accumulator chains x 6).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_f16.hip -o exp/ubench_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OFF>
__device__ __forceinline__ void rd(f16x8& d, uint32_t a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }

template <int NREADS, int OCC, bool BAR, int MODE = 0>
__global__ __launch_bounds__(256, OCC) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384 + 1024; i += 256) ((float*)smem)[i] = MODE ? __uint_as_float(0x3c003c00u ^ ((i * 2654435761u) & 0x83ff83ffu)) : 0.001f * (i & 15);
    __syncthreads();
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; a2[r] = 0; a3[r] = 0; }
    f16x8 q[2][12];
    for (int s = 0; s < 2; ++s) for (int j = 0; j < 12; ++j) for (int e = 0; e < 8; ++e) q[s][j][e] = (_Float16)(0.01f * (j + 1));
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + (MODE ? ((lane >> 5) * 7168 + ((tid >> 6) * 2 * 36 + (lane & 31)) * 16) : (lane * 16 + (tid >> 6) * 4096));
    const uint32_t wbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + (MODE ? 57344 + lane * 16 : lane * 16);
#define BODY(S, N)                                                                             \
    {                                                                                          \
        if (MODE == 0) {                                                                       \
        if (NREADS > 0) rd<0>(q[N][0], base);  if (NREADS > 1) rd<1024>(q[N][1], base);         \
        if (NREADS > 2) rd<2048>(q[N][2], base); if (NREADS > 3) rd<3072>(q[N][3], base);       \
        if (NREADS > 4) rd<16384>(q[N][4], base); if (NREADS > 5) rd<17408>(q[N][5], base);     \
        if (NREADS > 6) rd<18432>(q[N][6], base); if (NREADS > 7) rd<19456>(q[N][7], base);     \
        if (NREADS > 8) rd<32768>(q[N][8], base); if (NREADS > 9) rd<33792>(q[N][9], base);     \
        if (NREADS > 10) rd<34816>(q[N][10], base); if (NREADS > 11) rd<35840>(q[N][11], base); \
        } else {                                                                               \
        rd<0>(q[N][0], wbase); rd<1024>(q[N][1], wbase); rd<2048>(q[N][2], wbase); rd<3072>(q[N][3], wbase); \
        rd<608 + 0 * 7168>(q[N][4], base); rd<608 + 2 * 7168>(q[N][5], base);                   \
        rd<608 + 4 * 7168>(q[N][6], base); rd<608 + 6 * 7168>(q[N][7], base);                   \
        rd<608 + 576 + 0 * 7168>(q[N][8], base); rd<608 + 576 + 2 * 7168>(q[N][9], base);       \
        rd<608 + 576 + 4 * 7168>(q[N][10], base); rd<608 + 576 + 6 * 7168>(q[N][11], base);     \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][4], q[S][0], a0, 0, 0, 0);            \
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][5], q[S][0], a1, 0, 0, 0);            \
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][4], q[S][2], a2, 0, 0, 0);            \
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][5], q[S][2], a3, 0, 0, 0);            \
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][6], q[S][0], a2, 0, 0, 0);            \
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][7], q[S][0], a3, 0, 0, 0);            \
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][8], q[S][1], a0, 0, 0, 0);            \
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][9], q[S][1], a1, 0, 0, 0);            \
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][8], q[S][3], a2, 0, 0, 0);            \
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][9], q[S][3], a3, 0, 0, 0);            \
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][10], q[S][1], a2, 0, 0, 0);           \
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[S][11], q[S][1], a3, 0, 0, 0);           \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        if (BAR) __builtin_amdgcn_s_barrier();                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
    for (int it = 0; it < iters; it += 2) { BODY(0, 1) BODY(1, 0) }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NREADS, int OCC, bool BAR, int MODE = 0>
void run(const char* name) {
    const int grid = 256 * OCC, iters = 20000;
    float* out; hipMalloc(&out, grid * 256 * 4);
    hipFuncSetAttribute((const void*)k<NREADS, OCC, BAR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NREADS, OCC, BAR, MODE><<<grid, 256, 69632>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NREADS, OCC, BAR, MODE><<<grid, 256, 69632>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    const double lds = (double)grid * 4 * iters * NREADS * 1024.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s (%.0f%% of 2500)   LDS %.1f TB/s\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25, lds / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0, 1, false>("12 MFMA, 0 reads, 1 wave/SIMD");
    run<0, 2, false>("12 MFMA, 0 reads, 2 waves/SIMD");
    run<4, 2, false>("12 MFMA, 4 reads, 2 waves/SIMD");
    run<8, 2, false>("12 MFMA, 8 reads, 2 waves/SIMD");
    run<12, 2, false>("12 MFMA, 12 reads, 2 waves/SIMD");
    run<12, 2, true>("12 MFMA, 12 reads, 2 waves/SIMD + barrier");
    run<12, 1, false>("12 MFMA, 12 reads, 1 wave/SIMD");
    run<8, 1, false>("12 MFMA, 8 reads, 1 wave/SIMD");
    run<12, 2, false, 1>("kernel-like offsets + random data, no barrier");
    run<12, 2, true, 1>("kernel-like offsets + random data, barrier");
    run<0, 2, false, 1>("random data, 0 reads");
    // one workgroup per CU (one wave per SIMD), as in the lone-workgroup ablation of the column kernel
    run<12, 1, true>("12 MFMA, 12 reads, 1 wave/SIMD + barrier");
    run<12, 1, false, 1>("random data, kernel-like offsets, 1 wave/SIMD, no barrier");
    run<12, 1, true, 1>("random data, kernel-like offsets, 1 wave/SIMD, barrier");
    run<0, 1, true, 1>("random data, 0 reads, 1 wave/SIMD, barrier");
    run<0, 1, false, 1>("random data, 0 reads, 1 wave/SIMD, no barrier");
    return 0;
}
