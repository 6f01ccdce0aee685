// Feasibility probe for a Winograd F(2x2, 5x5) form of the 5x5 stages (point 4): can one wave per
// SIMD keep the f32 matrix pipe busy when EVERY v_mfma_f32_16x16x4_f32 needs a fresh A operand (a transformed
// input value, computed on the vector ALU from LDS data) and a fresh B operand (a transformed weight from LDS)?
// Per K-step (4 input channels) a wave: reads its 6x6 raw patch (36 ds_read_b32), transforms it (B^T d B,
// separable, ~100 VALU flops), and for each of the 36 positions issues 2 MFMAs (two halves of the 32 output
// channels), each with its own ds_read_b32 weight operand.  288 accumulator registers: one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_wino.hip -o exp/ubench_wino && exp/ubench_wino
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool TRANSFORM, bool BREADS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) smem[i] = 0.001f * (i & 31);
    __syncthreads();
    f32x4 acc[36][2];
#pragma unroll
    for (int p = 0; p < 36; ++p) { acc[p][0] = f32x4{0, 0, 0, 0}; acc[p][1] = f32x4{0, 0, 0, 0}; }
    const float* dbase = smem + wave * 2048 + lane;          // raw patch values of this lane's (tile, channel)
    const float* wbase = smem + 8192 + lane;                 // transformed weights
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            float d[36];
#pragma unroll
            for (int q = 0; q < 36; ++q) d[q] = dbase[q * 64 + (j & 1) * 8];
            float v[36];
            if (TRANSFORM) {
                // columns then rows with the F(2,5) input matrix pattern (adds / small constant multiplies)
                float t[36];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float a0 = d[c], a1 = d[6 + c], a2 = d[12 + c], a3 = d[18 + c], a4 = d[24 + c], a5 = d[30 + c];
                    t[c] = 4.f * a0 - 5.f * a2 + a4;
                    t[6 + c] = -4.f * (a1 + a2) + (a3 + a4);
                    t[12 + c] = 4.f * (a1 - a2) - (a3 - a4);
                    t[18 + c] = -2.f * (a1 - a3) - (a2 - a4);
                    t[24 + c] = 2.f * (a1 - a3) - (a2 - a4);
                    t[30 + c] = 4.f * a1 - 5.f * a3 + a5;
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const float a0 = t[6 * r], a1 = t[6 * r + 1], a2 = t[6 * r + 2], a3 = t[6 * r + 3], a4 = t[6 * r + 4], a5 = t[6 * r + 5];
                    v[6 * r] = 4.f * a0 - 5.f * a2 + a4;
                    v[6 * r + 1] = -4.f * (a1 + a2) + (a3 + a4);
                    v[6 * r + 2] = 4.f * (a1 - a2) - (a3 - a4);
                    v[6 * r + 3] = -2.f * (a1 - a3) - (a2 - a4);
                    v[6 * r + 4] = 2.f * (a1 - a3) - (a2 - a4);
                    v[6 * r + 5] = 4.f * a1 - 5.f * a3 + a5;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 36; ++q) v[q] = d[q];
            }
#pragma unroll
            for (int p = 0; p < 36; ++p) {
                const float b0 = BREADS ? wbase[(p * 2 + 0) * 64 + j * 4] : 0.5f;
                const float b1 = BREADS ? wbase[(p * 2 + 1) * 64 + j * 4] : 0.25f;
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[p], b0, acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[p], b1, acc[p][1], 0, 0, 0);
            }
        }
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 36; ++p) s += acc[p][0] + acc[p][1];
    out[blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
}

template <bool TRANSFORM, bool BREADS>
void run(const char* name) {
    const int grid = 256, iters = 2000;
    float* out; hipMalloc(&out, grid * 256 * 4);
    hipFuncSetAttribute((const void*)k<TRANSFORM, BREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<TRANSFORM, BREADS><<<grid, 256, 65536>>>(out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<TRANSFORM, BREADS><<<grid, 256, 65536>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 8 * 72;            // MFMAs issued
    const double tf = mfma * 2.0 * 16 * 16 * 4 / ms / 1e9;             // TFLOP/s in the Winograd domain
    printf("%-46s %8.3f ms  %6.1f TFLOP/s issued (%.0f%% of 157.3)  -> x2.78 = %.0f TFLOP/s direct-equivalent for a 5x5 layer\n",
           name, ms, tf, tf / 1.573, tf * 2.78);
    hipFree(out);
}

int main() {
    run<false, false>("MFMAs only (A from LDS raw, B constant)");
    run<false, true>("+ B operands from LDS");
    run<true, false>("+ input transform on the VALU (B constant)");
    run<true, true>("+ both (the real inner loop)");
    return 0;
}
