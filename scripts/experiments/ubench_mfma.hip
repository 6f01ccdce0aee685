// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this box under the
// structural ingredients of the conv kernel (LDS operand reads, per-tap barrier,
// LDS-DMA)?  hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, const float* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) ((float*)smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0; a1[r] = 0; }
    f32x4 av0 = {1.f, 2.f, 3.f, 4.f}, av1 = av0, b = {0.5f, 0.25f, 0.125f, 1.f};
    const char* base = smem + lane * 16;
    float junk[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    long clk0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + ((it & 15) * 256 + tid) * 4),
                                             (__attribute__((address_space(3))) void*)(smem + 32768 + (it & 1) * 4096 + wave * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE >= 1) {
                av0 = *(const f32x4*)(base + ((it + g) & 7) * 2048);
                av1 = *(const f32x4*)(base + ((it + g) & 7) * 2048 + 1024);
                b = *(const f32x4*)(base + 16384 + g * 1024);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], b[q], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], b[q], a1, 0, 0, 0);
            }
        }
        if (MODE == 8 || MODE == 9) {   // 64 (MODE 8) / 128 (MODE 9) independent VALU FMAs per 32 MFMAs
#pragma unroll
            for (int u = 0; u < (MODE == 8 ? 64 : 128); ++u)
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(junk[u & 7]) : "v"(b.x));
        }
        if (MODE == 2 || MODE == 3) __syncthreads();
        if (MODE == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }          // DMA never waited for (racy; cost of the DMA alone)
        if (MODE == 5 && (it % 5) == 4) __syncthreads();                                                             // one barrier per 5 taps
        if (MODE == 6) { asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }  // DMA 2 taps ahead, counted wait
        if (MODE == 7) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }  // like 3 but hand-written
    }
    long clk1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    for (int r = 0; r < 8; ++r) s += junk[r];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) ((long*)out)[gridDim.x * 128] = clk1 - clk0;
}

template <int MODE, int OCC>
void run(const char* name, int grid, int iters) {
    float *out, *w;
    hipMalloc(&out, (grid * 256 + 16) * 4);
    hipMalloc(&w, 16 * 1024 * 4);
    hipMemset(w, 0, 16 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const size_t lds = OCC == 1 ? 65536 : (OCC == 2 ? 65536 : 49152);
    k<MODE, OCC><<<grid, 256, lds>>>(out, w, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, OCC><<<grid, 256, lds>>>(out, w, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long cyc; hipMemcpy(&cyc, (char*)out + (size_t)grid * 128 * 8, 8, hipMemcpyDeviceToHost);
    double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("%-46s grid=%5d  %8.3f ms  %7.1f TFLOP/s  (wave0 cycles/iter %.0f, readcyclecounter ticks; %0.f MHz if tick=shader clk)\n",
           name, grid, ms, flops / ms / 1e9, (double)cyc / iters, (double)cyc / (ms * 1e3));
    hipFree(out); hipFree(w);
}

int main() {
    const int it = 4000;
    run<0, 1>("pure MFMA, 1 WG/CU (1 wave/SIMD)", 256, it);
    run<0, 2>("pure MFMA, 2 WG/CU", 512, it);
    run<1, 1>("+LDS operand reads, 1 WG/CU", 256, it);
    run<1, 2>("+LDS operand reads, 2 WG/CU", 512, it);
    run<2, 1>("+barrier per 32 MFMA, 1 WG/CU", 256, it);
    run<2, 2>("+barrier per 32 MFMA, 2 WG/CU", 512, it);
    run<3, 2>("+LDS-DMA 4KB per 32 MFMA, 2 WG/CU", 512, it);
    run<3, 3>("+LDS-DMA 4KB per 32 MFMA, 3 WG/CU", 768, it);
    run<8, 1>("pure MFMA + 64 VALU fma / 32 MFMA, 1 WG/CU", 256, it);
    run<8, 2>("pure MFMA + 64 VALU fma / 32 MFMA, 2 WG/CU", 512, it);
    run<9, 2>("pure MFMA + 128 VALU fma / 32 MFMA, 2 WG/CU", 512, it);
    run<4, 2>("LDS-DMA, raw barrier, no vmcnt wait, 2 WG/CU", 512, it);
    run<5, 2>("LDS-DMA, barrier per 5 taps, 2 WG/CU", 512, it);
    run<6, 2>("LDS-DMA, vmcnt(1)+raw barrier, 2 WG/CU", 512, it);
    run<7, 2>("LDS-DMA, vmcnt(0)+raw barrier, 2 WG/CU", 512, it);
    run<6, 1>("LDS-DMA, vmcnt(1)+raw barrier, 1 WG/CU", 256, it);
    run<7, 1>("LDS-DMA, vmcnt(0)+raw barrier, 1 WG/CU", 256, it);
    return 0;
}
