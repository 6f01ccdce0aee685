// Register layout of v_mfma_f32_16x16x32_f16 on gfx950, checked against a host product (round 5: the split-half stage loop is being
// moved to this shape).  Assumed: A lane l = row (l & 15), K 8 (l >> 4) .. + 7; B lane l = column (l & 15), K 8 (l >> 4) .. + 7;
// D lane l = column (l & 15), rows 4 (l >> 4) + r (r = 0..3).   hipcc --offload-arch=gfx950 -O2 probe_mfma16.hip -o probe_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {  // A[16][32], B[32][16], D[16][16]
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = (_Float16)B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
}
int main() {
    float hA[512], hB[512], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma_f32_16x16x32_f16 layout: %s (%d of 256 differ)\n", bad ? "NOT as assumed" : "as assumed", bad);
    return bad != 0;
}
