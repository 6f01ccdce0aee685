// Does v_mfma_f32_32x32x16_f16 flush subnormal half inputs on gfx950 (default kernel mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
    f16x8 a, b; f32x16 c;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)aval; b[e] = (_Float16)bval; }
    for (int r = 0; r < 16; ++r) c[r] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    for (float av : {1.0f, 3.0517578e-05f /*2^-15 subnormal*/, 5.9604645e-08f /*2^-24 smallest*/}) {
        k<<<1, 64>>>(d, av, 1024.0f);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.9g (as half %.9g), b = 1024, K = 16: mfma = %.9g, expected %.9g\n", av, h[1], h[0], 16.0 * h[1] * 1024.0);
    }
    return 0;
}
