// Does v_cvt_pk_u8_f32 equal data_to_img's quantiser (reference main.rs:175: clamp(floor(x), 0, 255) of x = 255 v + 0.5)?
// The ISA text says "convert to 8-bit unsigned integer" without naming a rounding mode; this settles it on the device:
//   hipcc --offload-arch=gfx950 -O2 scripts/experiments/probe_cvt_pk_u8.hip -o /tmp/probe_cvt && /tmp/probe_cvt
// Answer (MI355X, round 3): NO -- it rounds to nearest even (0.5 -> 0, 0.5017 -> 1, 1.5 -> 2): 523 769 of 1 084 995 probe values
// differ.  The last kernel therefore feeds it floor(x), an integer, and keeps only its clamp + convert + byte placement.
// Prints the number of inputs on which the instruction and the floor-clamp form disagree (every float on a fine grid over
// [-4, 260], the neighbourhood of every integer and half-integer boundary, huge values, infinities, NaN).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void probe(const float* x, unsigned* a, unsigned* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, 0u);
    float q = floorf(x[i]);
    q = fminf(fmaxf(q, 0.0f), 255.0f);
    b[i] = (unsigned)q;
}

int main() {
    std::vector<float> x;
    for (int k = -4 * 4096; k <= 260 * 4096; ++k) x.push_back((float)k / 4096.0f);
    for (int k = -2; k <= 257; ++k)
        for (int half = 0; half < 2; ++half) {
            float v = (float)k + 0.5f * half;
            for (int s = -3; s <= 3; ++s) {
                float w = v;
                for (int t = 0; t < (s < 0 ? -s : s); ++t) w = std::nextafter(w, s < 0 ? -1e9f : 1e9f);
                x.push_back(w);
            }
        }
    const float extra[] = {1e9f, -1e9f, 3e38f, -3e38f, INFINITY, -INFINITY, NAN, -0.0f, 1e-40f, -1e-40f};
    for (float e : extra) x.push_back(e);
    const int n = (int)x.size();
    float* dx; unsigned *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(dx, da, db, n);
    std::vector<unsigned> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i] && !(std::isnan(x[i]))) { if (bad < 10) printf("x = %.9g: cvt_pk_u8 %u, floor-clamp %u\n", x[i], a[i], b[i]); ++bad; }
    for (int i = 0; i < n; ++i) if (std::isnan(x[i])) printf("NaN: cvt_pk_u8 %u, floor-clamp %u\n", a[i], b[i]);
    printf("%d inputs, %d disagree\n", n, bad);
    return bad != 0;
}
