/* abi_smoke.c -- the drop-in boundary driven from plain C99: what a maintainer's FFI stub does, with no
 * Python, C++ or torch in the process.  Built and run by tests/test_c_abi_smoke.py on a GPU box:
 *   gcc -std=c99 -Wall -Iinclude tests/c/abi_smoke.c -Lrusty_sr_amd -lsrhip -o abi_smoke
 *   abi_smoke PARAMS.rsr OUT.bin        (writes the RGBA8 result of a fixed synthetic 40x70 image for the test to compare)
 * Exercises: sr_rsr_decode, sr_create, sr_upscale_rgba8 (host pointers), sr_upscale_f32, the 1-rank communicator +
 * sr_upscale_sharded_* through device memory obtained from HIP's C API is NOT needed here (device pointers are the
 * caller's business); instead the host-memory multi-context forms are called: sr_upscale_rgba8_multi and
 * sr_upscale_rgba8_batch_multi with two contexts on device 0. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srhip.h"

#define CHECK(expr) do { int rc_ = (expr); if (rc_ != SR_OK) { fprintf(stderr, "%s -> %d (%s)\n", #expr, rc_, sr_strerror(rc_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: abi_smoke PARAMS.rsr OUT.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* blob = (uint8_t*)malloc((size_t)len);
    if (fread(blob, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    size_t n = 0;
    CHECK(sr_rsr_decode(blob, (size_t)len, NULL, 0, &n));                 /* main.rs:146 */
    if ((int)n != SR_NUM_PARAMS) { fprintf(stderr, "unexpected parameter count %zu\n", n); return 1; }
    float* params = (float*)malloc(n * sizeof(float));
    CHECK(sr_rsr_decode(blob, (size_t)len, params, n, &n));
    if (sr_rsr_decode(blob, (size_t)len - 1, NULL, 0, &n) != SR_E_BYTEVEC) { fprintf(stderr, "truncated blob accepted\n"); return 1; }

    sr_ctx *a = NULL, *b = NULL;
    if (sr_create(&a, params, n - 1, SR_FACTOR, 0) != SR_E_PARAM_COUNT) { fprintf(stderr, "wrong count accepted\n"); return 1; }  /* main.rs:162 */
    CHECK(sr_create(&a, params, n, SR_FACTOR, 0));
    CHECK(sr_create(&b, params, n, SR_FACTOR, 0));

    enum { H = 40, W = 70, N = 3 };
    uint8_t* px = (uint8_t*)malloc((size_t)N * H * W * 3);
    uint32_t z = 2463534242u;
    for (size_t i = 0; i < (size_t)N * H * W * 3; ++i) { z ^= z << 13; z ^= z >> 17; z ^= z << 5; px[i] = (uint8_t)(z >> 24); }
    const size_t out_img = (size_t)9 * H * W * 4;
    uint8_t* out = (uint8_t*)malloc(N * out_img);
    uint8_t* out2 = (uint8_t*)malloc(N * out_img);
    CHECK(sr_upscale_rgba8(a, px, 3, N, H, W, out));                      /* img_to_data + graph.forward + data_to_img, main.rs:170-175 */
    for (size_t i = 3; i < N * out_img; i += 4) if (out[i] != 255) { fprintf(stderr, "alpha != 255\n"); return 1; }

    sr_ctx* both[2];
    both[0] = a; both[1] = b;
    CHECK(sr_upscale_rgba8_batch_multi(both, 2, px, 3, N, H, W, out2));   /* image i -> context i mod 2 */
    if (memcmp(out, out2, N * out_img)) { fprintf(stderr, "batch_multi differs from the single-context batch\n"); return 1; }
    CHECK(sr_upscale_rgba8_multi(both, 2, px, 3, H, W, out2));            /* one image, two row shares */
    if (memcmp(out, out2, out_img)) { fprintf(stderr, "multi differs from the single-context call\n"); return 1; }
    both[1] = a;
    if (sr_upscale_rgba8_multi(both, 2, px, 3, H, W, out2) != SR_E_INVALID) { fprintf(stderr, "duplicate context accepted\n"); return 1; }

    /* f32 seam (graph.forward proper): quantising it must give the fused u8 result */
    float* x = (float*)malloc((size_t)H * W * 3 * sizeof(float));
    float* y = (float*)malloc((size_t)9 * H * W * 3 * sizeof(float));
    for (size_t i = 0; i < (size_t)H * W * 3; ++i) x[i] = (float)px[i] / 255.0f;
    CHECK(sr_upscale_f32(a, x, 1, H, W, y));
    size_t bad = 0;
    for (size_t p = 0; p < (size_t)9 * H * W; ++p)
        for (int c = 0; c < 3; ++c) {
            float q = 255.0f * y[p * 3 + c] + 0.5f;
            int v = q < 0 ? 0 : q > 255 ? 255 : (int)q;
            if (v != out[p * 4 + c]) ++bad;
        }
    if (bad) { fprintf(stderr, "u8 path != quantise(f32 path) at %zu samples\n", bad); return 1; }

    int rank = -1, nranks = -1;
    CHECK(sr_comm_init_rank(a, NULL, 0, 0, 1));                           /* a 1-rank communicator needs no RCCL object */
    CHECK(sr_comm_rank(a, &rank, &nranks));
    if (rank != 0 || nranks != 1) { fprintf(stderr, "comm rank %d of %d\n", rank, nranks); return 1; }

    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 1, out_img, f) != out_img) { perror(argv[2]); return 2; }
    fclose(f);
    sr_destroy(a);
    sr_destroy(b);
    printf("abi_smoke ok: %d x %dx%d -> %dx%d RGBA8\n", N, W, H, 3 * W, 3 * H);
    return 0;
}
