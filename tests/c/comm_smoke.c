/* comm_smoke.c -- one image over every visible GPU from ONE plain-C process: no torch, no Python, no MPI.
 *   gcc -std=c99 -D_POSIX_C_SOURCE=199309L -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/comm_smoke.c -Lrusty_sr_amd -lsrhip \
 *       -L/opt/rocm/lib -lamdhip64 -o comm_smoke
 *   comm_smoke PARAMS.rsr [max_devices]
 * What it checks (SURVEY.md 8(b) "one RCCL communicator inside the context", 8(e) config C; the call being sharded is
 * graph.forward, reference main.rs:171):
 *   1. sr_create on every visible device (at most max_devices);
 *   2. the single-device result of a 3840-wide image = the truth;
 *   3. sr_comm_init_all (ncclCommInitAll inside libsrhip) + sr_upscale_sharded_rgba8_all over all devices, bands of
 *      UNEQUAL height, must equal (2) bit for bit, twice in a row (the second call meets warm buffers);
 *   4. the same through sr_comm_init_local (halos by hipMemcpyPeerAsync over xGMI);
 *   5. the calling thread's current HIP device is what it was before every call (the library restores it);
 *   6. error paths leave the set usable: duplicate devices refused by sr_comm_init_all, a band thinner than SR_HALO
 *      refused with SR_E_HALO, and after either the sharded call still works.
 *   7. NUMBERS, whatever lease first sees several devices: config C (3840x2160 over all devices, equal bands) five times per
 *      transport with per-stage profiling OFF (the bands run concurrently, as in production); per rank the device time of its
 *      whole step and of its halo exchange, from the two event pairs libsrhip records on the band's stream (sr_last_timing /
 *      sr_last_comm_ms), and the wall time of the call.
 * With one visible device the multi-device steps cannot run: RCCL admits one rank per device.  The program then runs
 * (4) with two contexts on device 0 (peer copy within one device), prints "skipped: 1 device" for (3) and exits 0. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "srhip.h"
#include "srhip_experimental.h" /* the "halo" switch: per-layer feature halos */

#define MAXDEV 16
#define CHECK(expr) do { int rc_ = (expr); if (rc_ != SR_OK) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #expr, rc_, sr_strerror(rc_)); return 1; } } while (0)
#define HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return 1; } } while (0)

enum { H = 211, W = 3840 };  /* config C's width; rows chosen so that the bands come out unequal */

static int current_device(void) { int d = -1; (void)hipGetDevice(&d); return d; }

/* cut H rows into n bands, deliberately unequal (each at least SR_HALO rows) */
static void cut(int n, int* h_band) {
    int left = H, k;
    for (k = 0; k < n; ++k) {
        int rows = k == n - 1 ? left : H / n + ((k & 1) ? -3 : 5);
        if (rows < SR_HALO) rows = SR_HALO;
        if (left - rows < SR_HALO * (n - 1 - k)) rows = left - SR_HALO * (n - 1 - k);
        h_band[k] = rows;
        left -= rows;
    }
}

/* bands of `px` up to the devices of ctxs, sharded call, bands back, compare with `want` */
static int sharded_equals(sr_ctx** ctxs, const int* dev, int n, const uint8_t* px, const uint8_t* want, const char* what) {
    int h_band[MAXDEV], k, y = 0, rc = 0;
    const uint8_t* d_in[MAXDEV];
    uint8_t* d_out[MAXDEV];
    uint8_t* got = (uint8_t*)malloc((size_t)9 * H * W * 4);
    const int home = current_device();
    cut(n, h_band);
    for (k = 0; k < n; ++k) {
        void *pi = NULL, *po = NULL;
        HIP(hipSetDevice(dev[k]));
        HIP(hipMalloc(&pi, (size_t)h_band[k] * W * 3));
        HIP(hipMalloc(&po, (size_t)9 * h_band[k] * W * 4));
        HIP(hipMemcpy(pi, px + (size_t)y * W * 3, (size_t)h_band[k] * W * 3, hipMemcpyHostToDevice));
        d_in[k] = (const uint8_t*)pi; d_out[k] = (uint8_t*)po;
        y += h_band[k];
    }
    HIP(hipSetDevice(home));
    for (int rep = 0; rep < 2 && !rc; ++rep) {
        CHECK(sr_upscale_sharded_rgba8_all(ctxs, n, d_in, 3, h_band, W, d_out));
        if (current_device() != home) { fprintf(stderr, "%s: the call left the thread on device %d (was %d)\n", what, current_device(), home); rc = 1; }
        for (k = 0, y = 0; k < n; ++k) {
            HIP(hipMemcpy(got + (size_t)9 * y * W * 4, d_out[k], (size_t)9 * h_band[k] * W * 4, hipMemcpyDeviceToHost));
            y += h_band[k];
        }
        if (memcmp(got, want, (size_t)9 * H * W * 4)) { fprintf(stderr, "%s: sharded over %d contexts differs from the single-device call (pass %d)\n", what, n, rep); rc = 1; }
    }
    for (k = 0; k < n; ++k) { (void)hipFree((void*)d_in[k]); (void)hipFree(d_out[k]); }
    free(got);
    if (!rc) printf("  %s over %d contexts: bit-identical (bands", what, n);
    if (!rc) { for (k = 0; k < n; ++k) printf(" %d", h_band[k]); printf(")\n"); }
    return rc;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static int cmp_double(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

/* config C on the devices of ctxs: 3840x2160 in n equal bands, timed; prints one line per rank (medians of 5 calls) */
static int timed_config_c(sr_ctx** ctxs, const int* dev, int n, const char* what) {
    enum { HC = 2160, WC = 3840, REPS = 5 };
    int h_band[MAXDEV], k, rep;
    const uint8_t* d_in[MAXDEV];
    uint8_t* d_out[MAXDEV];
    double band[MAXDEV][REPS], xchg[MAXDEV][REPS], expo[MAXDEV][REPS], wall[REPS];
    const int home = current_device();
    for (k = 0; k < n; ++k) h_band[k] = HC * (k + 1) / n - HC * k / n;
    for (k = 0; k < n; ++k) {
        void *pi = NULL, *po = NULL;
        HIP(hipSetDevice(dev[k]));
        HIP(hipMalloc(&pi, (size_t)h_band[k] * WC * 3));
        HIP(hipMalloc(&po, (size_t)9 * h_band[k] * WC * 4));
        HIP(hipMemset(pi, 0x5a, (size_t)h_band[k] * WC * 3));
        d_in[k] = (const uint8_t*)pi; d_out[k] = (uint8_t*)po;
    }
    HIP(hipSetDevice(home));
    for (rep = -2; rep < REPS; ++rep) {  /* two warm-up calls: workspaces, clocks, RCCL's lazy connections */
        const double t0 = now_ms();
        CHECK(sr_upscale_sharded_rgba8_all(ctxs, n, d_in, 3, h_band, WC, d_out));
        if (rep < 0) continue;
        wall[rep] = now_ms() - t0;
        for (k = 0; k < n; ++k) {
            double tot = 0, cm = 0, ex = 0;
            CHECK(sr_last_timing(ctxs[k], &tot, NULL, NULL, NULL));
            CHECK(sr_last_comm_ms(ctxs[k], &cm));
            CHECK(sr_last_comm_exposed_ms(ctxs[k], &ex));
            band[k][rep] = tot; xchg[k][rep] = cm; expo[k][rep] = ex;
        }
    }
    qsort(wall, REPS, sizeof(double), cmp_double);
    printf("  config C, %dx%d over %d contexts, %s: wall %.3f ms per call (median of %d) = %.0f output MP/s\n", WC, HC, n, what, wall[REPS / 2], REPS,
           9.0 * HC * WC / 1e6 / (wall[REPS / 2] / 1e3));
    for (k = 0; k < n; ++k) {
        qsort(band[k], REPS, sizeof(double), cmp_double);
        qsort(xchg[k], REPS, sizeof(double), cmp_double);
        qsort(expo[k], REPS, sizeof(double), cmp_double);
        printf("    rank %d (device %d, %d rows): step %.3f ms; halo exchange %.3f ms on its own stream, of which the band's stream waited %.3f ms\n", k, dev[k], h_band[k],
               band[k][REPS / 2], xchg[k][REPS / 2], expo[k][REPS / 2]);
    }
    for (k = 0; k < n; ++k) { (void)hipFree((void*)d_in[k]); (void)hipFree(d_out[k]); }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: comm_smoke PARAMS.rsr [max_devices]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* blob = (uint8_t*)malloc((size_t)len);
    if (fread(blob, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    size_t n = 0;
    CHECK(sr_rsr_decode(blob, (size_t)len, NULL, 0, &n));
    float* params = (float*)malloc(n * sizeof(float));
    CHECK(sr_rsr_decode(blob, (size_t)len, params, n, &n));

    int ndev = 0, k;
    HIP(hipGetDeviceCount(&ndev));
    if (ndev > MAXDEV) ndev = MAXDEV;
    if (argc > 2 && atoi(argv[2]) > 0 && atoi(argv[2]) < ndev) ndev = atoi(argv[2]);
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    printf("comm_smoke: %d device(s), image %dx%d\n", ndev, W, H);

    uint8_t* px = (uint8_t*)malloc((size_t)H * W * 3);
    uint32_t z = 2463534242u;
    for (size_t i = 0; i < (size_t)H * W * 3; ++i) { z ^= z << 13; z ^= z >> 17; z ^= z << 5; px[i] = (uint8_t)(z >> 24); }
    uint8_t* want = (uint8_t*)malloc((size_t)9 * H * W * 4);

    sr_ctx* ctxs[MAXDEV];
    int dev[MAXDEV];
    HIP(hipSetDevice(0));
    for (k = 0; k < ndev; ++k) {
        dev[k] = k;
        CHECK(sr_create(&ctxs[k], params, n, SR_FACTOR, k));
        if (current_device() != 0) { fprintf(stderr, "sr_create(device %d) left the thread on device %d\n", k, current_device()); return 1; }
    }
    CHECK(sr_upscale_rgba8(ctxs[0], px, 3, 1, H, W, want));              /* the single-device truth */
    for (k = 1; k < ndev; ++k) {                                          /* every device computes the same image */
        uint8_t* other = (uint8_t*)malloc((size_t)9 * H * W * 4);
        CHECK(sr_upscale_rgba8(ctxs[k], px, 3, 1, H, W, other));
        if (memcmp(other, want, (size_t)9 * H * W * 4)) { fprintf(stderr, "device %d differs from device 0\n", k); return 1; }
        free(other);
    }

    if (ndev >= 2) {
        if (!sr_comm_available()) { fprintf(stderr, "librccl not loadable\n"); return 1; }
        /* error paths first: they must leave every context released and usable */
        sr_ctx* dup[2];
        sr_ctx* twin = NULL;
        CHECK(sr_create(&twin, params, n, SR_FACTOR, 0));
        dup[0] = ctxs[0]; dup[1] = twin;                                  /* two contexts on ONE device: RCCL admits one rank per device */
        if (sr_comm_init_all(dup, 2) != SR_E_INVALID) { fprintf(stderr, "two ranks on one device accepted\n"); return 1; }
        sr_destroy(twin);
        CHECK(sr_comm_init_all(ctxs, ndev));
        if (current_device() != 0) { fprintf(stderr, "sr_comm_init_all left the thread on device %d\n", current_device()); return 1; }
        for (k = 0; k < ndev; ++k) {
            int r = -1, nr = -1;
            CHECK(sr_comm_rank(ctxs[k], &r, &nr));
            if (r != k || nr != ndev) { fprintf(stderr, "context %d is rank %d of %d\n", k, r, nr); return 1; }
        }
        {   /* a band thinner than the halo its neighbour needs is refused before anything is queued */
            int h_bad[MAXDEV];
            const uint8_t* din[MAXDEV];
            uint8_t* dout[MAXDEV];
            for (k = 0; k < ndev; ++k) { h_bad[k] = SR_HALO - 1; din[k] = (const uint8_t*)want; dout[k] = want; }
            if (sr_upscale_sharded_rgba8_all(ctxs, ndev, din, 3, h_bad, W, dout) != SR_E_HALO) { fprintf(stderr, "thin band accepted\n"); return 1; }
        }
        if (sharded_equals(ctxs, dev, ndev, px, want, "RCCL (sr_comm_init_all)")) return 1;
        CHECK(sr_comm_init_local(ctxs, ndev));
        if (sharded_equals(ctxs, dev, ndev, px, want, "peer copy (sr_comm_init_local)")) return 1;
        CHECK(sr_comm_init_all(ctxs, ndev));                              /* and back: a set may change transport */
        if (sharded_equals(ctxs, dev, ndev, px, want, "RCCL again")) return 1;
        if (timed_config_c(ctxs, dev, ndev, "RCCL")) return 1;
        /* feature rows after every stage instead of the recomputed overlap -- same bits, both transports */
        for (k = 0; k < ndev; ++k) CHECK(sr_set_experiment(ctxs[k], "halo", "layers"));
        if (sharded_equals(ctxs, dev, ndev, px, want, "RCCL, per-layer feature halos")) return 1;
        if (timed_config_c(ctxs, dev, ndev, "RCCL, per-layer feature halos")) return 1;
        CHECK(sr_comm_init_local(ctxs, ndev));
        if (sharded_equals(ctxs, dev, ndev, px, want, "peer copy, per-layer feature halos")) return 1;
        if (timed_config_c(ctxs, dev, ndev, "peer copy, per-layer feature halos")) return 1;
        for (k = 0; k < ndev; ++k) CHECK(sr_set_experiment(ctxs[k], "halo", "input"));
        if (timed_config_c(ctxs, dev, ndev, "peer copy")) return 1;
    } else {
        printf("  RCCL over several devices: skipped: 1 device\n");
        sr_ctx* two[2];
        int dev0[2] = {0, 0};
        two[0] = ctxs[0];
        CHECK(sr_create(&two[1], params, n, SR_FACTOR, 0));
        CHECK(sr_comm_init_local(two, 2));
        if (sharded_equals(two, dev0, 2, px, want, "peer copy, two contexts of device 0")) return 1;
        if (timed_config_c(two, dev0, 2, "peer copy, both contexts on device 0 (a rehearsal, not a measurement of two GPUs)")) return 1;
        CHECK(sr_set_experiment(two[0], "halo", "layers"));
        CHECK(sr_set_experiment(two[1], "halo", "layers"));
        if (sharded_equals(two, dev0, 2, px, want, "peer copy, per-layer feature halos, two contexts of device 0")) return 1;
        if (timed_config_c(two, dev0, 2, "peer copy, per-layer feature halos, both contexts on device 0 (a rehearsal)")) return 1;
        CHECK(sr_set_experiment(two[0], "halo", "input"));
        sr_destroy(two[1]);
        CHECK(sr_comm_init_rank(ctxs[0], NULL, 0, 0, 1));
    }
    for (k = 0; k < ndev; ++k) sr_destroy(ctxs[k]);
    if (current_device() != 0) { fprintf(stderr, "sr_destroy left the thread on device %d\n", current_device()); return 1; }
    printf("comm_smoke ok\n");
    return 0;
}
