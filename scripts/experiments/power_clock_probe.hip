// Round 6: board power and shader clock under the REAL kernels, long enough for the sensor (review, round 5: the 3-second probe of
// profiles/r5_power_probe.txt showed 314-325 W where the micro-benchmark of the same loop showed 1.2-1.38 kW -- "not evidence either way").
//
// Through the C ABI (libsrhip.so, no torch): the 1080p frame loops on one stream for WARM + SECONDS seconds per arithmetic mode, in
// fenced bursts like bench.py's `sustained`.  Meanwhile, every 100 ms, on a second stream a ONE-WAVE kernel reads the shader clock
// counter and the 100 MHz wall counter 20 us apart (clock64 / wall_clock64: s_memtime / s_memrealtime) -- the clock the chip actually
// runs at while the frame's kernels hold it, not the driver's level table -- and a host thread reads the board's hwmon power every
// 20 ms.  Printed: a row per second (watts: mean of the second's samples; GHz: mean of its probes; ms per frame), then per mode the
// means over the measured window.  A third leg runs the same schedule on a stream of random-operand v_mfma_f32_16x16x32_f16 (what
// stages 1-3 of the split-half mode issue) in 5 ms launches: the matrix-only rate under the same budget, the denominator bench.py quotes
// beside the nominal 2 500 TFLOP/s.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/power_clock_probe.hip -I include -L rusty_sr_amd -lsrhip \
//         -Wl,-rpath,$PWD/rusty_sr_amd -o /tmp/pcp && /tmp/pcp rusty_sr_amd/res/imagenet.rsr 20 10
#include <hip/hip_runtime.h>
#include <glob.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "srhip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define SR(x) do { int s_ = (x); if (s_ != SR_OK) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, sr_strerror(s_)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void clock_probe(unsigned long long* out) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000) r1 = wall_clock64();  // 20 us at 100 MHz
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// the matrix pipe alone on random operands: 24 x 16x16x32 per iteration and wave, 2 waves per SIMD
__global__ __launch_bounds__(256, 2) void mfma16_stream(float* out, int iters) {
    uint32_t seed = blockIdx.x * 256u + threadIdx.x;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return 0x3c003c00u ^ (seed & 0x83ff83ffu); };  // halves in +-[1, 2)
    f16x8 a[4], b[4];
    for (int k = 0; k < 4; ++k) {
        uint32_t wa[4] = {rnd(), rnd(), rnd(), rnd()}, wb[4] = {rnd(), rnd(), rnd(), rnd()};
        a[k] = __builtin_bit_cast(f16x8, wa); b[k] = __builtin_bit_cast(f16x8, wb);
    }
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + r) & 3], b[k & 3], acc[k], 0, 0, 0);
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static std::string find(const char* const* pats) {
    for (; *pats; ++pats) {
        glob_t g;
        if (glob(*pats, 0, nullptr, &g) == 0 && g.gl_pathc > 0) { std::string p = g.gl_pathv[0]; globfree(&g); return p; }
    }
    return "";
}
static double read_number(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

struct Sample { double t, watts, ghz; };

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: power_clock_probe PARAMS.rsr [seconds 20] [warm-up seconds 10]\n"); return 2; }
    const double secs = argc > 2 ? atof(argv[2]) : 20, warm = argc > 3 ? atof(argv[3]) : 10;
    const char* pw[] = {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", nullptr};
    const std::string power_path = find(pw);
    printf("power: %s\n", power_path.empty() ? "(none)" : power_path.c_str());
    // parameters
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> blob(1 << 21);
    blob.resize(fread(blob.data(), 1, blob.size(), f));
    fclose(f);
    size_t n = 0;
    SR(sr_rsr_decode(blob.data(), blob.size(), nullptr, 0, &n));
    std::vector<float> params(n);
    SR(sr_rsr_decode(blob.data(), blob.size(), params.data(), n, &n));
    const int H = 1080, W = 1920;
    uint8_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, (size_t)H * W * 3));
    CHECK(hipMalloc(&d_out, (size_t)9 * H * W * 4));
    {   // a smooth image (activations in the trained range): a low-frequency pattern + a little noise
        std::vector<uint8_t> px((size_t)H * W * 3);
        unsigned s = 7;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) { s = s * 1664525u + 1013904223u; px[((size_t)y * W + x) * 3 + c] = (uint8_t)(128 + 90 * __builtin_sinf(0.013f * x + 0.021f * y + c) + (s >> 28)); }
        CHECK(hipMemcpy(d_in, px.data(), px.size(), hipMemcpyHostToDevice));
    }
    hipStream_t work, side;
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    unsigned long long* d_probe;
    CHECK(hipHostMalloc((void**)&d_probe, 64, hipHostMallocMapped));
    float* d_junk;
    CHECK(hipMalloc(&d_junk, (size_t)512 * 256 * 4));

    for (int leg = 0; leg < 3; ++leg) {
        const char* name = leg == 0 ? "f32 (exact mode), 1080p frames" : leg == 1 ? "split_f16, 1080p frames" : "v_mfma_f32_16x16x32_f16 stream, random operands";
        sr_ctx* ctx = nullptr;
        if (leg < 2) {
            SR(sr_create(&ctx, params.data(), n, 3, 0));
            SR(sr_set_precision(ctx, leg == 0 ? SR_PRECISION_F32 : SR_PRECISION_SPLIT_F16));
            SR(sr_upscale_rgba8_dev(ctx, d_in, 3, 1, H, W, d_out, work));
            CHECK(hipStreamSynchronize(work));
        }
        int iters = 2000;
        if (leg == 2) {  // calibrate a launch to ~5 ms
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, work));
            mfma16_stream<<<512, 256, 0, work>>>(d_junk, iters);
            CHECK(hipEventRecord(e1, work));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            iters = (int)(iters * 5.0 / ms);
        }
        std::atomic<bool> run{true};
        std::vector<Sample> samples;
        const auto t_start = std::chrono::steady_clock::now();
        auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
        std::thread mon([&] {
            int k = 0;
            while (run) {
                Sample s{now(), power_path.empty() ? -1 : read_number(power_path) * 1e-6, -1};
                if (k++ % 5 == 0) {  // every 100 ms: the clock probe
                    clock_probe<<<1, 64, 0, side>>>(d_probe);
                    if (hipStreamSynchronize(side) == hipSuccess && d_probe[1] > 0) s.ghz = (double)d_probe[0] / (double)d_probe[1] * 0.1;
                }
                samples.push_back(s);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
        std::vector<std::pair<double, long>> bursts;  // (end time, units so far)
        long units = 0;
        while (now() < warm + secs) {
            if (leg < 2) {
                for (int k = 0; k < 25; ++k) SR(sr_upscale_rgba8_dev(ctx, d_in, 3, 1, H, W, d_out, work));
                units += 25;
            } else {
                for (int k = 0; k < 20; ++k) mfma16_stream<<<512, 256, 0, work>>>(d_junk, iters);
                units += 20;
            }
            CHECK(hipStreamSynchronize(work));
            bursts.push_back({now(), units});
        }
        run = false;
        mon.join();
        printf("== %s: %.0f s warm-up + %.0f s ==\n", name, warm, secs);
        double sw = 0, sg = 0; int nw = 0, ng = 0;
        for (int sec = 0; sec < (int)(warm + secs); ++sec) {
            double w = 0, g = 0; int cw = 0, cg = 0;
            for (const Sample& s : samples)
                if (s.t >= sec && s.t < sec + 1) { if (s.watts >= 0) { w += s.watts; ++cw; } if (s.ghz > 0) { g += s.ghz; ++cg; } }
            long u0 = 0, u1 = 0; double t0 = 0, t1 = 0;
            for (const auto& b : bursts) { if (b.first < sec) { u0 = b.second; t0 = b.first; } if (b.first < sec + 1) { u1 = b.second; t1 = b.first; } }
            const double per = u1 > u0 ? (t1 - t0) / (u1 - u0) * 1e3 : 0;
            if (sec % 2 == 0 || sec >= (int)(warm + secs) - 1)
                printf("  t %2d s  %7.1f W  %5.3f GHz  %s %.4f ms\n", sec, cw ? w / cw : -1, cg ? g / cg : -1, leg < 2 ? "frame" : "launch", per);
            if (sec >= warm) { if (cw) { sw += w / cw; ++nw; } if (cg) { sg += g / cg; ++ng; } }
        }
        long u0 = 0, u1 = 0; double t0 = 0, t1 = 0;
        for (const auto& b : bursts) { if (b.first < warm) { u0 = b.second; t0 = b.first; } u1 = b.second; t1 = b.first; }
        const double per_ms = (t1 - t0) / (u1 - u0) * 1e3;
        if (leg < 2) printf("  measured window: %.1f W, %.3f GHz, %.4f ms per frame (wall, bursts of 25)\n", nw ? sw / nw : -1, ng ? sg / ng : -1, per_ms);
        else {
            const double flops = 512.0 * 4 * iters * 24 * 2.0 * 16 * 16 * 32;
            printf("  measured window: %.1f W, %.3f GHz, %.1f TFLOP/s (launches of %.2f ms)\n", nw ? sw / nw : -1, ng ? sg / ng : -1, flops / (per_ms * 1e-3) * 1e-12, per_ms);
        }
        fflush(stdout);
        if (ctx) sr_destroy(ctx);
    }
    return 0;
}
