// Per-workgroup phase timeline of one conv stage kernel on a 1920x1080 map.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rusty_sr_amd/csrc scripts/timeline.hip -o exp/timeline
#define SR_TIMELINE 1
#include "../rusty_sr_amd/csrc/sr_kernels.hip"
#include <algorithm>
#include <cstring>
#include <map>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int stage = argc > 1 ? atoi(argv[1]) : 1, th = argc > 2 ? atoi(argv[2]) : 8, prec = argc > 3 ? atoi(argv[3]) : 0;
    const bool pipe = argc > 4 && !strcmp(argv[4], "pipe");  // the pipe form (8-row tiles, persistent)
    const int H = 1080, W = 1920;
    const int pitch = W + 4; const long img_stride = (long)(H + 14) * pitch;
    const size_t npx = (size_t)img_stride + 2 * pitch + 64;
    float *f[3], *dst, *w, *bias; void* out;
    for (auto& p : f) { hipMalloc(&p, npx * 128); hipMemset(p, 0, npx * 128); }
    hipMalloc(&dst, npx * 128); hipMalloc(&out, (size_t)H * W * 9 * 12);
    const size_t org = ((size_t)2 * pitch + 2) * 32;
    hipMalloc(&w, 64 * 4096); hipMemset(w, 0, 64 * 4096);
    hipMalloc(&bias, 256); hipMemset(bias, 0, 256);
    StageArgs a{};
    a.src[0] = f[0] + org; a.src[1] = f[1] + org; a.src[2] = f[2] + org; a.wpack = w; a.bias = bias; a.beta = bias; a.dst = dst + org;
    a.pitch = pitch; a.img_stride = img_stride;
    a.img = f[0]; a.out = out; a.H = H; a.W = W; a.img_ch = 3; a.y_begin = 0; a.y_end = H;
    a.tiles_x = W / 32; a.tiles_y = (H + th - 1) / th;
    a.div_tpi = make_tile_div(a.tiles_x * a.tiles_y); a.div_tx = make_tile_div(a.tiles_x);
    int nblk = a.tiles_x * a.tiles_y; a.n_img = 1; int* dq; hipMalloc(&dq, 64); hipMemset(dq, 0, 64); a.queue = dq;
    long long* tl; hipMalloc(&tl, (size_t)nblk * 128);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    if (prec == 1 || pipe || getenv("TL_PERSIST")) nblk = 512;  // persistent form: co-resident workgroups pull tiles from the queue
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemset(dq, 0, 64);
        hipEventRecord(e0, 0);
        if (pipe) sr_launch_stage_pipe(stage, 3, a, prec, nblk, false, false, 0);
        else sr_launch_stage(stage, 3, a, th, prec, nblk, false, false, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 1 && ms < best) best = ms;
    }
    hipDeviceSynchronize();
    printf("kernel time (best of 5, instrumented build, all-zero operands): %.1f us\n", best * 1e3f);
    std::vector<long long> h((size_t)nblk * 16);
    hipMemcpy(h.data(), tl, (size_t)nblk * 128, hipMemcpyDeviceToHost);
    auto stat = [&](const char* name, int k0, int k1) {
        std::vector<long long> d;
        for (int b = 0; b < nblk; ++b) d.push_back(h[b * 16 + k1] - h[b * 16 + k0]);
        std::sort(d.begin(), d.end());
        double s = 0; for (auto v : d) s += v;
        printf("  %-28s mean %9.0f  p10 %8lld  p50 %8lld  p90 %8lld  max %8lld cycles\n", name, s / nblk,
               d[nblk / 10], d[nblk / 2], d[nblk * 9 / 10], d[nblk - 1]);
    };
    if (pipe) {
        printf("pipe form, prec %d, stage %d, %d workgroups; sums over the tiles of a workgroup (thread 0, s_memtime cycles)\n", prec, stage, nblk);
        stat("source 0 (both halves)", 1, 2);
        if (stage >= 2) stat("source 1", 2, 3);
        if (stage >= 3) stat("source 2", 3, 4);
        if (stage == 4) stat("bilinear taps", 4, 5);
        stat("epilogue", 8, 7); stat("all tiles of the workgroup", 1, 7);
        double tiles = 0; for (int b = 0; b < nblk; ++b) tiles += (double)h[b * 16 + 12];
        printf("  tiles per workgroup: %.2f\n", tiles / nblk);
        return 0;
    }
    printf("prec %d, ", prec); printf("stage %d, TH=%d, %d workgroups (timestamps of thread 0; s_memtime shader cycles)\n", stage, th, nblk);
    stat("stage tile src0", 1, 2); stat("taps src0", 2, 3);
    if (stage >= 2) { stat("stage tile src1", 3, 4); stat("taps src1", 4, 5); stat("src2 (stage+taps)", 5, 6); }
#ifdef SR_TIMELINE_TAPS
    if (prec == 1) {  // sums over the 25 taps of source 0 of the last tile of each workgroup (wave 0)
        auto st1 = [&](const char* nm, int k) { std::vector<long long> d; for (int b = 0; b < nblk; ++b) d.push_back(h[b * 16 + k] / 25); std::sort(d.begin(), d.end());
            double sm = 0; for (auto v : d) sm += v; printf("  per tap: %-28s mean %7.0f p10 %6lld p50 %6lld p90 %6lld cycles\n", nm, sm / nblk, d[nblk / 10], d[nblk / 2], d[nblk * 9 / 10]); };
        st1("DMA req + 12 ds_read issue", 12); st1("wait for operands", 13); st1("12 MFMA issue", 14); st1("vmcnt wait + barrier", 15);
    }
#endif
    stat("next-tile request (persistent)", 6, 8);
    stat("epilogue", 8, 7); stat("all tiles of the workgroup", 1, 7);
    { double tiles = 0; for (int b = 0; b < nblk; ++b) tiles += (double)h[b * 16 + 12]; printf("  tiles per workgroup: %.2f (phase rows above are sums over them)\n", tiles / nblk); }
    long long t0 = h[0]; for (int b = 0; b < nblk; ++b) t0 = std::min(t0, h[b * 16]);
    std::vector<long long> starts; for (int b = 0; b < nblk; ++b) starts.push_back(h[b * 16] - t0);
    std::sort(starts.begin(), starts.end());
    printf("  WG start times (100 MHz ticks): first %lld, p25 %lld, p50 %lld, p75 %lld, last %lld  => kernel ~%.1f us to last start\n",
           starts[0], starts[nblk / 4], starts[nblk / 2], starts[nblk * 3 / 4], starts[nblk - 1], starts[nblk - 1] / 100.0);
    if (prec == 0) {
        // slot occupancy per CU: how much of the kernel's span each CU spends with 2 (1, 0) workgroups resident
        struct Ev { long long t; int d; };
        std::map<long long, std::vector<Ev>> per_cu;
        long long k_begin = h[11], k_end = 0; double prologue = 0;
        for (int b = 0; b < nblk; ++b) {
            const long long s = h[b * 16 + 11], e = h[b * 16 + 9], id = h[b * 16 + 10];
            prologue += (double)(h[b * 16 + 0] - s);
            const long long hw = id & 0xffffffffLL, xcc = (id >> 32) & 0xf;
            const long long cu = (xcc << 16) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 0x1) << 7) | ((hw >> 8) & 0xf);  // xcc, se, sh, cu
            per_cu[cu].push_back({s, +1}); per_cu[cu].push_back({e, -1});
            k_begin = std::min(k_begin, s); k_end = std::max(k_end, e);
        }
        double occ[4] = {0, 0, 0, 0}, gap_sum = 0; long long gaps = 0;
        for (auto& kv : per_cu) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Ev& a, const Ev& b) { return a.t < b.t || (a.t == b.t && a.d < b.d); });
            int cur = 0; long long last = k_begin, last_end = -1;
            for (auto& ev : v) {
                occ[std::min(cur, 3)] += (double)(ev.t - last); last = ev.t;
                if (ev.d < 0) last_end = ev.t; else if (last_end >= 0 && cur < 2) { gap_sum += (double)(ev.t - last_end); ++gaps; last_end = -1; }
                cur += ev.d;
            }
            occ[0] += (double)(k_end - last);
        }
        const double span = (double)(k_end - k_begin) * per_cu.size();
        printf("  %zu CUs seen; kernel span %.1f us; time with 0 / 1 / 2 / 3+ workgroups resident: %.1f %% / %.1f %% / %.1f %% / %.1f %%\n",
               per_cu.size(), (k_end - k_begin) / 100.0, 100 * occ[0] / span, 100 * occ[1] / span, 100 * occ[2] / span, 100 * occ[3] / span);
        printf("  retire -> next kernel entry on the same CU: mean %.2f us over %lld hand-overs; kernel entry -> first barrier: mean %.2f us\n",
               gaps ? gap_sum / gaps / 100.0 : 0.0, gaps, prologue / nblk / 100.0);
    }
    return 0;
}
