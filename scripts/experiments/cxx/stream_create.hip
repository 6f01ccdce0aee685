// What does a HIP stream cost at start-up, and do several of them cost less when created from several threads?
// hipcc --offload-arch=gfx950 -O2 scripts/cxx/stream_create.hip -o /tmp/stream_create -pthread && /tmp/stream_create
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }
__global__ void nop() {}
int main(int argc, char** argv) {
    const bool parallel = argc > 1 && !strcmp(argv[1], "parallel");
    auto t = std::chrono::steady_clock::now();
    hipSetDevice(0); hipFree(0);
    printf("runtime init %.1f ms\n", ms_since(t));
    hipStream_t s[4] = {};
    t = std::chrono::steady_clock::now();
    if (parallel) {
        std::vector<std::thread> th;
        for (int i = 0; i < 4; ++i) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); });
        for (auto& x : th) x.join();
        printf("4 streams, one thread each: %.1f ms\n", ms_since(t));
    } else {
        for (int i = 0; i < 4; ++i) { auto t1 = std::chrono::steady_clock::now(); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); printf("  stream %d: %.1f ms\n", i, ms_since(t1)); }
        printf("4 streams in sequence: %.1f ms\n", ms_since(t));
    }
    t = std::chrono::steady_clock::now();
    for (int i = 0; i < 4; ++i) { nop<<<1, 64, 0, s[i]>>>(); hipStreamSynchronize(s[i]); printf("  first launch on stream %d: %.1f ms\n", i, ms_since(t)); t = std::chrono::steady_clock::now(); }
    nop<<<1, 64, 0, 0>>>(); hipDeviceSynchronize(); printf("  launch on the null stream: %.1f ms\n", ms_since(t));
    return 0;
}
