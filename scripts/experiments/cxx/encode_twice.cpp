// First vs later PNG encode in one process (wall, CPU time, faults, context switches): is the first parallel burst of a process slow?
// g++ -O2 -std=c++17 scripts/cxx/encode_twice.cpp -L rusty_sr_amd -lsrpng -Wl,-rpath,$PWD/rusty_sr_amd -o /tmp/encode_twice && /tmp/encode_twice some.png
#include "../../rusty_sr_amd/host/png.hpp"
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    srpng::Image img; std::string err;
    if (!srpng::decode_file(argv[1], img, err)) { puts(err.c_str()); return 1; }
    for (int r = 0; r < 3; ++r) {
        rusage a, b; getrusage(RUSAGE_SELF, &a);
        auto t0 = std::chrono::steady_clock::now();
        srpng::encode_file("/dev/null", img.rgba.data(), img.w, img.h, err, 1);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        getrusage(RUSAGE_SELF, &b);
        auto tv = [](timeval t) { return t.tv_sec * 1e3 + t.tv_usec / 1e3; };
        printf("encode %d: %.1f ms wall, user %.0f ms, sys %.0f ms, minor faults %ld, vol ctx %ld, invol ctx %ld\n", r, ms, tv(b.ru_utime) - tv(a.ru_utime),
               tv(b.ru_stime) - tv(a.ru_stime), b.ru_minflt - a.ru_minflt, b.ru_nvcsw - a.ru_nvcsw, b.ru_nivcsw - a.ru_nivcsw);
    }
}
