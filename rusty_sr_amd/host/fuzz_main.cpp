// fuzz_main.cpp -- decoder robustness harness (test infrastructure of the host CLI, SURVEY.md section 5:
// "-fsanitize=address for host C++").  Built with AddressSanitizer + UBSan by rusty_sr_amd.build.build_sanitized().
//
//   srcodec_asan FILE...        decode every file the way the CLI's `image::open` stand-in does (main.rs:164) and
//                               print one line per file: "ok WxH" or "error: <message>"
//   srcodec_asan --roundtrip W H OUT.png   encode a synthetic RGBA image, decode it again, compare
//
// A malformed file must end in a clean "error:" line -- the counterpart of the reference's
// `.expect("Error opening input image file.")`.  Any out-of-bounds access, overflow or leak makes the sanitizers
// abort the process with a non-zero status, which tests/test_decoder_robustness.py treats as failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "png.hpp"

int main(int argc, char** argv) {
    if (argc >= 5 && !strcmp(argv[1], "--roundtrip")) {
        const int w = atoi(argv[2]), h = atoi(argv[3]);
        std::vector<uint8_t> px((size_t)w * h * 4);
        uint32_t z = 12345;
        for (int y = 0; y < h; ++y)  // noise, flat and ramp rows in turn: literals, long runs and every filter type
            for (int x = 0; x < w * 4; ++x) {
                z = z * 1664525u + 1013904223u;
                const int kind = (y / 3) % 4;
                px[(size_t)y * w * 4 + x] = kind == 0 ? (uint8_t)(z >> 24) : kind == 1 ? (uint8_t)(y * 7) : kind == 2 ? (uint8_t)(x / 4 + y) : (uint8_t)((x / 4) * (y & 3) + (z >> 31));
            }
        std::string err;
        if (!srpng::encode_file(argv[4], px.data(), w, h, err)) { printf("error: %s\n", err.c_str()); return 1; }
        srpng::Image img;
        if (!srpng::decode_file(argv[4], img, err)) { printf("error: %s\n", err.c_str()); return 1; }
        if (img.w != w || img.h != h || img.rgba != px) { printf("error: round trip differs\n"); return 1; }
        printf("ok %dx%d\n", w, h);
        return 0;
    }
    for (int i = 1; i < argc; ++i) {
        srpng::Image img;
        std::string err;
        if (srpng::decode_image_file(argv[i], img, err)) {
            if (img.w <= 0 || img.h <= 0 || img.rgba.size() != (size_t)img.w * img.h * 4) { printf("error: inconsistent image\n"); continue; }
            // touch every byte: a decoder that under-fills its buffer is caught by ASAN / MSAN-style checks here
            unsigned long sum = 0;
            for (uint8_t b : img.rgba) sum += b;
            printf("ok %dx%d %lu\n", img.w, img.h, sum);
        } else {
            printf("error: %s\n", err.c_str());
        }
    }
    return 0;
}
