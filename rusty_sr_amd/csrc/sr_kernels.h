// sr_kernels.h -- launch interface between sr_api.cpp and sr_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Conv0Args {
    const void* img;      // n*H*W*3 f32, or n*H*W*img_ch u8
    const float* wpack;   // 25 taps x [cin/2][cout 32][2]  (cin 3 zero-padded to 4)
    const float* bias;    // 32
    const float* beta;    // 32
    float* dst;           // n*H*W*32
    int H, W, img_ch;
    int y_begin, y_end;   // rows to compute
    int tiles_x, tiles_y;
};

struct StageArgs {
    const float* src[3];  // NHWC 32-channel feature maps
    const float* wpack;   // one 4 KB chunk per tap, sources concatenated: [cin/4][cout 32][4]
    const float* bias;    // 32 (expand_bias zero-padded from 27)
    const float* beta;    // 32 (unused by the final stage)
    float* dst;           // non-final: n*H*W*32
    const void* img;      // final: the input image again (bilinear residual)
    void* out;            // final: n*(3*rows)*(3W)*3 f32 or *4 u8 RGBA, rows = y_end-y_begin
    int H, W, img_ch;
    int y_begin, y_end;
    int tiles_x, tiles_y;
};

hipError_t sr_launch_conv0(const Conv0Args& a, int th, int nblk, bool img_u8, hipStream_t s);
hipError_t sr_launch_stage(int stage, const StageArgs& a, int th, int nblk, bool img_u8, bool out_u8,
                           hipStream_t s);
