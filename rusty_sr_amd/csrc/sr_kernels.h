// sr_kernels.h -- launch interface between sr_api.cpp and sr_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Conv0Args {
    const void* img;      // n*H*W*3 f32, or n*H*W*img_ch u8
    const float* wpack;   // 25 taps x [cin/2][cout 32][2]  (cin 3 zero-padded to 4)
    const float* bias;    // 32
    const float* beta;    // 32
    float* dst;           // padded feature map (see FeatGeom), pointer to pixel (0,0) of image 0
    int H, W, img_ch;
    int pitch;            // feature-map row pitch in pixels
    long img_stride;      // feature-map image stride in pixels
    int y_begin, y_end;   // rows to compute
    int tiles_x, tiles_y;
};

struct StageArgs {
    const float* src[3];  // NHWC 32-channel feature maps, zero-bordered (pointer to pixel (0,0))
    const uint32_t* voff5; // LDS-DMA gather tables: byte offset of tile pixel P from the tile origin,
    const uint32_t* voff3; //   36-wide (5x5 source) and 34-wide (3x3 source) tiles, 448 entries each
    const float* wpack;   // one 4 KB chunk per tap, sources concatenated: [cin/4][cout 32][4]
    const float* bias;    // 32 per N-tile (final stage: expand_bias in the triple layout, see sr_api.cpp)
    const float* beta;    // 32 (unused by the final stage)
    float* dst;           // non-final: padded feature map
    const void* img;      // final: the input image again (bilinear residual)
    void* out;            // final: n*(3*rows)*(3W)*3 f32 or *4 u8 RGBA, rows = y_end-y_begin
    int H, W, img_ch;
    int pitch;            // feature-map row pitch in pixels (>= tiles_x*32 + 4)
    long img_stride;      // feature-map image stride in pixels
    int y_begin, y_end;
    int tiles_x, tiles_y;
    int n_img;            // images in the batch (tiles = n_img * tiles_x * tiles_y)
    int* queue;           // persistent form: 8 per-XCD tile-queue heads, zeroed before the launch
};

// Feature maps live in HBM with a zero border so tile staging never tests bounds:
// rows [-kFeatPad, H + kFeatPadBottom), columns [-kFeatPad, pitch - kFeatPad); kernels only
// ever store inside [0,H) x [0,W), so the border keeps the reference's zero padding.
constexpr int kFeatPad = 2;          // 5x5 halo
constexpr int kFeatPadBottom = 12;   // last tile row may start at H-1: + 8 rows + 2 halo (+2 spare)
constexpr int kVoffEntries = 448;    // 7 groups of 64 tile pixels

struct AuxArgs {          // bilinear_net / downsample_net (parameter-free graphs)
    const void* img;      // n*H*W*3 f32 or n*H*W*img_ch u8
    void* out;            // bilinear: n*3H*3W*(3 f32 | 4 u8); downsample: n*(H/3)*(W/3)*(3 f32 | 4 u8)
    int n, H, W, img_ch;
};
hipError_t sr_launch_aux(int graph, const AuxArgs& a, bool img_u8, bool out_u8, hipStream_t s);  // graph: 1 bilinear, 2 downsample

// prec: 0 = exact f32 (v_mfma_f32_32x32x2_f32), 1 = split-half (3 x v_mfma_f32_32x32x16_f16)
hipError_t sr_launch_conv0(const Conv0Args& a, int th, int prec, int nblk, bool img_u8, hipStream_t s);
// factor (2, 3 or 4) only matters for stage 4 (3 f^2 expand channels, depth-to-space x f)
hipError_t sr_launch_stage(int stage, int factor, const StageArgs& a, int th, int prec, int nblk, bool img_u8, bool out_u8,
                           hipStream_t s);
