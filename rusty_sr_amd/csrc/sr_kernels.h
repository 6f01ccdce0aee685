// sr_kernels.h -- launch interface between sr_api.cpp and sr_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Division of a tile index by a launch constant d as multiply-high + add + shift (Granlund-Montgomery
// round-up form): q = (mulhi(t, m) + t) >> s, exact for t < 2^31.  Scalar-ALU only on the device -- the
// compiler's own expansion goes through v_rcp_f32 on the vector ALU, which the matrix pipe shares.
struct TileDiv {
    uint32_t m, s;
};
inline TileDiv make_tile_div(uint32_t d) {
    TileDiv r{1u, 0u};
    while ((1u << r.s) < d) ++r.s;
    r.m = (uint32_t)(((uint64_t)1 << 32) * (((uint64_t)1 << r.s) - d) / d + 1);
    return r;
}

struct Conv0Args {
    const void* img;      // n*H*W*3 f32, or n*H*W*img_ch u8
    const float* wpack;   // [ky 5][j/2 4][h 2][cout 32][2]: K packed per kernel row (15 of 16 slots), see pack_conv0
    const void* wpack_split;  // split-half mode: [K-block 7][hi | lo][h 2][cout 32][8 halves], slot 16 b + 8 h + e = 4 tap + channel (pack_conv0_split)
    const float* bias;    // 32
    const float* beta;    // 32
    float* dst;           // padded feature map (see FeatGeom), pointer to pixel (0,0) of image 0
    int H, W, img_ch;
    int pitch;            // feature-map row pitch in pixels
    long img_stride;      // feature-map image stride in pixels
    int y_begin, y_end;   // rows to compute
    int tiles_x, tiles_y;
    int n_tiles;          // images * tiles_x * tiles_y; the grid may be smaller (workgroups loop over tiles)
    TileDiv div_tpi, div_tx;
    int* queue_reset;     // the 5 x 8 tile-queue heads of this call's stage kernels: conv0 is the call's first launch and sets
                          // them (they are first read by stage 1, a later launch of the same stream) -- no memset node per call
    int* domain;          // split-half mode: the context's domain flag (mapped host memory), raised when a value leaves the f16 range
    int queue_grid[5];    // workgroups of each stage launch (entry 0 unused): head x of a stage starts at the number of its
                          // workgroups b with b % 8 == x, whose first tile is entry b / 8 of queue x without an atomic
};

// One class of tiles of a stage launch: `tiles_y` tile rows of `th` image rows each (every tile 32 px wide), the first at
// image row y0.  Tile ids of a class run over (image, tile row, tile column) in COLUMN-BLOCK order: blocks of `bw` tile
// columns, each walked row by row (bw = 0: plain row-major).  Consecutive ids -- what one XCD's workgroups hold at a time --
// then cover a few rows of one block instead of a slice of one long tile row, so the halo rows of vertically adjacent
// tiles are still in that XCD's L2 when they are needed again (at 3840 px a tile row of one 32-channel map is 3.9 MB,
// the L2 4 MB).  All divisions by launch constants are multiply-high forms (TileDiv).
struct TileGrid {
    int th, y0, tiles_y;
    int ntiles;                        // n_img * tiles_x * tiles_y
    TileDiv div_tpi;                   // tiles per image
    int bw, nfull;                     // block width in tiles; number of full-width blocks (tiles_x / bw)
    TileDiv div_tx, div_blk, div_bw, div_rem;  // divisors: tiles_x, bw * tiles_y, bw, tiles_x % bw (the last, narrower block)
};
inline TileGrid make_tile_grid(int th, int y0, int tiles_y, int tiles_x, int n_img, int bw) {
    TileGrid g{};
    g.th = th; g.y0 = y0; g.tiles_y = tiles_y; g.ntiles = n_img * tiles_x * tiles_y;
    const TileDiv one{1u, 0u};
    g.div_tpi = g.div_tx = g.div_blk = g.div_bw = g.div_rem = one;
    if (tiles_y <= 0) return g;
    g.div_tpi = make_tile_div((uint32_t)(tiles_x * tiles_y));
    g.div_tx = make_tile_div((uint32_t)tiles_x);
    if (bw <= 0 || bw >= tiles_x) return g;  // row-major
    g.bw = bw; g.nfull = tiles_x / bw;
    g.div_blk = make_tile_div((uint32_t)(bw * tiles_y));
    g.div_bw = make_tile_div((uint32_t)bw);
    const int rem = tiles_x % bw;
    g.div_rem = make_tile_div((uint32_t)(rem > 0 ? rem : 1));
    return g;
}

struct StageArgs {
    const float* src[3];  // NHWC 32-channel feature maps, zero-bordered (pointer to pixel (0,0))
    const float* wpack;   // one 4 KB chunk per step (2 taps x 16 channels [x N-tile]), sources concatenated: sr_api.cpp pack_steps
    const float* bias;    // 32 per N-tile (final stage: expand_bias in the triple layout, see sr_api.cpp)
    const float* beta;    // 32 (unused by the final stage)
    float* dst;           // non-final: padded feature map
    const void* img;      // final: the input image again (bilinear residual)
    void* out;            // final: n*(3*rows)*(3W)*3 f32 or *4 u8 RGBA, rows = y_end-y_begin
    int H, W, img_ch;
    int pitch;            // feature-map row pitch in pixels (>= tiles_x*32 + 4)
    long img_stride;      // feature-map image stride in pixels
    int y_begin, y_end;   // image rows this launch produces
    int tiles_x;
    int n_img;            // images in the batch
    // The rows [y_begin, y_end) are cut into 8-row tiles (grid[0]) and, below them, 4-row tiles (grid[1]); either class
    // may be empty.  The pipe kernel runs both in one launch, big tiles first: every XCD's queue is its run of big tiles
    // followed by its run of small ones, so that the last tiles a launch hands out are half-size -- the spread of the
    // workgroups' finishing times, which is what a launch loses at its end, halves.  The first kernel form runs one class.
    TileGrid grid[2];
    int* queue;           // persistent forms: 8 per-XCD tile-queue heads, set by conv0 (Conv0Args::queue_grid)
    int* domain;          // split-half mode: the context's domain flag (see Conv0Args)
};

// Feature maps live in HBM with a zero border so tile staging never tests bounds:
// rows [-kFeatPad, H + kFeatPadBottom), columns [-kFeatPad, pitch - kFeatPad); kernels only
// ever store inside [0,H) x [0,W), so the border keeps the reference's zero padding.
constexpr int kFeatPad = 2;          // 5x5 halo
constexpr int kFeatPadBottom = 12;   // last tile row may start at H-1: + 8 rows + 2 halo (+2 spare)

// Restore that border for a new geometry: zero every cell of the four maps that is not the interior of one of the n images
// (a few MB instead of the whole allocation -- a context that meets images of many sizes changes geometry per call).
struct ClearArgs {
    float* map[4];
    int n, H, W, pitch;
    int planar;       // the maps are row-planar (split-half mode)
    long img_stride;  // pixels
    long total_px;    // n * img_stride + the rows above image 0 + slack
};
hipError_t sr_launch_clear_borders(const ClearArgs& a, hipStream_t s);
// The split-half mode's maps are row-planar: [y][16-byte channel group c][x] (pixel (0,0) of group c sits
// (kFeatPad * pitch) * 128 + c * pitch * 16 + kFeatPad * 16 bytes into the map); the exact-f32 maps are pixel-major.  The split-half
// weight chunks of stages 1-3 are in the step order of v_mfma_f32_16x16x32_f16 (sr_kernels.hip half_steps_h16; sr_api.cpp
// pack_steps_h16), those of the last stage in that of 32x32x16 (pack_steps).

struct AuxArgs {          // bilinear_net / downsample_net (parameter-free graphs)
    const void* img;      // n*H*W*3 f32 or n*H*W*img_ch u8
    void* out;            // bilinear: n*3H*3W*(3 f32 | 4 u8); downsample: n*(H/3)*(W/3)*(3 f32 | 4 u8)
    int n, H, W, img_ch;
    const void* qtab;     // u8 output: the quantiser table of the context (sr_aux_build_tables), else unused
};
hipError_t sr_aux_build_tables(void** d_tab);  // sr_aux.hip: data_to_img(LinearToSrgb(l)) as a step table, built once per context
hipError_t sr_launch_aux(int graph, const AuxArgs& a, bool img_u8, bool out_u8, hipStream_t s);  // graph: 1 bilinear, 2 downsample

// prec: 0 = exact f32 (v_mfma_f32_32x32x2_f32), 1 = split-half (3 x v_mfma_f32_32x32x16_f16)
hipError_t sr_launch_conv0(const Conv0Args& a, int th, int prec, int nblk, bool img_u8, hipStream_t s);
// factor (2, 3 or 4) only matters for stage 4 (3 f^2 expand channels, depth-to-space x f)
// first form: ONE tile class (th = 8: a.grid[0], th = 4: a.grid[1]), whole source tiles resident
hipError_t sr_launch_stage(int stage, int factor, const StageArgs& a, int th, int prec, int nblk, bool img_u8, bool out_u8,
                           hipStream_t s);
// "pipe" form of the stage kernels (half-tile double buffering, persistent): both tile classes of `a` in one launch;
// grid = co-resident workgroups.
hipError_t sr_launch_stage_pipe(int stage, int factor, const StageArgs& a, int prec, int grid, bool img_u8, bool out_u8,
                                hipStream_t s);
