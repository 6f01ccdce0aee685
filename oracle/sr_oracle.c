/*
 * sr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the rusty_sr v1 upscale hot path (`graph.forward` at
 * reference src/main.rs:171 over the graph built by `sr_net(3, None)` at
 * reference src/network.rs:16-109).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path (libsrhip) never
 * links or calls it.
 *
 * Where the arithmetic comes from: the reference delegates every op to the
 * un-vendored crate `alumina ^0.1.1` (reference Cargo.toml:11) and the weight
 * container to `bytevec ^0.2.0` (Cargo.toml:10); neither source is under
 * /root/reference and no Rust toolchain exists in this image, so the reference
 * cannot be built here (oracle/_ref is therefore absent by necessity).  The op
 * semantics restated below are the ones SURVEY.md section 8(a)/(c) pinned
 * against the reference's only bit-level result pin:
 *     docs/cartoon_lr.png + src/res/anime.rsr -> docs/cartoon_rsa.png
 * (committed as tests/golden/cartoon_{lr,rsa}.png).  tests/test_oracle_golden.py
 * re-checks that pin on every run: >= 99.99 % of u8 samples equal, max |d| = 1.
 * For imagenet.rsr / imagenetlinear.rsr the code path is identical and only
 * the numbers in the blob differ; their docs images were produced by an
 * earlier weight snapshot, so they serve as PSNR sanity floors only.
 *
 * Build: see oracle/Makefile (gcc -O3 -march=x86-64-v3 -fopenmp -ffp-contract=off -shared).
 * REAL is float by default; -DSR_REAL_DOUBLE builds the f64 "truth" variant
 * used to measure how far any f32 evaluation order sits from exact arithmetic.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef SR_REAL_DOUBLE
typedef double real;
#define SQRT sqrt
#define FLOOR floor
#define SYM(n) n##_f64
#else
typedef float real;
#define SQRT sqrtf
#define FLOOR floorf
#define SYM(n) n
#endif

#define SR_FACTOR 3      /* reference src/main.rs:31  const FACTOR: usize = 3 */
#define SR_CH 3          /* reference src/network.rs:13 const CHANNELS: usize = 3 */
#define SR_FEAT 32       /* reference src/network.rs:29,41 Node::new_shaped(32, ..) */
#define SR_MAXC 64       /* widest conv output handled: 3*f*f = 48 expand channels at factor 4 */
#define SR_EXP (SR_CH * SR_FACTOR * SR_FACTOR) /* network.rs:37 expand node = 27 ch */
#define SR_NPARAMS 130459

/* Parameter segment offsets = op insertion order in reference
 * src/network.rs:33-72 (conv0, f_bias, f_activ, expand_bias, l{1,2,3}_bias,
 * l{1,2,3}_activ, conv1,2,3,5,6,7,8,9,10); Expand / LinearInterp /
 * ShapeConstraint own no parameters.  SURVEY.md 8(a) row W. */
enum {
    OFF_CONV0 = 0,        /* 32*5*5*3  = 2400  network.rs:33 */
    OFF_F_BIAS = 2400,    /* 32                network.rs:34 */
    OFF_F_ACTIV = 2432,   /* 32                network.rs:35 */
    OFF_EXP_BIAS = 2464,  /* 27                network.rs:38 */
    OFF_L1_BIAS = 2491,   /* 32                network.rs:50 */
    OFF_L2_BIAS = 2523,   /*                   network.rs:51 */
    OFF_L3_BIAS = 2555,   /*                   network.rs:52 */
    OFF_L1_ACTIV = 2587,  /*                   network.rs:54 */
    OFF_L2_ACTIV = 2619,  /*                   network.rs:55 */
    OFF_L3_ACTIV = 2651,  /*                   network.rs:56 */
    OFF_CONV1 = 2683,     /* 32*5*5*32 = 25600 network.rs:60 */
    OFF_CONV2 = 28283,    /*                   network.rs:61 */
    OFF_CONV3 = 53883,    /*                   network.rs:62 */
    OFF_CONV5 = 79483,    /* 32*3*3*32 = 9216  network.rs:65 */
    OFF_CONV6 = 88699,    /*                   network.rs:66 */
    OFF_CONV7 = 97915,    /* 27*3*3*32 = 7776  network.rs:67 */
    OFF_CONV8 = 105691,   /*                   network.rs:69 */
    OFF_CONV9 = 114907,   /*                   network.rs:70 */
    OFF_CONV10 = 122683,  /*                   network.rs:72 */
    OFF_END = 130459
};

/* ---- W: bytevec 0.2.0 `<Vec<f32>>::decode::<u32>` (call sites reference
 * src/main.rs:138,146,149,152).  Wire format (verified on
 * all three shipped blobs): u32 LE n | n x u32 LE element byte sizes (all 4) |
 * n x f32 LE.  Returns n (>= 0) or a negative error. */
long SYM(sr_oracle_rsr_decode)(const uint8_t* blob, size_t len, float* out, size_t cap) {
    if (len < 4) return -1;
    uint32_t n;
    memcpy(&n, blob, 4);
    if (len != 4 + (size_t)8 * n) return -2;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t sz;
        memcpy(&sz, blob + 4 + (size_t)4 * i, 4);
        if (sz != 4) return -3;
    }
    if (out) {
        if (cap < n) return -4;
        memcpy(out, blob + 4 + (size_t)4 * n, (size_t)4 * n);
    }
    return (long)n;
}

/* ---- G2: alumina Convolution, Padding::Same (reference network.rs:33,60-72).
 * Cross-correlation, stride 1, zero padding k/2, no bias, ACCUMULATES into dst:
 *   dst[y][x][o] += sum_{ky,kx,i} W[((o*k+ky)*k+kx)*Cin+i] * src[y+ky-k/2][x+kx-k/2][i]
 * NHWC tensors; weight layout [O][KH][KW][I].  Each output scalar is one
 * sequential f32 chain over (ky,kx,i) in that order, independent of threading. */
static void conv_same_acc(const real* src, int H, int W, int Cin, const float* wt, int k, int Cout,
                          real* dst) {
    const int r = k / 2;
    /* transpose weights to [ky][kx][i][o], o padded to 32 with zeros, so the
     * inner loop is a fixed-width vector op over o (padding lanes are discarded) */
    const int CW = Cout <= SR_FEAT ? SR_FEAT : SR_MAXC;  /* padded output width of the inner vector loop */
    real* wT = (real*)calloc((size_t)k * k * Cin * CW, sizeof(real));
    for (int o = 0; o < Cout; ++o)
        for (int t = 0; t < k * k; ++t)
            for (int i = 0; i < Cin; ++i)
                wT[((size_t)t * Cin + i) * CW + o] = (real)wt[((size_t)o * k * k + t) * Cin + i];
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            real acc[SR_MAXC];
            for (int o = 0; o < CW; ++o) acc[o] = 0;
            for (int ky = 0; ky < k; ++ky) {
                const int sy = y + ky - r;
                if (sy < 0 || sy >= H) continue; /* zero padding */
                for (int kx = 0; kx < k; ++kx) {
                    const int sx = x + kx - r;
                    if (sx < 0 || sx >= W) continue;
                    const real* s = src + ((size_t)sy * W + sx) * Cin;
                    const real* w = wT + (size_t)(ky * k + kx) * Cin * CW;
                    if (CW == SR_FEAT) {
                        for (int i = 0; i < Cin; ++i) {
                            const real sv = s[i];
#pragma omp simd
                            for (int o = 0; o < SR_FEAT; ++o) acc[o] += w[(size_t)i * SR_FEAT + o] * sv;
                        }
                    } else {
                        for (int i = 0; i < Cin; ++i) {
                            const real sv = s[i];
#pragma omp simd
                            for (int o = 0; o < SR_MAXC; ++o) acc[o] += w[(size_t)i * SR_MAXC + o] * sv;
                        }
                    }
                }
            }
            real* d = dst + ((size_t)y * W + x) * Cout;
            for (int o = 0; o < Cout; ++o) d[o] += acc[o];
        }
    }
    free(wT);
}

/* ---- G3: alumina Bias, ParamSharing::Spatial (network.rs:34,38,50-52) */
static void bias_add(real* node, size_t npx, int C, const float* b) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)npx; ++p)
        for (int c = 0; c < C; ++c) node[(size_t)p * C + c] += (real)b[c];
}

/* ---- G4: alumina BeLU, ParamSharing::Spatial (network.rs:35,54-56):
 *   dst = beta[c]*x + sqrt(x*x + 1) - 1     (form pinned by SURVEY.md 8(c) item 8) */
static void belu(const real* src, real* dst, size_t npx, int C, const float* beta) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)npx; ++p)
        for (int c = 0; c < C; ++c) {
            const real v = src[(size_t)p * C + c];
            dst[(size_t)p * C + c] = (real)beta[c] * v + SQRT(v * v + (real)1) - (real)1;
        }
}

/* ---- G1: alumina LinearInterp x3 (network.rs:27): bilinear, half-pixel
 * centres, edge clamped, on the sRGB values; ACCUMULATES into out.
 * For output index o along an axis: s=(o+0.5)/3-0.5, i0=floor(s), t=s-i0,
 * v=(1-t)*in[clamp(i0)]+t*in[clamp(i0+1)].  Evaluated per phase so the weights
 * are the exact constants {1/3,2/3,0,1} at every coordinate:
 *   phase 0: i0=i-1,t=2/3   phase 1: i0=i,t=0   phase 2: i0=i,t=1/3  */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static void linterp3_acc(const real* in, int H, int W, real* out) {
    const real T[3] = {(real)2 / (real)3, (real)0, (real)1 / (real)3};
    const int D[3] = {-1, 0, 0};
    const int OW = W * SR_FACTOR;
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < H * SR_FACTOR; ++oy) {
        const int y = oy / 3, py = oy % 3;
        const int y0 = clampi(y + D[py], 0, H - 1), y1 = clampi(y + D[py] + 1, 0, H - 1);
        const real ty = T[py];
        for (int ox = 0; ox < OW; ++ox) {
            const int x = ox / 3, px = ox % 3;
            const int x0 = clampi(x + D[px], 0, W - 1), x1 = clampi(x + D[px] + 1, 0, W - 1);
            const real tx = T[px];
            for (int c = 0; c < SR_CH; ++c) {
                const real a = ((real)1 - tx) * in[((size_t)y0 * W + x0) * SR_CH + c] +
                               tx * in[((size_t)y0 * W + x1) * SR_CH + c];
                const real b = ((real)1 - tx) * in[((size_t)y1 * W + x0) * SR_CH + c] +
                               tx * in[((size_t)y1 * W + x1) * SR_CH + c];
                out[((size_t)oy * OW + ox) * SR_CH + c] += ((real)1 - ty) * a + ty * b;
            }
        }
    }
}

/* ---- G5: alumina Expand x3 (network.rs:39): depth-to-space, ACCUMULATES:
 *   out[3y+dy][3x+dx][c] += e[y][x][(dy*3+dx)*3+c]   (colour fastest, then dx, dy) */
static void expand3_acc(const real* e, int H, int W, real* out) {
    const int OW = W * SR_FACTOR;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    for (int c = 0; c < SR_CH; ++c)
                        out[((size_t)(3 * y + dy) * OW + 3 * x + dx) * SR_CH + c] +=
                            e[((size_t)y * W + x) * SR_EXP + (dy * 3 + dx) * 3 + c];
}

/* Workspace kept between calls (grown on demand, zeroed per use): timing the oracle as a CPU
 * baseline should not charge it 2.4 GB of fresh page faults per 1080p call. */
static real* g_ws[9];
static size_t g_ws_cap[9];
static real* zalloc_slot(int slot, size_t n) {
    if (n > g_ws_cap[slot]) {
        free(g_ws[slot]);
        g_ws[slot] = (real*)malloc(sizeof(real) * (n ? n : 1));
        g_ws_cap[slot] = g_ws[slot] ? n : 0;
    }
    if (g_ws[slot]) {
#pragma omp parallel for schedule(static)
        for (long k = 0; k < (long)n; ++k) g_ws[slot][k] = 0;
    }
    return g_ws[slot];
}

/* ---- G: graph.forward(1, [input], params) (reference main.rs:171) for
 * sr_net(3, None).  in: n*H*W*3, out: n*3H*3W*3 (pre-quantisation), both in
 * `real`.  `taps` (optional, may be NULL) receives the post-activation /
 * pre-expand intermediates of image 0 in the order f, l1, l2, l3 (H*W*32 each)
 * then e (H*W*27) for per-stage parity tests.  Returns 0 or negative error. */
int SYM(sr_oracle_forward)(const float* params, size_t n_params, const real* in, int n, int H,
                           int W, real* out, real* taps) {
    if (n_params != SR_NPARAMS) return -1; /* reference main.rs:162 assert_eq!(params.len(), graph.num_params()) */
    if (n < 0 || H <= 0 || W <= 0) return -2;
    const size_t npx = (size_t)H * W;
    real* f_conv = zalloc_slot(0, npx * SR_FEAT);
    real* f = zalloc_slot(1, npx * SR_FEAT);
    real* l1c = zalloc_slot(2, npx * SR_FEAT);
    real* l1 = zalloc_slot(3, npx * SR_FEAT);
    real* l2c = zalloc_slot(4, npx * SR_FEAT);
    real* l2 = zalloc_slot(5, npx * SR_FEAT);
    real* l3c = zalloc_slot(6, npx * SR_FEAT);
    real* l3 = zalloc_slot(7, npx * SR_FEAT);
    real* e = zalloc_slot(8, npx * SR_EXP);
    if (!f_conv || !f || !l1c || !l1 || !l2c || !l2 || !l3c || !l3 || !e) return -3;
    const float* P = params;
    for (int b = 0; b < n; ++b) {
        const real* x = in + (size_t)b * npx * SR_CH;
        real* o = out + (size_t)b * npx * SR_CH * 9;
        memset(o, 0, sizeof(real) * npx * SR_CH * 9);
        if (b) {
            memset(f_conv, 0, sizeof(real) * npx * SR_FEAT);
            memset(l1c, 0, sizeof(real) * npx * SR_FEAT);
            memset(l2c, 0, sizeof(real) * npx * SR_FEAT);
            memset(l3c, 0, sizeof(real) * npx * SR_FEAT);
            memset(e, 0, sizeof(real) * npx * SR_EXP);
        }
        linterp3_acc(x, H, W, o);                                        /* network.rs:27 */
        conv_same_acc(x, H, W, SR_CH, P + OFF_CONV0, 5, SR_FEAT, f_conv); /* :33 */
        bias_add(f_conv, npx, SR_FEAT, P + OFF_F_BIAS);                   /* :34 */
        belu(f_conv, f, npx, SR_FEAT, P + OFF_F_ACTIV);                   /* :35 */
        conv_same_acc(f, H, W, SR_FEAT, P + OFF_CONV1, 5, SR_FEAT, l1c);  /* :60 */
        bias_add(l1c, npx, SR_FEAT, P + OFF_L1_BIAS);                     /* :50 */
        belu(l1c, l1, npx, SR_FEAT, P + OFF_L1_ACTIV);                    /* :54 */
        conv_same_acc(f, H, W, SR_FEAT, P + OFF_CONV2, 5, SR_FEAT, l2c);  /* :61 */
        conv_same_acc(l1, H, W, SR_FEAT, P + OFF_CONV5, 3, SR_FEAT, l2c); /* :65 */
        bias_add(l2c, npx, SR_FEAT, P + OFF_L2_BIAS);                     /* :51 */
        belu(l2c, l2, npx, SR_FEAT, P + OFF_L2_ACTIV);                    /* :55 */
        conv_same_acc(f, H, W, SR_FEAT, P + OFF_CONV3, 5, SR_FEAT, l3c);  /* :62 */
        conv_same_acc(l1, H, W, SR_FEAT, P + OFF_CONV6, 3, SR_FEAT, l3c); /* :66 */
        conv_same_acc(l2, H, W, SR_FEAT, P + OFF_CONV8, 3, SR_FEAT, l3c); /* :69 */
        bias_add(l3c, npx, SR_FEAT, P + OFF_L3_BIAS);                     /* :52 */
        belu(l3c, l3, npx, SR_FEAT, P + OFF_L3_ACTIV);                    /* :56 */
        conv_same_acc(l1, H, W, SR_FEAT, P + OFF_CONV7, 3, SR_EXP, e);    /* :67 */
        conv_same_acc(l2, H, W, SR_FEAT, P + OFF_CONV9, 3, SR_EXP, e);    /* :70 */
        conv_same_acc(l3, H, W, SR_FEAT, P + OFF_CONV10, 3, SR_EXP, e);   /* :72 */
        bias_add(e, npx, SR_EXP, P + OFF_EXP_BIAS);                       /* :38 */
        expand3_acc(e, H, W, o);                                          /* :39 */
        if (taps && b == 0) {
            memcpy(taps, f, sizeof(real) * npx * SR_FEAT);
            memcpy(taps + npx * SR_FEAT, l1, sizeof(real) * npx * SR_FEAT);
            memcpy(taps + 2 * npx * SR_FEAT, l2, sizeof(real) * npx * SR_FEAT);
            memcpy(taps + 3 * npx * SR_FEAT, l3, sizeof(real) * npx * SR_FEAT);
            memcpy(taps + 4 * npx * SR_FEAT, e, sizeof(real) * npx * SR_EXP);
        }
    }
    return 0;
}

/* ---- sr_net(factor, None) for factor != 3 (reference network.rs:16: `factor` is a parameter of
 * the graph builder; main.rs:31 hard-wires 3 "TODO: expose upscaling factor as argument", and no
 * weights for another factor ship).  UNPINNED for factor != 3: same op semantics with
 *   expand node = 3 f^2 channels (network.rs:37), channel = (dy*f+dx)*3+c (network.rs:39),
 *   LinearInterp x f with half-pixel centres: s = (o+0.5)/f - 0.5,
 * parameter order as in network.rs:33-72 with the f-dependent segment sizes.  At factor 3 it is
 * bit-identical to sr_oracle_forward (tested). */
static size_t nparams_for_factor(int f) {
    const size_t E = (size_t)SR_CH * f * f;
    return 2400 + 32 + 32 + E + 3 * 32 + 3 * 32 + 3 * 25600 + 2 * 9216 + E * 288 + 9216 + 2 * E * 288;
}
int SYM(sr_oracle_num_params_factor)(int f) { return f >= 1 && f <= 4 ? (int)nparams_for_factor(f) : -1; }

static void linterp_f_acc(const real* in, int H, int W, int f, real* out) {
    const int OW = W * f;
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < H * f; ++oy) {
        const int y = oy / f, py = oy % f, ny = 2 * py + 1 - f; /* (s - y) = ny / (2f) */
        const int ya = clampi(y + (ny < 0 ? -1 : 0), 0, H - 1), yb = clampi(y + (ny < 0 ? 0 : 1), 0, H - 1);
        const real ty = (real)(ny < 0 ? ny + 2 * f : ny) / (real)(2 * f);
        for (int ox = 0; ox < OW; ++ox) {
            const int x = ox / f, px = ox % f, nx = 2 * px + 1 - f;
            const int xa = clampi(x + (nx < 0 ? -1 : 0), 0, W - 1), xb = clampi(x + (nx < 0 ? 0 : 1), 0, W - 1);
            const real tx = (real)(nx < 0 ? nx + 2 * f : nx) / (real)(2 * f);
            for (int c = 0; c < SR_CH; ++c) {
                const real a = ((real)1 - tx) * in[((size_t)ya * W + xa) * SR_CH + c] + tx * in[((size_t)ya * W + xb) * SR_CH + c];
                const real b = ((real)1 - tx) * in[((size_t)yb * W + xa) * SR_CH + c] + tx * in[((size_t)yb * W + xb) * SR_CH + c];
                out[((size_t)oy * OW + ox) * SR_CH + c] += ((real)1 - ty) * a + ty * b;
            }
        }
    }
}

int SYM(sr_oracle_forward_factor)(const float* params, size_t n_params, int f, const real* in, int n, int H, int W,
                                  real* out) {
    if (f < 1 || f > 4) return -4;
    if (n_params != nparams_for_factor(f)) return -1;
    if (n < 0 || H <= 0 || W <= 0) return -2;
    const int E = SR_CH * f * f;
    const size_t npx = (size_t)H * W;
    size_t o = 0;
    const float* conv0 = params + o; o += 2400;
    const float* f_bias = params + o; o += 32;
    const float* f_act = params + o; o += 32;
    const float* e_bias = params + o; o += E;
    const float* lb[3]; for (int k = 0; k < 3; ++k) { lb[k] = params + o; o += 32; }
    const float* la[3]; for (int k = 0; k < 3; ++k) { la[k] = params + o; o += 32; }
    const float* c1 = params + o; o += 25600;
    const float* c2 = params + o; o += 25600;
    const float* c3 = params + o; o += 25600;
    const float* c5 = params + o; o += 9216;
    const float* c6 = params + o; o += 9216;
    const float* c7 = params + o; o += (size_t)E * 288;
    const float* c8 = params + o; o += 9216;
    const float* c9 = params + o; o += (size_t)E * 288;
    const float* c10 = params + o; o += (size_t)E * 288;
    real* fc = zalloc_slot(0, npx * SR_FEAT); real* ff = zalloc_slot(1, npx * SR_FEAT);
    real* l1c = zalloc_slot(2, npx * SR_FEAT); real* l1 = zalloc_slot(3, npx * SR_FEAT);
    real* l2c = zalloc_slot(4, npx * SR_FEAT); real* l2 = zalloc_slot(5, npx * SR_FEAT);
    real* l3c = zalloc_slot(6, npx * SR_FEAT); real* l3 = zalloc_slot(7, npx * SR_FEAT);
    real* e = zalloc_slot(8, npx * (size_t)E);
    if (!fc || !ff || !l1c || !l1 || !l2c || !l2 || !l3c || !l3 || !e) return -3;
    for (int b = 0; b < n; ++b) {
        const real* x = in + (size_t)b * npx * SR_CH;
        real* op = out + (size_t)b * npx * SR_CH * f * f;
        memset(op, 0, sizeof(real) * npx * SR_CH * f * f);
        if (b) {
            memset(fc, 0, sizeof(real) * npx * SR_FEAT); memset(l1c, 0, sizeof(real) * npx * SR_FEAT);
            memset(l2c, 0, sizeof(real) * npx * SR_FEAT); memset(l3c, 0, sizeof(real) * npx * SR_FEAT);
            memset(e, 0, sizeof(real) * npx * (size_t)E);
        }
        linterp_f_acc(x, H, W, f, op);
        conv_same_acc(x, H, W, SR_CH, conv0, 5, SR_FEAT, fc); bias_add(fc, npx, SR_FEAT, f_bias); belu(fc, ff, npx, SR_FEAT, f_act);
        conv_same_acc(ff, H, W, SR_FEAT, c1, 5, SR_FEAT, l1c); bias_add(l1c, npx, SR_FEAT, lb[0]); belu(l1c, l1, npx, SR_FEAT, la[0]);
        conv_same_acc(ff, H, W, SR_FEAT, c2, 5, SR_FEAT, l2c); conv_same_acc(l1, H, W, SR_FEAT, c5, 3, SR_FEAT, l2c);
        bias_add(l2c, npx, SR_FEAT, lb[1]); belu(l2c, l2, npx, SR_FEAT, la[1]);
        conv_same_acc(ff, H, W, SR_FEAT, c3, 5, SR_FEAT, l3c); conv_same_acc(l1, H, W, SR_FEAT, c6, 3, SR_FEAT, l3c);
        conv_same_acc(l2, H, W, SR_FEAT, c8, 3, SR_FEAT, l3c);
        bias_add(l3c, npx, SR_FEAT, lb[2]); belu(l3c, l3, npx, SR_FEAT, la[2]);
        conv_same_acc(l1, H, W, SR_FEAT, c7, 3, E, e); conv_same_acc(l2, H, W, SR_FEAT, c9, 3, E, e);
        conv_same_acc(l3, H, W, SR_FEAT, c10, 3, E, e);
        bias_add(e, npx, E, e_bias);
        const int OW = W * f;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                for (int dy = 0; dy < f; ++dy)
                    for (int dx = 0; dx < f; ++dx)
                        for (int c = 0; c < SR_CH; ++c)
                            op[((size_t)(f * y + dy) * OW + f * xx + dx) * SR_CH + c] += e[((size_t)y * W + xx) * E + (dy * f + dx) * 3 + c];
    }
    return 0;
}

/* ---- I: alumina supplier::imagefolder::img_to_data (reference main.rs:170):
 * v[y][x][c] = u8[c] / 255, c in {R,G,B}; alpha (if any) dropped. */
void SYM(sr_oracle_img_to_data)(const uint8_t* px, int in_channels, size_t npx, real* out) {
    for (size_t p = 0; p < npx; ++p)
        for (int c = 0; c < SR_CH; ++c) out[p * SR_CH + c] = (real)px[p * in_channels + c] / (real)255;
}

/* ---- O: alumina data_to_img(..).to_rgba() (reference main.rs:175):
 * u8 = clamp(floor(255*v + 0.5), 0, 255), alpha = 255 (rounding mode pinned by
 * the cartoon golden). */
void SYM(sr_oracle_data_to_rgba8)(const real* v, size_t npx, uint8_t* out) {
    for (size_t p = 0; p < npx; ++p) {
        for (int c = 0; c < SR_CH; ++c) {
            real q = FLOOR((real)255 * v[p * SR_CH + c] + (real)0.5);
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            if (q != q) q = 0; /* NaN: Rust's float -> u8 `as` cast saturates and sends NaN to 0 (a C cast of NaN is undefined) */
            out[p * 4 + c] = (uint8_t)q;
        }
        out[p * 4 + 3] = 255;
    }
}

/* ---- alumina SrgbToLinear / LinearToSrgb (reference network.rs:117,119,133,135):
 * the IEC 61966-2-1 piecewise transfer curve.  alumina's exact constants are not
 * visible (crate absent); this curve reproduces docs/logo_lin.png to knife-edge
 * level, so it is pinned up to quantiser rounding. */
#ifdef SR_REAL_DOUBLE
#define POW pow
#else
#define POW powf
#endif
static inline real srgb_to_linear(real s) {
    return s <= (real)0.04045 ? s / (real)12.92 : POW((s + (real)0.055) / (real)1.055, (real)2.4);
}
static inline real linear_to_srgb(real l) {
    return l <= (real)0.0031308 ? (real)12.92 * l : (real)1.055 * POW(l, (real)1 / (real)2.4) - (real)0.055;
}

/* ---- bilinear_net(3) (reference network.rs:111-123; `-p bilinear`, main.rs:153-155):
 * out = LinearToSrgb(LinearInterp3(SrgbToLinear(in))).  in n*H*W*3 -> out n*3H*3W*3. */
int SYM(sr_oracle_bilinear)(const real* in, int n, int H, int W, real* out) {
    if (n < 0 || H <= 0 || W <= 0) return -2;
    const size_t npx = (size_t)H * W;
    real* lin = (real*)malloc(sizeof(real) * npx * SR_CH);
    if (!lin) return -3;
    for (int b = 0; b < n; ++b) {
        const real* x = in + (size_t)b * npx * SR_CH;
        real* o = out + (size_t)b * npx * SR_CH * 9;
        for (size_t k = 0; k < npx * SR_CH; ++k) lin[k] = srgb_to_linear(x[k]);
        memset(o, 0, sizeof(real) * npx * SR_CH * 9);
        linterp3_acc(lin, H, W, o);
        for (size_t k = 0; k < npx * SR_CH * 9; ++k) o[k] = linear_to_srgb(o[k]);
    }
    free(lin);
    return 0;
}

/* ---- downsample_net(3) (reference network.rs:125-138; `-d`, main.rs:139-141):
 * out = LinearToSrgb(Pooling3x3(SrgbToLinear(in))), Pooling = mean over
 * non-overlapping 3x3 blocks.  UNPINNED: no reference image exercises it, and the
 * behaviour for sizes not divisible by 3 is unknown; this restatement drops the
 * remainder rows/columns (output floor(H/3) x floor(W/3)). */
int SYM(sr_oracle_downsample)(const real* in, int n, int H, int W, real* out) {
    if (n < 0 || H < 3 || W < 3) return -2;
    const int OH = H / 3, OW = W / 3;
    for (int b = 0; b < n; ++b) {
        const real* x = in + (size_t)b * H * W * SR_CH;
        real* o = out + (size_t)b * OH * OW * SR_CH;
        for (int y = 0; y < OH; ++y)
            for (int xx = 0; xx < OW; ++xx)
                for (int c = 0; c < SR_CH; ++c) {
                    real acc = 0;
                    for (int dy = 0; dy < 3; ++dy)
                        for (int dx = 0; dx < 3; ++dx)
                            acc += srgb_to_linear(x[((size_t)(3 * y + dy) * W + 3 * xx + dx) * SR_CH + c]);
                    o[((size_t)y * OW + xx) * SR_CH + c] = linear_to_srgb(acc / (real)9);
                }
    }
    return 0;
}

int SYM(sr_oracle_num_params)(void) { return SR_NPARAMS; }
