// sr_api.cpp -- the C ABI of libsrhip.so (include/srhip.h): context, weight
// re-packing, workspace, stage scheduling.  No CPU compute fallback exists:
// without a HIP device sr_create fails with SR_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srhip.h"
#include "../../include/srhip_experimental.h"
#include "sr_kernels.h"
#include "sr_internal.h"

namespace {

// Parameter segment offsets: op insertion order of reference src/network.rs:33-72
// (SURVEY.md 8(a) row W).  Only the expand node depends on the factor f: 3 f^2 channels
// (network.rs:37), so expand_bias and conv7 / conv9 / conv10 scale with it.
struct ParamLayout {
    size_t conv0, f_bias, f_activ, exp_bias, l_bias[3], l_activ[3], conv1, conv2, conv3, conv5, conv6, conv7, conv8,
        conv9, conv10, end;
    int E;  // expand channels
    explicit ParamLayout(int f) {
        E = 3 * f * f;
        size_t o = 0;
        conv0 = o; o += 2400;
        f_bias = o; o += 32;
        f_activ = o; o += 32;
        exp_bias = o; o += E;
        for (auto& b : l_bias) { b = o; o += 32; }
        for (auto& a : l_activ) { a = o; o += 32; }
        conv1 = o; o += 25600;
        conv2 = o; o += 25600;
        conv3 = o; o += 25600;
        conv5 = o; o += 9216;
        conv6 = o; o += 9216;
        conv7 = o; o += (size_t)E * 288;
        conv8 = o; o += 9216;
        conv9 = o; o += (size_t)E * 288;
        conv10 = o; o += (size_t)E * 288;
        end = o;
    }
};

constexpr int kChunk = 1024;  // floats per tap chunk (32 cin x 32 cout)
constexpr int kAutoBlockWidth = 16;  // default tile order of the stage kernels (StageArgs::bw): 16-tile (512 px) column blocks;
                                     // measured 1080p / 4K, both modes (scripts/bw_exp.py): 0-3 % faster than row-major, 8..32 alike

// w = hi + lo/2048 with hi, lo halves (see split_half in sr_kernels.hip)
void split_half_host(float v, _Float16& hi, _Float16& lo) {
    hi = std::fabs(v) < 6.103515625e-05f ? (_Float16)0.0f : (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}
// The expand node's 3 f^2 channels ((dy*f+dx)*3+c, network.rs:39) are laid out in whole RGB
// triples, 10 per 32-lane N-tile and 5 per 16-lane DPP row (lanes 15 and 31 idle), so that the u8 packing of the
// last kernel gathers G and B from the two lanes above R with row_shl DPP moves and never crosses a row:
// lane j of tile nt carries channel 3*(10 nt + 5 (j/16) + (j%16)/3) + (j%16)%3.  Returns that channel, or -1.
int expand_channel(int f, int nt, int j) {
    const int jj = j % 16, tr = nt * 10 + 5 * (j / 16) + jj / 3;
    return (jj < 15 && tr < f * f) ? tr * 3 + jj % 3 : -1;
}
int expand_tiles(int f) { return (f * f + 9) / 10; }
// The split-half mode's last stage computes the transposed tile (weights as the MFMA's A operand, sr_kernels.hip half_steps_h): lane
// (pixel, h) then holds output slots 8 a + 4 h + b (a, b = 0..3) in register 4 a + b.  Slot j of that form carries the channel that sits in
// column 16 h + 4 a + b of the layout above, so a lane's sixteen registers are the five whole triples of "row" h (stage_epilogue_final_t).
int expand_channel_t(int f, int nt, int j) { return expand_channel(f, nt, 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3)); }

// Weight chunks of the stage kernels (both forms): one 4 KB chunk per STEP = two taps of one 16-channel half
// (and, for a node with more than 32 output channels, per N-tile).  Steps run half 0 (channels 0-15) over the
// tap pairs (0,1), (2,3), ... then half 1; a lone last tap leaves its slot zero.
//   f32  : [q = 2 tapslot + rr][h 2][lane 32][4]    channel = 16 half + 8 rr + 4 h + e
//   split: [hi | lo] x [tapslot 2][h 2][lane 32][8]  channel = 16 half + 8 h + e
// `lane_channel(nt, j)` maps MFMA column j of N-tile nt to the output channel of w ([O][ks][ks][32]) or -1.
template <typename F>
void pack_steps(std::vector<float>& dst, const float* w, int ks, int ntn, bool split, F lane_channel) {
    const int nt = ks * ks, np = (nt + 1) / 2;
    for (int half = 0; half < 2; ++half)
        for (int p = 0; p < np; ++p)
            for (int tile = 0; tile < ntn; ++tile) {
                const size_t base = dst.size();
                dst.resize(base + kChunk, 0.0f);
                _Float16* hp = (_Float16*)(dst.data() + base);
                for (int ts = 0; ts < 2; ++ts) {
                    const int t = 2 * p + ts;
                    if (t >= nt) continue;
                    // the split loop walks the taps column by column (row re-use, half_steps_h), the f32 loop row by row
                    const int tap = split ? (t % ks) * ks + t / ks : t;
                    for (int j = 0; j < 32; ++j) {
                        const int o = lane_channel(tile, j);
                        if (o < 0) continue;
                        const float* src = w + ((size_t)o * nt + tap) * 32 + 16 * half;
                        for (int c = 0; c < 16; ++c) {
                            if (!split) {
                                const int rr = c / 8, h = (c / 4) % 2, e = c % 4;
                                dst[base + (size_t)(2 * ts + rr) * 256 + (h * 32 + j) * 4 + e] = src[c];
                            } else {
                                _Float16 hi, lo;
                                split_half_host(src[c], hi, lo);
                                const int h = c / 8, e = c % 8;
                                hp[ts * 512 + (h * 32 + j) * 8 + e] = hi;
                                hp[1024 + ts * 512 + (h * 32 + j) * 8 + e] = lo;
                            }
                        }
                    }
                }
            }
}

// Split-half weight chunks of stages 1-3 for the 16x16x32 step loop (sr_kernels.hip half_steps_h16): one 4 KB chunk per STEP,
//   [hi | lo][output-channel half 2][lane 64][8 halves],  lane = 16 g + n:  output channel 16 ch + n,  K block g: taps[g >> 1],
//   input channels 16 half + 8 (g & 1) + e  -- a step's K = 32 is two taps x the half's 16 channels.
// Taps run column by column (t = kx * ks + ky, as in the 32x32x16 split loop).  Steps of one source: the first half's tap pairs
// (0,1) (2,3) ..., then ONE step shared by the two halves' odd last tap (K 0-15: first half, K 16-31: second half), then the second
// half's pairs: ks^2 steps per source, none half empty.
void pack_steps_h16(std::vector<float>& dst, const float* w, int ks) {
    const int nt = ks * ks, np = (nt - 1) / 2;
    auto chunk = [&](int tap_a, int half_a, int tap_b, int half_b) {
        const size_t base = dst.size();
        dst.resize(base + kChunk, 0.0f);
        _Float16* hp = (_Float16*)(dst.data() + base);
        for (int ch = 0; ch < 2; ++ch)
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, o = 16 * ch + (lane & 15);
                const int t = (g >> 1) ? tap_b : tap_a, half = (g >> 1) ? half_b : half_a;
                const int tap = (t % ks) * ks + t / ks;  // column-major walk -> [ky][kx] index of w ([O][ks][ks][32])
                for (int e = 0; e < 8; ++e) {
                    _Float16 hi, lo;
                    split_half_host(w[((size_t)o * nt + tap) * 32 + 16 * half + 8 * (g & 1) + e], hi, lo);
                    hp[(ch * 64 + lane) * 8 + e] = hi;
                    hp[1024 + (ch * 64 + lane) * 8 + e] = lo;
                }
            }
    };
    for (int p = 0; p < np; ++p) chunk(2 * p, 0, 2 * p + 1, 0);
    chunk(nt - 1, 0, nt - 1, 1);
    for (int p = 0; p < np; ++p) chunk(2 * p, 1, 2 * p + 1, 1);
}

// conv0 [32][5][5][3]: K packed per kernel row -- slot k = 3 kx + c (15 used of 16) -- as
// [ky 5][jj 4][h 2][o 32][e 2] with k = 2 (2 jj + e) + h  (conv0_kernel: B[k][o] for MFMA j = 2 jj + e).
void pack_conv0(std::vector<float>& dst, const float* w) {
    dst.assign(5 * 8 * 64, 0.0f);
    for (int ky = 0; ky < 5; ++ky)
        for (int jj = 0; jj < 4; ++jj)
            for (int h = 0; h < 2; ++h)
                for (int o = 0; o < 32; ++o)
                    for (int e = 0; e < 2; ++e) {
                        const int k = 2 * (2 * jj + e) + h;
                        if (k < 15) dst[(((ky * 4 + jj) * 2 + h) * 32 + o) * 2 + e] = w[(((size_t)o * 5 + ky) * 5 + k / 3) * 3 + k % 3];
                    }
}

// conv0 for the split-half mode's own stage-0 kernel (conv0_split_kernel): K slot s = 4 tap + channel, tap = 5 ky + kx (100 slots, the
// fourth channel and slots 100-111 zero), seven K-blocks of 16: [b 7][hi | lo][h 2][o 32][e 8] halves with s = 16 b + 8 h + e.
void pack_conv0_split(std::vector<float>& dst, const float* w) {
    dst.assign(7 * 2 * 2 * 32 * 8 / 2, 0.0f);
    _Float16* hp = (_Float16*)dst.data();
    for (int b = 0; b < 7; ++b)
        for (int h = 0; h < 2; ++h)
            for (int o = 0; o < 32; ++o)
                for (int e = 0; e < 8; ++e) {
                    const int s = 16 * b + 8 * h + e, tap = s / 4, c = s % 4;
                    if (tap >= 25 || c >= 3) continue;
                    _Float16 hi, lo;
                    split_half_host(w[((size_t)o * 25 + tap) * 3 + c], hi, lo);
                    hp[(((b * 2 + 0) * 2 + h) * 32 + o) * 8 + e] = hi;
                    hp[(((b * 2 + 1) * 2 + h) * 32 + o) * 8 + e] = lo;
                }
}

// LinearInterp x f (network.rs:27) as a 3x3 convolution of the edge-replicated 3-channel
// input onto the expand channels: 9 taps x N-tiles x [h 2][lane 32][q 2], cin = 2h+q.
// Along one axis output phase p of pixel i sits at s - i = (2p+1-f)/(2f) (half-pixel centres,
// SURVEY.md 8(a) G1): negative -> inputs (i-1, i) with weights (1-t, t), t = (2p+1+f)/(2f);
// otherwise -> inputs (i, i+1) with weights (1-t, t), t = (2p+1-f)/(2f).  Same f32 constants
// as the oracle; the 2-D weight is their f32 product.
void pack_lin(std::vector<float>& dst, int f) {
    float w1[4][3] = {};
    for (int p = 0; p < f; ++p) {
        const int nn = 2 * p + 1 - f;
        const float t = (float)(nn < 0 ? nn + 2 * f : nn) / (float)(2 * f);
        if (nn < 0) { w1[p][0] = 1.0f - t; w1[p][1] = t; }
        else { w1[p][1] = 1.0f - t; w1[p][2] = t; }
    }
    const int ntn = expand_tiles(f);
    const size_t base = dst.size();
    dst.resize(base + (size_t)9 * ntn * 128, 0.0f);
    for (int u = 0; u < 3; ++u)
        for (int v = 0; v < 3; ++v)
            for (int nt = 0; nt < ntn; ++nt)
                for (int j = 0; j < 32; ++j) {
                    const int ch = expand_channel(f, nt, j);
                    if (ch < 0) continue;
                    const int c = ch % 3, tr = ch / 3, dy = tr / f, dx = tr % f;
                    dst[base + ((u * 3 + v) * ntn + nt) * 128 + ((c >> 1) * 32 + j) * 2 + (c & 1)] = w1[dy][u] * w1[dx][v];
                }
}

// The same residual for the split-half mode's last stage (sr_kernels.hip lin_mfma_h): the staged pixels are divided by (2 f)^2, which
// makes every weight a small integer -- exact in a half, no lo part.  K slot = 4 tap + colour (36 of 48 used), three K-blocks of 16:
// [b 3][N-tile][h 2][lane 32][e 8] halves, slot 16 b + 8 h + e.
void pack_lin_split(std::vector<float>& dst, int f) {
    float w1[4][3] = {};
    for (int p = 0; p < f; ++p) {
        const int nn = 2 * p + 1 - f;
        const float t = (float)(nn < 0 ? nn + 2 * f : nn) / (float)(2 * f);
        if (nn < 0) { w1[p][0] = 1.0f - t; w1[p][1] = t; }
        else { w1[p][1] = 1.0f - t; w1[p][2] = t; }
    }
    const int ntn = expand_tiles(f);
    const float scale = 4.0f * f * f;
    const size_t base = dst.size();
    dst.resize(base + (size_t)3 * ntn * 256, 0.0f);
    _Float16* hp = (_Float16*)(dst.data() + base);
    for (int b = 0; b < 3; ++b)
        for (int nt = 0; nt < ntn; ++nt)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 32; ++j)
                    for (int e = 0; e < 8; ++e) {
                        const int s = 16 * b + 8 * h + e, tap = s / 4, c = s % 4, ch = expand_channel_t(f, nt, j);
                        if (tap >= 9 || c >= 3 || ch < 0 || ch % 3 != c) continue;
                        const int tr = ch / 3, dy = tr / f, dx = tr % f;
                        hp[(((size_t)(b * ntn + nt) * 2 + h) * 32 + j) * 8 + e] = (_Float16)std::nearbyint(w1[dy][tap / 3] * w1[dx][tap % 3] * scale);
                    }
}

}  // namespace

extern "C" {

const char* sr_strerror(int s) {
    switch (s) {
        case SR_OK: return "ok";
        case SR_E_INVALID: return "invalid argument";
        case SR_E_PARAM_COUNT:
            return "Parameters selected do not have the size required by the neural net. Ensure that the "
                   "same sample factor is used for upscaling and training";  // reference main.rs:162
        case SR_E_FACTOR: return "unsupported upscaling factor (sr_net: 2, 3 or 4; the reference ships 3, main.rs:31)";
        case SR_E_NO_DEVICE: return "no HIP (gfx950) device available; libsrhip has no CPU fallback";
        case SR_E_HIP: return "HIP runtime error";
        case SR_E_NOMEM: return "out of device memory";
        case SR_E_BYTEVEC: return "ByteVec conversion failed";  // reference main.rs:138
        case SR_E_HALO: return "band halo must be 0 (true image edge) or >= SR_HALO, and a sharded band at least SR_HALO rows";
        case SR_E_COMM: return "RCCL communicator missing or failed (librccl not loadable, sr_comm_init_* not called, or an RCCL error)";
        case SR_E_DOMAIN: return "outside the domain of SR_PRECISION_SPLIT_F16: a weight, an input or an activation is not finite or reaches 65504 in magnitude (use SR_PRECISION_F32)";
        default: return "unknown error";
    }
}

int sr_rsr_decode(const uint8_t* blob, size_t len, float* out, size_t cap, size_t* n_out) {
    if (!blob || len < 4) return SR_E_BYTEVEC;
    uint32_t n;
    memcpy(&n, blob, 4);
    if (len != 4 + (size_t)8 * n) return SR_E_BYTEVEC;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t sz;
        memcpy(&sz, blob + 4 + (size_t)4 * i, 4);
        if (sz != 4) return SR_E_BYTEVEC;
    }
    if (n_out) *n_out = n;
    if (out) {
        if (cap < n) return SR_E_INVALID;
        memcpy(out, blob + 4 + (size_t)4 * n, (size_t)4 * n);
    }
    return SR_OK;
}

int sr_rsr_encode(const float* params, size_t n, uint8_t* out, size_t cap, size_t* len_out) {
    if (n > 0xffffffffu) return SR_E_INVALID;
    const size_t len = 4 + 8 * n;
    if (len_out) *len_out = len;
    if (!out) return SR_OK;
    if (!params || cap < len) return SR_E_INVALID;
    const uint32_t n32 = (uint32_t)n, four = 4;
    memcpy(out, &n32, 4);
    for (size_t i = 0; i < n; ++i) memcpy(out + 4 + 4 * i, &four, 4);
    memcpy(out + 4 + 4 * n, params, 4 * n);
    return SR_OK;
}

int sr_create(sr_ctx** out, const float* params, size_t n_params, int factor, int device) {
    return sr_create_graph(out, SR_GRAPH_SR_NET, params, n_params, factor, device);
}

int sr_create_graph(sr_ctx** out, int graph, const float* params, size_t n_params, int factor, int device) {
    if (!out) return SR_E_INVALID;
    *out = nullptr;
    if (graph != SR_GRAPH_SR_NET && graph != SR_GRAPH_BILINEAR && graph != SR_GRAPH_DOWNSAMPLE) return SR_E_INVALID;
    // the reference is hard-wired to 3 (main.rs:31); sr_net itself takes the factor as an argument
    // (network.rs:16), so user-trained 2x / 4x parameter files are accepted too
    if (graph == SR_GRAPH_SR_NET ? (factor < 2 || factor > 4) : factor != SR_FACTOR) return SR_E_FACTOR;
    // main.rs:162 assert_eq!(params.len(), graph.num_params()): 130459 for sr_net(3), 0 for the other two
    if (n_params != (graph == SR_GRAPH_SR_NET ? ParamLayout(factor).end : 0)) return SR_E_PARAM_COUNT;
    if (graph == SR_GRAPH_SR_NET && !params) return SR_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SR_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return SR_E_INVALID;
    sr_ctx* c = new (std::nothrow) sr_ctx();
    if (!c) return SR_E_NOMEM;
    sr_device_guard restore_device;
    c->device = device;
    c->graph = graph;
    c->factor = factor;
    // experiment switches: the environment gives the defaults, read here once; sr_set_experiment changes them
    static const char* const kSwitch[12][2] = {{"th", "SRHIP_TH"}, {"pipe", "SRHIP_PIPE"}, {"bw", "SRHIP_BW"}, {"tail", "SRHIP_TAIL"},
                                               {"bands", "SRHIP_BANDS"}, {"geo", "SRHIP_GEO"}, {"rows", "SRHIP_ROWS"}, {"fork", "SRHIP_FORK"},
                                               {"forkshare", "SRHIP_FORKSHARE"}, {"forkmin", "SRHIP_FORKMIN"}, {"forktune", "SRHIP_FORKTUNE"}, {"halo", "SRHIP_HALO"}};
    for (const auto& sw : kSwitch)
        if (const char* e = getenv(sw[1])) (void)sr_set_experiment(c, sw[0], e);
    {   // FNV-1a over the parameter bits: contexts that share a sharded call must hold the same parameters
        unsigned long long hsh = 1469598103934665603ull;
        const unsigned char* pb = (const unsigned char*)params;
        for (size_t i = 0; i < n_params * sizeof(float); ++i) { hsh ^= pb[i]; hsh *= 1099511628211ull; }
        c->params_hash = hsh;
    }
    // SRHIP_TRACE=1: where sr_create spends its time (a one-shot process pays it once per image)
    const bool trace = getenv("SRHIP_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[srhip] sr_create: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    int rc = [&]() -> int {
        HIPCHK(c, hipSetDevice(device));
        mark("hipSetDevice");
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, device));
        mark("hipGetDeviceProperties");
        snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
        c->cus = prop.multiProcessorCount;
        c->clock_mhz = prop.clockRate / 1000;
        // no stream yet: each costs 8-40 ms of start-up (a hardware queue), the device-pointer entry points run on the
        // caller's streams, a host call of one chunk needs one stream and only a pipelined one all three (sr_ensure_streams)
        for (auto& e : c->ev) HIPCHK(c, hipEventCreate(&e));
        mark("events");

        if (graph != SR_GRAPH_SR_NET) {  // parameter-free graphs: only the quantiser table of their u8 entry points
            const hipError_t te = sr_aux_build_tables(&c->d_qtab);
            // hipErrorUnknown: the builder's own plausibility checks of the device's powf failed -- no table, the u8 entry points of this
            // context answer SR_E_HIP (sr_launch_aux), its f32 entry points need none; anything else is a real HIP error
            if (te != hipSuccess && te != hipErrorUnknown) HIPCHK(c, te);
            if (te == hipErrorUnknown) { (void)hipGetLastError(); c->last_hip = (int)te; }
            mark("quantiser table");
            return SR_OK;
        }
        // ---- pack every parameter once, in the layouts the kernels read
        std::vector<float> host, w;
        auto push = [&](const std::vector<float>& v) {
            const size_t off = host.size();
            host.insert(host.end(), v.begin(), v.end());
            while (host.size() % 64) host.push_back(0.0f);  // keep 256-B alignment
            return off;
        };
        auto vec32 = [&](size_t off, int n) {
            std::vector<float> v(32, 0.0f);
            for (int i = 0; i < n; ++i) v[i] = params[off + i];
            return v;
        };
        const ParamLayout L(factor);
        pack_conv0(w, params + L.conv0);
        c->off_w0 = push(w);
        pack_conv0_split(w, params + L.conv0);
        c->off_w0h = push(w);
        for (int split = 0; split < 2; ++split) {  // exact-f32 chunks, then the same stages in split-half form
            auto ident = [](int, int j) { return j; };
            auto expand = [&](int nt, int j) { return split ? expand_channel_t(factor, nt, j) : expand_channel(factor, nt, j); };
            const bool h16 = split != 0;  // stages 1-3 of the split-half mode on 16x16x32 MFMAs: their own step order
            auto conv = [&](const float* wp, int ks) { if (h16) pack_steps_h16(w, wp, ks); else pack_steps(w, wp, ks, 1, split != 0, ident); };
            auto exp3 = [&](const float* wp) { pack_steps(w, wp, 3, expand_tiles(factor), split != 0, expand); };
            size_t* off = split ? c->off_wh : c->off_w;
            w.clear(); conv(params + L.conv1, 5);
            off[1] = push(w);
            w.clear(); conv(params + L.conv2, 5); conv(params + L.conv5, 3);
            off[2] = push(w);
            w.clear(); conv(params + L.conv3, 5); conv(params + L.conv6, 3); conv(params + L.conv8, 3);
            off[3] = push(w);
            w.clear();
            exp3(params + L.conv7); exp3(params + L.conv9); exp3(params + L.conv10);
            if (split) pack_lin_split(w, factor);  // the split-half mode: on the f16 pipe too, with exact integer weights (lin_mfma_h)
            else pack_lin(w, factor);
            off[4] = push(w);
        }
        const size_t boff[4] = {L.f_bias, L.l_bias[0], L.l_bias[1], L.l_bias[2]};
        const size_t aoff[4] = {L.f_activ, L.l_activ[0], L.l_activ[1], L.l_activ[2]};
        for (int s = 0; s < 4; ++s) c->off_bias[s] = push(vec32(boff[s], 32));
        for (int s = 0; s < 4; ++s) c->off_beta[s] = push(vec32(aoff[s], 32));
        {   // expand_bias in the triple layout, 32 floats per N-tile
            std::vector<float> eb((size_t)expand_tiles(factor) * 32, 0.0f);
            for (int nt = 0; nt < expand_tiles(factor); ++nt)
                for (int j = 0; j < 32; ++j) {
                    const int ch = expand_channel(factor, nt, j);
                    if (ch >= 0) eb[nt * 32 + j] = params[L.exp_bias + ch];
                }
            c->off_bias[4] = push(eb);
        }
        // the split-half mode carries every conv weight as a pair of halves: all of them finite and below the largest half, or the mode is refused
        for (size_t k = L.conv1; k < L.end && c->split_ok; ++k) c->split_ok = std::fabs(params[k]) < 65504.0f;   // (false for NaN too)
        for (size_t k = L.conv0; k < L.conv0 + 2400 && c->split_ok; ++k) c->split_ok = std::fabs(params[k]) < 65504.0f;
        mark("weight packing (host)");
        HIPCHK(c, hipHostMalloc((void**)&c->h_domain, 64, hipHostMallocMapped));
        *c->h_domain = 0;
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_domain, c->h_domain, 0));
        for (auto& w : c->ws) HIPCHK(c, hipMalloc((void**)&w.d_queue, 5 * 8 * sizeof(int)));

        HIPCHK(c, hipMalloc((void**)&c->d_params, host.size() * sizeof(float)));
        mark("hipMalloc x3");
        HIPCHK(c, hipMemcpy(c->d_params, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        mark("parameter upload");
        return SR_OK;
    }();
    if (rc != SR_OK) {
        sr_destroy(c);
        return rc;
    }
    *out = c;
    return SR_OK;
}

void sr_destroy(sr_ctx* c) {
    if (!c) return;
    sr_device_guard restore_device;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    sr_comm_release(c);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (auto& w : c->ws) {
        for (auto& p : w.d_feat) if (p) (void)hipFree(p);
        if (w.d_queue) (void)hipFree(w.d_queue);
    }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->d_qtab) (void)hipFree(c->d_qtab);
    if (c->h_domain) (void)hipHostFree(c->h_domain);
    for (auto& p : c->d_in) if (p) (void)hipFree(p);
    for (auto& p : c->d_out) if (p) (void)hipFree(p);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_fork) if (e) (void)hipEventDestroy(e);
    sr_fork_tune_clear(c);
    for (auto& e : c->pool) if (e) (void)hipEventDestroy(e);
    if (c->copy_in) { (void)hipStreamSynchronize(c->copy_in); (void)hipStreamDestroy(c->copy_in); }
    if (c->copy_out) { (void)hipStreamSynchronize(c->copy_out); (void)hipStreamDestroy(c->copy_out); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sr_last_hip_error(sr_ctx* c) { return c ? c->last_hip : 0; }

int sr_num_params_factor(int factor) { return factor >= 2 && factor <= 4 ? (int)ParamLayout(factor).end : -1; }

int sr_num_params(int graph) { return graph == SR_GRAPH_SR_NET ? SR_NUM_PARAMS : (graph == SR_GRAPH_BILINEAR || graph == SR_GRAPH_DOWNSAMPLE ? 0 : -1); }

int sr_set_precision(sr_ctx* c, int mode) {
    if (!c || (mode != SR_PRECISION_F32 && mode != SR_PRECISION_SPLIT_F16)) return SR_E_INVALID;
    if (mode == SR_PRECISION_SPLIT_F16 && !c->split_ok) return SR_E_DOMAIN;  // a weight that no pair of halves can carry: refused, not clamped
    if (mode != c->precision) {
        // the two modes lay their feature maps out differently (sr_kernels.h: row-planar against pixel-major): what was interior for one is border
        // for the other, so the workspaces' borders are cleared again before the next pass (ensure_features)
        for (auto& w : c->ws) w.geo_n = 0;
        c->last_h = c->last_w = 0;
    }
    c->precision = mode;
    return SR_OK;
}

int sr_set_experiment(sr_ctx* c, const char* key, const char* value) {
    if (!c || !key) return SR_E_INVALID;
    const char* v = value ? value : "";
    if (!strcmp(key, "th")) {          // tile height: "" automatic, one digit (4 | 8) for all stages, or five digits
        for (int k = 0; k < 5; ++k) {
            const char ch = strlen(v) == 5 ? v[k] : v[0];
            c->env_th[k] = ch == '4' ? 4 : (ch == '8' ? 8 : 0);
        }
    } else if (!strcmp(key, "pipe")) { // "" automatic; "none": first form of the stage kernels everywhere; "all": pipe form also for small launches
        c->env_pipe = !strcmp(v, "none") ? 0 : !strcmp(v, "all") ? 2 : 1;
    } else if (!strcmp(key, "bands")) {  // host pipeline: row bands of one large image ("" / "0": automatic)
        c->env_bands = *v ? atoi(v) : 0;
    } else if (!strcmp(key, "rows")) {  // host pipeline: the bands' heights themselves, "r0,r1,..." top to bottom (used when they add up to the
        c->env_rows.clear();            // rows of the call); computed in order on one stream, with a leading '=' on alternating streams
        c->env_rows_two = *v == '=';
        if (*v == '=') ++v;
        while (*v) {
            c->env_rows.push_back(atoi(v));
            while (*v && *v != ',') ++v;
            if (*v) ++v;
        }
    } else if (!strcmp(key, "geo")) {  // "0": equal bands also where the host pipeline would shrink them geometrically
        c->env_geo = strcmp(v, "0") != 0;
    } else if (!strcmp(key, "tail")) {  // how many 4-row tiles end a launch of 8-row tiles, in units of the resident workgroups ("" : automatic, "0": none)
        c->env_tail = *v ? (float)atof(v) : -1.0f;
    } else if (!strcmp(key, "fork")) {  // device entry points, one image as two bands on two streams: "" automatic, "0" never, "1" always, N > 1: always, N rows first
        c->env_fork = *v ? atoi(v) : -1;
    } else if (!strcmp(key, "forkshare")) {  // ... the first band's share of the rows ("" : 0.5)
        c->fork_share = *v ? std::min(0.9, std::max(0.1, atof(v))) : 0.5;
    } else if (!strcmp(key, "forkmin")) {  // ... automatic rule: fork from this many rounds of tiles on
        c->fork_min_rounds = *v ? atof(v) : 3.5;
    } else if (!strcmp(key, "forktune")) {  // ... automatic rule, mid-size shapes: "" / "1": measure both plans on the caller's calls and keep the faster (ForkTune); "0": the rule alone; either forgets what was measured
        c->fork_autotune = strcmp(v, "0") != 0;
        sr_device_guard restore_device;
        (void)hipSetDevice(c->device);
        sr_fork_tune_clear(c);
    } else if (!strcmp(key, "halo")) {  // sharded calls: "" / "input": 7 input rows per neighbour, the overlap recomputed; "layers": feature rows after every stage
        c->layer_halos = !strcmp(v, "layers");
    } else if (!strcmp(key, "bw")) {   // tile-order column-block width in tiles; "" / negative: automatic, 0: plain row-major
        c->env_bw = *v ? atoi(v) : -1;
    } else {
        return SR_E_INVALID;
    }
    return SR_OK;
}

int sr_get_experiment(sr_ctx* c, const char* key, char* buf, size_t cap) {
    if (!c || !key || !buf || cap == 0) return SR_E_INVALID;
    std::string out;
    if (!strcmp(key, "forktune")) {  // one line per shape the tuner has met: "HxW prec io state undivided_ms forked_ms"
        char line[160];
        for (const auto& t : c->fork_tune) {
            snprintf(line, sizeof line, "%dx%d+%d+%d %s %s %s %.4f %.4f\n", t.H, t.W, t.top, t.bot, t.precision == SR_PRECISION_F32 ? "f32" : "split_f16",
                     t.img_u8 ? "u8" : "f32", t.decided < 0 ? "measuring" : (t.decided ? "forked" : "undivided"),
                     t.best[0] < 1e29f ? t.best[0] : 0.f, t.best[1] < 1e29f ? t.best[1] : 0.f);
            out += line;
        }
    } else {
        return SR_E_INVALID;
    }
    if (out.size() + 1 > cap) return SR_E_INVALID;
    memcpy(buf, out.c_str(), out.size() + 1);
    return SR_OK;
}

int sr_set_pipeline(sr_ctx* c, int enabled) {
    if (!c) return SR_E_INVALID;
    c->pipeline = enabled != 0;
    return SR_OK;
}

int sr_host_alloc(void** out, size_t bytes) {
    if (!out || bytes == 0) return SR_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SR_E_NO_DEVICE;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return e == hipErrorOutOfMemory ? SR_E_NOMEM : SR_E_HIP; }
    return SR_OK;
}

void sr_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int sr_set_profiling(sr_ctx* c, int enabled) {
    if (!c) return SR_E_INVALID;
    c->profiling = enabled != 0;
    return SR_OK;
}

int sr_device_info(sr_ctx* c, char* name, size_t cap, int* cus, int* clock_mhz) {
    if (!c) return SR_E_INVALID;
    if (name && cap) snprintf(name, cap, "%s", c->name);
    if (cus) *cus = c->cus;
    if (clock_mhz) *clock_mhz = c->clock_mhz;
    return SR_OK;
}

}  // extern "C"

namespace {

// (Re)allocate the four zero-bordered feature maps for n images of H x W and make
// sure every border pixel is zero.  Kernels only store inside [0,H) x [0,W), so the
// borders stay clean until the geometry changes.
int ensure_features(sr_ctx* c, sr_ctx::Workspace& w, int n, int H, int W, int tiles_x, hipStream_t s) {
    const int pitch = tiles_x * 32 + 2 * kFeatPad;
    const long rows = (long)H + kFeatPad + kFeatPadBottom;
    const long img_stride = rows * pitch;
    const size_t npx = (size_t)n * img_stride + (size_t)kFeatPad * pitch + 64;  // slack for the row above image 0
    if (npx > w.feat_cap_px) {
        for (auto& p : w.d_feat) {
            if (p) HIPCHK(c, hipFree(p));
            p = nullptr;
        }
        w.feat_cap_px = 0; w.geo_n = 0;
        for (auto& p : w.d_feat) HIPCHK(c, hipMalloc((void**)&p, npx * 32 * sizeof(float)));
        w.feat_cap_px = npx;
        // fresh memory: everything once (interiors too: tile rows past a band's last row are computed and discarded,
        // and what they read should at least be numbers)
        for (auto& p : w.d_feat) HIPCHK(c, hipMemsetAsync(p, 0, npx * 32 * sizeof(float), s));
        w.geo_n = n; w.geo_h = H; w.geo_w = W;
    }
    if (w.geo_n != n || w.geo_h != H || w.geo_w != W) {
        // another geometry in the same allocation: what was interior may now be border.  Only the border is cleared
        // (1080p: 6 MB per map instead of 270 MB), so a context that meets a new image size per call pays microseconds.
        ClearArgs ca{};
        for (int k = 0; k < 4; ++k) ca.map[k] = w.d_feat[k];
        ca.n = n; ca.H = H; ca.W = W; ca.pitch = pitch; ca.img_stride = img_stride; ca.total_px = (long)npx;
        ca.planar = c->precision == SR_PRECISION_SPLIT_F16;
        HIPCHK(c, sr_launch_clear_borders(ca, s));
        w.geo_n = n; w.geo_h = H; w.geo_w = W;
    }
    w.pitch = pitch; w.img_stride = img_stride;
    return SR_OK;
}

}  // namespace

// The context's own streams, created when first needed: `stream` for everything the library runs by itself, and for a
// pipelined host call a second compute stream and the download stream.
int sr_ensure_streams(sr_ctx* c, bool pipelined) {
    if (!c->stream) HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (pipelined) {
        if (!c->stream2) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        if (!c->copy_out) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
    }
    return SR_OK;
}

int sr_ensure_buf(sr_ctx* c, void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return SR_OK;
    if (*p) HIPCHK(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    HIPCHK(c, hipMalloc(p, bytes));
    *cap = bytes;
    return SR_OK;
}

namespace {

// One pass of the conv stack over rows [top, bot) of n images of H x W on one stream with one workspace: planned first
// (tile classes, grids), then launched stage by stage -- so that two jobs (the two bands of a forked call) can be issued
// interleaved, stage for stage, each on its own stream.
struct StackJob {
    sr_ctx* c = nullptr;
    sr_ctx::Workspace* ws = nullptr;
    const void* d_img = nullptr;
    void* d_out = nullptr;
    bool img_u8 = false, out_u8 = false;
    int img_ch = 3, n = 1, H = 0, W = 0, top = 0, bot = 0, tiles_x = 0;
    bool forked = false;  // one of the two bands of sr_run_stack_auto: no 4-row tail (see prepare)
    bool layers = false;  // every stage computes rows [top, bot) only: the rows beyond come from the neighbours' maps (sr_band_pass)
    // rows [0, late_top) and [H - late_bot, H) of the image arrive with gate->ready (sr_internal.h sr_halo_gate); mark: record gate->mark around the wait
    const sr_halo_gate* gate = nullptr;
    int late_top = 0, late_bot = 0;
    bool mark = false;
    hipStream_t s = nullptr;
    struct Launch { int y0, y1, ty8, ty4, th, grid; bool pipe; } L[5];
    float* feat[4] = {nullptr, nullptr, nullptr, nullptr};
    int prepare();
    int launch(int st) const;
};

int StackJob::prepare() {
    static const int margin[5] = {5, 3, 2, 1, 0};
    tiles_x = (W + 31) / 32;
    const int rc = ensure_features(c, *ws, n, H, W, tiles_x, s);
    if (rc != SR_OK) return rc;
    const int cus = c->cus > 0 ? c->cus : 256;
    const int resident = 2 * cus;  // workgroups of a stage kernel that fit the chip at once (2 per CU: 76-78 KB of LDS each)
    // pointers to pixel (0,0) of image 0 inside the zero-bordered maps
    // (row-planar maps of the split-half mode: pixel (0,0) of channel group 0 -- kFeatPad rows down, kFeatPad 16-byte cells in)
    const bool planar = c->precision == SR_PRECISION_SPLIT_F16;
    for (int k = 0; k < 4; ++k) feat[k] = ws->d_feat[k] + (planar ? (size_t)kFeatPad * ws->pitch * 32 + kFeatPad * 4 : ((size_t)kFeatPad * ws->pitch + kFeatPad) * 32);
    // ---- plan every launch first: conv0 (the call's first launch) sets the tile-queue heads of the four stage kernels
    for (int st = 0; st < 5; ++st) {
        Launch& l = L[st];
        l.y0 = layers ? top : std::max(0, top - margin[st]);
        l.y1 = layers ? bot : std::min(H, bot + margin[st]);
        const int rows = l.y1 - l.y0;
        // Tile classes of the launch (sr_kernels.h TileGrid): `ty8` rows of 8-row tiles, then `ty4` rows of 4-row tiles.
        // Measured on MI355X (profiles/r3_tileplans_*, r3_queuefix_*; `rounds` = 8-row tiles per resident workgroup):
        //  * exact f32, a SMALL launch (rounds < 2): 4-row tiles on the first form of the stage kernel -- one tile per
        //    workgroup, no queue (256x256: 0.157 ms against 0.176-0.178 on either tile class of the pipe form);
        //  * split-half mode, small launch: the pipe form all the same (256x256: 0.091 ms against 0.24), 8-row tiles while
        //    they still give every CU one, else 4-row tiles;
        //  * otherwise the pipe form on 8-row tiles, and in exact f32 the LAST tiles of every XCD's queue are 4-row tiles, one per
        //    resident workgroup (sr_set_experiment "tail"): a persistent launch ends when its slowest workgroup does, and with
        //    half-size last tiles (and stealing) the workgroups finish closer together -- also when the tile count is an exact
        //    multiple of the workgroups (1024x768, 6.00 rounds: -4.9 %).  Measured, interleaved A/B (profiles/r3_tail_ab.txt):
        //    800x600 -4.7 %, 1000x600 -4.2 %, 1280x720 -2.4 %, 1920x1200 -1.4 %, the 8- / 4- / 2-way band of 3840x2160 -2.1 /
        //    -0.3 / -0.4 %, 2560x1440 -0.4 %, 3840x2160 0; but 512x512 (2.0 rounds) +5...9 % and 640x480 (2.3) +2 %: the 4-row
        //    tile body is code the launch would otherwise never touch (~30 us of cold instruction fetch per call), so no tail below
        //    3 rounds (round 3 also excluded launches above 14 rounds with a last round more than 70 % full -- 1920x1080, 15.8: 0 then; see below).  The
        //    split-half mode pays 17 % per 4-row tile (its B operands are re-read per tile row) and keeps 8-row tiles.
        //  conv0 and the first form run one class.
        const long tiles8 = (long)n * tiles_x * ((rows + 7) / 8);
        // (the last stage of factor 4 in the split-half mode exists with 4-row tiles only: two N-tiles of accumulators, sr_kernels.hip kBigTiles)
        const int forced = (st == 4 && c->factor == 4 && c->precision == SR_PRECISION_SPLIT_F16) ? 4 : c->env_th[st];
        const bool small_launch = tiles8 < 2L * resident;
        const bool split = c->precision == SR_PRECISION_SPLIT_F16;
        // (round 4: with the scalar overheads of the pipe form gone it also wins where every workgroup has exactly ONE 4-row tile and the
        // node has several sources -- their tiles arrive under the previous source's taps instead of between them: 256x256 stages 2 / 3
        // 40.5 / 49.0 -> 39.1 / 47.2 us; with more than one round of small tiles the first form still leads, 384x384 0.356 against 0.382 ms:
        // profiles/r4_ab_small_pipe.txt)
        const bool one_small_round = small_launch && (long)n * tiles_x * ((rows + 3) / 4) <= resident;
        l.pipe = st > 0 && c->env_pipe != 0 && (c->env_pipe == 2 || !small_launch || split || (st >= 2 && one_small_round));
        l.ty8 = (rows + 7) / 8; l.ty4 = 0;
        if (forced == 4 || (!forced && small_launch && (!split || tiles8 < cus))) {
            l.ty8 = 0; l.ty4 = (rows + 3) / 4;
        } else if (!forced && l.pipe && !small_launch) {
            const double rounds = (double)tiles8 / resident;
            float tail = c->env_tail;
            // (round 4, after the matrix stream lost its scalar overheads and 4-row tiles became relatively cheaper -- interleaved A/B,
            // profiles/r4_fork_tail.jsonl: the tail now also pays where round 3 excluded it, above 14 rounds with a nearly full last round
            // (1920x1080 undivided: 4.066 -> 4.052 ms); but NOT in the two bands of a forked call, whose launches run side by side and
            // end staggered anyway: 1920x1080 4.048 -> 4.018, 1600x900 2.853 -> 2.825, 1280x720 1.841 -> 1.834 ms without it)
            // (round 6: not in the exact mode's last stage either -- its 8-row tiles run on 4x4x1 MFMAs with 28 output columns, its 4-row
            // tiles still on 32x32x2 with 32, code the launch would otherwise never touch: 0.7775 -> 0.7695 ms at 1080p, profiles/r6_ab_quad.txt)
            if (tail < 0.0f) tail = (!split && rounds >= 3.0 && !forked && st != 4) ? 1.0f : 0.0f;
            if (tail > 0.0f) {
                const long per_row = (long)n * tiles_x;
                const int want = (int)((tail * resident + per_row - 1) / per_row);  // tile rows of small tiles
                l.ty8 = std::max(0, (rows - 4 * want) / 8);
                l.ty4 = std::max(0, (rows - 8 * l.ty8 + 3) / 4);
            }
        }
        if (!forced && l.pipe && !small_launch && !split && l.ty4 == 0 && (double)tiles8 / resident >= 3.0 && rows % 8 >= 1 && rows % 8 <= 4) {
            // the last 1-4 rows as ONE row of 4-row tiles instead of a mostly empty row of 8-row tiles (a band of a forked call, an image
            // height that is not a multiple of 8): half a tile row of matrix work saved
            l.ty8 = rows / 8; l.ty4 = 1;
        }
        if (l.ty8 > 0 && l.ty4 > 0 && (long)n * tiles_x * (l.ty8 + l.ty4) <= resident) {
            // (cannot happen with the rules above -- a tail is only added to launches of >= 2 rounds -- but a launch with a workgroup
            // per tile hands out tiles by workgroup number alone, which is only a bijection for ONE tile class)
            l.ty8 = 0; l.ty4 = (rows + 3) / 4;
        }
        l.th = l.ty8 > 0 ? 8 : 4;  // the one class of conv0 / the first form
        if (!l.pipe && l.ty8 > 0) { l.ty8 = (rows + 7) / 8; l.ty4 = 0; }
        const int ntiles = n * tiles_x * (l.ty8 + l.ty4);
        // the pipe form is persistent: one workgroup per resident slot, tiles from the queue; the first form one per tile
        l.grid = l.pipe ? std::min(ntiles, resident) : ntiles;
    }
    return SR_OK;
}

int StackJob::launch(int st) const {
    const int cus = c->cus > 0 ? c->cus : 256;
    const float* P = c->d_params;
    const int bw = c->env_bw >= 0 ? c->env_bw : kAutoBlockWidth;
    const Launch& l = L[st];
    const int y0 = l.y0, y1 = l.y1;
    if (st == 0) {
        auto rows = [&](int ya, int yb) -> int {  // f rows [ya, yb)
            if (ya >= yb) return SR_OK;
            const int tiles_y = (yb - ya + l.th - 1) / l.th;
            Conv0Args a{};
            a.img = d_img; a.wpack = P + c->off_w0; a.wpack_split = P + c->off_w0h; a.bias = P + c->off_bias[0]; a.beta = P + c->off_beta[0];
            a.dst = feat[0]; a.H = H; a.W = W; a.img_ch = img_ch;
            a.pitch = ws->pitch; a.img_stride = ws->img_stride;
            a.y_begin = ya; a.y_end = yb; a.tiles_x = tiles_x; a.tiles_y = tiles_y;
            a.div_tpi = make_tile_div((uint32_t)(tiles_x * tiles_y)); a.div_tx = make_tile_div((uint32_t)tiles_x);
            a.n_tiles = n * tiles_x * tiles_y;
            a.queue_reset = ws->d_queue;  // (every stage-0 launch of the call sets the same heads: the stage kernels follow them all)
            a.domain = c->d_domain;
            for (int k = 1; k < 5; ++k) a.queue_grid[k] = L[k].grid;
            // (grid: 8 workgroups per CU walking the tiles with a fixed stride; measured with 8 / 12 / 16 / 32 per CU, one per tile, and
            // a grid that divides the tile count evenly: conv0's time does not depend on it)
            HIPCHK(c, sr_launch_conv0(a, l.th, c->precision, std::min(a.n_tiles, 8 * cus), img_u8, s));
            return SR_OK;
        };
        if (!gate || !gate->ready) return rows(y0, y1);
        // Interior first (sr_halo_gate): f row y reads image rows y - 2 .. y + 2; those that touch none of the rows still on their way
        // are launched now, the rest -- at most 2 + the stage's margin rows either side -- behind the wait.
        const int lo = late_top > 0 ? std::max(y0, late_top + 2) : y0, hi = late_bot > 0 ? std::min(y1, H - late_bot - 2) : y1;
        int rc = lo < hi ? rows(lo, hi) : SR_OK;
        if (rc != SR_OK) return rc;
        if (mark && gate->mark[0]) HIPCHK(c, hipEventRecord(gate->mark[0], s));
        HIPCHK(c, hipStreamWaitEvent(s, gate->ready, 0));
        if (mark && gate->mark[1]) HIPCHK(c, hipEventRecord(gate->mark[1], s));
        if (lo >= hi) return rows(y0, y1);
        rc = rows(y0, lo);
        return rc != SR_OK ? rc : rows(hi, y1);
    }
    StageArgs a{};
    float* f = feat[0]; float* l1 = feat[1]; float* l2 = feat[2]; float* l3 = feat[3];
    a.pitch = ws->pitch; a.img_stride = ws->img_stride;
    switch (st) {
        case 1: a.src[0] = f; a.dst = l1; break;
        case 2: a.src[0] = f; a.src[1] = l1; a.dst = l2; break;
        case 3: a.src[0] = f; a.src[1] = l1; a.src[2] = l2; a.dst = l3; break;
        case 4: a.src[0] = l1; a.src[1] = l2; a.src[2] = l3; a.img = d_img; a.out = d_out; break;
    }
    a.wpack = P + (c->precision ? c->off_wh[st] : c->off_w[st]); a.bias = P + c->off_bias[st];
    a.beta = st < 4 ? P + c->off_beta[st] : nullptr;
    a.H = H; a.W = W; a.img_ch = img_ch;
    a.y_begin = y0; a.y_end = y1; a.tiles_x = tiles_x;
    a.n_img = n; a.queue = ws->d_queue + st * 8; a.domain = c->d_domain;
    a.grid[0] = make_tile_grid(8, y0, l.ty8, tiles_x, n, bw);
    a.grid[1] = make_tile_grid(4, y0 + 8 * l.ty8, l.ty4, tiles_x, n, bw);
    if (l.pipe) HIPCHK(c, sr_launch_stage_pipe(st, c->factor, a, c->precision, l.grid, img_u8, out_u8, s));
    else HIPCHK(c, sr_launch_stage(st, c->factor, a, l.th, c->precision, l.grid, img_u8, out_u8, s));
    return SR_OK;
}

}  // namespace

// The whole conv stack on device buffers.  Rows [halo_top, H - halo_bot) of each
// of the n images are produced; each earlier stage computes just the extra rows
// the later ones read (f +-5, l1 +-3, l2 +-2, l3 +-1 around the band).
int sr_run_stack(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int n, int H, int W, int halo_top,
                 int halo_bot, void* d_out, bool out_u8, hipStream_t s, int slot, const sr_halo_gate* gate) {
    if (!c || !d_img || !d_out || slot < 0 || slot > 1) return SR_E_INVALID;
    sr_device_guard restore_device;
    if (n <= 0 || H <= 0 || W <= 0) return SR_E_INVALID;
    if (img_u8 && img_ch != 3 && img_ch != 4) return SR_E_INVALID;
    if (c->graph != SR_GRAPH_SR_NET) {  // bilinear_net / downsample_net: one elementwise kernel
        if (halo_top || halo_bot || gate) return SR_E_INVALID;
        if (c->graph == SR_GRAPH_DOWNSAMPLE && (H < 3 || W < 3)) return SR_E_INVALID;
        if (img_u8 != out_u8) return SR_E_INVALID;
        HIPCHK(c, hipSetDevice(c->device));
        AuxArgs a{d_img, d_out, n, H, W, img_ch, c->d_qtab};
        HIPCHK(c, sr_launch_aux(c->graph, a, img_u8, out_u8, s));
        c->last_h = H; c->last_w = W;
        return SR_OK;
    }
    if ((halo_top != 0 && halo_top < SR_HALO) || (halo_bot != 0 && halo_bot < SR_HALO)) return SR_E_HALO;
    if (halo_top < 0 || halo_bot < 0 || halo_top + halo_bot >= H) return SR_E_INVALID;
    if ((halo_top || halo_bot) && n != 1) return SR_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    // s == nullptr is HIP's legacy default stream (what torch's default stream is);
    // the context's own non-blocking stream is used only by the host-pointer entry points.
    StackJob job;
    job.c = c; job.ws = &c->ws[slot]; job.d_img = d_img; job.d_out = d_out; job.img_u8 = img_u8; job.out_u8 = out_u8;
    job.img_ch = img_ch; job.n = n; job.H = H; job.W = W; job.top = halo_top; job.bot = H - halo_bot; job.s = s;
    if (gate) { job.gate = gate; job.late_top = gate->top; job.late_bot = gate->bot; job.mark = true; }
    int rc = job.prepare();
    if (rc != SR_OK) return rc;
    const bool prof = c->profiling;
    static const bool trace_stages = [] { const char* e = getenv("SRHIP_TRACE"); return e && atoi(e) >= 2; }();
    if (prof) HIPCHK(c, hipEventRecord(c->ev[0], s));
    for (int st = 0; st < 5; ++st) {
        rc = job.launch(st);
        if (rc != SR_OK) return rc;
        if (prof) HIPCHK(c, hipEventRecord(c->ev[st + 1], s));
        if (trace_stages) {  // SRHIP_TRACE=2: which launch a hang or a fault belongs to
            const StackJob::Launch& l = job.L[st];
            fprintf(stderr, "[srhip] stage %d launched: rows [%d,%d) th8 x%d th4 x%d grid %d %s ... ", st, l.y0, l.y1, l.ty8, l.ty4, l.grid, l.pipe ? "pipe" : "first");
            const hipError_t e = hipStreamSynchronize(s);
            fprintf(stderr, "%s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
        }
    }
    c->last_h = H; c->last_w = W;
    if (prof) {
        c->band_pending = false;
        HIPCHK(c, hipEventSynchronize(c->ev[5]));
        float ms = 0;
        for (int st = 0; st < 5; ++st) {
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[st], c->ev[st + 1]));
            c->stage_ms[st] = ms;
        }
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[5]));
        c->total_ms = ms;
    }
    return SR_OK;
}

struct sr_band_pass {
    StackJob job;
};

int sr_band_pass_begin(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int H, int W, int halo_top, int halo_bot, void* d_out, bool out_u8,
                       hipStream_t s, bool layers, const sr_halo_gate* gate, sr_band_pass** out) {
    if (!c || !d_img || !d_out || !out || c->graph != SR_GRAPH_SR_NET || H <= 0 || W <= 0) return SR_E_INVALID;
    *out = nullptr;
    if (img_u8 && img_ch != 3 && img_ch != 4) return SR_E_INVALID;
    if ((halo_top != 0 && halo_top < SR_HALO) || (halo_bot != 0 && halo_bot < SR_HALO)) return SR_E_HALO;
    if (halo_top < 0 || halo_bot < 0 || halo_top + halo_bot >= H) return SR_E_INVALID;
    sr_device_guard restore_device;
    HIPCHK(c, hipSetDevice(c->device));
    sr_band_pass* p = new (std::nothrow) sr_band_pass();
    if (!p) return SR_E_NOMEM;
    StackJob& job = p->job;
    job.c = c; job.ws = &c->ws[0]; job.d_img = d_img; job.d_out = d_out; job.img_u8 = img_u8; job.out_u8 = out_u8;
    job.img_ch = img_ch; job.n = 1; job.H = H; job.W = W; job.top = halo_top; job.bot = H - halo_bot; job.s = s; job.layers = layers;
    if (gate) { job.gate = gate; job.late_top = gate->top; job.late_bot = gate->bot; job.mark = true; }
    const int rc = job.prepare();
    if (rc != SR_OK) { delete p; return rc; }
    c->band_pending = false;
    c->last_h = c->last_w = 0;  // (with `layers` the maps hold this band's own rows and its neighbours' edge rows: not an image sr_read_feature could return)
    *out = p;
    return SR_OK;
}

int sr_band_pass_stage(sr_band_pass* p, int st) {
    if (!p || st < 0 || st > 4) return SR_E_INVALID;
    sr_device_guard restore_device;
    HIPCHK(p->job.c, hipSetDevice(p->job.c->device));
    return p->job.launch(st);
}

float* sr_band_pass_row(const sr_band_pass* p, int map, int y) {
    // pixel (y, -kFeatPad) of the map: in both layouts row y of the padded buffer begins (kFeatPad + y) x pitch pixels of 32 floats in
    return p->job.ws->d_feat[map] + (size_t)(kFeatPad + y) * p->job.ws->pitch * 32;
}

size_t sr_band_pass_row_floats(const sr_band_pass* p) { return (size_t)p->job.ws->pitch * 32; }

void sr_band_pass_end(sr_band_pass* p) { delete p; }

// One image as TWO row bands on two streams.  Each band carries the SR_HALO rows of the other it needs and is
// bit-identical to the undivided pass like every band; the second runs on the context's own second stream and workspace, forked
// from the caller's stream by an event and joined back by another, the launches of the two issued alternately -- the call stays
// asynchronous and ordered on the caller's stream.  What it buys, measured: the dispatcher runs the two bands' stage launches side
// by side (a workgroup of either per CU), so their ramps -- every workgroup's first gather from a cold L2, the partly filled last
// round of tiles -- overlap the other band's matrix work; against that stand ten launches instead of five, 14 recomputed rows
// and ~17 us of join / fork between consecutive calls.  Net, with the bands' launches free of 4-row tails: -0.8 ... -3.8 % for
// exact-f32 images from 3.5 rounds of tiles on, -0.1 % at 3840x2160; a loss in the split-half mode: hence the rule below.  Costs a
// second set of feature maps (each band's are half the size).
namespace {

// Whether a device call of this shape runs as two bands (the automatic rule or sr_set_experiment("fork")), and the first band's rows.
// `mode`: c->env_fork (-1: the automatic rule, 0: never, 1: always, > 1: always, that many rows first) -- or the tuner's 0 / 1.
bool plan_fork(const sr_ctx* c, int mode, bool img_u8, int img_ch, int n, int H, int W, int halo_top, int halo_bot, int* rows_a_out) {
    const int own = H - halo_top - halo_bot;
    bool fork = c->graph == SR_GRAPH_SR_NET && n == 1 && !c->profiling && mode != 0 && W > 0 && own >= 4 * SR_HALO &&
                (!img_u8 || img_ch == 3 || img_ch == 4) &&   // (anything sr_run_stack would refuse is left for it to refuse)
                halo_top >= 0 && halo_bot >= 0 && (halo_top == 0 || halo_top >= SR_HALO) && (halo_bot == 0 || halo_bot >= SR_HALO);
    if (fork && mode < 0) {
        // automatic: where the launches have enough rounds of tiles for two bands to fill the chip each (measured)
        const int cus = c->cus > 0 ? c->cus : 256;
        const double rounds = (double)((W + 31) / 32) * ((own + 7) / 8) / (2.0 * cus);
        // measured, interleaved in one process (scripts/fork_ab.py).  With the bands' launches free of 4-row tails (StackJob::prepare) the fork
        // wins wherever a band still has a few rounds of tiles, exact f32 (profiles/r4_fork_ab_f32_final_rule.jsonl): 800x600 (3.7 rounds)
        // -3.8 %, 1280x720 -1.6 %, 1920x1080 -0.8 %, 1920x1200 -0.9 %, 2560x1440 -0.4 %, 3840x2160 -0.1 %, a 276-row band of a 3840-wide
        // image -0.4 %; 960x540 (4.0 rounds) ties.  The split-half mode (tiles of 14 us) loses 0.6-2.4 % (r4_fork_ab_split.jsonl).
        fork = c->precision == SR_PRECISION_F32 && rounds >= c->fork_min_rounds && rounds < c->fork_max_rounds;
    }
    if (!fork) return false;
    // The first band's own rows: near the requested share, at the cut (within +-8 rows of it) that wastes the least matrix work in
    // partly filled tile rows.  Stage s of the first band computes rows_a + margin rows from the band's top, of the second band
    // own - rows_a + margin rows; a remainder of 1-4 rows costs a row of 4-row tiles (0.52 of an 8-row one), 5-7 rows a full one.
    int rows_a = mode > 1 ? mode : (int)(own * c->fork_share);
    rows_a = std::max(2 * SR_HALO, std::min(rows_a, own - 2 * SR_HALO));
    if (mode <= 1) {
        static const int margin[5] = {5, 3, 2, 1, 0};
        static const double weight[5] = {0.0, 25600.0, 34816.0, 44032.0, 28800.0};  // issued MACs per pixel of stages 1-4 (conv0: negligible)
        const bool fours = c->precision == SR_PRECISION_F32;
        auto tile_rows = [&](int rows) { const int r = rows % 8; return rows / 8 + (r == 0 ? 0.0 : (r <= 4 && fours) ? 0.52 : 1.0); };
        double best = 1e300;
        int best_rows = rows_a;
        for (int cand = rows_a - 8; cand <= rows_a + 8; ++cand) {
            if (cand < 2 * SR_HALO || own - cand < 2 * SR_HALO) continue;
            double cost = 0.0;
            for (int st = 1; st < 5; ++st) {
                const int ra = cand + margin[st] + std::min(halo_top, margin[st]), rb = own - cand + margin[st] + std::min(halo_bot, margin[st]);
                cost += weight[st] * (tile_rows(ra) + tile_rows(rb));
            }
            cost += 1e-3 * std::abs(cand - rows_a);  // ties: the cut nearest the requested share
            if (cost < best) { best = cost; best_rows = cand; }
        }
        rows_a = best_rows;
    }
    *rows_a_out = rows_a;
    return true;
}

// ---- the fork decision measured on the caller's own calls (sr_internal.h ForkTune) ----------------------------------------------
constexpr int kForkTuneBlock = 4;     // calls per plan: one dropped (the workspace meets a new geometry, first-use allocations), three timed
constexpr size_t kForkTuneShapes = 8;  // shapes remembered per context
constexpr double kForkTuneMinRounds = 0.55, kForkTuneMaxRounds = 12.0;  // outside: the rule (256x256 = 0.5 rounds: never; 1920x1080 = 15.8: exact f32 always)
constexpr float kForkTuneGain = 0.985f;  // the fork must win by 1.5 % to be chosen: it costs a second workspace

bool fork_tunable(const sr_ctx* c, int n, int H, int W, int halo_top, int halo_bot, const sr_halo_gate* gate) {
    if (!c->fork_autotune || c->env_fork >= 0 || c->graph != SR_GRAPH_SR_NET || n != 1 || c->profiling || gate || W <= 0) return false;
    const int own = H - halo_top - halo_bot;
    if (own < 4 * SR_HALO) return false;
    const int cus = c->cus > 0 ? c->cus : 256;
    const double rounds = (double)((W + 31) / 32) * ((own + 7) / 8) / (2.0 * cus);
    return rounds >= kForkTuneMinRounds && rounds < kForkTuneMaxRounds;
}

void fork_tune_release(sr_ctx::ForkTune& t) {
    for (auto& e : t.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
}

// The entry of this shape (created, with its two timing events, on first sight; the one used longest ago makes room).  Null: no tuning.
sr_ctx::ForkTune* fork_tune_entry(sr_ctx* c, bool img_u8, bool out_u8, int img_ch, int H, int W, int halo_top, int halo_bot) {
    ++c->fork_tune_clock;
    for (auto& t : c->fork_tune)
        if (t.H == H && t.W == W && t.top == halo_top && t.bot == halo_bot && t.img_u8 == img_u8 && t.out_u8 == out_u8 &&
            t.img_ch == (img_u8 ? img_ch : 3) && t.precision == c->precision) { t.used = c->fork_tune_clock; return &t; }
    if (c->fork_tune.size() >= kForkTuneShapes) {
        size_t oldest = 0;
        for (size_t k = 1; k < c->fork_tune.size(); ++k) if (c->fork_tune[k].used < c->fork_tune[oldest].used) oldest = k;
        fork_tune_release(c->fork_tune[oldest]);  // (an event still in flight is released when it completes: hipEventDestroy's contract)
        c->fork_tune.erase(c->fork_tune.begin() + (long)oldest);
    }
    sr_ctx::ForkTune t;
    t.H = H; t.W = W; t.top = halo_top; t.bot = halo_bot; t.img_u8 = img_u8; t.out_u8 = out_u8; t.img_ch = img_u8 ? img_ch : 3; t.precision = c->precision;
    t.used = c->fork_tune_clock;
    for (auto& e : t.ev)
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); fork_tune_release(t); return nullptr; }
    c->fork_tune.push_back(t);
    return &c->fork_tune.back();
}

// Read the sample in flight if it has finished -- a query, never a wait.  A sample that cannot be read (a capturing stream, a device
// error) ends the measurement in favour of the rule.
void fork_tune_harvest(sr_ctx::ForkTune& t, int rule) {
    if (!t.pending) return;
    const hipError_t q = hipEventQuery(t.ev[1]);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return; }
    float ms = 0.f;
    if (q != hipSuccess || hipEventElapsedTime(&ms, t.ev[0], t.ev[1]) != hipSuccess || !(ms > 0.f)) {
        (void)hipGetLastError();
        t.pending = false; t.decided = rule;
        return;
    }
    t.pending = false;
    const int plan = t.taken < kForkTuneBlock ? 0 : 1;
    if (t.taken % kForkTuneBlock != 0) t.best[plan] = std::min(t.best[plan], ms);
    if (++t.taken >= 2 * kForkTuneBlock) t.decided = t.best[1] < kForkTuneGain * t.best[0] ? 1 : 0;
}

}  // namespace

void sr_fork_tune_clear(sr_ctx* c) {
    for (auto& t : c->fork_tune) fork_tune_release(t);
    c->fork_tune.clear();
}

int sr_ensure_fork_resources(sr_ctx* c) {
    if (!c->stream2) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    for (auto& e : c->ev_fork) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return SR_OK;
}

int sr_run_stack_auto(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int n, int H, int W, int halo_top, int halo_bot,
                      void* d_out, bool out_u8, hipStream_t s, const sr_halo_gate* gate) {
    if (!c || !d_img || !d_out) return SR_E_INVALID;
    c->band_pending = false;  // (the event pairs of an earlier sharded call are not this call's: sr_comm.cpp sets the flag again behind its own)
    int rows_a = 0;
    sr_device_guard restore_device;
    int mode = c->env_fork;
    sr_ctx::ForkTune* tune = nullptr;
    bool sample = false;
    if (fork_tunable(c, n, H, W, halo_top, halo_bot, gate)) {
        HIPCHK(c, hipSetDevice(c->device));
        tune = fork_tune_entry(c, img_u8, out_u8, img_ch, H, W, halo_top, halo_bot);
        if (tune) {
            int rows_rule = 0;
            const int rule = plan_fork(c, -1, img_u8, img_ch, n, H, W, halo_top, halo_bot, &rows_rule) ? 1 : 0;
            if (tune->decided < 0) fork_tune_harvest(*tune, rule);
            if (tune->decided >= 0) {
                mode = tune->decided;
            } else {
                mode = tune->taken < kForkTuneBlock ? 0 : 1;  // this block's plan; timed unless the last sample is still in flight
                sample = !tune->pending;
            }
        }
    }
    if (sample && hipEventRecord(tune->ev[0], s) != hipSuccess) { (void)hipGetLastError(); sample = false; }
    auto sampled = [&](int rc) {  // the closing event of a timed call, on the caller's stream (behind the join of a forked one)
        if (sample && rc == SR_OK) {
            if (hipEventRecord(tune->ev[1], s) == hipSuccess) tune->pending = true; else (void)hipGetLastError();
        }
        return rc;
    };
    if (!plan_fork(c, mode, img_u8, img_ch, n, H, W, halo_top, halo_bot, &rows_a)) {
        if (tune && tune->decided < 0 && mode == 1) tune->decided = 0;  // (cannot be forked at all)
        return sampled(sr_run_stack(c, d_img, img_u8, img_ch, n, H, W, halo_top, halo_bot, d_out, out_u8, s, 0, gate));
    }
    HIPCHK(c, hipSetDevice(c->device));
    {
        const int rc = sr_ensure_fork_resources(c);
        if (rc != SR_OK) return rc;
    }
    const int cut = halo_top + rows_a;  // first row of the second band, in the coordinates of the caller's buffer
    const size_t in_px = img_u8 ? (size_t)img_ch : 3 * sizeof(float), out_px = out_u8 ? 4 : 3 * sizeof(float);
    const int f = c->factor;
    StackJob a, b;
    a.forked = b.forked = true;
    a.c = b.c = c; a.img_u8 = b.img_u8 = img_u8; a.out_u8 = b.out_u8 = out_u8; a.img_ch = b.img_ch = img_ch; a.W = b.W = W;
    a.ws = &c->ws[0]; a.s = s;
    a.d_img = d_img; a.d_out = d_out; a.H = cut + SR_HALO; a.top = halo_top; a.bot = cut;
    b.ws = &c->ws[1]; b.s = c->stream2;
    b.d_img = (const char*)d_img + (size_t)(cut - SR_HALO) * W * in_px;
    b.d_out = (char*)d_out + (size_t)rows_a * f * W * f * out_px;
    b.H = H - (cut - SR_HALO); b.top = SR_HALO; b.bot = b.H - halo_bot;
    if (gate) {  // the first band holds the late rows at the image's top, the second those at its bottom; the caller's stream is the one whose wait is timed
        a.gate = b.gate = gate; a.late_top = gate->top; b.late_bot = gate->bot; a.mark = true;
    }
    HIPCHK(c, hipEventRecord(c->ev_fork[0], s));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork[0], 0));
    // Both bands are planned (and their workspaces allocated) before anything is launched: should the second set of feature maps not fit,
    // the undivided pass on the first workspace still may -- the call must not fail for want of an optimisation's memory.
    int rc = a.prepare();
    if (rc != SR_OK) return rc;  // (the first band's maps do not fit: the undivided pass, which needs larger ones, cannot either)
    rc = b.prepare();
    if (rc == SR_E_NOMEM) {  // the second workspace is the optimisation's: give back what of it exists and run undivided on the first
        for (auto& p : c->ws[1].d_feat) { if (p) (void)hipFree(p); p = nullptr; }
        c->ws[1].feat_cap_px = 0; c->ws[1].geo_n = 0;
        if (tune && tune->decided < 0) tune->decided = 0;  // (no room for the fork's second workspace)
        sample = false;
        return sr_run_stack(c, d_img, img_u8, img_ch, n, H, W, halo_top, halo_bot, d_out, out_u8, s, 0, gate);
    }
    for (int st = 0; st < 5 && rc == SR_OK; ++st) {
        rc = a.launch(st);
        if (rc == SR_OK) rc = b.launch(st);
    }
    // join also on failure: whatever was queued on the second stream must be ordered before the caller's next work
    const hipError_t e1 = hipEventRecord(c->ev_fork[1], c->stream2);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, c->ev_fork[1], 0) : e1;
    if (rc != SR_OK) return rc;
    HIPCHK(c, e1); HIPCHK(c, e2);
    c->last_h = c->last_w = 0;  // each workspace holds one band: sr_read_feature refuses
    return sampled(SR_OK);
}

namespace {

// One unit of the host pipeline: some whole images of a batch, or a row band of a single image
// with the halo rows it needs (band == untiled bit for bit, see sr_upscale_band_*).
struct Chunk {
    size_t in_off, in_bytes, out_off, out_bytes;  // of the chunk's FIRST image / of the band, in the caller's buffers
    int n, h_ext, halo_top, halo_bot;
    size_t in_step, out_step;                     // n > 1: distance between consecutive images of the chunk in the caller's
                                                  // buffers (= the image size for a contiguous batch, stride x that for a deal)
};

// Which images of the caller's batch a call processes: first, first + stride, ... (count of them).  A plain call is
// {0, 1, n}; a context's share of a round-robin deal over N contexts is {k, N, ceil((n - k) / N)}.
struct Deal {
    int first, stride, count;
};

// Split the job.  Batches go in chunks of ~1M px of whole images.  A single large sr_net image goes as row bands: band k
// owns rows [y0,y1) and carries them plus SR_HALO rows on every side that is not an image edge.  Bands may differ in
// height -- a workspace that meets a new geometry only has its border re-cleared (ensure_features), microseconds.
// [y_lo, y_hi): the image rows this call is to produce (a whole image: 0, h; a device's share of a multi-GPU
// call: its rows, sr_net and n == 1 only).
std::vector<Chunk> plan_chunks(const sr_ctx* c, Deal deal, int h, int w, size_t in_px_bytes, size_t out_px_bytes, int y_lo, int y_hi,
                               bool* in_order) {
    std::vector<Chunk> plan;
    *in_order = false;
    const int f = c->factor, n = deal.count;
    const size_t in_img = (size_t)h * w * in_px_bytes;
    const size_t out_img = c->graph == SR_GRAPH_DOWNSAMPLE ? (size_t)(h / 3) * (w / 3) * out_px_bytes
                                                           : (size_t)h * f * w * f * out_px_bytes;
    const size_t in_step = in_img * deal.stride, out_step = out_img * deal.stride;
    const bool pipe = c->pipeline && !c->profiling;  // per-stage profiling times one undivided pass
    const int per = (int)std::max<size_t>(1, ((size_t)1 << 20) / ((size_t)h * w));  // images per chunk: ~1M px of work
    if (pipe && n > per) {
        for (int i = 0; i < n; i += per) {
            const int m = std::min(per, n - i);
            const size_t img = (size_t)deal.first + (size_t)i * deal.stride;
            plan.push_back({img * in_img, m * in_img, img * out_img, m * out_img, m, h, 0, 0, in_step, out_step});
        }
        return plan;
    }
    const bool part = y_lo > 0 || y_hi < h;  // a share of the image: always in band form (halo rows from the image itself)
    const int span = y_hi - y_lo;
    std::vector<int> rows;  // rows of each band, top to bottom
    int forced_sum = 0;
    for (int rk : c->env_rows) forced_sum += rk;
    const bool forced_plan = (!c->env_rows.empty() && forced_sum == span) || c->env_bands > 0;  // (sr_set_experiment "rows" / "bands": at any size)
    // Below 2^19 px a lone frame used to go as ONE chunk: upload, kernels, download, nothing overlapping.  Round 6, u8 output, from
    // ~200K px on: TWO bands in order on one stream -- the first band's download runs under the second band's kernels, which is worth more
    // than the second band's 7 recomputed rows and five launches cost (scripts/host_plan_sweep.py, profiles/r6_host_mid_plans.txt: exact f32,
    // 70 / 30: 640x480 0.970 -> 0.873 ms, 854x480 1.311 -> 1.138, 800x600 1.420 -> 1.330, 720x576 1.312 -> 1.186, 960x540 1.479 -> 1.342;
    // the split-half mode, whose kernels are shorter beside the same download, 60 / 40: 640x480 0.536 -> 0.468, 800x600 0.792 -> 0.660,
    // 960x540 0.837 -> 0.701; at 320x320 neither mode gains).  Equal bands on alternating streams are within 2 % of these on most shapes
    // and 6 % better on some (800x600), 6 % worse on others: the in-order plan is the even-tempered one.
    // (from 200K px in exact f32 -- 448x448 -5.4 %, 640x360 -6.5 %, 640x480 -5.5 % against one chunk measured alternately, but 430x419 +3 % -- and
    // from 180K px in the split-half mode: 430x419 -4 %, 448x448 -11 %)
    // f32 OUTPUT (three times the download, as long as the kernels or longer): in-order bands pay more still -- exact f32 60 / 40: 448x448
    // 0.982 -> 0.801 ms, 640x480 1.379 -> 1.143; three equal bands from 400K px: 800x600 2.186 -> 1.556, 960x540 2.253 -> 1.784; split-half
    // two equal bands: 448x448 0.686 -> 0.603, three from 300K px: 640x480 0.974 -> 0.845, 800x600 1.472 -> 1.234, 960x540 1.582 -> 1.328.
    const bool split_mode = c->precision == SR_PRECISION_SPLIT_F16;
    const size_t px_span = (size_t)span * w;
    // (f32 output, smaller frames: exact f32 60 / 40 at 384x384 0.737 -> 0.621, at 320x320 0.578 -> 0.500; split-half 50 / 50 at 384x384 0.523 -> 0.483,
    // at 320x320 -3 %: from 100K / 140K px)
    const size_t mid_lo = out_px_bytes == 4 ? (split_mode ? 180000u : 200000u) : (split_mode ? 140000u : 100000u);
    const bool mid_size = px_span >= mid_lo && px_span < ((size_t)1 << 19);
    if (pipe && n == 1 && c->graph == SR_GRAPH_SR_NET && (forced_plan || mid_size || (size_t)span * w >= ((size_t)1 << 19))) {
        // Kernel and download time per input pixel decide the shape of the plan (measured, page-locked buffers, PCIe 5 x16):
        const double kern_ns = (c->precision == SR_PRECISION_SPLIT_F16 ? 0.9 : 2.0) * (f == 4 ? 1.2 : 1.0);
        const double d2h_ns = (double)out_px_bytes * f * f / 52.0;
        const double rho = kern_ns / d2h_ns;
        int forced_rows = 0;
        for (int rk : c->env_rows) forced_rows += rk;
        if (!c->env_rows.empty() && forced_rows == span) {  // sr_set_experiment("rows"): exactly these bands
            rows = c->env_rows;
            *in_order = !c->env_rows_two;
        } else if (c->env_bands > 0) {  // sr_set_experiment("bands"): that many equal bands
            const int nb = std::min(c->env_bands, span / (2 * SR_HALO));
            for (int k = 0; k < nb; ++k) rows.push_back((span * (k + 1)) / nb - (span * k) / nb);
        } else if (mid_size) {
            const bool u8_out = out_px_bytes == 4;
            if (!u8_out && px_span >= (split_mode ? 300000u : 400000u) && span >= 6 * SR_HALO) {
                const int third = span / 3 / 8 * 8;
                rows = {third, third, span - 2 * third};
            } else {
                const double share = u8_out ? (split_mode ? 0.6 : 0.7) : (split_mode ? 0.5 : 0.6);
                const int first = (int)(span * share) / 8 * 8;
                if (first >= 2 * SR_HALO && span - first >= 2 * SR_HALO) rows = {first, span - first};
            }
            *in_order = !rows.empty();
        } else {
            // Compute-bound (f32 arithmetic, u8 output: rho = 2.9): only the LAST band's download is exposed, and band
            // i's download hides under band i+1's kernels as long as band i+1 is at least 1/rho of it -- bands that
            // shrink geometrically, as many as keep the last one >= 300K px (smaller bands no longer fill the chip: at
            // 1080p the 128-row third band costs more than it hides, 4.95 against 4.85 ms with four equal bands).
            // They compute IN ORDER on one stream: on two, band 1 runs beside band 0, both finish late and the
            // largest download is the exposed one (1080p 5.54 ms).  Measured (scripts/geo_exp.py), geometric
            // against equal bands: 2560x1440 8.17 / 8.37 ms, 3840x2160 17.50 / 17.78 ms.
            const double r = std::min(rho * 0.85, 3.0);
            int nb = 1;
            double sum = 1.0, term = 1.0;
            while (rho >= 2.0 && c->env_geo && nb < 5 && (double)span * w / (sum + term * r) >= 300e3) { term *= r; sum += term; ++nb; }
            if (nb >= 3) {
                int left = span;
                for (int k = 0; k < nb - 1; ++k) {
                    int rk = (int)((double)span * term / sum) / 8 * 8;  // whole 8-row tiles
                    rk = std::max(2 * SR_HALO, std::min(rk, left - 2 * SR_HALO));
                    rows.push_back(rk);
                    left -= rk;
                    term /= r;
                }
                rows.push_back(left);
                *in_order = true;
            } else if (rho >= 2.0 && c->env_geo && (double)span * w >= 800e3) {
                // Compute-bound, but too small for three bands in order (720p .. ~2.8 M px): on alternating streams, two equal
                // bands that keep the chip full, then a tail -- the exposed download is the last band's, so that one is
                // ~150K px (smaller no longer pays its five launches), from 1.8 M px on with a band of 2.5x that in front of
                // it under which the second big band's download finishes.  Measured round 3 (scripts/host_plan_sweep.py,
                // profiles/r3_host_plans.txt), against the equal bands of round 2: 1920x1080 400,400,200,80 = 4.60 against
                // 4.89 ms; 1600x900 3.36 / 3.54; 1280x720 2.22 / 2.36.  2560x1440 is the geometric plan's either way.
                const int last = std::max(16, (int)(150e3 / w + 4.0) / 8 * 8);
                const int mid = (double)span * w >= 1.8e6 ? (5 * last / 2) / 8 * 8 : 0;
                const int big = (span - last - mid) / 2 / 8 * 8;
                if (big >= 2 * last) {
                    rows = {big, span - last - mid - big};
                    if (mid) rows.push_back(mid);
                    rows.push_back(last);
                }
            }
        }
        if (rows.empty() && !mid_size) {
            // Download-bound or balanced (f32 output, the split-half mode): equal bands.  Few expose the first upload
            // and the last download, many pay 14 recomputed rows and five launches each.  Measured, f32 1080p
            // 1 / 2 / 4 / 5 / 8 bands = 5.96 / 5.14 / 4.93 / 5.12 / 5.22 ms; the split-half mode computes 2.2x faster
            // than the bus drains its output and prefers more: 4 / 5 / 6 / 8 bands = 3.77 / 2.7-3.4 / 2.82 / 3.05 ms.
            // Round 6, f32 OUTPUT re-measured on this round's kernels (scripts/host_plan_sweep.py, profiles/r6_host_mid_plans.txt 6.): five equal bands
            // IN ORDER on one stream, eight from 3 M px, beat the equal bands on alternating streams of rounds 2-3 in a process of its own (a C / Rust
            // host) -- exact f32 1280x720 3.53 -> 2.71 ms, 1920x1080 6.25 -> 5.69, 3840x2160 21.1 -> 19.9; split-half 2.29 -> 2.17, 4.74 -> 4.60,
            // 17.8 -> 17.2 -- and are within +-6 % of them inside bench.py's long-lived torch process (exact f32 1920x1080 6.17 against 5.85, split-half
            // 4.60 against 4.93: 7.).
            // The split-half mode with u8 output keeps its equal bands on alternating streams: bands in order that shrink by 0.85 are 3-6 % faster
            // in a fresh process (1920x1080 2.30 -> 2.12 ms) but read 3.37 ms for the first dozens of calls inside bench.py's process -- no overlap at
            // all between the one compute stream and the download stream, which two compute streams never lose entirely (7.); not adopted.
            if (out_px_bytes != 4) {
                const int nb = std::min(px_span < 3000000u ? 5 : 8, std::max(1, span / (2 * SR_HALO)));
                for (int k = 0; k < nb; ++k) rows.push_back((span * (k + 1)) / nb - (span * k) / nb);
                *in_order = true;
            } else {
                int nb = span / (split_mode ? 176 : 256);
                if (nb > 8) nb = 8;
                for (int k = 0; k < nb; ++k) rows.push_back((span * (k + 1)) / nb - (span * k) / nb);
            }
        }
    }
    const size_t img0_in = (size_t)deal.first * in_img, img0_out = (size_t)deal.first * out_img;
    if (rows.size() >= 2) {
        bool ok = true;
        int y0 = y_lo;
        for (int rk : rows) {
            const int y1 = y0 + rk;
            const int start = std::max(0, y0 - SR_HALO), end = std::min(h, y1 + SR_HALO);
            const int ht = y0 - start, hb = end - y1;
            // sr_run_stack's rules: a halo is SR_HALO rows or, at an image edge, none
            ok = ok && rk > 0 && (ht == 0 || ht == SR_HALO) && (hb == 0 || hb == SR_HALO);
            plan.push_back({img0_in + (size_t)start * w * in_px_bytes, (size_t)(end - start) * w * in_px_bytes,
                            img0_out + (size_t)y0 * f * w * f * out_px_bytes, (size_t)rk * f * w * f * out_px_bytes, 1, end - start, ht, hb,
                            0, 0});
            y0 = y1;
        }
        if (ok && y0 == y_hi) return plan;
        plan.clear();
    }
    if (part) {  // one band: the rows themselves plus SR_HALO rows on every side that is not an image edge
        const int start = std::max(0, y_lo - SR_HALO), end = std::min(h, y_hi + SR_HALO);
        plan.push_back({img0_in + (size_t)start * w * in_px_bytes, (size_t)(end - start) * w * in_px_bytes,
                        img0_out + (size_t)y_lo * f * w * f * out_px_bytes, (size_t)span * f * w * f * out_px_bytes, 1, end - start,
                        y_lo - start, end - y_hi, 0, 0});
        return plan;
    }
    plan.push_back({img0_in, (size_t)n * in_img, img0_out, (size_t)n * out_img, n, h, 0, 0, in_step, out_step});
    return plan;
}

// Host-pointer entry points: upload, conv stack, download -- software-pipelined over chunks on
// three streams.  Issue order is H2D(i+1), kernels(i+1), D2H(i): with pageable caller memory the
// runtime blocks the calling thread inside each copy, and this order keeps kernels queued behind
// it; with pinned memory (sr_host_alloc) all three engines run concurrently.
// reserve: no caller buffers -- everything else of the call happens (allocations, events, the kernels on whatever the
// staging buffers hold), so that the first real call costs what every later one does (sr_reserve_*).
int run_host(sr_ctx* c, const void* in, bool img_u8, int img_ch, Deal deal, int h, int w, void* out, bool out_u8, int y_lo = 0,
             int y_hi = -1, bool reserve = false) {
    const int n = deal.count;
    if (!c || ((!in || !out) && !reserve) || n <= 0 || h <= 0 || w <= 0 || deal.first < 0 || deal.stride < 1) return SR_E_INVALID;
    if (img_u8 && img_ch != 3 && img_ch != 4) return SR_E_INVALID;
    if (c->graph == SR_GRAPH_DOWNSAMPLE && (h < 3 || w < 3)) return SR_E_INVALID;
    if (c->graph != SR_GRAPH_SR_NET && img_u8 != out_u8) return SR_E_INVALID;
    sr_device_guard restore_device;
    HIPCHK(c, hipSetDevice(c->device));
    c->band_pending = false;  // (an earlier sharded call's event pairs are not this call's timing)
    const size_t in_px = img_u8 ? (size_t)img_ch : 3 * sizeof(float), out_px = out_u8 ? 4 : 3 * sizeof(float);
    if (y_hi < 0) y_hi = h;
    if (y_lo < 0 || y_hi > h || y_lo >= y_hi) return SR_E_INVALID;
    if ((y_lo > 0 || y_hi < h) && (n != 1 || c->graph != SR_GRAPH_SR_NET)) return SR_E_INVALID;
    if (y_lo > 0 && y_lo < SR_HALO) return SR_E_HALO;
    if (y_hi < h && h - y_hi < SR_HALO) return SR_E_HALO;
    // A fault still standing in the context's domain word was raised by an EARLIER call -- an unchecked *_dev call (a context has one
    // caller: nothing of it is still running once that caller is here) -- and is that call's to report (sr_check_domain); it must
    // not make this call, whose values may all be in range, recompute in f32.  Set it aside.
    if (c->h_domain && *(volatile int*)c->h_domain) { c->dev_fault = true; *(volatile int*)c->h_domain = 0; }
    bool in_order = false;
    const std::vector<Chunk> plan = plan_chunks(c, deal, h, w, in_px, out_px, y_lo, y_hi, &in_order);
    const int nch = (int)plan.size();
    const int slots = nch > 1 ? 2 : 1;
    {
        const int rc = sr_ensure_streams(c, nch > 1);
        if (rc != SR_OK) return rc;
    }
    // One chunk: upload, kernels and download in order on `stream`.  Several: chunk i uploads on the stream its kernels
    // follow on (the other compute stream is busy with chunk i-1 meanwhile) -- or, when all chunks compute in order on
    // `stream`, on the idle `stream2` -- and downloads on `copy_out`.
    // A dedicated upload stream (a third hardware queue, 8-40 ms to create) lets chunk i+1 arrive while both compute streams
    // are busy.  Measured A/B (profiles/r2_upload_stream_ab.txt): worth 1-2 % of an exact-f32 1080p call, nothing at 4K, and
    // 2 % SLOWER in the split-half mode (whose calls are download-bound) -- so only exact-f32 contexts get one, and only from
    // their second pipelined call on: a one-shot process never pays for it, a service does once.
    if (nch > 1 && !reserve && c->precision == SR_PRECISION_F32 && ++c->pipelined_calls >= 2 && !c->copy_in)
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
    const bool own_upload = nch > 1 && c->copy_in && c->precision == SR_PRECISION_F32;
    hipStream_t down = nch > 1 ? c->copy_out : c->stream;
    size_t in_max = 0, out_max = 0;
    for (const Chunk& k : plan) { in_max = std::max(in_max, k.in_bytes); out_max = std::max(out_max, k.out_bytes); }
    for (int sl = 0; sl < slots; ++sl) {
        int rc = sr_ensure_buf(c, &c->d_in[sl], &c->in_cap[sl], in_max);
        if (rc == SR_OK) rc = sr_ensure_buf(c, &c->d_out[sl], &c->out_cap[sl], out_max);
        if (rc != SR_OK) {  // a job that does not fit must not leave its partial staging buffers behind (they may be most of the device)
            for (int k = 0; k < 2; ++k) {
                if (c->d_in[k]) (void)hipFree(c->d_in[k]);
                if (c->d_out[k]) (void)hipFree(c->d_out[k]);
                c->d_in[k] = c->d_out[k] = nullptr;
                c->in_cap[k] = c->out_cap[k] = 0;
            }
            return rc;
        }
    }
    // events per chunk: 0 upload begins, 1 upload done, 2 kernels begin, 3 kernels done, 4 download done
    while (c->pool.size() < (size_t)nch * 5) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreate(&e));
        c->pool.push_back(e);
    }
    auto ev = [&](int i, int k) { return c->pool[(size_t)i * 5 + k]; };
    const char* src = (const char*)in;
    char* dst = (char*)out;
    // a chunk's images are contiguous on the device; in the caller's buffers they are in_step / out_step apart
    auto copy_images = [&](const Chunk& k, bool up, int sl, hipStream_t on) -> int {
        if (reserve) return SR_OK;
        const bool contiguous = k.n == 1 || (up ? k.in_step == k.in_bytes / k.n : k.out_step == k.out_bytes / k.n);
        const int pieces = contiguous ? 1 : k.n;
        const size_t in_img = k.in_bytes / (contiguous ? 1 : k.n), out_img = k.out_bytes / (contiguous ? 1 : k.n);
        for (int j = 0; j < pieces; ++j) {
            if (up) HIPCHK(c, hipMemcpyAsync((char*)c->d_in[sl] + j * in_img, src + k.in_off + j * k.in_step, in_img, hipMemcpyHostToDevice, on));
            else HIPCHK(c, hipMemcpyAsync(dst + k.out_off + j * k.out_step, (const char*)c->d_out[sl] + j * out_img, out_img, hipMemcpyDeviceToHost, on));
        }
        return SR_OK;
    };
    // chunk i computes on stream i % 2 with workspace i % 2: consecutive chunks' stage launches overlap (the tail of one
    // launch -- its last, partly filled round of workgroups -- runs beside the head of the other stream's next launch;
    // measured at 1080p, one stream: five bands cost 19 % more kernel time than the undivided pass)
    // (a geometric band plan wants the opposite: band i finished, and downloading, before band i+1 takes the chip)
    auto cstream = [&](int i) { return (slots == 2 && (i & 1) && !in_order) ? c->stream2 : c->stream; };
    auto issue_front = [&](int i) -> int {  // upload + kernels of chunk i
        const Chunk& k = plan[i];
        const int sl = i % slots;
        hipStream_t cs = cstream(i);
        hipStream_t upl = own_upload ? c->copy_in : (in_order && nch > 1) ? c->stream2 : cs;
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(upl, ev(i - 2, 3), 0));  // slot's previous reader
        HIPCHK(c, hipEventRecord(ev(i, 0), upl));
        int rc = copy_images(k, true, sl, upl);
        if (rc != SR_OK) return rc;
        HIPCHK(c, hipEventRecord(ev(i, 1), upl));
        if (upl != cs) HIPCHK(c, hipStreamWaitEvent(cs, ev(i, 1), 0));
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(cs, ev(i - 2, 4), 0));   // slot's previous download
        HIPCHK(c, hipEventRecord(ev(i, 2), cs));
        rc = sr_run_stack(c, c->d_in[sl], img_u8, img_ch, k.n, k.h_ext, w, k.halo_top, k.halo_bot, c->d_out[sl],
                          out_u8, cs, sl);
        if (rc != SR_OK) return rc;
        HIPCHK(c, hipEventRecord(ev(i, 3), cs));
        return SR_OK;
    };
    auto issue_back = [&](int i) -> int {  // download of chunk i
        if (down != cstream(i)) HIPCHK(c, hipStreamWaitEvent(down, ev(i, 3), 0));
        const int rc = copy_images(plan[i], false, i % slots, down);
        if (rc != SR_OK) return rc;
        HIPCHK(c, hipEventRecord(ev(i, 4), down));
        return SR_OK;
    };
    int rc = issue_front(0);
    for (int i = 0; i < nch && rc == SR_OK; ++i) {
        if (i + 1 < nch) rc = issue_front(i + 1);
        if (rc == SR_OK) rc = issue_back(i);
    }
    // drain everything before returning, ALSO on failure: copies into / out of the caller's buffers and kernels on
    // the context's stream must not be in flight once the call has returned
    const hipError_t e1 = own_upload ? hipStreamSynchronize(c->copy_in) : hipSuccess, e2 = hipStreamSynchronize(c->stream), e4 = nch > 1 ? hipStreamSynchronize(c->stream2) : hipSuccess,
                     e3 = nch > 1 ? hipStreamSynchronize(c->copy_out) : hipSuccess;
    if (rc != SR_OK) return rc;
    HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e4); HIPCHK(c, e3);
    double h2d = 0, ker = 0, d2h = 0;
    for (int i = 0; i < nch; ++i) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, ev(i, 0), ev(i, 1))); h2d += ms;
        // chunks overlap on two compute streams: kernel time = first kernel start -> last kernel end
        HIPCHK(c, hipEventElapsedTime(&ms, ev(0, 2), ev(i, 3))); ker = std::max(ker, (double)ms);
        HIPCHK(c, hipEventElapsedTime(&ms, ev(i, 3), ev(i, 4))); d2h += ms;  // includes waiting for the copy engine
    }
    c->h2d_ms = h2d; c->total_ms = ker; c->d2h_ms = d2h; c->band_pending = false;
    c->last_chunks = nch;
    if (nch > 1) c->last_h = c->last_w = 0;  // the feature maps hold one chunk only: sr_read_feature refuses
    if (c->precision == SR_PRECISION_SPLIT_F16 && c->h_domain && *(volatile int*)c->h_domain) {
        // Some value of this call left the split-half mode's domain (every stream has drained: the flag is final).  A synchronous
        // call never hands out clamped pixels: the whole job is computed again in exact f32 -- graph.forward takes any f32 (main.rs:171).
        *(volatile int*)c->h_domain = 0;
        if (reserve) return SR_OK;
        (void)sr_set_precision(c, SR_PRECISION_F32);
        const int rc2 = run_host(c, in, img_u8, img_ch, deal, h, w, out, out_u8, y_lo, y_hi, false);
        (void)sr_set_precision(c, SR_PRECISION_SPLIT_F16);
        ++c->domain_fallbacks;
        return rc2;
    }
    return SR_OK;
}

// Contexts that cooperate on one call must be distinct objects computing the same function: same graph, factor,
// arithmetic mode and parameters -- or the "bit-identical to the single-device call" guarantee silently breaks
// (a seam between shares), and one context driven from two threads at once races on its workspace.
int check_context_set(sr_ctx* const* ctxs, int n_ctx) {
    if (!ctxs || n_ctx <= 0) return SR_E_INVALID;
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k] || ctxs[k]->graph != SR_GRAPH_SR_NET) return SR_E_INVALID;
        if (ctxs[k]->factor != ctxs[0]->factor || ctxs[k]->precision != ctxs[0]->precision ||
            ctxs[k]->params_hash != ctxs[0]->params_hash) return SR_E_INVALID;
        for (int j = 0; j < k; ++j) if (ctxs[j] == ctxs[k]) return SR_E_INVALID;
    }
    return SR_OK;
}

// A share on its own host thread -- or, should the system refuse another thread, right here: no exception may cross the C ABI
// with joinable threads behind it.
template <class F>
void spawn_or_run(std::vector<std::thread>& th, F fn) {
    try {
        th.emplace_back(fn);
    } catch (const std::exception&) {
        fn();
    }
}

}  // namespace

int sr_check_context_set(sr_ctx* const* ctxs, int n_ctx) { return check_context_set(ctxs, n_ctx); }

extern "C" {

int sr_upscale_f32_dev(sr_ctx* c, const float* d_in, int n, int h, int w, float* d_out, void* stream) {
    return sr_run_stack_auto(c, d_in, false, 3, n, h, w, 0, 0, d_out, false, (hipStream_t)stream);
}

int sr_upscale_rgba8_dev(sr_ctx* c, const uint8_t* d_in, int in_channels, int n, int h, int w,
                         uint8_t* d_out, void* stream) {
    return sr_run_stack_auto(c, d_in, true, in_channels, n, h, w, 0, 0, d_out, true, (hipStream_t)stream);
}

int sr_upscale_band_f32_dev(sr_ctx* c, const float* d_in, int h_ext, int w, int halo_top, int halo_bot,
                            float* d_out, void* stream) {
    return sr_run_stack_auto(c, d_in, false, 3, 1, h_ext, w, halo_top, halo_bot, d_out, false, (hipStream_t)stream);
}

int sr_upscale_band_rgba8_dev(sr_ctx* c, const uint8_t* d_in, int in_channels, int h_ext, int w,
                              int halo_top, int halo_bot, uint8_t* d_out, void* stream) {
    return sr_run_stack_auto(c, d_in, true, in_channels, 1, h_ext, w, halo_top, halo_bot, d_out, true,
                             (hipStream_t)stream);
}

// ... and what a DEVICE-pointer call of that shape would create on its first use: if it runs as two bands (plan_fork), the second
// stream, the fork / join events and both workspaces at the bands' geometry -- allocations and a stream creation that would
// otherwise happen, synchronously, inside the first "asynchronous" sr_upscale_*_dev call after the reserve.
static int reserve_fork(sr_ctx* c, bool img_u8, int img_ch, int n, int h, int w) {
    int rows_a = 0;
    // (a shape the fork tuner will measure runs both ways: what the forked calls need is created here too)
    if (!plan_fork(c, fork_tunable(c, n, h, w, 0, 0, nullptr) ? 1 : c->env_fork, img_u8, img_ch, n, h, w, 0, 0, &rows_a)) return SR_OK;
    sr_device_guard restore_device;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sr_ensure_streams(c, false);
    if (rc == SR_OK) rc = sr_ensure_fork_resources(c);
    if (rc != SR_OK) return rc;
    StackJob a, b;
    a.forked = b.forked = true;
    a.c = b.c = c; a.W = b.W = w; a.img_u8 = b.img_u8 = img_u8; a.img_ch = b.img_ch = img_ch;
    a.ws = &c->ws[0]; a.s = c->stream; a.H = rows_a + SR_HALO; a.top = 0; a.bot = rows_a;
    b.ws = &c->ws[1]; b.s = c->stream; b.H = h - (rows_a - SR_HALO); b.top = SR_HALO; b.bot = b.H;
    rc = a.prepare();
    if (rc == SR_OK) rc = b.prepare();
    if (rc == SR_E_NOMEM) return SR_OK;  // the device call will run undivided (sr_run_stack_auto)
    if (rc != SR_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SR_OK;
}

int sr_reserve_f32(sr_ctx* c, int n, int h, int w) {
    const int rc = run_host(c, nullptr, false, 3, Deal{0, 1, n}, h, w, nullptr, false, 0, -1, true);
    return rc == SR_OK ? reserve_fork(c, false, 3, n, h, w) : rc;
}

int sr_reserve_rgba8(sr_ctx* c, int in_channels, int n, int h, int w) {
    const int rc = run_host(c, nullptr, true, in_channels, Deal{0, 1, n}, h, w, nullptr, true, 0, -1, true);
    return rc == SR_OK ? reserve_fork(c, true, in_channels, n, h, w) : rc;
}

int sr_upscale_f32(sr_ctx* c, const float* in, int n, int h, int w, float* out) {
    return run_host(c, in, false, 3, Deal{0, 1, n}, h, w, out, false);
}

int sr_upscale_rgba8(sr_ctx* c, const uint8_t* in, int in_channels, int n, int h, int w, uint8_t* out) {
    return run_host(c, in, true, in_channels, Deal{0, 1, n}, h, w, out, true);
}

// One image, several GPUs, one process: device k produces its share of the rows from the caller's image directly
// (the 7 halo rows either side are just more rows of the same host buffer, so nothing is exchanged between
// devices); one host thread per context drives its pipeline.  Shares are multiples of 8 rows (whole tiles).
static int run_multi(sr_ctx* const* ctxs, int n_ctx, const void* in, bool img_u8, int img_ch, int h, int w, void* out, bool out_u8) {
    if (!in || !out || h <= 0 || w <= 0) return SR_E_INVALID;
    const int chk = check_context_set(ctxs, n_ctx);
    if (chk != SR_OK) return chk;
    int rows = (h + n_ctx - 1) / n_ctx;
    rows = std::max(8, (rows + 7) / 8 * 8);
    const int used = (h + rows - 1) / rows;
    if (used == 1) return run_host(ctxs[0], in, img_u8, img_ch, Deal{0, 1, 1}, h, w, out, out_u8);
    // the last share must not be thinner than the halo its neighbour reads from it
    std::vector<int> lo(used), hi(used);
    for (int k = 0; k < used; ++k) { lo[k] = k * rows; hi[k] = std::min(h, lo[k] + rows); }
    if (hi[used - 1] - lo[used - 1] < SR_HALO) { hi[used - 2] = h; lo.pop_back(); hi.pop_back(); }
    const int parts = (int)lo.size();
    if (parts == 1) return run_host(ctxs[0], in, img_u8, img_ch, Deal{0, 1, 1}, h, w, out, out_u8);
    std::vector<int> rc(parts, SR_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < parts; ++k)
        spawn_or_run(th, [&, k] { rc[k] = run_host(ctxs[k], in, img_u8, img_ch, Deal{0, 1, 1}, h, w, out, out_u8, lo[k], hi[k]); });
    for (auto& t : th) t.join();
    for (int k = 0; k < parts; ++k)
        if (rc[k] != SR_OK) return rc[k];
    return SR_OK;
}

int sr_upscale_f32_multi(sr_ctx* const* ctxs, int n_ctx, const float* in, int h, int w, float* out) {
    return run_multi(ctxs, n_ctx, in, false, 3, h, w, out, false);
}

int sr_upscale_rgba8_multi(sr_ctx* const* ctxs, int n_ctx, const uint8_t* in, int in_channels, int h, int w, uint8_t* out) {
    return run_multi(ctxs, n_ctx, in, true, in_channels, h, w, out, true);
}

// Many images, several GPUs, one process (throughput mode, BASELINE configs[4]): image i goes to context i mod n_ctx,
// parameters are replicated, nothing is exchanged.  One host thread per context runs that context's images through
// its own upload / compute / download pipeline (chunks of ~1M px, so several small images share a launch).
static int run_batch_multi(sr_ctx* const* ctxs, int n_ctx, const void* in, bool img_u8, int img_ch, int n, int h, int w, void* out,
                           bool out_u8) {
    if (!in || !out || n <= 0 || h <= 0 || w <= 0) return SR_E_INVALID;
    const int chk = check_context_set(ctxs, n_ctx);
    if (chk != SR_OK) return chk;
    const int used = std::min(n_ctx, n);
    if (used == 1) return run_host(ctxs[0], in, img_u8, img_ch, Deal{0, 1, n}, h, w, out, out_u8);
    std::vector<int> rc(used, SR_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < used; ++k)
        spawn_or_run(th, [&, k] { rc[k] = run_host(ctxs[k], in, img_u8, img_ch, Deal{k, used, (n - k + used - 1) / used}, h, w, out, out_u8); });
    for (auto& t : th) t.join();
    for (int k = 0; k < used; ++k)
        if (rc[k] != SR_OK) return rc[k];
    return SR_OK;
}

int sr_upscale_f32_batch_multi(sr_ctx* const* ctxs, int n_ctx, const float* in, int n, int h, int w, float* out) {
    return run_batch_multi(ctxs, n_ctx, in, false, 3, n, h, w, out, false);
}

int sr_upscale_rgba8_batch_multi(sr_ctx* const* ctxs, int n_ctx, const uint8_t* in, int in_channels, int n, int h, int w,
                                 uint8_t* out) {
    return run_batch_multi(ctxs, n_ctx, in, true, in_channels, n, h, w, out, true);
}

int sr_read_feature(sr_ctx* c, int which, float* out_host, size_t cap_floats) {
    if (!c || which < 0 || which > 3 || !out_host) return SR_E_INVALID;
    const size_t nf = (size_t)c->last_h * c->last_w * 32;
    if (nf == 0 || cap_floats < nf || !c->ws[0].d_feat[which]) return SR_E_INVALID;
    sr_device_guard restore_device;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    const size_t pitch = (size_t)c->ws[0].pitch;
    if (c->precision == SR_PRECISION_SPLIT_F16) {
        // row-planar map: row y = 8 runs of `pitch` 16-byte groups (4 x 8 hi halves, then their lo halves); whole padded rows come over
        std::vector<_Float16> rows((size_t)c->last_h * pitch * 64);
        HIPCHK(c, hipMemcpy(rows.data(), c->ws[0].d_feat[which] + (size_t)kFeatPad * pitch * 32, rows.size() * sizeof(_Float16), hipMemcpyDeviceToHost));
        for (int y = 0; y < c->last_h; ++y)
            for (int x = 0; x < c->last_w; ++x)
                for (int k = 0; k < 32; ++k) {
                    const _Float16* row = rows.data() + (size_t)y * pitch * 64;
                    const size_t cell = ((size_t)(k >> 3) * pitch + kFeatPad + x) * 8 + (k & 7);
                    out_host[((size_t)y * c->last_w + x) * 32 + k] = (float)row[cell] + (float)row[cell + 4 * pitch * 8] * (1.0f / 2048.0f);
                }
        return SR_OK;
    }
    const float* src = c->ws[0].d_feat[which] + ((size_t)kFeatPad * pitch + kFeatPad) * 32;
    HIPCHK(c, hipMemcpy2D(out_host, (size_t)c->last_w * 128, src, pitch * 128, (size_t)c->last_w * 128,
                          c->last_h, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_check_domain(sr_ctx* c) {
    if (!c) return SR_E_INVALID;
    const bool fault = c->dev_fault || (c->h_domain && *(volatile int*)c->h_domain);
    c->dev_fault = false;
    if (c->h_domain) *(volatile int*)c->h_domain = 0;
    return fault ? SR_E_DOMAIN : SR_OK;
}

int sr_last_timing(sr_ctx* c, double* total_ms, double stage_ms[5], double* h2d_ms, double* d2h_ms) {
    if (!c) return SR_E_INVALID;
    if (c->band_pending) {  // a sharded call (sr_comm.cpp): its whole step on this context -- band copy, halo exchange, conv stack -- waited for here
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->ev_band[1]));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_band[0], c->ev_band[1]));
        c->total_ms = ms;
        for (auto& v : c->stage_ms) v = 0;
        c->h2d_ms = c->d2h_ms = 0;
        c->band_pending = false;
    }
    if (total_ms) *total_ms = c->total_ms;
    if (stage_ms) for (int i = 0; i < 5; ++i) stage_ms[i] = c->stage_ms[i];
    if (h2d_ms) *h2d_ms = c->h2d_ms;
    if (d2h_ms) *d2h_ms = c->d2h_ms;
    return SR_OK;
}

}  // extern "C"
