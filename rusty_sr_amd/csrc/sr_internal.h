// sr_internal.h -- the context behind include/srhip.h's opaque sr_ctx, shared by sr_api.cpp (engine) and
// sr_comm.cpp (RCCL communicator + sharded entry points).  Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/srhip.h"

struct sr_ctx {
    int device = 0;
    int cus = 0, clock_mhz = 0;
    char name[128] = {0};
    hipStream_t stream = nullptr;
    float* d_params = nullptr;  // all packed parameters, one allocation
    void* d_qtab = nullptr;     // bilinear_net / downsample_net: data_to_img(LinearToSrgb(l)) as a step table (sr_aux.hip)
    size_t off_w0 = 0, off_w0h = 0, off_w[5] = {0}, off_wh[5] = {0}, off_bias[5] = {0}, off_beta[5] = {0};
    int precision = 0;  // SR_PRECISION_F32 / SR_PRECISION_SPLIT_F16
    // Domain of the split-half mode (include/srhip.h, sr_set_precision): values are carried as pairs of HALVES, so every weight, input
    // and activation must be finite and below 65504 in magnitude.  Weights are checked once (split_ok); inputs and activations by the
    // kernels, which raise *h_domain -- one word of mapped host memory, read by the host without a copy once the stream has drained.
    bool split_ok = true;
    int* h_domain = nullptr;   // host address of the flag
    int* d_domain = nullptr;   // the same word as the device sees it
    int domain_fallbacks = 0;  // host-pointer calls that were recomputed in exact f32 because of it
    bool dev_fault = false;    // a fault an earlier *_dev call left in *h_domain, set aside at the start of a host-pointer call: sr_check_domain's
    int graph = SR_GRAPH_SR_NET;
    int factor = SR_FACTOR;
    // Device workspace of one pass of the conv stack.  Two of them: the host pipeline (run_host) alternates chunks between
    // two compute streams so that the tail of one chunk's stage launches overlaps the head of the next chunk's; every
    // other entry point uses ws[0] only (ws[1] is allocated on first use).
    struct Workspace {
        float* d_feat[4] = {nullptr, nullptr, nullptr, nullptr};  // f, l1, l2, l3 (zero-bordered, see sr_kernels.h)
        size_t feat_cap_px = 0;       // allocated padded pixels per map
        int geo_n = 0, geo_h = 0, geo_w = 0;  // geometry the borders were last zeroed for
        int pitch = 0; long img_stride = 0;
        int* d_queue = nullptr;       // 5 stages x 8 per-XCD tile-queue heads (persistent kernels)
    } ws[2];
    hipStream_t stream2 = nullptr;    // second compute stream of the host pipeline
    // host-pointer entry points: two in / out slots so that chunk i+1 uploads and chunk i-1
    // downloads while chunk i computes (run_host)
    void* d_in[2] = {nullptr, nullptr};  size_t in_cap[2] = {0, 0};
    void* d_out[2] = {nullptr, nullptr}; size_t out_cap[2] = {0, 0};
    hipStream_t copy_out = nullptr;   // downloads of a pipelined host call
    hipStream_t copy_in = nullptr;    // uploads: exact-f32 contexts, from their second pipelined call on (else on the compute streams)
    int pipelined_calls = 0;
    std::vector<hipEvent_t> pool;  // per-chunk timing / ordering events of run_host, grown on demand
    int pipeline = 1;              // 0: one upload, one pass, one download
    int last_chunks = 0;
    hipEvent_t ev[8] = {nullptr};
    bool profiling = false;
    double total_ms = 0, stage_ms[5] = {0}, h2d_ms = 0, d2h_ms = 0;
    int last_h = 0, last_w = 0;
    int last_hip = 0;
    // experiment switches, read once at sr_create (none changes results): SRHIP_TH, SRHIP_PIPE, SRHIP_BW, SRHIP_TAIL
    int env_th[5] = {0, 0, 0, 0, 0};  // 0: automatic
    int env_pipe = 1;                 // 0: first form everywhere, 1: pipe form except for small launches, 2: pipe form everywhere
    int env_bw = -1;                  // tile-order column-block width in tiles (-1: automatic)
    float env_tail = -1.0f;           // 4-row tiles at the end of a launch, in resident workgroups (< 0: automatic = 1 where the tail rule applies, 0: none)
    int env_fork = -1;                // device entry points, one image: two row bands on two streams (sr_run_stack_auto); -1 automatic, 0 never,
                                      // 1 always, > 1: always, with this many rows in the first band
    double fork_min_rounds = 3.5;     //   automatic: fork from this many rounds of 8-row tiles per resident workgroup on ...
    double fork_max_rounds = 1e9;     //   ... and below this many (no upper bound by default)
    double fork_share = 0.5;          //   the first band's share of the rows
    hipEvent_t ev_fork[2] = {nullptr, nullptr};  // fork (caller's stream -> stream2) and join (stream2 -> caller's stream)
    // Mid-size shapes (0.55-12 rounds of 8-row tiles): whether the fork pays depends on how the tiles of the image and of its two bands
    // happen to fill the chip's last round -- -15 ... +20 % from one shape to the next (profiles/r6_fork_tune.txt), which no rule in
    // `rounds` predicts.  So it is MEASURED, per shape and context, on the caller's own calls: a block of undivided calls, then a block of
    // forked ones, each timed by an event pair on the caller's stream that is read -- never waited for -- at a later call of that shape;
    // then the faster plan stays.  Results do not depend on the plan (bit-identical), only the time does.  sr_set_experiment "forktune".
    struct ForkTune {
        int H = 0, W = 0, top = 0, bot = 0, img_ch = 0, precision = 0;
        bool img_u8 = false, out_u8 = false;
        int taken = 0;                    // samples read so far: [0, kForkTuneBlock) undivided, [kForkTuneBlock, 2 kForkTuneBlock) forked; a block's first is dropped (warm-up)
        float best[2] = {1e30f, 1e30f};   // fastest sample of either plan, ms
        int decided = -1;                 // -1: still measuring, 0: undivided, 1: forked
        bool pending = false;             // ev[] bracket a call whose time has not been read yet
        hipEvent_t ev[2] = {nullptr, nullptr};
        unsigned long long used = 0;      // for eviction: the entry used longest ago goes
    };
    std::vector<ForkTune> fork_tune;
    bool fork_autotune = true;
    unsigned long long fork_tune_clock = 0;
    int env_bands = 0;                // host pipeline: forced number of row bands (0: automatic)
    std::vector<int> env_rows;        // host pipeline: forced band heights (empty: automatic)
    bool env_rows_two = false;        //   ... computed on alternating streams instead of in order
    bool env_geo = true;              // host pipeline: geometric band plan where the call is compute-bound
    unsigned long long params_hash = 0;  // FNV-1a of the parameter vector: contexts of one sharded call must agree
    // ---- multi-GPU (sr_comm.cpp): one RCCL communicator per context, neighbour halo exchange
    void* comm = nullptr;             // ncclComm_t
    bool comm_local = false;          // sr_comm_init_local: neighbours are contexts of this process, halos go by peer copy
    int comm_rank = 0, comm_nranks = 1;
    void* d_ext = nullptr; size_t ext_cap = 0;  // band + halo rows, the exchange lands here
    hipEvent_t ev_comm[2] = {nullptr, nullptr};  // around the halo exchange of a sharded call, on the stream it runs on (round 6: the context's stream2)
    hipEvent_t ev_wait[2] = {nullptr, nullptr};  // on the band's stream, either side of its wait for the exchange: what of the exchange was NOT hidden
    hipEvent_t ev_xfork = nullptr;               // band's stream -> exchange stream: the caller's band is complete
    bool wait_pending = false;
    double comm_exposed_ms = 0;
    hipEvent_t ev_band[2] = {nullptr, nullptr};  // around the whole sharded step of this context (band copy, exchange, conv stack)
    bool comm_pending = false, band_pending = false;  // the events of the last sharded call have not been read yet (sr_last_comm_ms / sr_last_timing)
    double comm_ms = 0;
    int last_nccl = 0;
    bool comm_broken = false;         // an exchange failed half-posted: the communicator was aborted, sharded calls return SR_E_COMM
    bool layer_halos = false;         // sharded calls exchange FEATURE rows after every stage instead of recomputing the overlap (sr_set_experiment "halo")
    hipEvent_t ev_layer[4] = {nullptr, nullptr, nullptr, nullptr};  // one-process sharded call in that mode: "stage st of this context is done"
};

// The library never leaves the calling thread on another device than it found it on: torch (and any HIP host) takes
// "the current device" from hipGetDevice, and the one-process multi-GPU calls walk over every context's device.
// Every extern "C" entry point that may call hipSetDevice holds one of these for its duration.
struct sr_device_guard {
    int saved = -1;
    sr_device_guard() { if (hipGetDevice(&saved) != hipSuccess) { (void)hipGetLastError(); saved = -1; } }
    ~sr_device_guard() { if (saved >= 0) (void)hipSetDevice(saved); }
    sr_device_guard(const sr_device_guard&) = delete;
    sr_device_guard& operator=(const sr_device_guard&) = delete;
};

#define HIPCHK(ctx, expr)                         \
    do {                                          \
        hipError_t e__ = (expr);                  \
        if (e__ != hipSuccess) {                  \
            (void)hipGetLastError(); /* the runtime keeps the error until it is read: the next launch check must not find it */ \
            (ctx)->last_hip = (int)e__;           \
            return e__ == hipErrorOutOfMemory ? SR_E_NOMEM : SR_E_HIP; \
        }                                         \
    } while (0)


// Rows of the image that are still on their way when the call is made -- the halo rows of a sharded band, which a neighbour's GPU
// sends while this one already works (sr_comm.cpp): the first `top` and the last `bot` rows of d_img are in place once `ready`
// (recorded on another stream) has fired.  Only stage 0 reads the image's halo rows directly (f rows within 2 of them), so the stack
// launches stage 0 for the rows that need none of them FIRST, waits for the event on its own stream, and then runs stage 0's few
// edge rows and the other stages: the exchange hides under the band copy and the interior of stage 0.  mark[0..1] (optional) are
// recorded on the waiting stream either side of the wait.
struct sr_halo_gate {
    hipEvent_t ready = nullptr;
    int top = 0, bot = 0;
    hipEvent_t mark[2] = {nullptr, nullptr};
};

// The whole conv stack on device buffers (sr_api.cpp): rows [halo_top, H - halo_bot) of each image are produced.
int sr_run_stack(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int n, int H, int W, int halo_top, int halo_bot,
                 void* d_out, bool out_u8, hipStream_t s, int slot = 0, const sr_halo_gate* gate = nullptr);
// ... the same for one image as two row bands forked onto the context's second stream where that pays (the device entry points)
int sr_run_stack_auto(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int n, int H, int W, int halo_top, int halo_bot,
                      void* d_out, bool out_u8, hipStream_t s, const sr_halo_gate* gate = nullptr);
int sr_ensure_fork_resources(sr_ctx* c);  // the second stream + the fork / join events
void sr_fork_tune_clear(sr_ctx* c);      // forget what the fork tuner has measured (its events with it); the context's device is current

// One pass of the conv stack over a band, stage by stage -- for a caller that has something to do BETWEEN the stages (sr_comm.cpp:
// the per-layer feature-halo exchange).  `layers`: every stage computes the band's OWN rows only (no recompute
// margin); the rows of f / l1 / l2 / l3 that the next stage reads beyond them (2 / 1 / 1 / 1 either side) are the caller's to put
// into this context's maps before it launches that stage.
struct sr_band_pass;
int sr_band_pass_begin(sr_ctx* c, const void* d_img, bool img_u8, int img_ch, int H, int W, int halo_top, int halo_bot, void* d_out, bool out_u8,
                       hipStream_t s, bool layers, const sr_halo_gate* gate, sr_band_pass** out);
int sr_band_pass_stage(sr_band_pass* p, int st);                 // launch stage st (0..4) on the pass's stream
float* sr_band_pass_row(const sr_band_pass* p, int map, int y);  // row y (the pass's image coordinates) of map 0..3 = f, l1, l2, l3, border columns included
size_t sr_band_pass_row_floats(const sr_band_pass* p);           // ... its length: pitch x 32 floats, contiguous in both map layouts
void sr_band_pass_end(sr_band_pass* p);
int sr_ensure_buf(sr_ctx* c, void** p, size_t* cap, size_t bytes);
int sr_ensure_streams(sr_ctx* c, bool pipelined);  // the context's own streams are created on first use
void sr_comm_release(sr_ctx* c);  // sr_comm.cpp: destroy the communicator and its buffers (called by sr_destroy)
