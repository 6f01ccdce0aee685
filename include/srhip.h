/*
 * srhip.h -- C ABI of libsrhip.so, the MI355X (gfx950) engine behind rusty_sr's
 * upscale hot path.
 *
 * The reference (millardjn/rusty_sr v1) has no FFI/plugin seam of its own; the
 * boundary this library replaces is the single call
 *     let output = graph.forward(1, vec![input], &params).remove(0);
 * at reference src/main.rs:171, together with the tensor conversions on either
 * side of it (img_to_data main.rs:170, data_to_img main.rs:175) and the weight
 * decode that feeds it (`<Vec<f32>>::decode::<u32>` main.rs:138,146,149,152).
 * The conventions of that call site are kept: the caller owns `params` and all
 * image buffers, a call is synchronous, a context is single-caller, errors are
 * reported where the reference panics (main.rs:134-138,162,164,175).
 *
 * Everything is plain C: pointers, sizes, ints.  No torch / HIP types appear
 * in a signature (`stream` is an opaque hipStream_t passed as void*; NULL =
 * HIP's default stream).  A Rust host binds this with a 30-line
 * `extern "C"` block (INTEGRATION.md shows it).
 *
 * Tensor layout everywhere: NHWC, channel fastest -- alumina's
 * DataShape::new(channels, &[W, H], n) order (reference main.rs:168).
 * The up-scaling factor of the reference binary is 3 (main.rs:31 `const FACTOR: usize = 3`;
 * the bundled weights only fit factor 3, network.rs:37); the engine also takes 2 and 4.
 */
#ifndef SRHIP_H
#define SRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_FACTOR 3
#define SR_NUM_PARAMS 130459 /* graph.num_params() of sr_net(3, None); main.rs:162 */
/* sr_net(f, None) has 2400 + 64 + 3f^2 + 192 + 3*25600 + 3*9216 + 3*(3f^2*288) parameters */
#define SR_HALO 7            /* receptive-field radius of the conv stack in input px */

typedef struct sr_ctx sr_ctx;

enum sr_status {
    SR_OK = 0,
    SR_E_INVALID = -1,     /* NULL pointer / non-positive dimension / bad channel count */
    SR_E_PARAM_COUNT = -2, /* main.rs:162 assert_eq!(params.len(), graph.num_params()) */
    SR_E_FACTOR = -3,      /* sr_net: factor must be 2, 3 or 4 (the reference ships 3, main.rs:31) */
    SR_E_NO_DEVICE = -4,   /* no gfx950 device visible: the engine has NO CPU fallback */
    SR_E_HIP = -5,         /* a HIP runtime call failed; sr_last_hip_error() has the code */
    SR_E_NOMEM = -6,
    SR_E_BYTEVEC = -7,     /* main.rs:138 "ByteVec conversion failed" */
    SR_E_HALO = -8,        /* band call with a halo that is neither 0 nor >= SR_HALO (or a band thinner than SR_HALO) */
    SR_E_COMM = -9,        /* librccl missing, no communicator on the context, or an RCCL call failed;
                              sr_last_comm_error() has the ncclResult_t */
    SR_E_DOMAIN = -10      /* SR_PRECISION_SPLIT_F16 only: a weight, an input or an activation cannot be carried as a pair of
                              halves (not finite, or 65504 and beyond); see sr_set_precision / sr_check_domain */
};

/* Replaces: `<Vec<f32>>::decode::<u32>(blob)` (bytevec 0.2.0; reference
 * main.rs:138,146,149,152).  Wire format: u32 LE n | n x u32 LE sizes (=4) |
 * n x f32 LE.  Pass out = NULL to query *n_out.  Host-side, no GPU needed. */
int sr_rsr_decode(const uint8_t* blob, size_t len, float* out, size_t cap, size_t* n_out);

/* Replaces: `.encode::<u32>()` (reference main.rs:213), the inverse of the above.
 * Pass out = NULL to query the byte length in *len_out. */
int sr_rsr_encode(const float* params, size_t n, uint8_t* out, size_t cap, size_t* len_out);

/* Replaces: `sr_net(FACTOR, None)` + the parameter-count assert (reference
 * main.rs:146,162; network.rs:16-109).  Validates, selects HIP device
 * `device`, uploads the parameters once re-packed into the MFMA B-operand
 * layout.  Fails with SR_E_NO_DEVICE when no GPU is present. */
int sr_create(sr_ctx** out, const float* params, size_t n_params, int factor, int device);
void sr_destroy(sr_ctx* ctx);

/* The reference's upscale() picks one of three graphs (main.rs:133-158):
 *   SR_GRAPH_SR_NET      sr_net(FACTOR, None)      network.rs:16-109   130459 params
 *   SR_GRAPH_BILINEAR    bilinear_net(FACTOR)      network.rs:111-123  `-p bilinear`, 0 params
 *                        sRGB->linear, bilinear x3, linear->sRGB
 *   SR_GRAPH_DOWNSAMPLE  downsample_net(FACTOR)    network.rs:125-138  `-d`, 0 params
 *                        sRGB->linear, mean over non-overlapping 3x3 blocks, linear->sRGB;
 *                        output (h/3) x (w/3), remainder rows / columns dropped
 * sr_create_graph replaces the `(params, graph)` selection + the count assert
 * (main.rs:162): n_params must equal sr_num_params(graph).  Every sr_upscale_*
 * entry point below then means `graph.forward` for whichever graph the context
 * holds (for SR_GRAPH_DOWNSAMPLE the output is n*(h/3)*(w/3) pixels); the band
 * entry points exist for SR_GRAPH_SR_NET only. */
enum sr_graph { SR_GRAPH_SR_NET = 0, SR_GRAPH_BILINEAR = 1, SR_GRAPH_DOWNSAMPLE = 2 };
int sr_create_graph(sr_ctx** out, int graph, const float* params, size_t n_params, int factor, int device);
int sr_num_params(int graph); /* graph.num_params() at factor 3; -1 for an unknown graph */

/* `sr_net(factor, ..)` takes the factor as an argument (network.rs:16) although main.rs:31
 * hard-wires 3 ("TODO: expose upscaling factor as argument"): sr_create / sr_create_graph accept
 * factor 2, 3 or 4 for SR_GRAPH_SR_NET given a parameter vector of sr_num_params_factor(factor)
 * entries in the same op order (the expand node has 3 f^2 channels, network.rs:37), and every
 * entry point then produces f*h x f*w outputs.  No 2x / 4x weights ship with the reference, so
 * those factors are checked against the CPU restatement only (UNPINNED). */
int sr_num_params_factor(int factor); /* -1 unless 2 <= factor <= 4 */

/* Sizes: nothing but device memory limits an image.  One pass of the conv stack keeps four 32-channel f32 feature maps
 * (512 B per input pixel) beside input and output: 1920x1080 1.1 GB, 3840x2160 4.3 GB, 11 000 x 11 000 62 GB (tested; byte
 * offsets and outputs beyond 4 GiB are fine).  The host-pointer entry points cut large jobs into chunks / row bands, so their
 * workspace is that of a band.  A failed allocation is SR_E_NOMEM and leaves the context usable.
 *
 * Replaces: graph.forward(n, vec![input], &params) (reference main.rs:171).
 * in : n*h*w*3 f32 in [0,1] (what img_to_data produced), host memory.
 * out: n*(3h)*(3w)*3 f32, pre-quantisation, host memory. */
int sr_upscale_f32(sr_ctx* ctx, const float* in, int n, int h, int w, float* out);

/* Replaces: img_to_data + graph.forward + data_to_img(..).to_rgba() (reference
 * main.rs:170-175) as one fused device pass: u8/255 on load, and
 * clamp(floor(255 v + 0.5)) with alpha = 255 on store.
 * in : n*h*w*in_channels u8, in_channels 3 (RGB) or 4 (RGBA, alpha dropped).
 * out: n*(3h)*(3w)*4 u8 RGBA.  Host memory. */
int sr_upscale_rgba8(sr_ctx* ctx, const uint8_t* in, int in_channels, int n, int h, int w,
                     uint8_t* out_rgba);

/* Optional: everything the matching sr_upscale_* call of that shape would do EXCEPT touching caller memory -- the
 * workspace and staging allocations, the events, and one pass of its kernels over whatever the staging buffers hold
 * (code objects load, clocks come up) -- and what a sr_upscale_*_dev call of that shape creates on first use (see there).  The first real call then costs what every later one does; a host calls this at
 * start-up, or -- like the CLI -- on one thread while another still decodes the input file.  The reference has no
 * counterpart (alumina allocates inside graph.forward, main.rs:171). */
int sr_reserve_f32(sr_ctx* ctx, int n, int h, int w);
int sr_reserve_rgba8(sr_ctx* ctx, int in_channels, int n, int h, int w);

/* One image across several GPUs from one process: ctxs[k] (sr_net contexts of the same parameters, normally one
 * per device, created with sr_create(.., device k)) produces a contiguous share of the rows, a multiple of 8.
 * The SR_HALO rows a share needs from its neighbours are read from the caller's image itself, so the devices
 * exchange nothing; results are bit-identical to the single-device call.  This is the host-memory counterpart
 * of the RCCL halo exchange of device-resident bands (sr_upscale_sharded_* below); the reference has neither
 * (one CPU, main.rs:171).  Shares run concurrently, one host thread per context. */
int sr_upscale_f32_multi(sr_ctx* const* ctxs, int n_ctx, const float* in, int h, int w, float* out);
int sr_upscale_rgba8_multi(sr_ctx* const* ctxs, int n_ctx, const uint8_t* in, int in_channels, int h, int w,
                           uint8_t* out_rgba);

/* Many images across several GPUs from one process -- throughput mode: image i goes to ctxs[i mod n_ctx]
 * (contexts of the same parameters and arithmetic mode, one per device), parameters replicated, no exchange;
 * each context runs its images through its own upload / compute / download pipeline on its own host thread.
 * in / out are the whole batch (n images, host memory).  The reference processes one image per process
 * (main.rs:164-171); its users loop over files. */
int sr_upscale_f32_batch_multi(sr_ctx* const* ctxs, int n_ctx, const float* in, int n, int h, int w, float* out);
int sr_upscale_rgba8_batch_multi(sr_ctx* const* ctxs, int n_ctx, const uint8_t* in, int in_channels, int n, int h, int w,
                                 uint8_t* out_rgba);

/* The host-pointer entry points (sr_upscale_f32 / sr_upscale_rgba8 and their _multi forms) run upload / conv stack / download as a software
 * pipeline on up to four HIP streams of the context's own, created on first need (a call that is one chunk uses one): a batch goes in
 * chunks of whole images, one sr_net image of about 200K pixels or more (100K with f32 output) as two or more row bands with SR_HALO
 * halo rows, so that a band's download runs under the next band's kernels (bit-identical to the undivided pass, see sr_upscale_band_*).  Results do not depend on the setting; 0 = one upload, one pass, one download.  The reference has no counterpart
 * (its tensors never leave host memory, main.rs:168-175); buffers from sr_host_alloc are page-locked, which lets the copies run at
 * PCIe rate and truly overlap -- any host memory is accepted.
 * (The device-pointer entry points below run on the CALLER'S stream; where they cut one image into two bands they also use one stream
 * of the context's own and a second set of feature maps, see there.) */
int sr_set_pipeline(sr_ctx* ctx, int enabled);         /* default: enabled */
int sr_host_alloc(void** out, size_t bytes);           /* SR_E_NO_DEVICE without a GPU */
void sr_host_free(void* p);

/* Same two operations on buffers already resident in this context's device
 * memory (HBM); asynchronous on `stream` (opaque hipStream_t; NULL = HIP's
 * default stream, which is also torch's default stream).  The caller orders
 * its own producers / consumers of d_in / d_out on that stream.  These are what bench.py times and what the multi-GPU
 * driver calls after its halo exchange.  One image (n = 1) of enough rows may run as TWO row bands, the second on a stream
 * of the context's own that is forked from `stream` and joined back to it by events (one band's launches drain while the
 * other's fill; bit-identical): the call is still asynchronous and ordered on `stream` alone.  What that costs:
 * a second set of feature maps (each band's are half the size; should they not fit, the call runs undivided), and -- on its first
 * use -- one stream creation and the allocations, inside the call (sr_reserve_* of the same shape does both ahead of time).
 * After such a call sr_read_feature refuses (each workspace holds one band), as it does after a pipelined host call.
 * u8 device images are READ as whole aligned 32-bit words by the parameter-free graphs' kernels: up to 3 bytes in front of the
 * image's first byte and behind its last one -- bytes of the same aligned word, hence of the same allocation granule -- may be
 * read (never written, never used). */
int sr_upscale_f32_dev(sr_ctx* ctx, const float* d_in, int n, int h, int w, float* d_out,
                       void* stream);
int sr_upscale_rgba8_dev(sr_ctx* ctx, const uint8_t* d_in, int in_channels, int n, int h, int w,
                         uint8_t* d_out_rgba, void* stream);

/* Row-band form for images sharded across GPUs.  d_in holds h_ext = halo_top +
 * h_band + halo_bot input rows of ONE image of width w; halo_top / halo_bot are
 * the rows that belong to the neighbouring bands (0 = this edge is the true
 * image edge, where the reference's per-layer zero padding applies; otherwise
 * must be >= SR_HALO).  d_out receives only the band's 3*h_band output rows.
 * Bit-identical to the corresponding rows of the un-sharded call. */
int sr_upscale_band_f32_dev(sr_ctx* ctx, const float* d_in, int h_ext, int w, int halo_top,
                            int halo_bot, float* d_out, void* stream);
int sr_upscale_band_rgba8_dev(sr_ctx* ctx, const uint8_t* d_in, int in_channels, int h_ext, int w,
                              int halo_top, int halo_bot, uint8_t* d_out_rgba, void* stream);

/* ---- One image sharded over several GPUs, DEVICE-RESIDENT: RCCL halo exchange inside the library ----
 * The reference has no counterpart (one CPU, main.rs:171).  A context can own one RCCL communicator
 * (librccl is dlopen'ed on first use; a host needs no torch and no MPI).  Rank r of n holds a contiguous
 * row band of the image in its GPU's memory (bands in rank order, every band >= SR_HALO rows, same width);
 * sr_upscale_sharded_*_dev sends the band's first / last SR_HALO rows to ranks r-1 / r+1 and receives theirs
 * (one grouped ncclSend / ncclRecv pair per neighbour over xGMI, queued on `stream`), then runs the band form
 * of the conv stack (sr_upscale_band_*_dev) and writes the band's 3*h_band output rows to d_out.  The rows of
 * all ranks together are bit-identical to the single-GPU call.
 * Interior first: the exchange runs on a second stream of the context's own (forked from `stream` by an event and joined back
 * to it by another, like the two-band form above) while `stream` copies the band and computes the first layer on every row that
 * reads no halo row; only then does `stream` wait for the halos.  The call stays asynchronous and ordered on `stream` alone.
 *
 * One process per GPU (the normal form):  rank 0 calls sr_comm_unique_id and hands the 128 bytes to the other
 * ranks by whatever means the host has (a file, a socket, an environment variable, torch.distributed);
 * every rank then calls sr_comm_init_rank (collective: returns when all n ranks have joined).
 * One process, n GPUs: sr_comm_init_all on n contexts of distinct devices (ncclCommInitAll; rank = index), then
 * sr_upscale_sharded_*_all drives all bands from the calling thread (grouped exchange, synchronous).
 * One process, without RCCL: sr_comm_init_local on n contexts (rank = index; the same device may appear more than
 * once).  sr_upscale_sharded_*_all then has every context PULL its two halos from the neighbours' bands with
 * hipMemcpyPeerAsync on its own stream (SDMA over xGMI, peer access enabled where the devices allow it): no
 * rendezvous, no compute unit spent on the exchange.  The *_dev entry points return SR_E_COMM on such a context
 * (a lone rank cannot see its neighbours' buffers). */
#define SR_COMM_ID_BYTES 128
int sr_comm_available(void);                            /* 1 if librccl could be loaded */
int sr_comm_unique_id(uint8_t* id, size_t cap);         /* cap >= SR_COMM_ID_BYTES */
int sr_comm_init_rank(sr_ctx* ctx, const uint8_t* id, size_t id_len, int rank, int nranks);
int sr_comm_init_all(sr_ctx* const* ctxs, int n);
int sr_comm_init_local(sr_ctx* const* ctxs, int n);
void sr_comm_destroy(sr_ctx* ctx);                      /* sr_destroy does this too */
int sr_comm_rank(sr_ctx* ctx, int* rank, int* nranks);  /* 0 of 1 without a communicator */
int sr_last_comm_error(sr_ctx* ctx);                    /* ncclResult_t of the last failed RCCL call */
int sr_last_comm_ms(sr_ctx* ctx, double* comm_ms);      /* device time of the last sharded call's halo exchange (an event pair on the stream
                                                         * it ran on, recorded on every call; waits for the exchange, not for the kernels).
                                                         * sr_last_timing after a sharded call: total_ms = this context's whole step */
int sr_last_comm_exposed_ms(sr_ctx* ctx, double* exposed_ms);  /* ... and how long the band's stream stood waiting for it (an event pair
                                                         * either side of its wait): the part of comm_ms that was NOT hidden under the
                                                         * band copy and the first layer's interior rows */
int sr_upscale_sharded_f32_dev(sr_ctx* ctx, const float* d_band, int h_band, int w, float* d_out, void* stream);
int sr_upscale_sharded_rgba8_dev(sr_ctx* ctx, const uint8_t* d_band, int in_channels, int h_band, int w,
                                 uint8_t* d_out_rgba, void* stream);
int sr_upscale_sharded_f32_all(sr_ctx* const* ctxs, int n, const float* const* d_bands, const int* h_bands, int w,
                               float* const* d_outs);
int sr_upscale_sharded_rgba8_all(sr_ctx* const* ctxs, int n, const uint8_t* const* d_bands, int in_channels,
                                 const int* h_bands, int w, uint8_t* const* d_outs);

/* Arithmetic of the conv stack.
 *   SR_PRECISION_F32       (default) v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate --
 *                          the same arithmetic class as the reference's f32 CPU path.  Domain: any f32, like
 *                          graph.forward (main.rs:171); infinities and NaNs propagate as IEEE arithmetic has them.
 *   SR_PRECISION_SPLIT_F16 every activation / weight is carried as a pair of halves
 *                          (hi + lo/2048, ~2^-23 relative) and each product is three f16 MFMAs with
 *                          f32 accumulation on the matrix cores; outputs stay within the 1e-4 bar
 *                          (tests/test_gpu_parity.py runs every parity test in both modes).
 *                          DOMAIN: every weight, input value and activation finite and below 65504 in magnitude
 *                          (u8 images through the bundled weights stay below 100).  Nothing outside it is clamped
 *                          silently:
 *                            - sr_set_precision returns SR_E_DOMAIN for a parameter vector with such a weight and leaves
 *                              the context in its previous mode;
 *                            - the kernels notice an input or activation that leaves the domain.  The synchronous
 *                              host-pointer entry points (sr_upscale_f32 / _rgba8 and their _multi / _batch_multi forms) then
 *                              compute the whole call again in SR_PRECISION_F32 and return its result;
 *                            - the asynchronous *_dev entry points cannot: their output is unspecified where the overflow
 *                              reached, and the context keeps a fault that sr_check_domain reports. */
enum sr_precision { SR_PRECISION_F32 = 0, SR_PRECISION_SPLIT_F16 = 1 };
int sr_set_precision(sr_ctx* ctx, int mode);
/* After the stream(s) of earlier *_dev calls have been synchronised: SR_E_DOMAIN if any of them left the domain of
 * SR_PRECISION_SPLIT_F16 since the last check (the fault is cleared), else SR_OK.  No call made in SR_PRECISION_F32 raises a
 * fault; one left by earlier split-mode calls is kept -- across sr_set_precision and across host-pointer calls, which neither
 * report nor act on it -- until it is checked.  (A host-pointer call over several contexts recomputes per context: only the
 * contexts whose rows left the domain return f32-mode rows; both modes meet the same 1e-4 bar.) */
int sr_check_domain(sr_ctx* ctx);

/* (A/B tuning switches that change no result bit, and their environment defaults, are NOT part of this interface:
 * include/srhip_experimental.h.) */

/* Test hook: copy the post-activation feature maps of the most recent call
 * (image 0) to host: which = 0..3 -> f, l1, l2, l3 (h*w*32 f32 each).  The
 * reference exposes the same values as graph node data (network.rs:30,43-48).  Refused (SR_E_INVALID) after a call that ran the image
 * as bands -- a pipelined host call, a forked device call: sr_set_pipeline(ctx, 0) / the undivided device call leave whole maps. */
int sr_read_feature(sr_ctx* ctx, int which, float* out_host, size_t cap_floats);

/* Device time of the most recent call, measured with HIP events on the stream
 * the kernels ran on.  stage_ms[5] = conv0, l1, l2, l3, expand stage kernels
 * (enable with sr_set_profiling; off by default -- it inserts events, and the host-pointer
 * entry points then run undivided).  After a pipelined host call total = first kernel start to last
 * kernel end (chunks overlap on two compute streams), h2d / d2h = sums over the chunks, and
 * sr_read_feature refuses (the maps hold the last chunks only).
 * h2d / d2h are zero for the *_dev entry points. */
int sr_set_profiling(sr_ctx* ctx, int enabled);
int sr_last_timing(sr_ctx* ctx, double* total_ms, double stage_ms[5], double* h2d_ms,
                   double* d2h_ms);

/* Device facts for reports: name (e.g. "gfx950..."), CU count, clock MHz. */
int sr_device_info(sr_ctx* ctx, char* name, size_t cap, int* compute_units, int* clock_mhz);

int sr_last_hip_error(sr_ctx* ctx);
const char* sr_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif /* SRHIP_H */
