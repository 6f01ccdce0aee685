// sr_comm.cpp -- multi-GPU part of the C ABI: one RCCL communicator per context and the sharded entry points
// (SURVEY.md 8(b) "one RCCL communicator inside the context", 8(e)(i) halo-recompute exchange).
//
// The reference runs graph.forward on one CPU (main.rs:171) and has no collective of any kind.  What shards is
// the output: every output pixel depends on a 15x15 input window (SR_HALO = 7), so an image splits into
// contiguous row bands; a band needs the 7 input rows either side of it from its neighbours, recomputes the
// overlap and writes its own rows -- bit-identical to the undivided call.  The exchange is one grouped
// ncclSend / ncclRecv pair per neighbour (<= 7 * W * 12 B each: latency-bound, one xGMI link per direction).
// Round 6, interior first: the exchange is queued on the context's SECOND stream (forked from the band's stream by an event, so
// that it starts when the caller's band is complete) while the band's own stream copies the band and runs stage 0 on every row that
// reads no halo row; that stream then waits for the exchange's event and runs stage 0's few edge rows and the other stages
// (sr_internal.h sr_halo_gate).  No host synchronisation anywhere; what of the exchange was not hidden is timed by an event pair
// around the wait (sr_last_comm_exposed_ms).
//
// A host that drives all its GPUs from ONE process has a second transport (sr_comm_init_local): the neighbour's rows
// are the caller's own device buffers, so each context pulls its two halos with hipMemcpyPeerAsync on its own stream
// -- the SDMA engines move <= 322 KB over the xGMI link while no compute unit is taken from the band kernels, and
// nothing has to rendezvous.  It also lets the whole sharded path run on a one-GPU box (several contexts of one
// device), which is how tests/ cover it bit for bit.
//
// librccl is loaded lazily (dlopen of the SONAME): a process that already holds RCCL -- a torch process -- gets
// that same instance, a plain C / Rust host gets the system one, and a single-GPU user never needs it at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "sr_internal.h"

int sr_check_context_set(sr_ctx* const* ctxs, int n_ctx);  // sr_api.cpp

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        bool all = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, name));
            all = all && fn != nullptr;
        };
        sym(r.GetUniqueId, "ncclGetUniqueId");
        sym(r.CommInitRank, "ncclCommInitRank");
        sym(r.CommInitAll, "ncclCommInitAll");
        sym(r.CommDestroy, "ncclCommDestroy");
        sym(r.CommAbort, "ncclCommAbort");
        sym(r.Send, "ncclSend");
        sym(r.Recv, "ncclRecv");
        sym(r.GroupStart, "ncclGroupStart");
        sym(r.GroupEnd, "ncclGroupEnd");
        sym(r.GetErrorString, "ncclGetErrorString");
        r.ok = all;
    });
    return r.ok ? &r : nullptr;
}

#define NCCLCHK(ctx, expr)                              \
    do {                                                \
        ncclResult_t r__ = (expr);                      \
        if (r__ != ncclSuccess) {                       \
            if (ctx) (ctx)->last_nccl = (int)r__;       \
            return SR_E_COMM;                           \
        }                                               \
    } while (0)

// An exchange that failed after some of its sends / receives were queued would leave them waiting for partners nobody
// posted, and the stream drain behind them would never return.  ncclCommAbort ends whatever the communicator still has
// in flight; the context then refuses sharded calls (SR_E_COMM) until a new sr_comm_init_*.
void abort_comm(sr_ctx* c, Rccl* R) {
    if (!c->comm) return;
    if (R) (void)R->CommAbort((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_broken = true;
}

struct BandGeom {
    int top, bot, h_ext;
    size_t row_bytes;
};

BandGeom band_geom(const sr_ctx* c, int h_band, int w, size_t px_bytes) {
    BandGeom g;
    g.top = c->comm_rank > 0 ? SR_HALO : 0;
    g.bot = c->comm_rank < c->comm_nranks - 1 ? SR_HALO : 0;
    g.h_ext = g.top + h_band + g.bot;
    g.row_bytes = (size_t)w * px_bytes;
    return g;
}

// Queue the band copy and this rank's four point-to-point operations on `s`.  The caller brackets the calls of
// all ranks it drives with ONE ncclGroupStart / ncclGroupEnd (a single-threaded multi-GPU host must, or the first
// rank's send blocks for a receive nobody has posted yet).
int post_exchange(sr_ctx* c, Rccl* R, const void* d_band, int h_band, const BandGeom& g, hipStream_t s) {
    const char* band = (const char*)d_band;
    char* ext = (char*)c->d_ext;
    const size_t halo = (size_t)SR_HALO * g.row_bytes;
    ncclComm_t comm = (ncclComm_t)c->comm;
    if (g.top) {
        NCCLCHK(c, R->Send(band, halo, ncclUint8, c->comm_rank - 1, comm, s));
        NCCLCHK(c, R->Recv(ext, halo, ncclUint8, c->comm_rank - 1, comm, s));
    }
    if (g.bot) {
        NCCLCHK(c, R->Send(band + (size_t)(h_band - SR_HALO) * g.row_bytes, halo, ncclUint8, c->comm_rank + 1, comm, s));
        NCCLCHK(c, R->Recv(ext + (size_t)(g.top + h_band) * g.row_bytes, halo, ncclUint8, c->comm_rank + 1, comm, s));
    }
    return SR_OK;
}

int prepare_band(sr_ctx* c, const void* d_band, int h_band, int w, size_t px_bytes, BandGeom& g, hipStream_t s) {
    if (!c || !d_band || h_band <= 0 || w <= 0) return SR_E_INVALID;
    if (c->graph != SR_GRAPH_SR_NET) return SR_E_INVALID;
    if (c->comm_broken || (c->comm_nranks > 1 && !c->comm && !c->comm_local)) return SR_E_COMM;
    if (c->comm_nranks > 1 && h_band < SR_HALO) return SR_E_HALO;  // a neighbour reads SR_HALO rows of this band
    HIPCHK(c, hipSetDevice(c->device));
    g = band_geom(c, h_band, w, px_bytes);
    const int rc = sr_ensure_buf(c, &c->d_ext, &c->ext_cap, (size_t)g.h_ext * g.row_bytes);
    if (rc != SR_OK) return rc;
    // Two event pairs time every sharded step on the band's stream, profiling or not (an event record costs the stream nothing):
    // the whole step of this context and the halo exchange inside it.  They are read -- with a wait for the second event -- only when
    // the caller asks (sr_last_timing / sr_last_comm_ms), so the call itself stays asynchronous and the bands of a one-process call
    // are not serialised the way per-stage profiling serialises them.
    for (auto& e : c->ev_comm) if (!e) HIPCHK(c, hipEventCreate(&e));
    for (auto& e : c->ev_band) if (!e) HIPCHK(c, hipEventCreate(&e));
    for (auto& e : c->ev_wait) if (!e) HIPCHK(c, hipEventCreate(&e));
    if (!c->ev_xfork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_xfork, hipEventDisableTiming));
    c->comm_pending = c->band_pending = c->wait_pending = false;
    HIPCHK(c, hipEventRecord(c->ev_band[0], s));
    if (c->comm_nranks > 1) {
        // the exchange stream starts where the band's stream stands now: the caller's producers of d_band are done, and so is every
        // earlier call's reader of d_ext (the receives land there)
        const int rc2 = sr_ensure_fork_resources(c);
        if (rc2 != SR_OK) return rc2;
        HIPCHK(c, hipEventRecord(c->ev_xfork, s));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_xfork, 0));
    }
    HIPCHK(c, hipMemcpyAsync((char*)c->d_ext + (size_t)g.top * g.row_bytes, d_band, (size_t)h_band * g.row_bytes,
                             hipMemcpyDeviceToDevice, s));
    return SR_OK;
}

// the gate the band's conv stack waits behind: the exchange's end event, the halo rows either side, the wait's own event pair
sr_halo_gate gate_of(sr_ctx* c, const BandGeom& g) {
    sr_halo_gate gate;
    gate.ready = c->ev_comm[1]; gate.top = g.top; gate.bot = g.bot;
    gate.mark[0] = c->ev_wait[0]; gate.mark[1] = c->ev_wait[1];
    return gate;
}

// ---- per-layer feature halos (sr_set_experiment "halo" = "layers") -------------------------------------------
// Instead of recomputing 14 input rows' worth of every stage per interior band, every stage computes the band's own rows and the
// neighbours' edge rows of its OUTPUT are exchanged before the next stage: f 2 rows (the 5x5 layers read them), l1 / l2 / l3 one row
// each.  Whole padded map rows travel (pitch x 128 B: 984 KB for f, 492 KB for the others at 3840 px), so the receiver's border columns
// get the sender's zeros.  The input exchange stays as it is (stage 0 reads 2 of its 7 rows, the residual of the last stage 1).  Same
// values computed once each instead of twice: bit-identical to the recompute form and to the undivided call.
constexpr int layer_rows(int st) { return st == 0 ? 2 : 1; }

struct LayerRows {  // what stage st's exchange moves for one context: null where there is no neighbour
    float *top_send, *top_recv, *bot_send, *bot_recv;
    size_t count;  // floats per direction
};
LayerRows layer_rows_of(const sr_band_pass* bp, const BandGeom& g, int h_band, int st) {
    const int k = layer_rows(st);
    LayerRows r{nullptr, nullptr, nullptr, nullptr, (size_t)k * sr_band_pass_row_floats(bp)};
    if (g.top) { r.top_send = sr_band_pass_row(bp, st, g.top); r.top_recv = sr_band_pass_row(bp, st, g.top - k); }
    if (g.bot) { r.bot_send = sr_band_pass_row(bp, st, g.top + h_band - k); r.bot_recv = sr_band_pass_row(bp, st, g.top + h_band); }
    return r;
}

int post_layer_exchange(sr_ctx* c, Rccl* R, const LayerRows& r, hipStream_t s) {
    ncclComm_t comm = (ncclComm_t)c->comm;
    if (r.top_send) {
        NCCLCHK(c, R->Send(r.top_send, r.count, ncclFloat, c->comm_rank - 1, comm, s));
        NCCLCHK(c, R->Recv(r.top_recv, r.count, ncclFloat, c->comm_rank - 1, comm, s));
    }
    if (r.bot_send) {
        NCCLCHK(c, R->Send(r.bot_send, r.count, ncclFloat, c->comm_rank + 1, comm, s));
        NCCLCHK(c, R->Recv(r.bot_recv, r.count, ncclFloat, c->comm_rank + 1, comm, s));
    }
    return SR_OK;
}

// one rank's band pass in that mode: stage, exchange, stage, ... all on `s`
int run_layers_rank(sr_ctx* c, Rccl* R, bool u8, int img_ch, int h_band, int w, const BandGeom& g, void* d_out, const sr_halo_gate* gate, hipStream_t s) {
    sr_band_pass* bp = nullptr;
    int rc = sr_band_pass_begin(c, c->d_ext, u8, img_ch, g.h_ext, w, g.top, g.bot, d_out, u8, s, true, gate, &bp);
    for (int st = 0; st < 5 && rc == SR_OK; ++st) {
        rc = sr_band_pass_stage(bp, st);
        if (rc != SR_OK || st == 4) break;
        const LayerRows r = layer_rows_of(bp, g, h_band, st);
        ncclResult_t gs = R->GroupStart();
        if (gs != ncclSuccess) { c->last_nccl = (int)gs; rc = SR_E_COMM; break; }
        rc = post_layer_exchange(c, R, r, s);
        const ncclResult_t ge = R->GroupEnd();
        if (rc == SR_OK && ge != ncclSuccess) { c->last_nccl = (int)ge; rc = SR_E_COMM; }
        if (rc != SR_OK) abort_comm(c, R);
    }
    if (bp) sr_band_pass_end(bp);
    return rc;
}

// one process per GPU: exchange + band pass of this rank, asynchronous on `s`
int run_sharded(sr_ctx* c, const void* d_band, bool u8, int img_ch, int h_band, int w, void* d_out, hipStream_t s) {
    if (!d_out) return SR_E_INVALID;
    if (u8 && img_ch != 3 && img_ch != 4) return SR_E_INVALID;
    if (c && c->comm_nranks > 1 && c->comm_local) return SR_E_COMM;  // the neighbours' rows are only known to sr_upscale_sharded_*_all
    sr_device_guard restore_device;
    BandGeom g;
    int rc = prepare_band(c, d_band, h_band, w, u8 ? (size_t)img_ch : 3 * sizeof(float), g, s);
    if (rc != SR_OK) return rc;
    sr_halo_gate gate;
    if (c->comm_nranks > 1) {
        Rccl* R = rccl();
        if (!R) return SR_E_COMM;
        hipStream_t xs = c->stream2;
        HIPCHK(c, hipEventRecord(c->ev_comm[0], xs));
        NCCLCHK(c, R->GroupStart());
        rc = post_exchange(c, R, d_band, h_band, g, xs);
        const ncclResult_t ge = R->GroupEnd();
        if (rc != SR_OK || ge != ncclSuccess) {  // possibly half-posted: nothing of it may stay queued
            if (rc == SR_OK) c->last_nccl = (int)ge;
            abort_comm(c, R);
            (void)hipStreamWaitEvent(s, c->ev_xfork, 0);
            return SR_E_COMM;
        }
        HIPCHK(c, hipEventRecord(c->ev_comm[1], xs));
        gate = gate_of(c, g);
    }
    if (c->layer_halos && c->comm_nranks > 1) rc = run_layers_rank(c, rccl(), u8, img_ch, h_band, w, g, d_out, &gate, s);
    else rc = sr_run_stack_auto(c, c->d_ext, u8, img_ch, 1, g.h_ext, w, g.top, g.bot, d_out, u8, s, c->comm_nranks > 1 ? &gate : nullptr);
    if (rc != SR_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev_band[1], s));
    c->comm_pending = c->wait_pending = c->comm_nranks > 1;
    // (with per-stage profiling on, sr_last_timing reports the conv stack's own events -- which the stage-by-stage pass of the layer-halo form does not record)
    c->band_pending = !c->profiling || (c->layer_halos && c->comm_nranks > 1);
    return SR_OK;
}

// one process, all ranks: every band of one image, synchronous
int run_sharded_all(sr_ctx* const* ctxs, int n, const void* const* d_bands, const int* h_bands, bool u8, int img_ch, int w,
                    void* const* d_outs) {
    if (!d_bands || !h_bands || !d_outs) return SR_E_INVALID;
    int rc = sr_check_context_set(ctxs, n);
    if (rc != SR_OK) return rc;
    if (u8 && img_ch != 3 && img_ch != 4) return SR_E_INVALID;
    for (int k = 0; k < n; ++k)
        if (ctxs[k]->comm_nranks != n || ctxs[k]->comm_rank != k || !d_outs[k]) return SR_E_INVALID;
    const bool local = ctxs[0]->comm_local;
    for (int k = 1; k < n; ++k)
        if (ctxs[k]->comm_local != local || ctxs[k]->layer_halos != ctxs[0]->layer_halos) return SR_E_INVALID;
    sr_device_guard restore_device;
    Rccl* R = n > 1 && !local ? rccl() : nullptr;
    if (n > 1 && !local && !R) return SR_E_COMM;
    std::vector<BandGeom> g(n);
    const size_t px = u8 ? (size_t)img_ch : 3 * sizeof(float);
    if (n > 1)
        for (int k = 0; k < n; ++k)
            if (h_bands[k] < SR_HALO) return SR_E_HALO;  // a neighbour reads SR_HALO rows of every band
    // From here on work is queued on the contexts' streams: whatever fails, every stream is drained before the call
    // returns (the copies read the caller's buffers).
    auto hip = [&](sr_ctx* c, hipError_t e) -> int {
        if (e == hipSuccess) return SR_OK;
        c->last_hip = (int)e;
        return e == hipErrorOutOfMemory ? SR_E_NOMEM : SR_E_HIP;
    };
    for (int k = 0; k < n && rc == SR_OK; ++k) {
        rc = hip(ctxs[k], hipSetDevice(ctxs[k]->device));
        if (rc == SR_OK) rc = sr_ensure_streams(ctxs[k], false);
        if (rc == SR_OK) rc = prepare_band(ctxs[k], d_bands[k], h_bands[k], w, px, g[k], ctxs[k]->stream);
    }
    if (rc == SR_OK && local) {
        // every context pulls its halos from the neighbours' bands -- caller buffers that are complete before this
        // (synchronous) call, so no cross-stream ordering is needed
        for (int k = 0; k < n && rc == SR_OK; ++k) {
            sr_ctx* c = ctxs[k];
            const size_t halo = (size_t)SR_HALO * g[k].row_bytes;
            char* ext = (char*)c->d_ext;
            rc = hip(c, hipSetDevice(c->device));
            if (rc == SR_OK) rc = hip(c, hipEventRecord(c->ev_comm[0], c->stream2));
            if (rc == SR_OK && g[k].top)
                rc = hip(c, hipMemcpyPeerAsync(ext, c->device, (const char*)d_bands[k - 1] + (size_t)(h_bands[k - 1] - SR_HALO) * g[k].row_bytes,
                                               ctxs[k - 1]->device, halo, c->stream2));
            if (rc == SR_OK && g[k].bot)
                rc = hip(c, hipMemcpyPeerAsync(ext + (size_t)(g[k].top + h_bands[k]) * g[k].row_bytes, c->device, d_bands[k + 1],
                                               ctxs[k + 1]->device, halo, c->stream2));
            if (rc == SR_OK) rc = hip(c, hipEventRecord(c->ev_comm[1], c->stream2));
        }
    } else if (rc == SR_OK && n > 1) {
        for (int k = 0; k < n && rc == SR_OK; ++k) {
            rc = hip(ctxs[k], hipSetDevice(ctxs[k]->device));
            if (rc == SR_OK) rc = hip(ctxs[k], hipEventRecord(ctxs[k]->ev_comm[0], ctxs[k]->stream2));
        }
        ncclResult_t gs = rc == SR_OK ? R->GroupStart() : ncclSuccess;
        if (gs != ncclSuccess) { ctxs[0]->last_nccl = (int)gs; rc = SR_E_COMM; }
        const bool grouped = rc == SR_OK;
        for (int k = 0; k < n && rc == SR_OK; ++k) {
            (void)hipSetDevice(ctxs[k]->device);
            rc = post_exchange(ctxs[k], R, d_bands[k], h_bands[k], g[k], ctxs[k]->stream2);
        }
        if (grouped) {
            const ncclResult_t ge = R->GroupEnd();
            if (rc == SR_OK && ge != ncclSuccess) { ctxs[0]->last_nccl = (int)ge; rc = SR_E_COMM; }
        }
        for (int k = 0; k < n && rc == SR_OK; ++k) {
            (void)hipSetDevice(ctxs[k]->device);
            rc = hip(ctxs[k], hipEventRecord(ctxs[k]->ev_comm[1], ctxs[k]->stream2));
        }
        // a failure inside the group may have left some ranks' operations queued without partners: abort every
        // communicator of the set, so that the drain below returns instead of waiting for them for ever
        if (rc != SR_OK)
            for (int k = 0; k < n; ++k) { (void)hipSetDevice(ctxs[k]->device); abort_comm(ctxs[k], R); }
    }
    const bool layers = n > 1 && ctxs[0]->layer_halos;
    if (rc == SR_OK && layers) {
        // Feature halos, all ranks from this thread: the stages go out stage by stage over all contexts, and between two stages every
        // context gets its neighbours' edge rows -- by peer copy on its own stream behind the neighbours' "stage done" events (local
        // transport), or by one grouped send / receive round (RCCL).
        std::vector<sr_band_pass*> bp(n, nullptr);
        std::vector<sr_halo_gate> gates(n);
        for (int k = 0; k < n && rc == SR_OK; ++k) {
            sr_ctx* c = ctxs[k];
            rc = hip(c, hipSetDevice(c->device));
            for (auto& e : c->ev_layer) if (rc == SR_OK && !e) rc = hip(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            gates[k] = gate_of(c, g[k]);
            if (rc == SR_OK) rc = sr_band_pass_begin(c, c->d_ext, u8, img_ch, g[k].h_ext, w, g[k].top, g[k].bot, d_outs[k], u8, c->stream, true, &gates[k], &bp[k]);
        }
        for (int st = 0; st < 5 && rc == SR_OK; ++st) {
            for (int k = 0; k < n && rc == SR_OK; ++k) {
                rc = sr_band_pass_stage(bp[k], st);
                if (rc == SR_OK && st < 4) { rc = hip(ctxs[k], hipSetDevice(ctxs[k]->device)); if (rc == SR_OK) rc = hip(ctxs[k], hipEventRecord(ctxs[k]->ev_layer[st], ctxs[k]->stream)); }
            }
            if (st == 4 || rc != SR_OK) break;
            if (local) {
                for (int k = 0; k < n && rc == SR_OK; ++k) {
                    sr_ctx* c = ctxs[k];
                    const LayerRows mine = layer_rows_of(bp[k], g[k], h_bands[k], st);
                    rc = hip(c, hipSetDevice(c->device));
                    if (rc == SR_OK && mine.top_recv) {  // the upper neighbour's last own rows
                        const LayerRows theirs = layer_rows_of(bp[k - 1], g[k - 1], h_bands[k - 1], st);
                        rc = hip(c, hipStreamWaitEvent(c->stream, ctxs[k - 1]->ev_layer[st], 0));
                        if (rc == SR_OK) rc = hip(c, hipMemcpyPeerAsync(mine.top_recv, c->device, theirs.bot_send, ctxs[k - 1]->device, mine.count * sizeof(float), c->stream));
                    }
                    if (rc == SR_OK && mine.bot_recv) {  // the lower neighbour's first own rows
                        const LayerRows theirs = layer_rows_of(bp[k + 1], g[k + 1], h_bands[k + 1], st);
                        rc = hip(c, hipStreamWaitEvent(c->stream, ctxs[k + 1]->ev_layer[st], 0));
                        if (rc == SR_OK) rc = hip(c, hipMemcpyPeerAsync(mine.bot_recv, c->device, theirs.top_send, ctxs[k + 1]->device, mine.count * sizeof(float), c->stream));
                    }
                }
            } else {
                ncclResult_t gs = R->GroupStart();
                if (gs != ncclSuccess) { ctxs[0]->last_nccl = (int)gs; rc = SR_E_COMM; break; }
                for (int k = 0; k < n && rc == SR_OK; ++k) {
                    (void)hipSetDevice(ctxs[k]->device);
                    rc = post_layer_exchange(ctxs[k], R, layer_rows_of(bp[k], g[k], h_bands[k], st), ctxs[k]->stream);
                }
                const ncclResult_t ge = R->GroupEnd();
                if (rc == SR_OK && ge != ncclSuccess) { ctxs[0]->last_nccl = (int)ge; rc = SR_E_COMM; }
                if (rc != SR_OK)
                    for (int k = 0; k < n; ++k) { (void)hipSetDevice(ctxs[k]->device); abort_comm(ctxs[k], R); }
            }
        }
        for (int k = 0; k < n; ++k) {
            if (bp[k]) sr_band_pass_end(bp[k]);
            if (rc == SR_OK) rc = hip(ctxs[k], hipSetDevice(ctxs[k]->device));
            if (rc == SR_OK) rc = hip(ctxs[k], hipEventRecord(ctxs[k]->ev_band[1], ctxs[k]->stream));
            if (rc == SR_OK) { ctxs[k]->comm_pending = ctxs[k]->wait_pending = true; ctxs[k]->band_pending = true; }  // (no per-stage events in this form)
        }
    }
    for (int k = 0; k < n && rc == SR_OK && !layers; ++k) {
        const sr_halo_gate gate = gate_of(ctxs[k], g[k]);
        rc = sr_run_stack_auto(ctxs[k], ctxs[k]->d_ext, u8, img_ch, 1, g[k].h_ext, w, g[k].top, g[k].bot, d_outs[k], u8, ctxs[k]->stream,
                               n > 1 ? &gate : nullptr);
        if (rc == SR_OK) rc = hip(ctxs[k], hipSetDevice(ctxs[k]->device));
        if (rc == SR_OK) rc = hip(ctxs[k], hipEventRecord(ctxs[k]->ev_band[1], ctxs[k]->stream));
        if (rc == SR_OK) { ctxs[k]->comm_pending = ctxs[k]->wait_pending = n > 1; ctxs[k]->band_pending = !ctxs[k]->profiling; }
    }
    int first = rc;
    for (int k = 0; k < n; ++k) {  // drain every device, also on failure (both streams: a failed call may have left the exchange unjoined)
        (void)hipSetDevice(ctxs[k]->device);
        hipError_t e = hipStreamSynchronize(ctxs[k]->stream);
        if (e == hipSuccess && n > 1 && ctxs[k]->stream2) e = hipStreamSynchronize(ctxs[k]->stream2);
        if (e != hipSuccess && first == SR_OK) { ctxs[k]->last_hip = (int)e; first = SR_E_HIP; }
    }
    return first;
}

}  // namespace

void sr_comm_release(sr_ctx* c) {
    if (!c) return;
    if (c->comm) {
        if (Rccl* R = rccl()) (void)R->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr;
    }
    c->comm_rank = 0; c->comm_nranks = 1; c->comm_local = false; c->comm_broken = false;
    if (c->d_ext) { (void)hipFree(c->d_ext); c->d_ext = nullptr; c->ext_cap = 0; }
    for (auto& e : c->ev_comm) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : c->ev_band) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : c->ev_wait) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (c->ev_xfork) { (void)hipEventDestroy(c->ev_xfork); c->ev_xfork = nullptr; }
    for (auto& e : c->ev_layer) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    c->comm_pending = c->band_pending = c->wait_pending = false;
}

extern "C" {

int sr_comm_available(void) { return rccl() ? 1 : 0; }

int sr_comm_unique_id(uint8_t* id, size_t cap) {
    if (!id || cap < SR_COMM_ID_BYTES) return SR_E_INVALID;
    static_assert(SR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "srhip.h and rccl.h disagree on the id size");
    Rccl* R = rccl();
    if (!R) return SR_E_COMM;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return SR_E_COMM;
    memcpy(id, u.internal, SR_COMM_ID_BYTES);
    return SR_OK;
}

static int comm_events(sr_ctx* c) {
    for (auto& e : c->ev_comm) if (!e) HIPCHK(c, hipEventCreate(&e));
    return SR_OK;
}

int sr_comm_init_rank(sr_ctx* c, const uint8_t* id, size_t id_len, int rank, int nranks) {
    if (!c || nranks < 1 || rank < 0 || rank >= nranks) return SR_E_INVALID;
    if (c->graph != SR_GRAPH_SR_NET) return SR_E_INVALID;
    sr_device_guard restore_device;
    HIPCHK(c, hipSetDevice(c->device));
    sr_comm_release(c);
    int rc = comm_events(c);
    if (rc != SR_OK) return rc;
    if (nranks > 1) {
        if (!id || id_len < SR_COMM_ID_BYTES) return SR_E_INVALID;
        Rccl* R = rccl();
        if (!R) return SR_E_COMM;
        ncclUniqueId u;
        memcpy(u.internal, id, SR_COMM_ID_BYTES);
        ncclComm_t comm = nullptr;
        NCCLCHK(c, R->CommInitRank(&comm, nranks, u, rank));
        c->comm = comm;
    }
    c->comm_rank = rank; c->comm_nranks = nranks;
    return SR_OK;
}

int sr_comm_init_all(sr_ctx* const* ctxs, int n) {
    int rc = sr_check_context_set(ctxs, n);
    if (rc != SR_OK) return rc;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < k; ++j)
            if (ctxs[j]->device == ctxs[k]->device) return SR_E_INVALID;  // RCCL: one rank per device
    // All or nothing: whatever fails below, EVERY context of the set is left in the released state (rank 0 of 1, no
    // communicator, single-device calls work as before) -- never some of them ranked and others not.
    sr_device_guard restore_device;
    std::vector<int> devs(n);
    auto release_all = [&]() {
        for (int k = 0; k < n; ++k) { (void)hipSetDevice(ctxs[k]->device); sr_comm_release(ctxs[k]); }
    };
    release_all();
    for (int k = 0; k < n; ++k) {
        devs[k] = ctxs[k]->device;
        hipError_t e = hipSetDevice(devs[k]);
        if (e == hipSuccess) rc = comm_events(ctxs[k]); else { (void)hipGetLastError(); ctxs[k]->last_hip = (int)e; rc = SR_E_HIP; }
        if (rc != SR_OK) { release_all(); return rc; }
    }
    if (n > 1) {
        Rccl* R = rccl();
        if (!R) { release_all(); return SR_E_COMM; }
        std::vector<ncclComm_t> comms(n, nullptr);
        const ncclResult_t r = R->CommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            for (int k = 0; k < n; ++k) {  // RCCL may have created some of them before it failed
                if (comms[k]) (void)R->CommAbort(comms[k]);
                ctxs[k]->last_nccl = (int)r;
            }
            release_all();
            return SR_E_COMM;
        }
        for (int k = 0; k < n; ++k) ctxs[k]->comm = comms[k];
    }
    for (int k = 0; k < n; ++k) { ctxs[k]->comm_rank = k; ctxs[k]->comm_nranks = n; }
    return SR_OK;
}

int sr_comm_init_local(sr_ctx* const* ctxs, int n) {
    int rc = sr_check_context_set(ctxs, n);
    if (rc != SR_OK) return rc;
    sr_device_guard restore_device;
    for (int k = 0; k < n; ++k) {
        if (ctxs[k]->graph != SR_GRAPH_SR_NET) return SR_E_INVALID;
        HIPCHK(ctxs[k], hipSetDevice(ctxs[k]->device));
        sr_comm_release(ctxs[k]);
        rc = comm_events(ctxs[k]);
        if (rc != SR_OK) return rc;
        for (int j : {k - 1, k + 1}) {  // direct xGMI access to the two neighbours (without it the copy is staged)
            if (j < 0 || j >= n || ctxs[j]->device == ctxs[k]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ctxs[k]->device, ctxs[j]->device) != hipSuccess || !can) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ctxs[k]->last_hip = (int)e; return SR_E_HIP; }
            (void)hipGetLastError();
        }
    }
    for (int k = 0; k < n; ++k) { ctxs[k]->comm_rank = k; ctxs[k]->comm_nranks = n; ctxs[k]->comm_local = n > 1; }
    return SR_OK;
}

void sr_comm_destroy(sr_ctx* c) {
    if (!c) return;
    sr_device_guard restore_device;
    (void)hipSetDevice(c->device);
    sr_comm_release(c);
}

int sr_comm_rank(sr_ctx* c, int* rank, int* nranks) {
    if (!c) return SR_E_INVALID;
    if (rank) *rank = c->comm_rank;
    if (nranks) *nranks = c->comm_nranks;
    return SR_OK;
}

int sr_last_comm_error(sr_ctx* c) { return c ? c->last_nccl : 0; }

int sr_last_comm_ms(sr_ctx* c, double* comm_ms) {
    if (!c || !comm_ms) return SR_E_INVALID;
    if (c->comm_pending) {  // waits for the exchange of the last sharded call (not for its kernels)
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->ev_comm[1]));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_comm[0], c->ev_comm[1]));
        c->comm_ms = ms;
        c->comm_pending = false;
    }
    *comm_ms = c->comm_ms;
    return SR_OK;
}

int sr_last_comm_exposed_ms(sr_ctx* c, double* exposed_ms) {
    if (!c || !exposed_ms) return SR_E_INVALID;
    if (c->wait_pending) {  // waits for the band's stream to get past its wait for the exchange (not for the kernels behind it)
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(c->ev_wait[1]));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_wait[0], c->ev_wait[1]));
        c->comm_exposed_ms = ms;
        c->wait_pending = false;
    }
    *exposed_ms = c->comm_exposed_ms;
    return SR_OK;
}

int sr_upscale_sharded_f32_dev(sr_ctx* c, const float* d_band, int h_band, int w, float* d_out, void* stream) {
    return run_sharded(c, d_band, false, 3, h_band, w, d_out, (hipStream_t)stream);
}

int sr_upscale_sharded_rgba8_dev(sr_ctx* c, const uint8_t* d_band, int in_channels, int h_band, int w, uint8_t* d_out,
                                 void* stream) {
    return run_sharded(c, d_band, true, in_channels, h_band, w, d_out, (hipStream_t)stream);
}

int sr_upscale_sharded_f32_all(sr_ctx* const* ctxs, int n, const float* const* d_bands, const int* h_bands, int w,
                               float* const* d_outs) {
    return run_sharded_all(ctxs, n, (const void* const*)d_bands, h_bands, false, 3, w, (void* const*)d_outs);
}

int sr_upscale_sharded_rgba8_all(sr_ctx* const* ctxs, int n, const uint8_t* const* d_bands, int in_channels,
                                 const int* h_bands, int w, uint8_t* const* d_outs) {
    return run_sharded_all(ctxs, n, (const void* const*)d_bands, h_bands, true, in_channels, w, (void* const*)d_outs);
}

}  // extern "C"
