/*
 * srhip_experimental.h -- tuning switches of libsrhip.so that are NOT part of the drop-in C ABI (include/srhip.h).
 *
 * Nothing here has a counterpart in the reference (millardjn/rusty_sr has no tuning surface: src/main.rs:33-127 is its
 * whole CLI) and nothing here changes a result bit.  The switches exist for interleaved A/B timing (scripts/ab_libs.py,
 * scripts/fork_ab.py, scripts/band_profile.py) and for the tests that prove bit-identity across kernel forms and tile
 * plans (tests/test_gpu_parity.py).  Keys, values and defaults may change between builds; a host that binds
 * include/srhip.h alone (rust_host/src/srhip.rs, the C++ CLI) never needs this file.
 */
#ifndef SRHIP_EXPERIMENTAL_H
#define SRHIP_EXPERIMENTAL_H

#include "srhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sr_set_experiment(ctx, key, value).  key "th": tile height, value "" (automatic: 8-row tiles ended by 4-row tiles, or 4-row tiles only for
 * small launches), "4" / "8" (all stages) or five digits (one per stage); "tail": how many 4-row tiles end a launch of
 * 8-row tiles, in units of the resident workgroups ("" automatic: 1 where it pays, "0" none); "pipe": "none" forces the first
 * form of the stage kernels (one tile class per launch), "all" the pipe form also for small launches ("" automatic); "bw": width in tiles of the column blocks the tile queue walks
 * ("" automatic, "0" plain row-major); "bands": the host pipeline cuts one large image into that many equal row bands
 * ("" / "0": its own plan); "rows": the bands' heights themselves, "r0,r1,..." top to bottom, computed in order on one stream, or with a
 * leading '=' on alternating streams (used when they add up to the rows of the call); "geo": "0" keeps equal bands where the plan would
 * shrink them geometrically; "halo": what a sharded call (sr_upscale_sharded_*) exchanges -- "" / "input": 7 input rows per
 * neighbour, the overlap recomputed; "layers": additionally the edge rows of every layer's output after its stage
 * (f 2 rows, l1 / l2 / l3 one each), nothing recomputed (8(e)(ii)); every context of a call must say the same.
 * "fork": a lone image through a device entry point as two row bands on two streams -- "" automatic, "0" never, "1" always, N > 1 always
 * with N rows in the first band; "forkshare" / "forkmin": the first band's share of the rows, the automatic rule's threshold in rounds of
 * tiles; "forktune": "" / "1" -- for mid-size shapes (0.55-12 rounds of tiles) the automatic choice is MEASURED on the caller's own calls
 * (four undivided, four forked, timed by event pairs that are queried, never waited for; then the faster plan stays for that shape),
 * "0" -- the rule alone; setting it forgets what was measured.
 * Defaults come from SRHIP_TH / SRHIP_TAIL / SRHIP_PIPE / SRHIP_BW / SRHIP_BANDS / SRHIP_ROWS / SRHIP_GEO / SRHIP_HALO, read once in sr_create.
 * Unknown key: SR_E_INVALID. */
int sr_set_experiment(sr_ctx* ctx, const char* key, const char* value);

/* What a switch has learned, as text (NUL-terminated, into buf[cap]).  key "forktune": one line per shape the fork tuner has met,
 * "HxW+halo_top+halo_bot precision io state undivided_ms forked_ms" with state measuring | undivided | forked (0.0000: not measured yet).
 * Unknown key, or a buffer too small: SR_E_INVALID. */
int sr_get_experiment(sr_ctx* ctx, const char* key, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SRHIP_EXPERIMENTAL_H */
