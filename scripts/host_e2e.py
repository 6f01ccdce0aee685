"""Wall time of the host-pointer entry point (upload + kernels + download) on the GPU box:
pageable vs page-locked caller memory, pipeline off / on.  These are the PCIe-inclusive
rates DESIGN.md quotes beside bench.py's HBM-resident `value`."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import rusty_sr_amd as r  # noqa: E402
from rusty_sr_amd.engine import host_alloc  # noqa: E402
from conftest import synth_u8  # noqa: E402

params = r.rsr.builtin("imagenet")
rows = []
for precision in ("f32", "split_f16"):
    eng = r.Engine(params, precision=precision)
    for name, (n, h, w) in (("1080p", (1, 1080, 1920)), ("4K", (1, 2160, 3840)), ("64x512x512", (64, 512, 512))):
        px = synth_u8(9, 1, h, w).repeat(n, 0) if n > 1 else synth_u8(9, 1, h, w)
        pin_in = host_alloc(px.shape)
        pin_in.array[...] = px
        pin_out = host_alloc((n, 3 * h, 3 * w, 4))
        page_out = np.zeros((n, 3 * h, 3 * w, 4), np.uint8)
        for label, pipe, src, dst in (("pageable, undivided", False, px, page_out), ("pageable, pipelined", True, px, page_out),
                                      ("pinned, undivided", False, pin_in.array, pin_out.array),
                                      ("pinned, pipelined", True, pin_in.array, pin_out.array)):
            eng.set_pipeline(pipe)
            ts = []
            for it in range(6):
                t0 = time.perf_counter()
                eng.upscale_rgba8(src, out=dst)
                ts.append(time.perf_counter() - t0)
            ms = 1e3 * min(ts[1:])
            t = eng.last_timing()
            tot, h2d, d2h = t["total_ms"], t["h2d_ms"], t["d2h_ms"]
            rows.append({"precision": precision, "workload": name, "mode": label, "wall_ms": round(ms, 3),
                         "out_MP_per_s": round(n * 9 * h * w / 1e6 / (ms / 1e3), 1), "kernels_ms": round(tot, 3),
                         "h2d_ms": round(h2d, 3), "d2h_ms": round(d2h, 3)})
            print(json.dumps(rows[-1]), flush=True)
        pin_in.close(); pin_out.close()
    eng.close()
json.dump(rows, open("gpurun_out/host_e2e.json", "w"), indent=1)
