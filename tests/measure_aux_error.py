"""How far the parameter-free graphs' f32 entry points (pow as exp2(p * log2 x) on v_log_f32 / v_exp_f32) are from the oracle's libm,
and how many u8 outputs differ (only rounding knife-edges may).  A measurement for DESIGN.md 4d, run by hand on the GPU box:
    python tests/measure_aux_error.py          (lives under tests/ because it uses the oracle)"""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, rusty_sr_amd as r
from bench import synth_u8 as _synth
def synth_u8(seed, n, h, w):
    return _synth(seed, h, w, n=n)
bl, ds = r.bilinear_net(r.FACTOR), r.downsample_net(r.FACTOR)
worst_b = worst_d = 0.0
rng = np.random.default_rng(3)
for h, w in ((130, 257), (64, 512), (301, 99)):
    for x in (oracle.img_to_data(synth_u8(30 + h, 2, h, w)), rng.random((1, h, w, 3), dtype=np.float32)):
        worst_b = max(worst_b, float(np.abs(bl.upscale_f32(x) - oracle.bilinear(x)).max()))
        worst_d = max(worst_d, float(np.abs(ds.upscale_f32(x) - oracle.downsample(x)).max()))
        px = (x * 255).round().astype(np.uint8)
        d = bl.upscale_rgba8(px)[..., :3].astype(int) - oracle.data_to_rgba8(oracle.bilinear(oracle.img_to_data(px)))[..., :3].astype(int)
        print("u8 bilinear mismatches", int((d != 0).sum()), "of", d.size, "max", int(np.abs(d).max()))
print("max |f32 bilinear - oracle|", worst_b, " max |f32 downsample - oracle|", worst_d)
