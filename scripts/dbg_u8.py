import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle, rusty_sr_amd as r
from conftest import synth_u8
p = r.rsr.builtin("imagenet")
eng = r.Engine(p)
for (h, w) in ((64, 96), (256, 256), (1080, 1920)):
    px = synth_u8(2, 1, h, w)
    x = oracle.img_to_data(px)
    f32 = eng.upscale_f32(x)
    f0a = eng.read_feature(0, h, w)
    u8 = eng.upscale_rgba8(px)
    f0b = eng.read_feature(0, h, w)
    q = oracle.data_to_rgba8(f32)
    d = (u8[..., :3].astype(int) - q[..., :3].astype(int))
    print(h, w, "u8 mismatches", (d != 0).sum(), "of", d.size, "feature0 maxdiff", np.abs(f0a - f0b).max(),
          "x div check", np.abs(x - px.astype(np.float32) / np.float32(255)).max())
    if (d != 0).any():
        idx = np.argwhere(d != 0)[:5]
        for i in idx:
            v = f32[tuple(i)]
            print("   at", i, "f32", v, "255v+.5", 255.0 * float(v) + 0.5, "u8 path", u8[tuple(i[:3])], "q", q[tuple(i[:3])])
