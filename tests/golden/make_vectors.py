"""Generate tests/golden/vectors_torch_f64.npz: node-by-node outputs of the sr_net(3) graph from an
INDEPENDENT second restatement ("PyTorch in this container can be used as an
independent second restatement when authoring fixtures").  Nothing here shares code with
oracle/sr_oracle.c: convolutions are torch.nn.functional.conv2d in float64, the residual is
F.interpolate(bilinear, align_corners=False), the Expand op is written as an index shuffle.
The C oracle, and the HIP engine, are then tested against these vectors.

    python tests/golden/make_vectors.py        # needs only torch + the .rsr blobs in rusty_sr_amd/res

Cases (all u8 inputs, seeded): a 24x20 crop of butterfly_lr.png (imagenet weights), a 15x15 image
whose every pixel is within 7 px of a border (anime weights: zero padding on all four sides of every
layer), and the degenerate 1x1 and 2x3 images (imagenetlinear weights)."""
import os
import struct

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# op insertion order of the reference's network.rs: name, count, shape
LAYOUT = [("conv0", (32, 5, 5, 3)), ("f_bias", (32,)), ("f_activ", (32,)), ("expand_bias", (27,)), ("l1_bias", (32,)),
          ("l2_bias", (32,)), ("l3_bias", (32,)), ("l1_activ", (32,)), ("l2_activ", (32,)), ("l3_activ", (32,)),
          ("conv1", (32, 5, 5, 32)), ("conv2", (32, 5, 5, 32)), ("conv3", (32, 5, 5, 32)), ("conv5", (32, 3, 3, 32)),
          ("conv6", (32, 3, 3, 32)), ("conv7", (27, 3, 3, 32)), ("conv8", (32, 3, 3, 32)), ("conv9", (27, 3, 3, 32)),
          ("conv10", (27, 3, 3, 32))]


def read_rsr(name):
    blob = open(os.path.join(ROOT, "rusty_sr_amd", "res", name + ".rsr"), "rb").read()
    (n,) = struct.unpack_from("<I", blob, 0)
    assert len(blob) == 4 + 8 * n and all(s == 4 for s in struct.unpack_from(f"<{n}I", blob, 4))
    flat = np.frombuffer(blob, dtype="<f4", count=n, offset=4 + 4 * n)
    out, pos = {}, 0
    for key, shape in LAYOUT:
        cnt = int(np.prod(shape))
        out[key] = torch.from_numpy(flat[pos:pos + cnt].astype(np.float64).reshape(shape))
        pos += cnt
    assert pos == n
    return out


def conv(x, w):  # x (1,C,H,W); w stored [O][KH][KW][I]; cross-correlation, zero "same" padding
    return F.conv2d(x, w.permute(0, 3, 1, 2).contiguous(), padding=w.shape[1] // 2)


def belu(x, bias, beta):
    x = x + bias.view(1, -1, 1, 1)
    return beta.view(1, -1, 1, 1) * x + torch.sqrt(x * x + 1.0) - 1.0


def graph(p, px_u8):
    x = torch.from_numpy(px_u8.astype(np.float64) / 255.0).permute(2, 0, 1)[None]
    f = belu(conv(x, p["conv0"]), p["f_bias"], p["f_activ"])
    l1 = belu(conv(f, p["conv1"]), p["l1_bias"], p["l1_activ"])
    l2 = belu(conv(f, p["conv2"]) + conv(l1, p["conv5"]), p["l2_bias"], p["l2_activ"])
    l3 = belu(conv(f, p["conv3"]) + conv(l1, p["conv6"]) + conv(l2, p["conv8"]), p["l3_bias"], p["l3_activ"])
    e = conv(l1, p["conv7"]) + conv(l2, p["conv9"]) + conv(l3, p["conv10"]) + p["expand_bias"].view(1, -1, 1, 1)
    _, _, H, W = x.shape
    out = F.interpolate(x, scale_factor=3, mode="bilinear", align_corners=False)
    # Expand: out[3y+dy][3x+dx][c] += e[y][x][(dy*3+dx)*3+c]   (colour fastest; NOT pixel_shuffle's order)
    out = out + e.view(1, 3, 3, 3, H, W).permute(0, 3, 4, 1, 5, 2).reshape(1, 3, 3 * H, 3 * W)
    nhwc = lambda t: t[0].permute(1, 2, 0).contiguous().numpy()
    return {"f": nhwc(f), "l1": nhwc(l1), "l2": nhwc(l2), "l3": nhwc(l3), "e": nhwc(e), "out": nhwc(out)}


def main():
    from PIL import Image
    rng = np.random.default_rng(20260926)
    bf = np.array(Image.open(os.path.join(HERE, "butterfly_lr.png")).convert("RGB"))
    cases = {
        "crop": ("imagenet", bf[60:84, 100:120].copy()),
        "border": ("anime", rng.integers(0, 256, (15, 15, 3), dtype=np.uint8)),
        "one": ("imagenetlinear", rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)),
        "twothree": ("imagenetlinear", rng.integers(0, 256, (2, 3, 3), dtype=np.uint8)),
    }
    blob = {}
    for name, (weights, px) in cases.items():
        nodes = graph(read_rsr(weights), px)
        blob[f"{name}.weights"] = np.array(weights)
        blob[f"{name}.px"] = px
        for k, v in nodes.items():
            # node data as f32 (what the engines produce); the feature maps of the big case are kept
            # for 8 of the 32 channels to keep the fixture small
            v = v.astype(np.float32)
            blob[f"{name}.{k}"] = v[..., ::4] if (name == "crop" and k in ("f", "l1", "l2", "l3")) else v
    np.savez_compressed(os.path.join(HERE, "vectors_torch_f64.npz"), **blob)
    print({k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
