#!/bin/bash
# End-to-end wall time of the CLI on a 1920x1080 PNG (host I/O included), on the GPU box.
set -e
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth_u8
from PIL import Image
Image.fromarray(synth_u8(2, 1, 1080, 1920)[0]).save("/tmp/in1080.png")
print("input", os.path.getsize("/tmp/in1080.png") / 1e6, "MB")
PY
for p in f32 split_f16; do
  time rusty_sr_amd/bin/rusty_sr /tmp/in1080.png /tmp/out1080.png --precision $p --timing
done
ls -la /tmp/out1080.png
