#!/bin/bash
# End-to-end wall time of the CLI (t_cli of SURVEY.md 8(d): process start-up, decode, device, upscale, encode, write) on
# synthetic 1920x1080 and 3840x2160 PNGs, on the GPU box.  Second run of each = warm page cache.
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth_u8
from PIL import Image
for name, seed, h, w in (("in1080", 2, 1080, 1920), ("in2160", 3, 2160, 3840)):
    Image.fromarray(synth_u8(seed, 1, h, w)[0]).save(f"/tmp/{name}.png")
    print(name, os.path.getsize(f"/tmp/{name}.png") / 1e6, "MB")
PY
for img in in1080 in2160; do
  for p in f32 split_f16; do
    for rep in 1 2; do
      t0=$(date +%s.%N)
      rusty_sr_amd/bin/rusty_sr /tmp/$img.png /tmp/out_$img.png --precision $p --timing 2>&1 | grep "timing"
      echo "$img $p run $rep: process wall $(echo "$(date +%s.%N) - $t0" | bc -l | cut -c1-6) s"
    done
  done
  ls -la /tmp/out_$img.png
done
for ext in jpg bmp ppm; do
  t0=$(date +%s.%N)
  rusty_sr_amd/bin/rusty_sr /tmp/in1080.png /tmp/out.$ext --timing 2>&1 | grep "wall"
  echo "in1080 f32 -> .$ext: process wall $(echo "$(date +%s.%N) - $t0" | bc -l | cut -c1-6) s"
done
