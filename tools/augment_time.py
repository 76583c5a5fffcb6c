#!/usr/bin/env python3
"""Time fr_augment_u8 (GPU-side train transform) on one batch and the host transform it replaces on one core."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import numpy as np
import torch
from frhip.input_pipeline import GpuTrainTransform
from dataset import TrainTransform
from PIL import Image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tf = GpuTrainTransform(112)
host = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 112, 112, 3), dtype=np.uint8)).pin_memory()
u8 = host.cuda()
crop, flip = tf.draw(B)
crop, flip = crop.cuda(), flip.cuda()
for _ in range(5):
    out = tf(u8, crop, flip)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    out = tf(u8, crop, flip)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
algo = B * (112 * 112 * 3 + 3 * 112 * 112 * 4)  # staged bytes read once + float32 batch written
print("fr_augment_u8 B=%d: %.1f us per batch (incl. host checks), %.1f M images/s, %.0f GB/s algorithmic"
      % (B, ms * 1e3, B / ms / 1e3, algo / ms / 1e6))
t0 = time.perf_counter()
for _ in range(20):
    x = host.cuda(non_blocking=True)
torch.cuda.synchronize()
print("H2D of the staged uint8 batch (pinned): %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
ht = TrainTransform(112)
imgs = [Image.fromarray(host[i].numpy()) for i in range(64)]
t0 = time.perf_counter()
for im in imgs:
    ht(im)
dt = (time.perf_counter() - t0) / len(imgs)
print("host transform (PIL resize + crop + flip + normalise), one core: %.0f us per image = %.0f images/s per core"
      % (dt * 1e6, 1 / dt))
