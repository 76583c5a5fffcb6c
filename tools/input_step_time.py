#!/usr/bin/env python3
"""Training step with the GPU-side input pipeline in the loop: every step copies a staged uint8 batch from pinned host
memory, runs fr_augment_u8 (fresh crops / flips) and then the usual step.  Compared with the resident-batch bench step."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import bench
from frhip.input_pipeline import GpuTrainTransform


class A: batch = 256; classes = 7000; dtype = "bf16"; model = "IR_50"; sharded_head = False


dev = torch.device("cuda", 0)
model, head, loss_fn, opt, x, y = bench.build_job(A, dev, 0)
step = bench.make_step(model, head, loss_fn, opt, None)
tf = GpuTrainTransform(112)
gen = torch.Generator().manual_seed(1)
staged = [torch.from_numpy(np.random.default_rng(i).integers(0, 256, (A.batch, 112, 112, 3), dtype=np.uint8)).pin_memory()
          for i in range(4)]


def run(with_input, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if with_input:
            xb = tf(staged[i % 4].to(dev, non_blocking=True), generator=gen)
        else:
            xb = x
        step(xb, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(True, 5); run(False, 5)
for _ in range(2):
    print("resident batch: %.2f ms/step   staged uint8 + GPU transform per step: %.2f ms/step" % (run(False, 30), run(True, 30)))
