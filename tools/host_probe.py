#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one training step (no device sync inside the loop)?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch
import bench
class A: batch=256; classes=7000; dtype="bf16"; model="IR_50"
dev = torch.device("cuda", 0)
model, head, loss_fn, opt, x, y = bench.build_job(A, dev, 0)
step = bench.make_step(model, head, loss_fn, opt, None)
for _ in range(5): step(x, y)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): step(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
