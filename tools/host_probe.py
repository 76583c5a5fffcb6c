#!/usr/bin/env python3
"""Where is the host relative to the GPU inside one training step?  For every phase boundary of the bench.py step
(forward enqueued, head + loss, zero_grad, backward enqueued, optimizer) prints the host time at which the boundary was
reached and the GPU time at which the work enqueued up to there finished (HIP events): a boundary where host >= GPU is a
place where the GPU idles waiting for launches.

    python tools/host_probe.py
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
from frhip import functional as FRF  # noqa: E402
from util.utils import accuracy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="IR_50")
    ap.add_argument("--head", default="ArcFace")
    ap.add_argument("--classes", type=int, default=7000)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    args = argparse.Namespace(model=a.model, head=a.head, classes=a.classes, batch=a.batch, dtype="bf16", sharded_head=False,
                              resident_batches=4)
    dev = torch.device("cuda", 0)
    model, head, loss_fn, opt, xs, ys = bench.build_job(args, dev, 0)
    FRF.CHECK_LABELS = False
    names = ["forward", "head", "loss", "accuracy", "zero_grad", "backward", "opt.step"]

    def step(x, y, rec=None):
        def mark():
            if rec is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                rec.append((time.perf_counter(), ev))
        mark()
        feats = model(x)
        mark()
        logits = head(feats, y)
        mark()
        loss, _ = loss_fn(logits, y)
        mark()
        accuracy(logits.data, y, topk=(1, 5))
        mark()
        opt.zero_grad(set_to_none=False)
        mark()
        loss.backward()
        mark()
        opt.step()
        mark()

    for i in range(6):
        step(xs[i % 4], ys[i % 4])
    torch.cuda.synchronize()
    t_host0 = time.perf_counter()
    for rep in range(3):
        step(xs[rep % 4], ys[rep % 4])  # steady state: the host runs ahead
    recs = []
    for rep in range(3):
        r = []
        step(xs[rep % 4], ys[rep % 4], r)
        recs.append(r)
    torch.cuda.synchronize()
    base_t, base_ev = recs[0][0]
    print("phase boundary     host reached (ms)   GPU reached (ms)   host lead (ms)   [times relative to the first boundary]")
    for r in recs:
        for k, (t, ev) in enumerate(r):
            th = (t - base_t) * 1e3
            tg = base_ev.elapsed_time(ev)
            # GPU times are relative to the GPU reaching the first boundary; the host reached it `lead0` earlier
            print("  %-14s %12.3f %18.3f" % ("start" if k == 0 else names[k - 1], th, tg))
        print("  --")


if __name__ == "__main__":
    main()
