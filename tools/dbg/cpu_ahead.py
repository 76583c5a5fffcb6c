"""How far ahead of the GPU is the host?  Per-step host enqueue time (no sync inside the loop) against the GPU's step time, and
the host time of each phase of a step (forward / head+loss / zero_grad / backward / optimizer).  python tools/dbg/cpu_ahead.py"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    a = argparse.Namespace(model=os.environ.get("MODEL", "IR_50"), head=os.environ.get("HEAD", "ArcFace"),
                           classes=int(os.environ.get("CLASSES", 7000)), batch=int(os.environ.get("BATCH", 256)), dtype="bf16",
                           sharded_head=False, resident_batches=4)
    dev = torch.device("cuda:0")
    model, head, loss_fn, opt, xs, ys = bench.build_job(a, dev, 0)
    from frhip import functional as FRF
    from util.utils import accuracy
    FRF.CHECK_LABELS = False
    ph = {k: 0.0 for k in ("forward", "head", "zero_grad", "backward", "step")}

    def step(x, y, acc=None):
        t = [time.perf_counter()]
        feats = model(x); t.append(time.perf_counter())
        logits = head(feats, y)
        loss, _ = loss_fn(logits, y)
        accuracy(logits.data, y, topk=(1, 5)); t.append(time.perf_counter())
        opt.zero_grad(); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if acc is not None:
            for k, d in zip(ph, [t[i + 1] - t[i] for i in range(5)]):
                acc[k] += d
        return loss
    for i in range(6):
        step(xs[i % 4], ys[i % 4])
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    host = []
    for i in range(n):
        s = time.perf_counter()
        step(xs[i % 4], ys[i % 4], ph)
        host.append(time.perf_counter() - s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per step: first %.2f ms, median %.2f ms; loop %.2f ms/step before the final sync, %.2f ms/step with it"
          % (host[0] * 1e3, sorted(host)[n // 2] * 1e3, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    print("host ms per phase:", {k: round(v / n * 1e3, 3) for k, v in ph.items()})


main()
