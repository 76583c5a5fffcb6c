#!/usr/bin/env python3
"""Digest of a few training steps under the current FRHIP_* switches: the loss of every step as hex + a SHA-1 over every
parameter after the last step.  Two switch settings that print the same lines run bit-identical steps.

    python tools/step_digest.py [--model IR_50 --head ArcFace --classes 7000 --batch 256 --steps 3]
"""
import argparse
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="IR_50")
    ap.add_argument("--head", default="ArcFace")
    ap.add_argument("--classes", type=int, default=7000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    device = torch.device("cuda:0")
    args = argparse.Namespace(dtype="bf16", sharded_head=False, resident_batches=4, model=a.model, head=a.head,
                              classes=a.classes, batch=a.batch)
    model, head, loss_fn, opt, xs, ys = bench.build_job(args, device, 0)
    step = bench.make_step(model, head, loss_fn, opt, None)
    losses = []
    for i in range(a.steps):
        losses.append(float(step(xs[i % len(xs)], ys[i % len(ys)])[0]))
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for p in list(model.parameters()) + list(head.parameters()):
        h.update(p.detach().float().cpu().numpy().tobytes())
    sw = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("FRHIP_")) or "default"
    print("%-32s losses %s  parameters sha1 %s" % (sw, " ".join(float(l).hex() for l in losses), h.hexdigest()))


if __name__ == "__main__":
    main()
