#!/usr/bin/env python3
"""In-situ stamps of the warp-specialised weight-gradient kernel: the LAST such launch of a full bf16 training step
(two streams, everything co-running as in bench.py).  Diagnostic build only:

    FRHIP_LIB=stylegan-for-facerec_amd/frhip/lib/libfrhip_stamps.so python tools/stamps_step.py
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from frhip import _lib  # noqa: E402


def main():
    args = argparse.Namespace(model="IR_50", head="ArcFace", classes=7000, batch=256, dtype="bf16", sharded_head=False,
                              resident_batches=4)
    dev = torch.device("cuda", 0)
    model, head, loss_fn, opt, xs, ys = bench.build_job(args, dev, 0)
    step = bench.make_step(model, head, loss_fn, opt, None)
    for i in range(8):
        step(xs[i % 4], ys[i % 4])
    torch.cuda.synchronize()
    buf = torch.zeros(4 * 4096 * 16, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_wgr.argtypes = [ctypes.c_void_p]
    assert dbg.fr_debug_set_stamp_buffer_wgr(ctypes.c_void_p(buf.data_ptr())) == 0
    for i in range(3):
        step(xs[i % 4], ys[i % 4])
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(4, 4096, 16)
    med = lambda v: float(np.median(v))  # noqa: E731
    for region, W in enumerate((14, 28, 56, 112)):
        s = raw[region]
        s = s[s[:, 6] != 0].astype(np.float64)
        if not len(s):
            continue
        nimg = s[:, 6]
        mhz = np.median(s[:, 0] / (s[:, 5] * 10e-9)) / 1e6
        print("last %dx%d weight-gradient launch of the step: %d workgroups, %d images / phases each, in-kernel clock %.0f MHz"
              % (W, W, len(s), int(nimg[0]), mhz))
        print("  kernel (per workgroup)  median %7.0f cycles = %6.2f us   p90 %6.2f us" % (
            med(s[:, 0]), med(s[:, 0]) / mhz, float(np.percentile(s[:, 0], 90)) / mhz))
        print("  loop                    %7.0f cycles = %6.2f us = %6.0f cycles per image / phase" % (
            med(s[:, 1]), med(s[:, 1]) / mhz, med(s[:, 1] / nimg)))
        if s[:, 8].any():
            print("  before the loop (wave 0): LDS zero fill %.2f us, previous launch's slab sum %.2f us, wait for the first "
                  "tiles %.2f us; after the loop (slab store) %.2f us" % (
                      med(s[:, 8]) / mhz, med(s[:, 9]) / mhz, med(s[:, 10]) / mhz,
                      med(s[:, 0] - s[:, 1] - s[:, 8] - s[:, 9] - s[:, 10]) / mhz))
        print("  computing wave waits in barriers   %4.1f %% of the loop" % (100 * med(s[:, 2] / s[:, 1])))
        print("  data-moving wave waits in barriers %4.1f %% of its loop" % (100 * med(s[:, 3] / np.maximum(s[:, 4], 1))))


if __name__ == "__main__":
    main()
