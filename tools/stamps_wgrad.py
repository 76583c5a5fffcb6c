#!/usr/bin/env python3
"""Cycle stamps of the warp-specialised weight-gradient kernel (diagnostic build: make -C .../csrc stamps).

    FRHIP_LIB=stylegan-for-facerec_amd/frhip/lib/libfrhip_stamps.so python tools/stamps_wgrad.py 256 256 14 [pro]

Per workgroup: cycles of the whole kernel / of the image loop (wave 0), cycles wave 0 (computing) and wave 4 (data-moving)
wait inside the hand-over barriers, and the wall time (s_memrealtime, 100 MHz) -> the clock the kernel ran at.
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kbench  # noqa: E402
from frhip import _lib  # noqa: E402


def main():
    cout, cin, W = (int(v) for v in sys.argv[1:4])
    pro = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    B = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    buf = torch.zeros(4 * 4096 * 16, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_wgr.argtypes = [ctypes.c_void_p]
    for _ in range(3):  # ~2 s of back-to-back launches first: the clock the chip settles at under this load
        kbench.wgrad_case("wgs", cout, cin, W, B, pro=pro, iters=200)
    assert dbg.fr_debug_set_stamp_buffer_wgr(ctypes.c_void_p(buf.data_ptr())) == 0
    ms, tf = kbench.wgrad_case("wgs", cout, cin, W, B, pro=pro, iters=20)
    torch.cuda.synchronize()
    region = {14: 0, 28: 1, 56: 2, 112: 3}[W]
    s = buf.cpu().numpy().reshape(4, 4096, 16)[region]
    s = s[s[:, 6] != 0].astype(np.float64)
    nimg = s[:, 6]
    mhz = np.median(s[:, 0] / (s[:, 5] * 10e-9)) / 1e6
    print("wgrad %dx%d @%d pro %d B=%d: %.4f ms per launch pair (kernel + slab sum), %.0f TFLOP/s; %d workgroups, %d images "
          "each; in-kernel clock %.0f MHz" % (cout, cin, W, pro, B, ms, tf, len(s), int(nimg[0]), mhz))
    med = lambda v: float(np.median(v))  # noqa: E731
    print("  kernel            %8.0f cycles = %6.2f us" % (med(s[:, 0]), med(s[:, 0]) / mhz))
    print("  image loop        %8.0f cycles = %6.2f us = %6.0f cycles per image (MFMA floor 4032 at 14x14 x 64x64)"
          % (med(s[:, 1]), med(s[:, 1]) / mhz, med(s[:, 1] / nimg)))
    print("  computing wave waits in barriers   %8.0f cycles = %4.1f %% of the loop" % (med(s[:, 2]), 100 * med(s[:, 2] / s[:, 1])))
    print("  data-moving wave waits in barriers %8.0f cycles = %4.1f %% of its loop (%0.f cycles)"
          % (med(s[:, 3]), 100 * med(s[:, 3] / np.maximum(s[:, 4], 1)), med(s[:, 4])))


if __name__ == "__main__":
    main()
