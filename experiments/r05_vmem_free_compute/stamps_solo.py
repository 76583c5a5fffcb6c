#!/usr/bin/env python3
"""Phase stamps of conv3x3_solo (diagnostic library: conv3x3_solo.hip built with -DFRHIP_STAMPS).

    FRHIP_LIB=.../libfrhip_stamps.so python tools/stamps_solo.py strip_256_256_14_fwd_bn [B]

Wave 0 of every workgroup: 0 start, 1 image written to LDS, 2 barrier, 3 K loop done, 4 everything in flight landed,
6 epilogue cells + stores issued.  s_memrealtime runs at 100 MHz."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kbench  # noqa: E402
from frhip import _lib  # noqa: E402


def main():
    label = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    fn = dict(kbench.suite_cases(B))[label]
    nblk = 1 << 12
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_solo.argtypes = [ctypes.c_void_p]
    fn(3)
    assert dbg.fr_debug_set_stamp_buffer_solo(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    buf.zero_()
    fn(1)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(nblk, 8)
    s = s[s[:, 0] != 0]
    t = s.astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    print("%s: %d workgroups, kernel span %.1f us, starts spread %.1f us" % (label, len(s), t[:, 6].max() - t0, t[:, 0].max() - t0))
    for a, b, n in ((0, 1, "load+write"), (1, 2, "barrier"), (2, 3, "K loop"), (3, 4, "drain"), (4, 6, "cells+stores"), (0, 6, "lifetime")):
        d = t[:, b] - t[:, a]
        print("  %-12s median %7.2f  p10 %7.2f  p90 %7.2f us" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    cyc = (s[:, 7] - s[:, 5]).astype(np.float64)
    us = t[:, 4] - t[:, 2]
    print("  K loop: %.0f shader cycles median = %.2f GHz in-kernel clock; %.2f cycles per MFMA (3744 per wave)"
          % (np.median(cyc), np.median(cyc / us) / 1e3, np.median(cyc) / 3744.0))


if __name__ == "__main__":
    main()
