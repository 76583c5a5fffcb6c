#!/usr/bin/env python3
"""Phase stamps of conv3x3_relay (diagnostic library: conv3x3_relay.hip built with -DFRHIP_STAMPS).

    FRHIP_LIB=.../libfrhip_stamps.so python tools/stamps_relay.py strip_256_256_14_fwd_bn [B]

Wave 0 (a computing wave) of every workgroup: 0 start, 1 image written to LDS, 2 barrier, 3 K loop of pass 0 done,
4 epilogue of pass 0 done, 5 K loop of pass 1 done, 6 end; shader-clock stamps 8 (loop start), 11 (pass-0 loop end)."""
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
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_relay.argtypes = [ctypes.c_void_p]
    fn(3)
    assert dbg.fr_debug_set_stamp_buffer_relay(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    buf.zero_()
    fn(1)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(nblk, 16)
    s = s[s[:, 0] != 0]
    t = s.astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    print("%s: %d workgroups, kernel span %.1f us, starts spread %.1f us" % (label, len(s), t[:, 6].max() - t0, t[:, 0].max() - t0))
    for a, b, n in ((0, 1, "load+write"), (1, 2, "barrier"), (2, 3, "K loop 0"), (3, 4, "epilogue 0"), (4, 5, "K loop 1"),
                    (5, 6, "epilogue 1"), (0, 6, "lifetime")):
        d = t[:, b] - t[:, a]
        print("  %-12s median %7.2f  p10 %7.2f  p90 %7.2f us" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    cyc = (s[:, 11] - s[:, 8]).astype(np.float64)
    us = t[:, 3] - t[:, 2]
    print("  K loop 0: %.0f shader cycles median = %.2f GHz in-kernel clock; %.2f cycles per MFMA (1872 per wave and pass)"
          % (np.median(cyc), np.median(cyc / us) / 1e3, np.median(cyc) / 1872.0))


if __name__ == "__main__":
    main()
