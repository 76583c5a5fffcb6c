#!/usr/bin/env python3
"""Phase stamps of the stride-2 strip kernel (diagnostic build: make -C stylegan-for-facerec_amd/frhip/csrc stamps).

    FRHIP_LIB=stylegan-for-facerec_amd/frhip/lib/libfrhip_stamps.so python tools/stamps_s2.py s2_128_28_fwd

Forward (no plane prefetch, 64 / 128 channels): 0 start, 1 plane 0 loads committed, 2 barrier, 3 taps of plane 0 (wave 0),
4 barrier, 5 plane 1 resident, 6 taps 1 + barrier, 7 plane 2 resident, 8 taps 2 + barrier, 9 plane 3 committed, 10 barrier,
11 taps 3 (wave 0), 12 barrier, 13 epilogue done.  Data gradient: 0 start, 1 g strip committed, 2 barrier, then per output
class: taps done (3, 5, 7, 9), epilogue done (4, 6, 8, 10).  s_memrealtime runs at 100 MHz."""
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


def main_ws(label, B):
    """conv3x3_s2_ws.hip: computing wave 0 and data-moving wave 4 of every workgroup (48 slots)."""
    fn = dict(kbench.suite_cases(B))[label]
    nblk = 1 << 14
    buf = torch.zeros(nblk * 48, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_ws.argtypes = [ctypes.c_void_p]
    fn(3)
    assert dbg.fr_debug_set_stamp_buffer_ws(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    buf.zero_()
    fn(1)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(nblk, 48)
    s = s[s[:, 0] != 0]
    t = s.astype(np.float64) * 0.01
    nc = int(np.max(np.nonzero(s[0, :24])[0]))
    nl = int(np.max(np.nonzero(s[0, 24:])[0]))
    print("%s (warp-specialised): %d workgroups, span %.1f us, lifetime median %.2f us" % (
        label, len(s), t[:, nc].max() - t[:, 0].min(), np.median(t[:, nc] - t[:, 0])))
    names = ["wait phase 0"] + ["taps %d" % k for k in range(nc - 4)] + ["cells", "barrier", "stores"]
    for k in range(nc):
        d = t[:, k + 1] - t[:, k]
        print("  computing   %-14s median %6.2f  p90 %6.2f us   (cumulative %6.2f)" % (names[k] if k < len(names) else k, np.median(d),
              np.percentile(d, 90), np.median(t[:, k + 1] - t[:, 0])))
    for k in range(nl):
        d = t[:, 24 + k + 1] - t[:, 24 + k]
        print("  data-moving stage %2d        median %6.2f  p90 %6.2f us   (done at %6.2f)" % (k, np.median(d), np.percentile(d, 90),
              np.median(t[:, 24 + k + 1] - t[:, 0])))


def main():
    label = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 256
    if "--ws" in sys.argv:
        return main_ws(label, B)
    fn = dict(kbench.suite_cases(B))[label]
    nblk = 1 << 15
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    dbg.fr_debug_set_stamp_buffer_s2.argtypes = [ctypes.c_void_p]
    fn(3)
    assert dbg.fr_debug_set_stamp_buffer_s2(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    buf.zero_()
    fn(1)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(nblk, 16)
    s = s[s[:, 0] != 0]
    last = int(np.max(np.nonzero(s[0])[0]))
    t = s[:, :last + 1].astype(np.float64) * 0.01
    print("%s: %d workgroups, kernel span %.1f us, lifetime median %.2f us" % (
        label, len(s), t[:, last].max() - t[:, 0].min(), np.median(t[:, last] - t[:, 0])))
    for k in range(last):
        d = t[:, k + 1] - t[:, k]
        print("  stamp %2d -> %2d  median %6.2f  p10 %6.2f  p90 %6.2f us" % (k, k + 1, np.median(d), np.percentile(d, 10),
                                                                          np.percentile(d, 90)))


if __name__ == "__main__":
    main()
