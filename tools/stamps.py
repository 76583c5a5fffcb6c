#!/usr/bin/env python3
"""Phase stamps of the LDS-strip convolution (diagnostic build: make -C stylegan-for-facerec_amd/frhip/csrc stamps).

    FRHIP_LIB=stylegan-for-facerec_amd/frhip/lib/libfrhip_stamps.so python tools/stamps.py strip_64_64_112_fwd

Per workgroup (wave 0): 0 start, 1 strip loads issued + written, 2 barrier (strip resident), 3 main loop done (wave 0),
4 barrier (all waves done), 5 epilogue cells written + barrier, 6 stores issued; 7 = hardware id.  s_memrealtime runs
at 100 MHz (10 ns).  Prints the median / p90 of every phase and how many workgroups a CU runs at once.
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
    label = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 256
    fn = dict(kbench.suite_cases(B))[label]
    nblk = 1 << 16
    buf = torch.zeros(2 * nblk * 8, dtype=torch.int64, device="cuda")  # second plane: s_memtime (shader clock) of the strip kernel
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    roll = "--roll" in sys.argv
    setter = dbg.fr_debug_set_stamp_buffer_roll if roll else dbg.fr_debug_set_stamp_buffer
    setter.argtypes = [ctypes.c_void_p]
    fn(3)  # warm up without stamps
    assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    buf.zero_()
    fn(1)  # timeit runs 3 warm-up + 1 timed launch: the last one's stamps remain
    torch.cuda.synchronize()
    both = buf.cpu().numpy().reshape(2, nblk, 8)
    s, clk = both[0], both[1]
    if roll:  # conv3x3_roll64: per workgroup, phase times summed over its nit iterations
        s = s[s[:, 7] != 0]
        nit = s[:, 7].astype(np.float64)
        names = ["main", "cells", "barrier"]
        print("%s (rolling kernel): %d workgroups, %d iterations each" % (label, len(s), int(nit[0])))
        for k, n in enumerate(names):
            d = s[:, k] * 0.01 / nit
            print("  %-10s median %6.2f  p90 %6.2f us per iteration" % (n, np.median(d), np.percentile(d, 90)))
        life = s[:, 6] * 0.01
        print("  lifetime   median %7.1f us = %.2f us per iteration" % (np.median(life), np.median(life / nit)))
        return
    clk = clk[s[:, 0] != 0]
    s = s[s[:, 0] != 0]
    if not roll and clk[:, 2].any():  # the clock the K loop holds: shader cycles per 10-ns tick, wave 0 of every workgroup
        ghz = (clk[:, 3] - clk[:, 2]).astype(np.float64) / np.maximum(1, (s[:, 3] - s[:, 2]).astype(np.float64)) * 0.1
        print("  in-kernel clock over the K loop: median %.2f GHz  p10 %.2f  p90 %.2f" % (np.median(ghz), np.percentile(ghz, 10),
                                                                                      np.percentile(ghz, 90)))
    t = s[:, :7].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    names = ["load", "barrier1", "main", "barrier2", "cells", "store"]
    print("%s: %d workgroups, kernel span %.1f us" % (label, len(s), t[:, 6].max() - t0))
    for k, n in enumerate(names):
        d = t[:, k + 1] - t[:, k]
        print("  %-9s median %7.2f  p10 %7.2f  p90 %7.2f us" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    life = t[:, 6] - t[:, 0]
    print("  lifetime  median %7.2f  p90 %7.2f us" % (np.median(life), np.percentile(life, 90)))
    hw = s[:, 7]
    cu = (hw >> 32) * 1000 + ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 0x7) + 128 * ((hw >> 16) & 0x3)  # xcc, cu, sh, se
    ncu = len(np.unique(cu))
    # concurrency: for a sample of CUs, how many workgroups overlap in time
    starts, ends = t[:, 0], t[:, 6]
    occ = []
    for c in np.unique(cu)[:32]:
        m = cu == c
        ev = sorted([(a, 1) for a in starts[m]] + [(b, -1) for b in ends[m]])
        cur = best = 0
        tot, last, acc = 0.0, ev[0][0], 0.0
        for x, d in ev:
            acc += cur * (x - last)
            tot += (x - last) if cur > 0 else 0.0
            last = x
            cur += d
            best = max(best, cur)
        occ.append((best, acc / max(tot, 1e-9), m.sum()))
    print("  %d distinct CU ids; per CU: max resident %s, mean resident while busy %.2f, workgroups per CU %.1f"
          % (ncu, max(o[0] for o in occ), np.mean([o[1] for o in occ]), np.mean([o[2] for o in occ])))
    gap = []
    for c in np.unique(cu)[:32]:
        m = cu == c
        order = np.argsort(starts[m])
        st, en = starts[m][order], ends[m][order]
        if len(st) > 2:
            gap.append(np.median(st[2:] - en[:-2]))  # slot reuse gap with two resident: start of k+2 vs end of k
    if gap:
        print("  median (start of workgroup k+2) - (end of workgroup k) on a CU: %.2f us" % np.median(gap))


if __name__ == "__main__":
    main()
