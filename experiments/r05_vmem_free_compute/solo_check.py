#!/usr/bin/env python3
"""conv3x3_solo (one wave per SIMD) against conv3x3_strip (8 waves) at 256 -> 256 @14: bit identity of outputs, partial rows
and the stored residual stream for every prologue / epilogue pair the engine launches, then timings (warm and cold)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch  # noqa: E402

import kbench  # noqa: E402
from frhip import _lib, ops  # noqa: E402

BF = torch.bfloat16
OPT = os.environ.get("CHECK_OPT", "FRHIP_SOLO")  # the switch under test (its 0 setting = the 8-wave strip kernel)


def setopt(name, v):
    return _lib.lib.fr_set_option(name.encode(), v)


def run(B, pro, epi, mode, solo):
    torch.manual_seed(7)
    st = ops.current_stream_ptr()
    C = 256
    src = kbench.rnd(B, 14, 14, C)
    src2 = kbench.rnd(B, 14, 14, C)
    w = kbench.rnd(C, 9, C) * 0.05
    aux = kbench.rnd(B, 14, 14, C)
    out = torch.zeros(B, 14, 14, C, device="cuda", dtype=BF)
    pro_out = torch.zeros(B, 14, 14, C, device="cuda", dtype=BF)
    nv = 3 if epi == 8 else 2
    part = torch.zeros(B * nv * C + 64, device="cuda")
    va, vb = torch.rand(512, device="cuda") + 0.5, torch.rand(512, device="cuda") - 0.5
    vc, vd = torch.rand(512, device="cuda") + 0.5, torch.rand(512, device="cuda") - 0.5
    vg = torch.rand(B, C, device="cuda")
    kw = dict(src=src, w=w, out=out, B=B, RH=14, RW=14, SH=14, SW=14, SC=C, N=C, KH=3, KW=3, stride=1, pad=1, mode=mode,
              lda=C, ldc=C, ldaux=C, pro=pro, pro_a=va, pro_b=vb, epi=epi, epi_a=va, epi_b=vb, aux=aux, part=part)
    if pro in (4, 5):
        kw.update(src2=src2, pro_c=vc, pro_d=vd, pro_out=pro_out)
    if pro == 5:
        kw.update(pro_g=vg)
    for o in ("FRHIP_SOLO", "FRHIP_RELAY"):
        setopt(o, 0)
    setopt(OPT, solo)
    ops.conv_strip(st, **kw)()
    torch.cuda.synchronize()
    return out, part, pro_out


def main():
    bad = 0
    combos = [(1, 0, 0), (2, 1, 0), (2, 8, 0), (4, 0, 0), (5, 0, 0), (0, 2, 1), (0, 3, 1), (1, 7, 0), (0, 0, 0), (1, 1, 0),
              (4, 1, 0), (0, 8, 0)]
    for B in (162, 256):
        for pro, epi, mode in combos:
            a = run(B, pro, epi, mode, 0)
            b = run(B, pro, epi, mode, 1)
            ok = all(torch.equal(x, y) for x, y in zip(a, b))
            nz = float(a[0].float().abs().mean())
            print("B=%d pro=%d epi=%d mode=%d: %s (mean|out| %.3f)" % (B, pro, epi, mode, "bit-identical" if ok else "DIFFERENT", nz), flush=True)
            if not ok:
                bad += 1
                for x, y, n in zip(a, b, ("out", "part", "pro_out")):
                    d = (x.float() - y.float()).abs()
                    print("   %s: max diff %.4g, mismatches %d of %d" % (n, float(d.max()), int((d > 0).sum()), d.numel()))
    cases = "strip_256_256_14_fwd_bn,strip_256_256_14_fwd_prelu,strip_256_256_14_fwd_resbn,strip_256_256_14_dgrad"
    for cold in (0, 1):
        kbench.COLD = bool(cold)
        for rep in range(2):
            for solo in (0, 1):
                for o in ("FRHIP_SOLO", "FRHIP_RELAY"):
                    setopt(o, 0)
                setopt(OPT, solo)
                for label, fn in kbench.suite_cases(256):
                    if label in cases.split(","):
                        ms, tf = fn(20 if cold else 50)
                        print("cold=%d solo=%d %-28s %.4f ms %7.1f TFLOP/s" % (cold, solo, label, ms, tf), flush=True)
    print("SOLO_CHECK", "FAIL" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
