#!/usr/bin/env python3
"""Stand-alone timing of single HIP kernels at training shapes (for A/B tuning and rocprofv3 --pmc runs).

    python tools/kbench.py strip 256 256 14          # conv3x3_strip fwd, B=256
    python tools/kbench.py wgs 256 256 14            # conv_wgrad_strip
    python tools/kbench.py igemm 256 256 14 --stride 1
    python tools/kbench.py all                       # the IR-50 layer table, strip vs igemm

Prints TFLOP/s from HIP events over --iters launches (random bf16 data, never zeros: DVFS).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch  # noqa: E402

from frhip import _lib, ops  # noqa: E402

BF = torch.bfloat16


def timeit(launch, iters, flops):
    if COLD:
        return timeit_cold(launch, iters, flops)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, flops / ms / 1e9


_FLUSH = None
TOUCH = []  # operands of the case being timed (KBENCH_TOUCH)
COLD = os.environ.get("KBENCH_COLD", "0") == "1"


def timeit_cold(launch, iters, flops):
    """Every launch behind a 1-GiB write that evicts L2 and the 256-MB memory-side cache: what a kernel costs INSIDE the
    training step, where its operands were written several launches ago (kbench's back-to-back loop re-reads them from the
    memory-side cache: the 14x14 weight gradient runs 47-50 us that way and ~70 in the step)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 * 1024 * 1024, device="cuda")
    launch()
    torch.cuda.synchronize()
    tot = 0.0
    mb = int(os.environ.get("KBENCH_FLUSH_MB", "1024"))  # 48: evicts the eight 4-MB L2s only, not the memory-side cache
    touch = os.environ.get("KBENCH_TOUCH", "0") == "1"  # after the full flush READ the operands again, then evict L2 only
    for _ in range(iters):
        _FLUSH[:mb * 262144].fill_(1.0)
        if touch:
            for t in TOUCH:
                t.view(torch.int16).bitwise_and(1).sum() if t.dtype == BF else t.sum()
            _FLUSH[:48 * 262144].fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / iters
    return ms, flops / ms / 1e9


def rnd(*shape):
    return (torch.rand(*shape, device="cuda") * 2 - 1).to(BF)


def conv_case(kind, cin, cout, W, B, stride=1, pro=1, epi=1, mode=0, iters=20):
    st = ops.current_stream_ptr()
    Ho = W // stride
    if mode == 0:
        src, RH, SH, SC, N = rnd(B, W, W, cin), Ho, W, cin, cout
        w = rnd(cout, 9, cin) * 0.05
    else:  # data gradient: src = g [B,Ho,Ho,cout] -> out [B,W,W,cin]
        src, RH, SH, SC, N = rnd(B, Ho, Ho, cout), W, Ho, cout, cin
        w = rnd(cin, 9, cout) * 0.05
    out = torch.empty(B, RH, RH, N, device="cuda", dtype=BF)
    aux = rnd(B, RH, RH, N)
    part = torch.zeros(4 * 1024 * 1024, device="cuda")
    va, vb = torch.rand(512, device="cuda") + 0.5, torch.rand(512, device="cuda") - 0.5
    kw = dict(src=src, w=w, out=out, B=B, RH=RH, RW=RH, SH=SH, SW=SH, SC=SC, N=N, KH=3, KW=3, stride=stride, pad=1,
              mode=mode, lda=SC, ldc=N, ldaux=N, pro=pro, pro_a=va, pro_b=vb, epi=epi, epi_a=va, epi_b=vb, aux=aux,
              part=part)
    TOUCH[:] = [src, aux]
    flops = 2.0 * B * Ho * Ho * N * 9 * SC  # stride 2: both directions do 9 taps per LOW-res pixel
    if pro == 4:  # FR_PRO_RESBN: the residual sum formed by its consumer (two sources, two coefficient pairs, stored once)
        kw.update(src2=rnd(*src.shape), pro_c=va, pro_d=vb, pro_out=torch.empty_like(src))
    # the engine packs the weights of these kernels in MFMA-fragment order (FrConvArgs.w_frag; random data: same bytes).
    # KBENCH_FRAG=0: the plain layout
    frag = os.environ.get("KBENCH_FRAG", "1") != "0"
    if kind == "s2":  # stride-2 parity-plane kernel: mode 0 forward (W = input side), mode 2 all-class data gradient
        if mode == 2:
            kw.update(par_h=-1, par_w=-1)
        if frag and _lib.lib.fr_conv3x3_s2_strip_takes_frag(B, SC, Ho, mode):
            kw.update(w_frag=1)
        l = ops.conv_s2_strip(st, **kw)
    elif kind == "strip":
        if frag and _lib.lib.fr_conv3x3_strip_takes_frag(B, SC, N, W):
            kw.update(w_frag=1)
        l = ops.conv_strip(st, **kw)
    else:
        l = ops.conv(st, ops.FR_BF16, **kw)
    return timeit(l, iters, flops)


def wgrad_case(kind, cout, cin, W, B, stride=1, pro=1, iters=20):
    st = ops.current_stream_ptr()
    Ho = W // stride
    g, x = rnd(B, Ho, Ho, cout), rnd(B, W, W, cin)
    dw = torch.zeros(cout, 9, cin, device="cuda")
    va, vb = torch.rand(512, device="cuda") + 0.5, torch.rand(512, device="cuda") - 0.5
    kw = dict(g=g, src=x, dw=dw, B=B, GH=Ho, GW=Ho, Cout=cout, SH=W, SW=W, SC=cin, KH=3, KW=3, stride=stride, pad=1,
              ldg=cout, lda=cin, pro=pro, pro_a=va, pro_b=vb)
    TOUCH[:] = [g, x]
    flops = 2.0 * B * Ho * Ho * cout * cin * 9
    if kind == "wgs":
        tiles = (cout // 64) * (cin // 64)
        if stride == 2:  # the engine's group count (engine.py _wgrad)
            rows, nimg = {56: (2, 1), 28: (4, 1), 14: (7, 1), 7: (7, 2)}[Ho]
            fills = (B * (Ho // rows) + nimg - 1) // nimg
        else:
            rows = {112: 2, 56: 4, 28: 7, 14: 14, 7: 7}[W]
            fills = B * (W // rows) // (4 if W == 7 else 1)
        groups = max(1, min(fills, 256 // tiles))
        slab = torch.empty(groups * cout * 9 * cin, device="cuda")
        l = ops.wgrad_strip(st, nsplit=groups, slab=slab, **kw)
    else:
        tiles = ((cout + 127) // 128) * ((cin + 127) // 128) * 9
        l = ops.wgrad(st, ops.FR_BF16, nsplit=max(1, min(256, 768 // tiles)), **kw)
    return timeit(l, iters, flops)


# The launches of one IR-50 training step that the round-2 profile work looks at (B = 256): (label, callable args).
# conv1 of a unit: forward pro=BN epi=STORE, data gradient mode 1 epi=BNBWD; conv2: forward pro=PRELU epi=STATS,
# data gradient epi=PRELU_BWD (engine.py _build_forward / _build_backward).
def suite_cases(B):
    return [
        ("strip_64_64_112_fwd", lambda it: conv_case("strip", 64, 64, 112, B, pro=1, epi=0, iters=it)),
        ("strip_64_64_112_dgrad", lambda it: conv_case("strip", 64, 64, 112, B, pro=0, epi=3, mode=1, iters=it)),
        ("s2_64_56_fwd", lambda it: conv_case("s2", 64, 64, 112, B, stride=2, pro=2, epi=1, iters=it)),
        ("s2_64_56_dgrad", lambda it: conv_case("s2", 64, 64, 112, B, stride=2, pro=0, epi=2, mode=2, iters=it)),
        ("s2_128_28_fwd", lambda it: conv_case("s2", 128, 128, 56, B, stride=2, pro=2, epi=1, iters=it)),
        ("s2_128_28_dgrad", lambda it: conv_case("s2", 128, 128, 56, B, stride=2, pro=0, epi=2, mode=2, iters=it)),
        ("s2_256_14_fwd", lambda it: conv_case("s2", 256, 256, 28, B, stride=2, pro=2, epi=1, iters=it)),
        ("s2_256_14_dgrad", lambda it: conv_case("s2", 256, 256, 28, B, stride=2, pro=0, epi=2, mode=2, iters=it)),
        ("s2_512_7_fwd", lambda it: conv_case("s2", 512, 512, 14, B, stride=2, pro=2, epi=1, iters=it)),
        ("s2_512_7_dgrad", lambda it: conv_case("s2", 512, 512, 14, B, stride=2, pro=0, epi=2, mode=2, iters=it)),
        ("strip_64_64_56_fwd_bn", lambda it: conv_case("strip", 64, 64, 56, B, pro=1, epi=0, iters=it)),
        ("strip_64_64_56_fwd_prelu", lambda it: conv_case("strip", 64, 64, 56, B, pro=2, epi=1, iters=it)),
        ("strip_64_64_56_dgrad", lambda it: conv_case("strip", 64, 64, 56, B, pro=0, epi=2, mode=1, iters=it)),
        ("wgs_64_64_112", lambda it: wgrad_case("wgs", 64, 64, 112, B, pro=1, iters=it)),
        ("wgs_64_64_56", lambda it: wgrad_case("wgs", 64, 64, 56, B, pro=2, iters=it)),
        ("strip_128_128_28_fwd", lambda it: conv_case("strip", 128, 128, 28, B, pro=1, epi=0, iters=it)),
        ("strip_256_256_14_fwd_bn", lambda it: conv_case("strip", 256, 256, 14, B, pro=1, epi=0, iters=it)),
        ("strip_256_256_14_fwd_prelu", lambda it: conv_case("strip", 256, 256, 14, B, pro=2, epi=1, iters=it)),
        ("strip_256_256_14_fwd_resbn", lambda it: conv_case("strip", 256, 256, 14, B, pro=4, epi=0, iters=it)),
        # (FR_EPI_STATS_X -- `strip 256 256 14 --pro 2 --epi 8` -- shares its kernel name with the case above: not in the suite,
        # whose PMC rows are matched by name)
        ("strip_256_256_14_dgrad", lambda it: conv_case("strip", 256, 256, 14, B, pro=0, epi=2, mode=1, iters=it)),
        ("strip_128_128_28_dgrad", lambda it: conv_case("strip", 128, 128, 28, B, pro=0, epi=2, mode=1, iters=it)),
        ("wgs_256_256_14", lambda it: wgrad_case("wgs", 256, 256, 14, B, pro=2, iters=it)),
        ("wgs_256_256_14_bn", lambda it: wgrad_case("wgs", 256, 256, 14, B, pro=1, iters=it)),
        ("wgs_512_512_7", lambda it: wgrad_case("wgs", 512, 512, 7, B, pro=2, iters=it)),
        ("wgs_128_128_28", lambda it: wgrad_case("wgs", 128, 128, 28, B, pro=2, iters=it)),
        ("wgs_s2_64_56", lambda it: wgrad_case("wgs", 64, 64, 112, B, stride=2, pro=2, iters=it)),
        ("wgs_s2_128_28", lambda it: wgrad_case("wgs", 128, 128, 56, B, stride=2, pro=2, iters=it)),
        ("wgs_s2_256_14", lambda it: wgrad_case("wgs", 256, 256, 28, B, stride=2, pro=2, iters=it)),
        ("wgs_s2_512_7", lambda it: wgrad_case("wgs", 512, 512, 14, B, stride=2, pro=2, iters=it)),
        ("strip_512_512_7_fwd", lambda it: conv_case("strip", 512, 512, 7, B, pro=1, epi=0, iters=it)),
    ]


LAYERS = [(64, 64, 112), (64, 64, 56), (64, 128, 56), (128, 128, 28), (128, 256, 28), (256, 256, 14), (256, 512, 14),
          (512, 512, 7)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind")
    ap.add_argument("dims", nargs="*", type=int)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--pro", type=int, default=1)
    ap.add_argument("--epi", type=int, default=1)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="suite: comma-separated case labels")
    a = ap.parse_args()
    if a.kind == "suite":  # one process, many kernels: what tools/pmc_round.sh profiles
        import json
        only = set(x for x in a.only.split(",") if x)
        res = {}
        for label, fn in suite_cases(a.batch):
            if only and label not in only:
                continue
            ms, tf = fn(a.iters)
            res[label] = {"ms": round(ms, 4), "tflops": round(tf, 1)}
            print("%-28s %.4f ms  %7.1f TFLOP/s" % (label, ms, tf), flush=True)
        print("KBENCH_SUITE " + json.dumps(res))
        return
    if a.kind == "all":
        for cin, cout, W in LAYERS:
            r = ["%3d->%3d @%3d" % (cin, cout, W)]
            for kind in ("strip", "igemm"):
                ms, tf = conv_case(kind, cin, cout, W, a.batch, iters=a.iters, epi=0 if cout == 512 and cin == 256 else 1)
                r.append("%s fwd %.3f ms %6.0f TF/s" % (kind, ms, tf))
            ms, tf = conv_case("strip", cin, cout, W, a.batch, pro=0, epi=3, mode=1, iters=a.iters)
            r.append("strip dgrad %.3f ms %6.0f TF/s" % (ms, tf))
            for kind in ("wgs", "wgrad"):
                ms, tf = wgrad_case(kind, cout, cin, W, a.batch, iters=a.iters)
                r.append("%s %.3f ms %6.0f TF/s" % (kind, ms, tf))
            print(" | ".join(r), flush=True)
        return
    if a.kind in ("strip", "igemm", "s2"):
        cin, cout, W = a.dims
        ms, tf = conv_case(a.kind, cin, cout, W, a.batch, a.stride, a.pro, a.epi, a.mode, a.iters)
    else:
        cout, cin, W = a.dims
        ms, tf = wgrad_case(a.kind, cout, cin, W, a.batch, a.stride, a.pro, a.iters)
    print("%s %s B=%d: %.4f ms  %.1f TFLOP/s" % (a.kind, a.dims, a.batch, ms, tf))


if __name__ == "__main__":
    main()
