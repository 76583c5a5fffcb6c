#!/usr/bin/env python3
"""fr_conv3x3_pair against the two fr_conv3x3_strip launches it replaces (B = 256, 256 channels, 14x14): HIP-event times."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd")); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
from frhip import ops
import kbench
B, C, W = 256, 256, 14
st = ops.current_stream_ptr()
x = kbench.rnd(B, W, W, C); w1 = kbench.rnd(C, 9, C) * 0.05; w2 = kbench.rnd(C, 9, C) * 0.05
y1 = torch.empty_like(x); y2 = torch.empty_like(x)
va, vb = torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") - 0.5
part = torch.zeros(4 * 1024 * 1024, device="cuda")
common = dict(B=B, RH=W, RW=W, SH=W, SW=W, SC=C, N=C, KH=3, KW=3, stride=1, pad=1, mode=0, lda=C, ldc=C)
pair = ops.conv_strip_pair(st, src=x, w=w1, out=y1, w2=w2, out2=y2, slope2=va, pro=1, pro_a=va, pro_b=vb, epi=1, part=part, **common)
a = ops.conv_strip(st, src=x, w=w1, out=y1, pro=1, pro_a=va, pro_b=vb, epi=0, **common)
b = ops.conv_strip(st, src=y1, w=w2, out=y2, pro=2, pro_a=va, epi=1, part=part, **common)
fl = 2 * 2.0 * B * W * W * C * C * 9
for name, fn in (("two launches", lambda: (a(), b())), ("pair", pair), ("two launches", lambda: (a(), b())), ("pair", pair)):
    ms, tf = kbench.timeit(fn, 50, fl)
    print("%-14s %.4f ms  %.1f TFLOP/s" % (name, ms, tf))
