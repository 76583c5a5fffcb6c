#!/usr/bin/env python3
"""Do an MFMA-bound strip kernel and an HBM-bound elementwise kernel overlap when issued on two HIP streams?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch
from frhip import ops
BF = torch.bfloat16
B, W, C = 256, 14, 256
def rnd(*s): return (torch.rand(*s, device="cuda") * 2 - 1).to(BF)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g, x = rnd(B, W, W, C), rnd(B, W, W, C)
dw = torch.zeros(C, 9, C, device="cuda")
va = torch.rand(512, device="cuda")
slab = torch.empty(16 * C * 9 * C, device="cuda")
src = rnd(B, W, W, C); wgt = rnd(C, 9, C) * 0.05; out = torch.empty(B, W, W, C, device="cuda", dtype=BF)
part = torch.zeros(1 << 20, device="cuda")
a, b_, c = rnd(64 << 20), rnd(64 << 20), torch.empty(64 << 20, device="cuda", dtype=BF)   # 128 MB each
def mk(stream):
    st = __import__("ctypes").c_void_p(stream.cuda_stream)
    wg = ops.wgrad_strip(st, g=g, src=x, dw=dw, B=B, GH=W, GW=W, Cout=C, SH=W, SW=W, SC=C, KH=3, KW=3, stride=1, pad=1,
                         ldg=C, lda=C, pro=1, pro_a=va, pro_b=va, nsplit=16, slab=slab)
    cv = ops.conv_strip(st, src=src, w=wgt, out=out, B=B, RH=W, RW=W, SH=W, SW=W, SC=C, N=C, KH=3, KW=3, stride=1, pad=1,
                        mode=0, lda=C, ldc=C, pro=1, pro_a=va, pro_b=va, epi=1, part=part)
    return wg, cv
wg1, cv1 = mk(s1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def ew():
    with torch.cuda.stream(s2): torch.add(a, b_, out=c)
def k_wg():
    for _ in range(4): wg1()
def k_cv():
    for _ in range(4): cv1()
def both(k):
    def f():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        k(); ew()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return f
def only(k):
    def f():
        s1.wait_stream(torch.cuda.current_stream()); k(); torch.cuda.current_stream().wait_stream(s1)
    return f
def only_ew():
    s2.wait_stream(torch.cuda.current_stream()); ew(); torch.cuda.current_stream().wait_stream(s2)
print("elementwise add 3x128MB alone: %.3f ms" % t(only_ew))
for name, k in (("4x wgrad_strip", k_wg), ("4x conv_strip", k_cv)):
    ta, tb = t(only(k)), t(both(k))
    print("%s alone %.3f ms | with elementwise on 2nd stream %.3f ms" % (name, ta, tb))
