"""autograd.Function wrappers over the HIP kernels for the margin head, the focal loss and top-k accuracy.

These are what ``head/metrics.py``, ``loss/focal.py`` and ``util/utils.py`` call.  Inputs must be ROCm device
tensors; a host tensor raises (no CPU fallback -- the CPU restatement is ``oracle/``, test-only).
The head always computes in fp32 (FR_F32): it is <0.3 % of the step's FLOPs and the 1e-3 logits bar of
BASELINE.json is an fp32 bar (bf16 operands drift the logits by ~0.2).
"""
import math
import os

import torch

from . import ops
from ._lib import FR_F32
from .engine import _side_stream


def _head_side_stream(dev):
    """The device's weight-gradient stream (engine.py: one per device), or None: FRHIP_HEAD_SIDE=0 / FRHIP_SINGLE_STREAM=1."""
    if os.environ.get("FRHIP_HEAD_SIDE", "1") == "0" or os.environ.get("FRHIP_SINGLE_STREAM", "0") != "0":
        return None
    return _side_stream(dev, 1)


def _pad(n, m):
    return (n + m - 1) // m * m


def margin_forward(x, weight, label, kind, s, m, easy_margin):
    """logits = s * where(j == label, phi(cos), cos) for fp32 device tensors.  Returns (logits, saved, cfg) for
    ``margin_backward``.  A label outside [0, N) selects nothing in its row (the class-sharded head passes -1 for rows
    whose label lives on another rank)."""
    B, D = x.shape
    N = weight.shape[0]
    dev = x.device
    st = ops.current_stream_ptr()
    x = x.contiguous().float()
    w = weight.contiguous().float()
    label = label.contiguous().long()
    Np = _pad(N, 32)
    xn = torch.empty(B, D, device=dev)
    inv_x = torch.empty(B, device=dev)
    wn = torch.empty(Np, D, device=dev)
    wt = torch.empty(D, Np, device=dev)
    inv_w = torch.empty(N, device=dev)
    ops.call("fr_row_normalize", x, xn, None, inv_x, B, B, D, 0, FR_F32, st)()
    ops.call("fr_row_normalize", w, wn, wt, inv_w, N, Np, D, Np, FR_F32, st)()
    ld = _pad(N, 4)  # 16-byte row pitch for the GEMM's vector stores; [B, N] is a view when N is not a multiple of 4
    store = torch.empty(B, ld, device=dev)
    logits = store if ld == N else store[:, :N]
    cos_t = torch.zeros(B, device=dev)
    if kind == 0:
        cos_m, sin_m = math.cos(m), math.sin(m)
        th, mm = math.cos(math.pi - m), math.sin(math.pi - m) * m
    else:
        cos_m, sin_m, th, mm = m, 0.0, 0.0, 0.0
    ops.conv(st, FR_F32, src=xn, w=wn, out=logits, B=B, RH=1, RW=1, SH=1, SW=1, SC=D, N=N, KH=1, KW=1, stride=1,
             pad=0, mode=0, lda=D, ldc=ld, pro=0, epi=ops.EPI_MARGIN, out_f32=1, margin_kind=kind,
             easy_margin=int(bool(easy_margin)), cos_m=cos_m, sin_m=sin_m, th=th, mm=mm, scale=float(s),
             label=label, cos_t=cos_t)()
    saved = (x, w, label, xn, wt, inv_x, inv_w, cos_t)
    cfg = (kind, float(s), cos_m, sin_m, th, int(bool(easy_margin)), Np)
    return logits, saved, cfg


def margin_backward(saved, cfg, g, need_x, need_w, raw_x_grad=False):
    """(gx, gw) of ``margin_forward``.  ``raw_x_grad``: return G = d loss / d normalize(x) instead of gx (the
    class-sharded head sums G over ranks before it goes through the normalisation backward, which is linear in G)."""
    x, w, label, xn, wt, inv_x, inv_w, cos_t = saved
    kind, s, cos_m, sin_m, th, easy, Np = cfg
    B, D = x.shape
    N = w.shape[0]
    dev = x.device
    st = ops.current_stream_ptr()
    g = g.contiguous().float()
    gcos = torch.empty(B, Np, device=dev)
    ops.call("fr_margin_bwd", g, label, cos_t, gcos, B, N, Np, kind, easy, cos_m, sin_m, th, s, FR_F32, st)()
    gx = gw = None
    # Round 6: the two halves (three launches each, 55 us each at 7000 classes, neither fills the chip) run side by side: the
    # weight's on the weight-gradient stream, which is idle until the backbone's backward pass starts.  Every buffer is
    # allocated on the calling stream, which waits for the side stream before this function returns.
    side = _head_side_stream(dev) if (need_x and need_w) else None
    if side is not None:
        main = torch.cuda.current_stream(dev)
        N4 = _pad(N, 4)
        GW = torch.empty(N4, D, device=dev)
        gw = torch.empty(N, D, device=dev)
        side.wait_stream(main)
        sp = ops.stream_ptr(side)
        ops.call("fr_fill_rows", GW, None, N4, D, sp)()
        ops.wgrad(sp, FR_F32, g=gcos, src=xn, dw=GW, B=B, GH=1, GW=1, Cout=N4, SH=1, SW=1, SC=D, KH=1, KW=1,
                  stride=1, pad=0, ldg=Np, lda=D, pro=0, nsplit=1)()
        ops.call("fr_normalize_bwd", GW, w, inv_w, gw, N, D, sp)()
        need_w = False
    if need_x:
        Gx = torch.empty(B, D, device=dev)
        nk = Np // 32
        splitk = max(1, min(nk, 64, nk // 8))
        slab = torch.empty(splitk, B, D, device=dev)  # K slices to slabs, added in a fixed order (reproducible)
        ops.conv(st, FR_F32, src=gcos, w=wt, out=slab, B=B, RH=1, RW=1, SH=1, SW=1, SC=Np, N=D, KH=1, KW=1,
                 stride=1, pad=0, mode=0, lda=Np, ldc=D, pro=0, epi=ops.EPI_SLAB, out_f32=1, splitk=splitk)()
        ops.call("fr_reduce_parts", slab, splitk, 1, B * D, Gx, None, None, st)()
        if raw_x_grad:
            gx = Gx
        else:
            gx = torch.empty(B, D, device=dev)
            ops.call("fr_normalize_bwd", Gx, x, inv_x, gx, B, D, st)()
    if need_w:
        N4 = _pad(N, 4)  # the weight-gradient GEMM wants 16-byte channel counts; gcos columns >= N are zero
        GW = torch.zeros(N4, D, device=dev)
        ops.wgrad(st, FR_F32, g=gcos, src=xn, dw=GW, B=B, GH=1, GW=1, Cout=N4, SH=1, SW=1, SC=D, KH=1, KW=1,
                  stride=1, pad=0, ldg=Np, lda=D, pro=0, nsplit=1)()
        gw = torch.empty(N, D, device=dev)
        ops.call("fr_normalize_bwd", GW, w, inv_w, gw, N, D, st)()
    if side is not None:
        main.wait_stream(side)
    return gx, gw


class MarginHeadFn(torch.autograd.Function):
    """logits = s * where(j == label, phi(cos), cos),  cos = normalize(x) . normalize(W)^T

    head/metrics.py:97-140 (ArcFace: phi = cos(theta+m) with the cos>th fallback / easy margin) and
    :164-191 (CosFace: phi = cos - m).  Backward per SURVEY.md App. D.
    """

    @staticmethod
    def forward(ctx, x, weight, label, kind, s, m, easy_margin):
        logits, saved, cfg = margin_forward(x, weight, label, kind, s, m, easy_margin)
        ctx.save_for_backward(*saved)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(label)
        return logits

    @staticmethod
    def backward(ctx, g):
        gx, gw = margin_backward(ctx.saved_tensors, ctx.cfg, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None, None, None, None


CHECK_LABELS = True  # host-side range check of the labels (one device sync per call); loops with validated data clear it


def margin_head(x, weight, label, kind, s, m, easy_margin=False):
    if x.shape[0] == 0:  # the reference returns empty logits (F.linear / scatter_ on zero rows); nothing to launch
        ops.ptr(x)  # host tensors still fail loudly
        return x.new_zeros((0, weight.shape[0]), dtype=torch.float32) + 0.0 * (x.sum() + weight.sum())
    if CHECK_LABELS and (label.min() < 0 or label.max() >= weight.shape[0]):  # reference: RuntimeError from scatter_ (metrics.py:134)
        raise RuntimeError("index %d is out of bounds for dimension 1 with size %d"
                           % (int(label.max()), weight.shape[0]))
    return MarginHeadFn.apply(x, weight, label, kind, s, m, easy_margin)


class FocalLossFn(torch.autograd.Function):
    """loss = (1 - exp(-l))^gamma * l,  l = mean_i CE(logits_i, y_i)   -- loss/focal.py:17-21."""

    @staticmethod
    def forward(ctx, logits, target, gamma):
        B, N = logits.shape
        dev = logits.device
        st = ops.current_stream_ptr()
        logits = logits.contiguous().float()
        target = target.contiguous().long()
        lse = torch.empty(B, device=dev)
        ce = torch.empty(B, device=dev)
        rank = torch.empty(B, device=dev, dtype=torch.int32)
        scalars = torch.empty(8, device=dev)
        ops.call("fr_ce_rows", logits, target, lse, ce, rank, B, N, N, st)()
        ops.call("fr_focal_finalize", ce, rank, B, float(gamma), scalars, st)()
        ctx.save_for_backward(logits, target, lse, scalars)
        return scalars[0].clone()

    @staticmethod
    def backward(ctx, gup):
        logits, target, lse, scalars = ctx.saved_tensors
        B, N = logits.shape
        st = ops.current_stream_ptr()
        grad = torch.empty_like(logits)
        gup = gup.contiguous().float().reshape(1)
        ops.call("fr_focal_bwd", logits, target, lse, scalars, gup, grad, B, N, N, st)()
        return grad, None, None


def focal_loss(logits, target, gamma=2.0):
    if logits.shape[0] == 0:  # mean cross entropy of no rows is NaN in the reference (loss/focal.py:18)
        ops.ptr(logits)
        return logits.sum() * float("nan")
    return FocalLossFn.apply(logits, target, gamma)


def topk_precision(rank, topk):
    """[precision@k in percent for k in topk] from the label ranks (device float32 [len(topk)]); the arithmetic of
    ``(rank < k).float().sum().mul_(100.0 / n)`` (reference util/utils.py:343-358) in one launch."""
    ks = [int(k) for k in topk] + [0] * (4 - len(topk))
    out = torch.empty(len(topk), device=rank.device, dtype=torch.float32)
    ops.call("fr_topk_precision", rank, rank.numel(), len(topk), ks[0], ks[1], ks[2], ks[3], 100.0 / rank.numel(), out,
             ops.current_stream_ptr())()
    return out


def topk_ranks(logits, target):
    """rank[m] = number of classes scoring strictly above the label's logit (device int32 [B])."""
    B, N = logits.shape
    if B == 0:
        ops.ptr(logits)
        return torch.empty(0, device=logits.device, dtype=torch.int32)
    st = ops.current_stream_ptr()
    logits = logits.contiguous().float()
    rank = torch.empty(B, device=logits.device, dtype=torch.int32)
    ops.call("fr_rank_rows", logits, target.contiguous().long(), rank, B, N, N, st)()
    return rank
