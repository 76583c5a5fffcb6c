"""Margin-softmax heads with the reference's import path, constructor signatures and state-dict keys.

    from head.metrics import ArcFace, CosFace, SphereFace, Am_softmax        (reference train.py:9)

``ArcFace`` / ``CosFace`` (the two heads the shipped configs and BASELINE.json name) run on the HIP kernels
(row normalise -> MFMA cosine GEMM with the margin / label-select / scale epilogue -> closed-form backward).
``SphereFace`` / ``Am_softmax`` are constructed eagerly by the reference driver (train.py:178-181) and are
therefore provided too, as small plain-PyTorch modules outside the accelerated path.

Differences from the reference that a caller can observe:
  * ``device_id`` is accepted for signature compatibility but the class-dimension ``.cuda(i)`` split of
    head/metrics.py:104-113 is not reproduced: one process drives one GPU and the whole weight lives there
    (data parallelism is one process per GPU over RCCL, see frhip/parallel.py).
  * the reference's ArcFace allocates its one-hot on ``'cuda'`` unconditionally (metrics.py:133) and takes an
    optional ``onehot_vec``; here the label select happens inside the GEMM epilogue (bit-exact equivalent of
    the blend for finite values), ``onehot_vec`` is accepted and ignored.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter

from frhip import functional as FRF


class _MarginHead(nn.Module):
    _kind = 0

    def __init__(self, in_features, out_features, device_id, s, m):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.device_id = device_id
        self.s = s
        self.m = m
        self.weight = Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)  # metrics.py:87-88 / :163-164

    def _logits(self, input, label, easy_margin=False):
        w = self.weight
        if w.device != input.device:
            # the reference keeps HEAD on the host and copies W every step (train.py never calls HEAD.to);
            # here the parameter is moved once, next to the features
            self.to(input.device)
            w = self.weight
        return FRF.margin_head(input, w, label.to(input.device), self._kind, self.s, self.m, easy_margin)

    def __repr__(self):
        return "%s(in_features = %d, out_features = %d, s = %s, m = %s)" % (
            self.__class__.__name__, self.in_features, self.out_features, self.s, self.m)


class ArcFace(_MarginHead):
    """cos(theta + m) margin -- reference head/metrics.py:66-140."""
    _kind = 0

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.50, easy_margin=False):
        super().__init__(in_features, out_features, device_id, s, m)
        self.easy_margin = easy_margin
        self.cos_m = math.cos(m)
        self.sin_m = math.sin(m)
        self.th = math.cos(math.pi - m)
        self.mm = math.sin(math.pi - m) * m
        self.eps = 1e-10

    def forward(self, input, label, onehot_vec=None):
        return self._logits(input, label, self.easy_margin)


class CosFace(_MarginHead):
    """cos(theta) - m margin, default m = 0.50 as in the reference (head/metrics.py:143-191, :155)."""
    _kind = 1

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.50):
        super().__init__(in_features, out_features, device_id, s, m)

    def forward(self, input, label):
        return self._logits(input, label)


class SphereFace(nn.Module):
    """cos(m*theta) head (reference head/metrics.py:200-277).  Not on the accelerated path; plain PyTorch."""

    def __init__(self, in_features, out_features, device_id, m=4):
        super().__init__()
        self.in_features, self.out_features, self.device_id, self.m = in_features, out_features, device_id, m
        self.base, self.gamma, self.power, self.LambdaMin, self.iter = 1000.0, 0.12, 1, 5.0, 0
        self.weight = Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)

    @staticmethod
    def _cheb(m, c):
        return {0: lambda x: x ** 0, 1: lambda x: x, 2: lambda x: 2 * x ** 2 - 1, 3: lambda x: 4 * x ** 3 - 3 * x,
                4: lambda x: 8 * x ** 4 - 8 * x ** 2 + 1, 5: lambda x: 16 * x ** 5 - 20 * x ** 3 + 5 * x}[m](c)

    def forward(self, input, label):
        self.iter += 1
        self.lamb = max(self.LambdaMin, self.base * (1 + self.gamma * self.iter) ** (-1 * self.power))
        w = self.weight.to(input.device)
        c = F.linear(F.normalize(input), F.normalize(w)).clamp(-1, 1)
        k = (self.m * c.detach().acos() / 3.14159265).floor()
        phi = ((-1.0) ** k) * self._cheb(self.m, c) - 2 * k
        hot = torch.zeros_like(c).scatter_(1, label.view(-1, 1), 1)
        out = hot * (phi - c) / (1 + self.lamb) + c
        return out * input.norm(2, 1).view(-1, 1)


class Am_softmax(nn.Module):
    """Additive-margin softmax with a [in, out] ``kernel`` (reference head/metrics.py:287-333). Plain PyTorch."""

    def __init__(self, in_features, out_features, device_id, m=0.35, s=30.0):
        super().__init__()
        self.in_features, self.out_features, self.device_id, self.m, self.s = in_features, out_features, device_id, m, s
        self.kernel = Parameter(torch.empty(in_features, out_features))
        self.kernel.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)

    def forward(self, embbedings, label):
        kn = self.kernel.to(embbedings.device)
        c = torch.mm(embbedings, kn / kn.norm(2, 0, True)).clamp(-1, 1)
        hot = torch.zeros_like(c).scatter_(1, label.view(-1, 1), 1).bool()
        return torch.where(hot, c - self.m, c) * self.s
