"""Test stand-in for ``frhip.sharded_head.HipKernels`` built on the oracle (torch CPU): lets the world_size-2 ``gloo``
test run the sharded head's collective choreography without a GPU.  Same method signatures, same meaning of every
return value; test infrastructure only."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import irse_ref as O  # noqa: E402


class OracleKernels(object):
    def logits(self, x_all, w, label_local, kind, s, m, easy_margin):
        xl = x_all.clone().requires_grad_(True)
        wl = w.clone().requires_grad_(True)
        own = (label_local >= 0) & (label_local < w.shape[0])
        lab = label_local.clamp(min=0)
        with torch.enable_grad():  # called from inside autograd.Function.forward, where grad mode is off
            if kind == 0:
                sel = O.arcface_forward(xl, wl, lab, s=s, m=m, easy_margin=easy_margin)
            else:
                sel = O.cosface_forward(xl, wl, lab, s=s, m=m)
            plain = O.cosine_logits(xl, wl) * s
            logits = torch.where(own[:, None], sel, plain)
        # saved[5] must be a per-row tensor (the HIP path keeps 1/|x| there); the stand-in's G is already d loss / d x
        saved = (xl, wl, label_local, None, None, torch.ones(x_all.shape[0]), None, logits)
        return logits.detach(), saved, None

    def row_stats(self, logits, label_local):
        own = (label_local >= 0) & (label_local < logits.shape[1])
        mx = logits.max(1).values
        se = torch.exp(logits - mx[:, None]).sum(1)
        zl = torch.where(own, logits.gather(1, label_local.clamp(min=0)[:, None])[:, 0], torch.zeros_like(mx))
        return torch.stack([mx, se, zl])

    def combine(self, stats_all, world, rows):
        st = stats_all.view(world, 3, rows)
        gmax = st[:, 0].max(0).values
        tot = (st[:, 1] * torch.exp(st[:, 0] - gmax[None])).sum(0)
        t = st[:, 2].sum(0)
        lse = gmax + torch.log(tot)
        return lse, lse - t, t

    def shard_rank(self, logits, tlogit):
        return (logits > tlogit[:, None]).sum(1).to(torch.int32)

    def focal(self, ce, rank, gamma):
        l = ce.double().mean().float()
        p = torch.exp(-l)
        w = (1 - p) ** gamma
        sc = torch.zeros(8)
        sc[0] = w * l
        sc[1] = gamma * (1 - p) ** (gamma - 1) * p * l + w
        sc[2] = 100.0 * (rank < 1).sum() / ce.shape[0]
        sc[3] = 100.0 * (rank < 5).sum() / ce.shape[0]
        sc[4] = l
        return sc

    def dlogits(self, logits, label_local, lse, scalars, gup):
        k = gup.reshape(()) * scalars[1] / logits.shape[0]
        hot = torch.arange(logits.shape[1])[None, :] == label_local[:, None]
        return k * (torch.exp(logits - lse[:, None]) - hot.float())

    def head_bwd(self, saved, cfg, g, need_x, need_w):
        xl, wl, logits = saved[0], saved[1], saved[7]
        gx, gw = torch.autograd.grad(logits, [xl, wl], g)
        return (gx if need_x else None), (gw if need_w else None)

    def normalize_bwd(self, G, x, inv_x):
        return G  # head_bwd already went through the normalisation (linear in G, so summing over ranks first is equal)
