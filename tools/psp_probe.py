"""Phase timing of a pSp (IR-SE-50, 6-channel stem) training step: host enqueue vs GPU time per phase."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylegan-for-facerec_amd"))
import torch
from backbone.restyle_psp import pSp
from head.metrics import ArcFace
from loss.focal import FocalLoss
from frhip import synth
from frhip.optim import SGD
from util.utils import separate_irse_bn_paras
import frhip.functional as FRF
FRF.CHECK_LABELS = False
B = 256
avg = synth.uniform(2, "avg", (3, 112, 112))
m = pSp(size=112, encoder_type="BackboneEncoder", avg_image=avg)
m.encoder.compute_dtype = torch.bfloat16
m = m.cuda().train()
head = ArcFace(512, 7000, None).cuda()
bn, wo = separate_irse_bn_paras(m)
print("param groups", len(wo), len(bn))
opt = SGD([{"params": wo + list(head.parameters()), "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
x = synth.uniform(1, "x", (B, 3, 112, 112)).cuda(); y = synth.labels(1, "y", B, 7000).cuda()
def phases():
    t = [time.perf_counter()]
    f = m(x); t.append(time.perf_counter())
    loss, _ = FocalLoss()(head(f, y), y); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
for _ in range(4): phases()
acc = [0.0] * 6
for _ in range(10):
    for i, v in enumerate(phases()): acc[i] += v / 10
print("host ms: forward %.2f head+loss %.2f zero_grad %.2f backward %.2f opt.step %.2f final-sync %.2f  total %.2f" % (*acc, sum(acc)))
