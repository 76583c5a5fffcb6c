import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd")); sys.path.insert(0, REPO)
import torch
from backbone.model_irse import IR_50, IR_SE_50, IR_SE_101
from head.metrics import ArcFace
from loss.focal import FocalLoss
from frhip.optim import SGD
from frhip import synth
from util.utils import separate_irse_bn_paras
import frhip.functional as FRF
FRF.CHECK_LABELS = False
for name, ctor, B in (("IR_50", IR_50, 256), ("IR_SE_50", IR_SE_50, 256), ("IR_SE_101", IR_SE_101, 128)):
    m = ctor([112, 112]); m.compute_dtype = torch.bfloat16; m = m.cuda().train()
    head = ArcFace(512, 7000, None).cuda()
    bn, wo = separate_irse_bn_paras(m)
    opt = SGD([{"params": wo + list(head.parameters()), "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
    x = synth.uniform(1, "x", (B, 3, 112, 112)).cuda(); y = synth.labels(1, "y", B, 7000).cuda()
    def step():
        loss, _ = FocalLoss()(head(m(x), y), y); opt.zero_grad(); loss.backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s B=%d: %.2f ms/step, %.0f img/s" % (name, B, dt * 1e3, B / dt), flush=True)
    del m, head, opt; torch.cuda.empty_cache()

# pSp (IR-SE-50 trunk, 6-channel stem with the average image): BASELINE configs 1 and 5
from backbone.restyle_psp import pSp
B = 256
avg = synth.uniform(2, "avg", (3, 112, 112))
m = pSp(size=112, encoder_type="BackboneEncoder", avg_image=avg)
m.encoder.compute_dtype = torch.bfloat16
m = m.cuda().train()
head = ArcFace(512, 7000, None).cuda()
bn, wo = separate_irse_bn_paras(m)
opt = SGD([{"params": wo + list(head.parameters()), "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
x = synth.uniform(1, "x", (B, 3, 112, 112)).cuda(); y = synth.labels(1, "y", B, 7000).cuda()
def step():
    loss, _ = FocalLoss()(head(m(x), y), y); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("pSp(IR_SE_50, 6ch) B=%d: %.2f ms/step, %.0f img/s" % (B, dt * 1e3, B / dt), flush=True)
