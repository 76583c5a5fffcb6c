"""debug: per-tensor gradient error of the fp32 HIP path vs the float64 oracle, with and without dropout."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from frhip import synth
from backbone.model_irse import IR_50
from head.metrics import ArcFace
from loss.focal import FocalLoss
from oracle import irse_ref as O
from test_gpu_dropout import host_keep_mask

def run(p):
    B, N = 8, 100
    model = IR_50([112, 112]); synth.fill_state_dict(model.state_dict(), 15)
    sd64 = {k: (v.detach().double().requires_grad_("running" not in k) if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    model.output_layer[1].p = p
    model.compute_dtype = torch.float32
    model = model.cuda().train()
    head = ArcFace(512, N, None).cuda(); hw = synth.uniform(16, "full.head", (N, 512), -0.1, 0.1)
    with torch.no_grad(): head.weight.copy_(hw)
    x = synth.uniform(16, "full.x", (B, 3, 112, 112)); label = synth.labels(16, "full.label", B, N)
    feats = model(x.cuda()); seed = model._runner[0].step_seed
    loss, _ = FocalLoss()(head(feats, label.cuda()), label.cuda()); loss.backward(); torch.cuda.synchronize()
    mask = torch.from_numpy(host_keep_mask(seed, B, 512, 49, p)).double() if p > 0 else None
    if mask is not None and p != 0.5: mask = mask * (1.0 / (1 - p)) / 2.0
    _f, _l, _loss, g64 = O.train_step(sd64, x.double(), label, hw.double().requires_grad_(True), drop_mask=mask)
    named = dict(model.named_parameters()); named["head.weight"] = head.weight
    rows = []
    for k, t in g64.items():
        mine = named[k].grad.cpu().double()
        rows.append((float((mine - t).norm() / (t.norm() + 1e-30)), k, float(t.norm()), float((mine-t).abs().max()), float(t.abs().max())))
    rows.sort(reverse=True)
    print("p =", p, "loss", float(loss.detach()), float(_loss.detach()))
    for r in [r for r in rows if r[2] > 1e-5][:30]:
        if True: print("  rel %.2e  %-34s |ref| %.2e  max|err| %.2e max|ref| %.2e" % r)
    k = [r for r in rows if r[2] > 1e-5][0][1]
    mine, t = named[k].grad.cpu().double().reshape(-1), g64[k].reshape(-1)
    e = (mine - t).abs(); idx = torch.argsort(e, descending=True)[:8]
    print("  worst", k, [(int(i), float(mine[i]), float(t[i])) for i in idx])

run(0.0)
