import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd")); sys.path.insert(0, REPO)
import torch
from frhip import synth
from backbone.model_irse import IR_50
from oracle import irse_ref as O
m = IR_50([112,112]); synth.fill_state_dict(m.state_dict(), 15)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
for k in sd:
    if k.endswith("running_mean"): sd[k] = synth.uniform(31, k, tuple(sd[k].shape), -0.2, 0.2)
    elif k.endswith("running_var"): sd[k] = synth.uniform(31, k, tuple(sd[k].shape), 0.5, 1.5)
m.load_state_dict(sd); m = m.cuda().eval(); m.compute_dtype = torch.float32
x = synth.uniform(16, "full.x", (5, 3, 112, 112))
with torch.no_grad():
    taps = {}
    ref = O.backbone_forward({k: v.clone() for k, v in sd.items()}, x, 50, False, bn_train=False, taps=taps)
    ref64 = O.backbone_forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double(), 50, False, bn_train=False)
    got = m(x.cuda()).cpu()
    plan = m._runner[0].plan
    print("fold", plan.fold, "feat scale", float(ref.abs().max()), "fold vs oracle32", float((got-ref).abs().max()), "oracle32 vs 64", float((ref.double()-ref64).abs().max()), "fold vs 64", float((got.double()-ref64).abs().max()))
    for i in (0, 1, 2, 3, 7, 12, 20, 23):
        o = plan.ubuf[i]["out"].float().cpu().view(5, plan.units[i].Ho, plan.units[i].Ho, -1).permute(0,3,1,2)
        r = taps["body.%d" % i]
        print(i, "unit out scale %.3e  max abs err %.3e" % (float(r.abs().max()), float((o-r).abs().max())))
    os.environ["FRHIP_NO_FOLD"]="1"; m._runner[0].plans = {}
    unf = m(x.cuda()).cpu()
    print("unfolded vs oracle32", float((unf-ref).abs().max()), "vs 64", float((unf.double()-ref64).abs().max()))
