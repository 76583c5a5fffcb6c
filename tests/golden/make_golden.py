#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REFERENCE (/root/reference) on CPU.

Runs only in the build container (the reference is absent on the GPU box).  Nothing from the reference is
copied: this script imports its modules (with import-only stand-ins for the absent, unused packages
torchvision / imageio / bcolz / wandb -- SURVEY.md App. C), feeds them inputs and weights from the
repo's own counter-based generator (stylegan-for-facerec_amd/frhip/synth.py) and stores the outputs.

    python tests/golden/make_golden.py            # writes next to this file

Fixtures: g1_head, g2_focal, g3_blocks, g4_se, g5_ir50, g6_psp, g7_sgd, g8_structure.json,
g9_stage2, g10_verification (8f rank 2: the k-fold verification metrics of util/verification.py),
g11_dataset.json (dataset.py FacesDataset on a tiny tree with `Race^id` directory names), g12_resnet_structure.json
(state-dict keys / shapes of backbone/model_resnet.py, which train.py:6 imports), g13_* (round 5: one training step of every
BASELINE config AT ITS OWN SIZE -- batch 100 / 128 / 256, 100 / 7000 / 28000 identities -- run by the reference itself in fp32
on the host cores: `python tests/golden/make_golden.py g13` takes ~10 minutes and ~40 GB; one config per process).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.dont_write_bytecode = True


def _load_synth():
    spec = importlib.util.spec_from_file_location(
        "frhip_synth", os.path.join(REPO, "stylegan-for-facerec_amd", "frhip", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth = _load_synth()


def _install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, name):
            return _Any()

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = mk("torchvision")
    tv.models = mk("torchvision.models")
    tv.models.resnet = mk("torchvision.models.resnet", resnet34=_Any)
    tv.transforms = mk("torchvision.transforms", Compose=_Any, ToPILImage=_Any, ToTensor=_Any, Normalize=_Any,
                       Resize=_Any, CenterCrop=_Any, functional=_Any())
    mk("imageio")
    mk("bcolz")
    mk("wandb")
    mk("turbojpeg", TurboJPEG=_Any)  # dataset.py:8 imports it and never uses it


_install_stubs()
sys.path.insert(0, REF)
torch.manual_seed(0)
torch.set_num_threads(8)

from backbone import model_irse as ref_irse  # noqa: E402
from backbone import restyle_psp as ref_psp  # noqa: E402
from backbone import restyle_psp_helpers as ref_helpers  # noqa: E402
from head import metrics as ref_heads  # noqa: E402
from loss.focal import FocalLoss as RefFocal  # noqa: E402
from util import utils as ref_utils  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: live buffers are mutated later


def onehot(label, n):
    return torch.zeros(label.shape[0], n).scatter_(1, label.view(-1, 1), 1)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------ G1


def head_inputs():
    """x [8,512], W [100,512], labels with edge cases; rows 4..7 force cos(label) to +1, -1, th, just above th."""
    n, d = 100, 512
    x = synth.normal(11, "g1.x", (8, d))
    w = synth.uniform(11, "g1.w", (n, d), -0.1, 0.1)
    label = torch.tensor([0, n - 1, 7, 7, 13, 21, 34, 55], dtype=torch.int64)
    wn = torch.nn.functional.normalize(w)
    x[4] = 3.0 * wn[13]  # cos = +1
    x[5] = -2.0 * wn[21]  # cos = -1  (< th)
    th = np.cos(np.pi - 0.5)
    for row, cls, c in ((6, 34, th), (7, 55, th + 1e-4)):
        u = wn[cls]
        v = synth.normal(11, "g1.v%d" % row, (d,))
        v = v - (v @ u) * u
        v = v / v.norm()
        x[row] = 1.7 * (float(c) * u + float(np.sqrt(1 - c * c)) * v)
    return x, w, label


def g1_head():
    x0, w0, label = head_inputs()
    out = {"x": npy(x0), "w": npy(w0), "label": npy(label)}
    gout = synth.normal(11, "g1.gout", (8, 100))
    out["gout"] = npy(gout)
    for kind in ("ArcFace", "CosFace"):
        x = x0.clone().requires_grad_(True)
        head = getattr(ref_heads, kind)(512, 100, None)
        with torch.no_grad():
            head.weight.copy_(w0)
        if kind == "ArcFace":
            y = head(x, label, onehot_vec=onehot(label, 100))
        else:
            y = head(x, label)
        gx, gw = torch.autograd.grad(y, [x, head.weight], gout)
        out[kind + ".logits"] = npy(y)
        out[kind + ".gx"] = npy(gx)
        out[kind + ".gw"] = npy(gw)
    # easy-margin + non-default s/m variant
    head = ref_heads.ArcFace(512, 100, None, s=30.0, m=0.35, easy_margin=True)
    with torch.no_grad():
        head.weight.copy_(w0)
    out["ArcFace.easy.logits"] = npy(head(x0, label, onehot_vec=onehot(label, 100)))
    save("g1_head", **out)


# ------------------------------------------------------------------------------------------------ G2


def g2_focal():
    logits = synth.normal(12, "g2.logits", (8, 100), std=8.0).requires_grad_(True)
    label = synth.labels(12, "g2.label", 8, 100)
    loss, aux = RefFocal()(logits, label)
    assert aux is None
    (g,) = torch.autograd.grad(loss, [logits])
    p1, p5 = ref_utils.accuracy(logits.data, label, topk=(1, 5))
    save("g2_focal", logits=npy(logits), label=npy(label), loss=npy(loss), grad=npy(g), prec1=npy(p1),
         prec5=npy(p5))


# ------------------------------------------------------------------------------------------------ G3/G4

BLOCKS = [
    ("ir_64_64_1", False, 64, 64, 1, 16),
    ("ir_64_128_2", False, 64, 128, 2, 16),
    ("irse_128_128_1", True, 128, 128, 1, 8),
    ("irse_256_512_2", True, 256, 512, 2, 8),
]


def g3_blocks():
    out = {}
    for tag, se, cin, depth, stride, hw in BLOCKS:
        cls = ref_irse.bottleneck_IR_SE if se else ref_irse.bottleneck_IR
        blk = cls(cin, depth, stride)
        sd = {k: v.clone() for k, v in blk.state_dict().items()}  # clone: train pass mutates live buffers
        synth.fill_state_dict(sd, 13)
        blk.load_state_dict(sd)
        x = synth.normal(13, "g3.x." + tag, (4, cin, hw, hw)).requires_grad_(True)
        ho = hw // stride
        gout = synth.normal(13, "g3.g." + tag, (4, depth, ho, ho))
        for mode in ("train", "eval"):
            blk.load_state_dict(sd)
            blk.train(mode == "train")
            y = blk(x)
            names = [n for n, _ in blk.named_parameters()]
            gs = torch.autograd.grad(y, [x] + [p for _, p in blk.named_parameters()], gout)
            out["%s.%s.y" % (tag, mode)] = npy(y)
            out["%s.%s.gx" % (tag, mode)] = npy(gs[0])
            for n, g in zip(names, gs[1:]):
                # big conv-weight grads are pinned by their L2 norm + the first 2048 elements (fixture size)
                out["%s.%s.gnorm.%s" % (tag, mode, n)] = np.array(float(g.double().norm()))
                out["%s.%s.g.%s" % (tag, mode, n)] = npy(g.reshape(-1)[:2048])
            if mode == "train":
                for n, b in blk.named_buffers():
                    out["%s.train.buf.%s" % (tag, n)] = npy(b)
    save("g3_blocks", **out)


def g4_se():
    se = ref_irse.SEModule(128, 16)
    sd = se.state_dict()
    synth.fill_state_dict(sd, 14)
    se.load_state_dict(sd)
    x = synth.normal(14, "g4.x", (4, 128, 14, 14)).requires_grad_(True)
    gout = synth.normal(14, "g4.g", (4, 128, 14, 14))
    y = se(x)
    gx, g1, g2 = torch.autograd.grad(y, [x, se.fc1.weight, se.fc2.weight], gout)
    save("g4_se", y=npy(y), gx=npy(gx), gfc1=npy(g1), gfc2=npy(g2))


# ------------------------------------------------------------------------------------------------ G5/G6/G7


def _disable_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def full_model(kind):
    """Returns (model, prefix, avg_image|None).  Weights from synth.fill_state_dict(seed 15)."""
    if kind == "ir50":
        model = ref_irse.IR_50([112, 112])
        prefix = ""
        avg = None
    elif kind == "irse101":
        model = ref_irse.IR_SE_101([112, 112])
        prefix = ""
        avg = None
    else:
        model = ref_psp.pSp(size=112, checkpoint_path=None, avg_image=None, include_dropout=False)
        prefix = "encoder."
        avg = synth.uniform(15, "avg_image", (3, 112, 112))
        model.avg_image = avg
    sd = model.state_dict()
    synth.fill_state_dict(sd, 15)
    model.load_state_dict(sd)
    _disable_dropout(model)
    return model, prefix, avg


def run_full(kind, tag, batch=8, nclass=100, extra_grads=()):
    """extra_grads: further parameter names whose full fp32 + float64 gradient tensors are stored (g6c: the squeeze-excite
    MLP weights of the first units, whose batch-4 gradients in g6b are sums of 4 cancelling per-image terms)."""
    model, prefix, avg = full_model(kind)
    model.train()
    x = synth.uniform(16, "full.x", (batch, 3, 112, 112))
    label = synth.labels(16, "full.label", batch, nclass)
    head = ref_heads.ArcFace(512, nclass, None, s=64.0)
    with torch.no_grad():
        head.weight.copy_(synth.uniform(16, "full.head", (nclass, 512), -0.1, 0.1))
    feats = model(x)
    logits = head(feats, label, onehot_vec=onehot(label, nclass))
    loss, _ = RefFocal()(logits, label)
    named = list(model.named_parameters())
    gs = torch.autograd.grad(loss, [p for _, p in named] + [head.weight])
    out = {"features": npy(feats), "logits": npy(logits), "loss": npy(loss)}
    out["grad_names"] = np.array([n for n, _ in named] + ["head.weight"])
    out["grad_norms"] = np.array([float(g.double().norm()) for g in gs])
    gd = dict(zip([n for n, _ in named], gs))
    for n in (prefix + "input_layer.0.weight", prefix + "output_layer.4.weight", prefix + "output_layer.4.bias",
              prefix + "body.0.res_layer.2.weight", prefix + "body.3.shortcut_layer.0.weight") + tuple(extra_grads):
        out["g." + n] = npy(gd[n])
    out["g.head.weight"] = npy(gs[-1])
    # the same step in float64 = "truth" for judging fp32 implementations against the reference's own fp32 noise
    model64, _, avg64 = full_model(kind)
    model64 = model64.double().train()
    if avg64 is not None:
        model64.avg_image = avg64.double()
    head64 = ref_heads.ArcFace(512, nclass, None, s=64.0).double()
    with torch.no_grad():
        head64.weight.copy_(head.weight.double())
    feats64 = model64(x.double())
    logits64 = head64(feats64, label, onehot_vec=onehot(label, nclass).double())
    loss64, _ = RefFocal()(logits64, label)
    named64 = list(model64.named_parameters())
    gs64 = torch.autograd.grad(loss64, [p for _, p in named64] + [head64.weight])
    gd64 = dict(zip([n for n, _ in named64], gs64))
    out["features64"], out["logits64"] = npy(feats64), npy(logits64)
    out["grad_norms64"] = np.array([float(g.norm()) for g in gs64])
    for n in list(out):
        if n.startswith("g.") and n != "g.head.weight":
            out["g64." + n[2:]] = npy(gd64[n[2:]])
    out["g64.head.weight"] = npy(gs64[-1])
    bufs = dict(model.named_buffers())
    for n in (prefix + "input_layer.1", prefix + "body.7.res_layer.4", prefix + "output_layer.4"):
        out["buf." + n + ".running_mean"] = npy(bufs[n + ".running_mean"])
        out["buf." + n + ".running_var"] = npy(bufs[n + ".running_var"])
        out["buf." + n + ".num_batches_tracked"] = npy(bufs[n + ".num_batches_tracked"])
    save(tag, **out)


def g7_sgd():
    """Two SGD steps, lr 0.03 / momentum 0.9 / wd 2e-3 on the non-BN group (train.py:188-196, 296-316)."""
    model, prefix, _ = full_model("ir50")
    model.train()
    nclass, batch = 100, 8
    head = ref_heads.ArcFace(512, nclass, None, s=64.0)
    with torch.no_grad():
        head.weight.copy_(synth.uniform(16, "full.head", (nclass, 512), -0.1, 0.1))
    bn, wo = ref_utils.separate_irse_bn_paras(model)
    _, hwo = ref_utils.separate_irse_bn_paras(head)
    opt = torch.optim.SGD([{"params": wo + hwo, "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
    out = {"loss": [], "prec1": [], "prec5": []}
    for step in range(2):
        x = synth.uniform(17, "sgd.x%d" % step, (batch, 3, 112, 112))
        label = synth.labels(17, "sgd.label%d" % step, batch, nclass)
        feats = model(x)
        logits = head(feats, label, onehot_vec=onehot(label, nclass))
        loss, _ = RefFocal()(logits, label)
        p1, p5 = ref_utils.accuracy(logits.data, label, topk=(1, 5))
        out["loss"].append(float(loss))
        out["prec1"].append(float(p1))
        out["prec5"].append(float(p5))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step == 0:  # state after ONE step: fp32 implementations agree tightly here, step 2 amplifies rounding
            ps1 = [p for _, p in model.named_parameters()] + [head.weight]
            out["param_norms_step1"] = [float(p.double().norm()) for p in ps1]
            snap1 = {"w1.input_layer.0.weight": npy(model.input_layer[0].weight),
                     "w1.body.10.res_layer.1.weight.head": npy(model.body[10].res_layer[1].weight.reshape(-1)[:4096]),
                     "w1.head.weight.rows0_3": npy(head.weight[:4])}
    res = {k: np.array(v) for k, v in out.items()}
    res.update(snap1)
    names = [n for n, _ in model.named_parameters()]
    res["param_names"] = np.array(names + ["head.weight"])
    ps = [p for _, p in model.named_parameters()] + [head.weight]
    res["param_sums"] = np.array([float(p.double().sum()) for p in ps])
    res["param_norms"] = np.array([float(p.double().norm()) for p in ps])
    res["buf_norms"] = np.array([float(opt.state[p]["momentum_buffer"].double().norm()) for p in ps])
    res["w.input_layer.0.weight"] = npy(model.input_layer[0].weight)
    res["w.output_layer.4.weight"] = npy(model.output_layer[4].weight)
    res["w.head.weight.rows0_3"] = npy(head.weight[:4])
    save("g7_sgd", **res)


# ------------------------------------------------------------------------------------------------ G8/G9


def g8_structure():
    info = {}
    specs = {
        "IR_50": lambda: ref_irse.IR_50([112, 112]),
        "IR_SE_50": lambda: ref_irse.IR_SE_50([112, 112]),
        "IR_SE_101": lambda: ref_irse.IR_SE_101([112, 112]),
        "IR_101": lambda: ref_irse.IR_101([112, 112]),
        "pSp": lambda: ref_psp.pSp(size=112),
        "pSp34": lambda: ref_psp.pSp(size=112, encoder_type="BackboneEncoder34"),
    }
    for name, ctor in specs.items():
        m = ctor()
        bn, wo = ref_utils.separate_irse_bn_paras(m)
        info[name] = {
            "keys": [[k, list(v.shape)] for k, v in m.state_dict().items()],
            "param_names": [n for n, _ in m.named_parameters()],
            "n_bn": len(bn), "n_wo": len(wo),
            "bn_numel": int(sum(p.numel() for p in bn)), "wo_numel": int(sum(p.numel() for p in wo)),
        }
    for name in ("ArcFace", "CosFace", "SphereFace", "Am_softmax"):
        h = getattr(ref_heads, name)(512, 100, None)
        bn, wo = ref_utils.separate_irse_bn_paras(h)
        info[name] = {"keys": [[k, list(v.shape)] for k, v in h.state_dict().items()], "n_bn": len(bn),
                      "n_wo": len(wo)}
    # pSp with additional dropouts (restyle_psp.py:401-405, helpers :201-209): child order after insertion
    m = ref_psp.pSp(size=112, include_dropout=0.15)
    blk = m.encoder.body[0]
    info["pSp.dropout.body0.res_layer"] = [type(c).__name__ for c in blk.res_layer]
    blk = m.encoder.body[3]
    info["pSp.dropout.body3.shortcut_layer"] = [type(c).__name__ for c in blk.shortcut_layer]
    try:
        ref_psp.pSp(size=112, encoder_type="nope")
    except Exception as e:  # noqa: BLE001
        info["pSp.bad_encoder_error"] = str(e)
    # LR helpers (util/utils.py:184-196)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.03)
    ref_utils.warm_up_lr(3, 10, 0.03, opt)
    info["warm_up_lr_3_10_0.03"] = opt.param_groups[0]["lr"]
    opt.param_groups[0]["lr"] = 0.03
    ref_utils.schedule_lr(opt)
    info["schedule_lr_0.03"] = opt.param_groups[0]["lr"]
    with open(os.path.join(HERE, "g8_structure.json"), "w") as f:
        json.dump(info, f)
    print("wrote g8_structure.json")


def g9_stage2():
    """Stage-2 checkpoint import (restyle_psp.py:419-437): only encoder.input_layer/body load."""
    src = ref_psp.pSp(size=112)
    sd = {"encoder." + k[len("encoder."):]: v.clone() for k, v in src.state_dict().items()
          if k.startswith("encoder.input_layer") or k.startswith("encoder.body")}
    synth.fill_state_dict(sd, 19)
    ckpt_sd = dict(sd)
    ckpt_sd["encoder.styles.0.convs.0.weight"] = torch.ones(2, 2)
    ckpt_sd["decoder.style.1.weight"] = torch.ones(3)
    path = os.path.join("/tmp", "g9_stage2_ckpt.pt")
    torch.save({"state_dict": ckpt_sd, "latent_avg": torch.zeros(18, 512), "opts": {"x": 1}}, path)
    torch.manual_seed(5)
    m = ref_psp.pSp(size=112, checkpoint_path=path)
    got = m.state_dict()
    loaded = [k for k in sd if torch.equal(got[k], sd[k])]
    out_keys = [k for k in got if k.startswith("encoder.output_layer")]
    res = {
        "n_ckpt_encoder_keys": len(sd), "n_loaded_equal": len(loaded),
        "output_layer_keys": out_keys,
        "ignored": ["encoder.styles.0.convs.0.weight", "decoder.style.1.weight"],
        "sum.encoder.input_layer.0.weight": float(got["encoder.input_layer.0.weight"].double().sum()),
        "sum.encoder.body.23.res_layer.3.weight": float(got["encoder.body.23.res_layer.3.weight"].double().sum()),
    }
    with open(os.path.join(HERE, "g9_stage2.json"), "w") as f:
        json.dump(res, f)
    os.remove(path)
    print("wrote g9_stage2.json")


def verification_inputs(n_pairs, dim, seed, tag):
    """Seeded pair embeddings with a real same/different signal: l2-normalised rows, pairs interleaved (2i, 2i+1)."""
    base = synth.normal(seed, tag + ".a", (n_pairs, dim))
    noise = synth.normal(seed, tag + ".n", (n_pairs, dim))
    other = synth.normal(seed, tag + ".o", (n_pairs, dim))
    same = (synth.uniform(seed, tag + ".s", (n_pairs,), 0.0, 1.0) < 0.5)
    second = torch.where(same.view(-1, 1), base + 0.9 * noise, other)
    emb = torch.stack([base, second], 1).reshape(2 * n_pairs, dim)
    emb = emb / emb.norm(dim=1, keepdim=True)
    return emb.double().numpy(), same.numpy()


def g10_verification():
    """util/verification.py evaluate(): k-fold ROC / accuracy / best threshold on seeded embeddings."""
    from util import verification as V
    out = {}
    for tag, n_pairs, folds in (("a", 600, 10), ("b", 203, 10), ("c", 57, 5)):
        emb, same = verification_inputs(n_pairs, 32, 77, "ver." + tag)
        tpr, fpr, acc, best = V.evaluate(emb, same, nrof_folds=folds)
        # (calculate_val is dead code in the reference -- commented out of evaluate() -- and its interp1d call
        #  rejects the duplicate FAR values with the scipy of this container: not captured)
        out.update({tag + "_tpr": tpr, tag + "_fpr": fpr, tag + "_acc": acc, tag + "_best": best})
    save("g10_verification", **out)


# directory name -> file names.  Covers: ethnicity prefixes (label order = order of the BARE ids, dataset.py:45-49), the
# same id under two prefixes (one class), a name without '^', a directory holding no .jpg (contributes no class), and
# non-.jpg files (ignored by the '*/*.jpg' glob, dataset.py:38)
DATASET_TREE = {
    "Caucasian^m49.r8743": ["0001.jpg", "0002.jpg"],
    "African^m99.zz": ["a.jpg"],
    "Asian^m00.first": ["x.jpg", "y.jpg", "notes.txt"],
    "Indian^m49.r8743": ["dup.jpg"],
    "plain_identity": ["p.jpg", "q.png"],
    "only_png": ["z.png"],
    "empty_dir": [],
}


def make_dataset_tree(root, tree):
    from PIL import Image
    for d, files in tree.items():
        os.makedirs(os.path.join(root, d), exist_ok=True)
        for k, f in enumerate(files):
            path = os.path.join(root, d, f)
            if f.endswith(".txt"):
                open(path, "w").write("x")
            else:
                Image.fromarray(np.full((8, 8, 3), 10 * k + len(d), np.uint8)).save(path)


def g11_dataset():
    """dataset.py:17-91 on DATASET_TREE: file order, classes, id2label, per-sample labels."""
    import tempfile
    import dataset as ref_dataset
    with tempfile.TemporaryDirectory() as root:
        make_dataset_tree(root, DATASET_TREE)
        ds = ref_dataset.FacesDataset(root)
        items = [ds[i] for i in range(len(ds))]
        info = {"tree": DATASET_TREE, "filenames": [os.path.relpath(f, root) for f in ds.filenames],
                "classes": list(ds.classes), "id2label": dict(ds.id2label), "n_identities": ds.n_identities,
                "orig_n_samples": ds.orig_n_samples, "dims": list(ds.dims), "labels": [int(it[1]) for it in items],
                "len": len(ds), "item0_type": type(items[0][0]).__module__.split(".")[0]}
    with open(os.path.join(HERE, "g11_dataset.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print("g11_dataset.json", info["classes"], info["labels"])


def g12_resnet_structure():
    """backbone/model_resnet.py:91-188: ordered state-dict keys + shapes, output shape, init facts."""
    from backbone import model_resnet as ref_resnet
    info = {}
    for name, size in (("ResNet_50", 112), ("ResNet_101", 112), ("ResNet_152", 224)):
        m = getattr(ref_resnet, name)([size, size])
        sd = m.state_dict()
        info[name] = {"input": size, "keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()],
                      "n_params": int(sum(p.numel() for p in m.parameters()))}
        if name == "ResNet_50":
            m.eval()
            with torch.no_grad():
                info[name]["out_shape"] = list(m(torch.zeros(2, 3, size, size)).shape)
            info[name]["bn3_weight_zero"] = bool(float(m.layer1[0].bn3.weight.abs().sum()) == 0.0)
    with open(os.path.join(HERE, "g12_resnet_structure.json"), "w") as f:
        json.dump(info, f)
    print("g12_resnet_structure.json", {k: (len(v["keys"]), v["n_params"]) for k, v in info.items()})


# ------------------------------------------------------------------------------------------------ G13
# One training step of the BASELINE configs at their own sizes (BASELINE.json configs[0..4]; bench.py times exactly these), by
# the reference modules in fp32.  What is stored is small: loss, features, a sample of the logits (label column + 63 fixed
# columns per row), every per-parameter gradient norm, a fixed sample of <= 16384 elements of gradient tensors along the depth
# (+ the head weight), and the running statistics of three BatchNorms.
BENCH_SIZE = {
    "g13_configs1_ir50_arc7000_b256": ("ir50", "ArcFace", 7000, 256),
    "g13_configs2_ir50_arc28000_b256": ("ir50", "ArcFace", 28000, 256),
    "g13_configs3_irse101_cos28000_b128": ("irse101", "CosFace", 28000, 128),
    "g13_configs4_psp_arc28000_b256": ("psp", "ArcFace", 28000, 256),
    "g13_configs0_psp_arc100_b100": ("psp", "ArcFace", 100, 100),
}


def bench_probe_names(prefix):
    return ([prefix + "input_layer.0.weight"] +
            [prefix + "body.%d.res_layer.%d.weight" % (u, k) for u, k in ((0, 1), (0, 3), (3, 3), (12, 1), (21, 3), (23, 1))] +
            [prefix + "output_layer.3.weight", prefix + "output_layer.4.weight", prefix + "body.7.res_layer.4.weight"])


def sample_index(name, numel, n=16384):
    """A fixed sample of element indices of a tensor (all of it when it is small): a counter hash of the name, stored with the
    fixture so that no random-number library is part of the contract."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    seed = sum((i + 1) * ord(c) for i, c in enumerate(name)) % 1000003
    k = np.arange(n, dtype=np.uint64)
    h = (k * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(31)
    h = (h * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(29)
    return np.unique((h % np.uint64(numel)).astype(np.int64))


def run_bench_size(tag):
    kind, head_name, nclass, batch = BENCH_SIZE[tag]
    model, prefix, avg = full_model(kind)
    model.train()
    x = synth.uniform(33, "big.x", (batch, 3, 112, 112))
    label = synth.labels(33, "big.y", batch, nclass)
    head = getattr(ref_heads, head_name)(512, nclass, None)
    with torch.no_grad():
        head.weight.copy_(synth.uniform(33, "big.head", (nclass, 512), -0.05, 0.05))
    feats = model(x)
    logits = head(feats, label, onehot_vec=onehot(label, nclass)) if head_name == "ArcFace" else head(feats, label)
    loss, _ = RefFocal()(logits, label)
    named = list(model.named_parameters())
    gs = torch.autograd.grad(loss, [p for _, p in named] + [head.weight])
    out = {"features": npy(feats), "loss": npy(loss), "labels": npy(label)}
    cols = np.stack([np.concatenate(([int(label[r])], (np.arange(63) * 109 + 7 * r + 1) % nclass)) for r in range(batch)])
    out["logit_cols"] = cols.astype(np.int32)
    out["logit_vals"] = npy(logits)[np.arange(batch)[:, None], cols]
    out["logit_absmax"] = np.array(float(logits.detach().abs().max()))
    out["grad_names"] = np.array([n for n, _ in named] + ["head.weight"])
    out["grad_norms"] = np.array([float(g.double().norm()) for g in gs])
    gd = dict(zip([n for n, _ in named] + ["head.weight"], gs))
    for n in bench_probe_names(prefix) + ["head.weight"]:
        idx = sample_index(n, gd[n].numel())
        out["gi." + n] = idx
        out["g." + n] = npy(gd[n]).reshape(-1)[idx]
    bufs = dict(model.named_buffers())
    # running statistics after the step: the stem, one BN2, the output BatchNorm1d -- and BN1 of EVERY unit: on the bf16 path
    # those are derived from moments along chains of up to 30 identity units (FR_PRO_RESBN), never measured on the tensor
    nunits = sum(1 for k in bufs if k.endswith(".res_layer.0.running_mean"))
    for n in [prefix + "input_layer.1", prefix + "body.7.res_layer.4", prefix + "output_layer.4"] + \
            [prefix + "body.%d.res_layer.0" % u for u in range(nunits)]:
        out["buf." + n + ".running_mean"] = npy(bufs[n + ".running_mean"])
        out["buf." + n + ".running_var"] = npy(bufs[n + ".running_var"])
    save(tag, **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g6b", "g6c", "g7", "g8", "g9", "g10", "g11", "g12"]
    if "g1" in which:
        g1_head()
    if "g2" in which:
        g2_focal()
    if "g3" in which:
        g3_blocks()
    if "g4" in which:
        g4_se()
    if "g5" in which:
        run_full("ir50", "g5_ir50")
    if "g6" in which:
        run_full("psp", "g6_psp")
    if "g6b" in which:
        run_full("irse101", "g6b_irse101", batch=4)
    if "g6c" in which:  # round 3: IR-SE-101 at batch 16 -- SE-MLP gradients that are not 4 cancelling terms
        run_full("irse101", "g6c_irse101_b16", batch=16,
                 extra_grads=["body.%d.res_layer.5.%s.weight" % (u, fc) for u in (0, 1, 2, 16) for fc in ("fc1", "fc2")])
    if "g7" in which:
        g7_sgd()
    if "g8" in which:
        g8_structure()
    if "g9" in which:
        g9_stage2()
    if "g10" in which:
        g10_verification()
    if "g11" in which:
        g11_dataset()
    if "g12" in which:
        g12_resnet_structure()
    if "g13" in which:  # tens of GB each: one process per config
        import subprocess
        for tag in BENCH_SIZE:
            subprocess.run([sys.executable, os.path.abspath(__file__), tag], check=True)
    for tag in which:
        if tag in BENCH_SIZE:
            run_bench_size(tag)
