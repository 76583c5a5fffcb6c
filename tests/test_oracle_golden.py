"""Pins the oracle (oracle/irse_ref.py) against outputs captured from the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py importing /root/reference).

CPU only.  Tolerance: the oracle calls the same torch CPU kernels as the reference, so 1e-5 abs on
logits is loose; most comparisons are bit-equal.
"""
import json
import os

import numpy as np
import pytest
import torch

from frhip import synth
from oracle import irse_ref as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize("kind", ["ArcFace", "CosFace"])
def test_g1_head(golden_dir, kind):
    g = _load(golden_dir, "g1_head")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    w = torch.from_numpy(g["w"]).requires_grad_(True)
    label = torch.from_numpy(g["label"])
    f = O.arcface_forward if kind == "ArcFace" else O.cosface_forward
    y = f(x, w, label)
    assert np.array_equal(y.detach().numpy(), g[kind + ".logits"])  # bit-exact incl. label scatter
    gx, gw = torch.autograd.grad(y, [x, w], torch.from_numpy(g["gout"]))
    _close(gx, g[kind + ".gx"], 1e-5)
    _close(gw, g[kind + ".gw"], 1e-5)


def test_g1_head_easy_margin(golden_dir):
    g = _load(golden_dir, "g1_head")
    y = O.arcface_forward(torch.from_numpy(g["x"]), torch.from_numpy(g["w"]), torch.from_numpy(g["label"]),
                          s=30.0, m=0.35, easy_margin=True)
    assert np.array_equal(y.numpy(), g["ArcFace.easy.logits"])


def test_g2_focal_and_accuracy(golden_dir):
    g = _load(golden_dir, "g2_focal")
    logits = torch.from_numpy(g["logits"]).requires_grad_(True)
    label = torch.from_numpy(g["label"])
    loss = O.focal_loss(logits, label)
    _close(loss, g["loss"], 1e-6)
    (gr,) = torch.autograd.grad(loss, [logits])
    _close(gr, g["grad"], 1e-7)
    p1, p5 = O.topk_accuracy(logits.detach(), label)
    assert float(p1) == float(g["prec1"]) and float(p5) == float(g["prec5"])


BLOCKS = [
    ("ir_64_64_1", False, 64, 64, 1, 16),
    ("ir_64_128_2", False, 64, 128, 2, 16),
    ("irse_128_128_1", True, 128, 128, 1, 8),
    ("irse_256_512_2", True, 256, 512, 2, 8),
]


def block_state(tag, se, cin, depth, stride):
    """State dict of one residual unit with the reference's key names, filled by synth (seed 13)."""
    sd = {}
    if cin != depth:
        sd["shortcut_layer.0.weight"] = torch.empty(depth, cin, 1, 1)
        for k, n in (("weight", depth), ("bias", depth), ("running_mean", depth), ("running_var", depth)):
            sd["shortcut_layer.1." + k] = torch.empty(n)
        sd["shortcut_layer.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def bn(key, n):
        for k in ("weight", "bias", "running_mean", "running_var"):
            sd[key + "." + k] = torch.empty(n)
        sd[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    bn("res_layer.0", cin)
    sd["res_layer.1.weight"] = torch.empty(depth, cin, 3, 3)
    sd["res_layer.2.weight"] = torch.empty(depth)
    sd["res_layer.3.weight"] = torch.empty(depth, depth, 3, 3)
    bn("res_layer.4", depth)
    if se:
        sd["res_layer.5.fc1.weight"] = torch.empty(depth // 16, depth, 1, 1)
        sd["res_layer.5.fc2.weight"] = torch.empty(depth, depth // 16, 1, 1)
    synth.fill_state_dict(sd, 13)
    return sd


@pytest.mark.parametrize("spec", BLOCKS, ids=[b[0] for b in BLOCKS])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_g3_blocks(golden_dir, spec, mode):
    tag, se, cin, depth, stride, hw = spec
    g = _load(golden_dir, "g3_blocks")
    sd = {"u." + k: v for k, v in block_state(tag, se, cin, depth, stride).items()}
    pnames = [k for k in sd if sd[k].is_floating_point() and "running" not in k]
    for k in pnames:
        sd[k].requires_grad_(True)
    x = synth.normal(13, "g3.x." + tag, (4, cin, hw, hw)).requires_grad_(True)
    gout = synth.normal(13, "g3.g." + tag, (4, depth, hw // stride, hw // stride))
    y = O.residual_unit(sd, "u", x, cin, depth, stride, se, mode == "train")
    _close(y, g["%s.%s.y" % (tag, mode)], 2e-5)
    gs = torch.autograd.grad(y, [x] + [sd[k] for k in pnames], gout)
    _close(gs[0], g["%s.%s.gx" % (tag, mode)], 1e-4, 1e-4)
    for k, gr in zip(pnames, gs[1:]):
        ref = g["%s.%s.g.%s" % (tag, mode, k[2:])]
        _close(gr.reshape(-1)[:2048], ref, 2e-3, 1e-4)
        nrm = float(g["%s.%s.gnorm.%s" % (tag, mode, k[2:])])
        assert abs(float(gr.double().norm()) - nrm) <= 1e-4 * max(1.0, nrm)
    if mode == "train":
        for k in sd:
            if "running" in k or "num_batches" in k:
                _close(sd[k], g["%s.train.buf.%s" % (tag, k[2:])], 1e-6)


def test_g4_se(golden_dir):
    g = _load(golden_dir, "g4_se")
    sd = {"s.fc1.weight": torch.empty(8, 128, 1, 1), "s.fc2.weight": torch.empty(128, 8, 1, 1)}
    synth.fill_state_dict({k[2:]: v for k, v in sd.items()}, 14)
    for v in sd.values():
        v.requires_grad_(True)
    x = synth.normal(14, "g4.x", (4, 128, 14, 14)).requires_grad_(True)
    y = O.se_module(sd, "s", x)
    _close(y, g["y"], 1e-6)
    gx, g1, g2 = torch.autograd.grad(y, [x, sd["s.fc1.weight"], sd["s.fc2.weight"]],
                                     synth.normal(14, "g4.g", (4, 128, 14, 14)))
    _close(gx, g["gx"], 1e-5)
    _close(g1, g["gfc1"], 1e-4)
    _close(g2, g["gfc2"], 1e-4)


def build_state(golden_dir, model, seed=15):
    """Full state dict (reference key order/shapes from g8_structure.json) filled by synth."""
    with open(os.path.join(golden_dir, "g8_structure.json")) as f:
        info = json.load(f)[model]
    sd = {}
    for k, shape in info["keys"]:
        sd[k] = torch.zeros(shape, dtype=torch.int64) if k.endswith("num_batches_tracked") else torch.empty(shape)
    synth.fill_state_dict(sd, seed)
    for k in info["param_names"]:
        sd[k].requires_grad_(True)
    return sd, info


FULL = [("g5_ir50", "IR_50", 50, False, "", 8), ("g6_psp", "pSp", 50, True, "encoder.", 8),
        ("g6b_irse101", "IR_SE_101", 100, True, "", 4), ("g6c_irse101_b16", "IR_SE_101", 100, True, "", 16)]


@pytest.mark.parametrize("spec", FULL, ids=[f[0] for f in FULL])
def test_full_models(golden_dir, spec):
    fixture, model, nl, se, prefix, batch = spec
    g = _load(golden_dir, fixture)
    sd, info = build_state(golden_dir, model)
    x = synth.uniform(16, "full.x", (batch, 3, 112, 112))
    label = synth.labels(16, "full.label", batch, 100)
    hw = synth.uniform(16, "full.head", (100, 512), -0.1, 0.1).requires_grad_(True)
    avg = synth.uniform(15, "avg_image", (3, 112, 112)) if model == "pSp" else None
    feats, logits, loss, grads = O.train_step(sd, x, label, hw, num_layers=nl, se=se, prefix=prefix,
                                              avg_image=avg)
    _close(feats, g["features"], 1e-5)
    _close(logits, g["logits"], 1e-5 * 64)
    _close(loss, g["loss"], 1e-5)
    names = list(g["grad_names"])
    assert names == info["param_names"] + ["head.weight"]
    norms = np.array([float(grads[n].double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("g."):
            _close(grads[k[2:]], g[k], 1e-4, 2e-3)
        if k.startswith("buf."):
            _close(sd[k[4:]], g[k], 1e-5)


def test_oracle_at_a_baseline_config_size(golden_dir):
    """Round 5: the oracle against the reference's own step at a BASELINE size -- configs[0]: pSp (IR-SE-50 trunk, 6-channel
    stem, average image) + ArcFace(100), batch 100 (g13 capture; the four larger configs take a minute and tens of GB each and
    are replayed by the GPU suite against the HIP path instead)."""
    g = _load(golden_dir, "g13_configs0_psp_arc100_b100")
    sd, info = build_state(golden_dir, "pSp")
    B, N = 100, 100
    x = synth.uniform(33, "big.x", (B, 3, 112, 112))
    label = synth.labels(33, "big.y", B, N)
    hw = synth.uniform(33, "big.head", (N, 512), -0.05, 0.05).requires_grad_(True)
    avg = synth.uniform(15, "avg_image", (3, 112, 112))
    feats, logits, loss, grads = O.train_step(sd, x, label, hw, num_layers=50, se=True, prefix="encoder.", avg_image=avg,
                                              head="ArcFace", s=64.0, m=0.5)
    _close(feats, g["features"], 1e-5)
    _close(loss, g["loss"], 1e-5)
    cols = torch.from_numpy(g["logit_cols"].astype(np.int64))
    _close(logits.detach().gather(1, cols), g["logit_vals"], 1e-5 * 64)
    names = list(g["grad_names"])
    norms = np.array([float(grads[n].double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("gi."):
            _close(grads[k[3:]].detach().reshape(-1)[torch.from_numpy(g[k])], g["g." + k[3:]], 1e-4, 2e-3)


def test_g7_two_sgd_steps(golden_dir):
    """A0/A15: two full steps with the param-group split (is_bn_key) and oracle sgd_step."""
    g = _load(golden_dir, "g7_sgd")
    sd, info = build_state(golden_dir, "IR_50")
    hw = synth.uniform(16, "full.head", (100, 512), -0.1, 0.1).requires_grad_(True)
    names = info["param_names"] + ["head.weight"]
    assert names == list(g["param_names"])
    bufs = {n: None for n in names}
    for step in range(2):
        x = synth.uniform(17, "sgd.x%d" % step, (8, 3, 112, 112))
        label = synth.labels(17, "sgd.label%d" % step, 8, 100)
        feats, logits, loss, grads = O.train_step(sd, x, label, hw)
        p1, p5 = O.topk_accuracy(logits.detach(), label)
        assert abs(float(loss.detach()) - g["loss"][step]) < 1e-4
        assert float(p1) == g["prec1"][step] and float(p5) == g["prec5"][step]
        with torch.no_grad():
            for n in names:
                p = hw if n == "head.weight" else sd[n]
                wd = 0.0 if (n != "head.weight" and O.is_bn_key(n)) else 2e-3
                b = O.sgd_step([p], [grads[n]], [bufs[n]], 0.03, 0.9, wd)
                bufs[n] = b[0]
    ps = [hw if n == "head.weight" else sd[n] for n in names]
    np.testing.assert_allclose([float(p.double().norm()) for p in ps], g["param_norms"], rtol=1e-4)
    np.testing.assert_allclose([float(bufs[n].double().norm()) for n in names], g["buf_norms"], rtol=5e-3,
                               atol=1e-6)
    _close(sd["input_layer.0.weight"], g["w.input_layer.0.weight"], 1e-4)
    _close(hw[:4], g["w.head.weight.rows0_3"], 1e-4)


def test_g8_param_split_matches_is_bn_key(golden_dir):
    with open(os.path.join(golden_dir, "g8_structure.json")) as f:
        info = json.load(f)
    for model, (nbn, nwo) in {"IR_50": (108, 79), "pSp": (108, 127)}.items():
        names = info[model]["param_names"]
        assert sum(O.is_bn_key(n) for n in names) == nbn == info[model]["n_bn"]
        assert sum(not O.is_bn_key(n) for n in names) == nwo == info[model]["n_wo"]
    assert len(info["IR_50"]["keys"]) == 349 and len(info["IR_SE_50"]["keys"]) == 397
    assert len(O.unit_table(50)) == 24 and len(O.unit_table(100)) == 49
