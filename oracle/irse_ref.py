"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain PyTorch fp32/fp64, functional, state-dict driven) of the reference's Stage-3
training step: IR / IR-SE backbone, ArcFace / CosFace margin heads, focal loss, top-k accuracy, SGD.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``stylegan-for-facerec_amd/``) never does and has no CPU fallback.

Parity pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference itself, captured in this container by
``tests/golden/make_golden.py`` (imports /root/reference with import-only stubs) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference file:line it restates (paths relative to /root/reference).
The arithmetic primitives (conv2d, batch_norm, prelu, linear, normalize, cross_entropy) are PyTorch's,
exactly as in the reference, which is itself pure PyTorch (third-party arithmetic = torch, unpinned by
the reference for Stage 3; this container runs torch 2.10.0 CPU kernels).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- architecture tables


def unit_table(num_layers):
    """[(in_channel, depth, stride)] per residual unit.

    backbone/model_irse.py:98-126 (get_block/get_blocks) and backbone/restyle_psp_helpers.py:33-64
    (adds the 34-layer variant).  First unit of each stage has stride 2.
    """
    stages = {
        34: (3, 4, 6, 3),
        50: (3, 4, 14, 3),
        100: (3, 13, 30, 3),
        152: (3, 8, 36, 3),
    }[num_layers]
    widths = ((64, 64), (64, 128), (128, 256), (256, 512))
    units = []
    for (cin, depth), n in zip(widths, stages):
        units.append((cin, depth, 2))
        units.extend((depth, depth, 1) for _ in range(n - 1))
    return units


# ----------------------------------------------------------------------------- building blocks


def _bn(sd, key, x, train, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d/1d forward incl. running-stat update (torch defaults)."""
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    if train and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], train, momentum, eps)


def se_module(sd, key, x):
    """SEModule: x * sigmoid(fc2(relu(fc1(avgpool(x)))))  -- backbone/model_irse.py:23-46,
    backbone/restyle_psp_helpers.py:67-83."""
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.relu(F.conv2d(s, sd[key + ".fc1.weight"]))
    s = torch.sigmoid(F.conv2d(s, sd[key + ".fc2.weight"]))
    return x * s


def residual_unit(sd, key, x, cin, depth, stride, se, bn_train):
    """bottleneck_IR / bottleneck_IR_SE -- backbone/model_irse.py:49-91,
    backbone/restyle_psp_helpers.py:97-199 (ModuleList variant, same child indices).

    shortcut: MaxPool2d(1, stride) == x[:, :, ::s, ::s] when cin == depth, else conv1x1(stride) -> BN.
    residual: BN -> conv3x3(s1) -> PReLU -> conv3x3(stride) -> BN [-> SE].
    """
    if cin == depth:
        sc = x[:, :, ::stride, ::stride] if stride > 1 else x
    else:
        sc = F.conv2d(x, sd[key + ".shortcut_layer.0.weight"], stride=stride)
        sc = _bn(sd, key + ".shortcut_layer.1", sc, bn_train)
    r = _bn(sd, key + ".res_layer.0", x, bn_train)
    r = F.conv2d(r, sd[key + ".res_layer.1.weight"], stride=1, padding=1)
    r = F.prelu(r, sd[key + ".res_layer.2.weight"])
    r = F.conv2d(r, sd[key + ".res_layer.3.weight"], stride=stride, padding=1)
    r = _bn(sd, key + ".res_layer.4", r, bn_train)
    if se:
        r = se_module(sd, key + ".res_layer.5", r)
    return r + sc


def backbone_forward(sd, x, num_layers=50, se=False, bn_train=True, drop_mask=None, prefix="",
                     avg_image=None, taps=None):
    """Backbone.forward (backbone/model_irse.py:167-172) and, with ``prefix='encoder.'`` +
    ``avg_image``, pSp.forward -> BackboneEncoderDiffHead.forward
    (backbone/restyle_psp.py:439-452, :193-216).

    sd         state dict (tensors may require grad; running stats are updated in place when bn_train)
    drop_mask  None = Dropout in eval mode (identity); else a {0,1} float [B, 25088] mask in NCHW-flatten
               order, applied as x*mask/(1-p) with p = 0.5 (model_irse.py:145)
    avg_image  [3,H,W] tensor concatenated on the channel axis (restyle_psp.py:445-447)
    taps       optional dict that receives named intermediates (for block-level parity tests)
    """
    if avg_image is not None:
        x = torch.cat([x, avg_image.unsqueeze(0).expand(x.shape[0], -1, -1, -1).to(x.dtype)], dim=1)
    p = prefix
    h = F.conv2d(x, sd[p + "input_layer.0.weight"], stride=1, padding=1)
    h = _bn(sd, p + "input_layer.1", h, bn_train)
    h = F.prelu(h, sd[p + "input_layer.2.weight"])
    if taps is not None:
        taps["stem"] = h
    for i, (cin, depth, stride) in enumerate(unit_table(num_layers)):
        h = residual_unit(sd, "%sbody.%d" % (p, i), h, cin, depth, stride, se, bn_train)
        if taps is not None:
            taps["body.%d" % i] = h
    h = _bn(sd, p + "output_layer.0", h, bn_train)
    h = h.flatten(1)  # C-major: c*49 + h*7 + w  
    if drop_mask is not None:
        h = h * drop_mask * 2.0
    h = F.linear(h, sd[p + "output_layer.3.weight"], sd[p + "output_layer.3.bias"])
    h = _bn(sd, p + "output_layer.4", h, bn_train)
    return h


# ----------------------------------------------------------------------------- margin heads


def cosine_logits(x, w):
    """F.linear(F.normalize(x), F.normalize(w)), eps 1e-12 -- head/metrics.py:103 / :167."""
    return F.linear(F.normalize(x), F.normalize(w))


def arcface_constants(m):
    """head/metrics.py:90-93."""
    return math.cos(m), math.sin(m), math.cos(math.pi - m), math.sin(math.pi - m) * m


def arcface_forward(x, w, label, s=64.0, m=0.5, easy_margin=False):
    """ArcFace.forward, device_id=None branch -- head/metrics.py:97-140.

    The reference blends with a dense one-hot: out = onehot*phi + (1-onehot)*cos; this restatement
    keeps that exact arithmetic (not torch.where) so signed zeros match bit for bit.
    """
    cos_m, sin_m, th, mm = arcface_constants(m)
    cosine = cosine_logits(x, w)
    sine = torch.sqrt((1.0 - torch.pow(cosine, 2)).clamp(1e-10, 1 - 1e-10))
    phi = cosine * cos_m - sine * sin_m
    if easy_margin:
        phi = torch.where(cosine > 0, phi, cosine)
    else:
        phi = torch.where(cosine > th, phi, cosine - mm)
    one_hot = torch.zeros_like(cosine).scatter_(1, label.view(-1, 1).long(), 1)
    out = (one_hot * phi) + ((1.0 - one_hot) * cosine)
    return out * s


def cosface_forward(x, w, label, s=64.0, m=0.5):
    """CosFace.forward -- head/metrics.py:164-191 (default m = 0.50, :155)."""
    cosine = cosine_logits(x, w)
    phi = cosine - m
    one_hot = torch.zeros_like(cosine).scatter_(1, label.view(-1, 1).long(), 1)
    out = (one_hot * phi) + ((1.0 - one_hot) * cosine)
    return out * s


# ----------------------------------------------------------------------------- loss / metrics / optimiser


def focal_loss(logits, target, gamma=2):
    """FocalLoss.forward -- loss/focal.py:17-21: focal modulation of the *batch-mean* CE scalar."""
    logp = F.cross_entropy(logits, target)
    p = torch.exp(-logp)
    return ((1 - p) ** gamma * logp).mean()


def topk_accuracy(output, target, topk=(1, 5)):
    """accuracy -- util/utils.py:343-358: percentages of rows whose label is within the top-k."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def is_bn_key(name):
    """True for parameters that util/utils.py:118-139 (separate_irse_bn_paras) puts in the no-decay group.

    In the reference the split is by module class name containing 'batchnorm'; in the IR/IR-SE key layout
     those are exactly: input_layer.1, output_layer.0, output_layer.4,
    res_layer.0, res_layer.4, shortcut_layer.1.
    """
    stem = name.rsplit(".", 1)[0]
    return stem.endswith(("input_layer.1", "output_layer.0", "output_layer.4", "res_layer.0",
                          "res_layer.4", "shortcut_layer.1"))


def sgd_step(params, grads, bufs, lr, momentum, weight_decay):
    """torch.optim.SGD step (train.py:196, torch defaults: dampening 0, no nesterov) -- SURVEY App. D.

    d = g + wd*p ; buf = d (first step) else momentum*buf + d ; p -= lr*buf.  In place; bufs[i] may be None.
    """
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        d = g + weight_decay * p if weight_decay != 0 else g.clone()
        if bufs[i] is None:
            bufs[i] = d.clone()
        else:
            bufs[i].mul_(momentum).add_(d)
        p.sub_(lr * bufs[i])
    return bufs


def train_step(sd, x, label, head_w, *, num_layers=50, se=False, prefix="", avg_image=None, head="ArcFace",
               s=64.0, m=0.5, gamma=2, drop_mask=None):
    """One Stage-3 step up to the gradients -- train.py:296-315.

    Returns (features, logits, loss, grads) where grads maps every float entry of ``sd`` that requires
    grad, plus 'head.weight', to its gradient.
    """
    feats = backbone_forward(sd, x, num_layers, se, True, drop_mask, prefix, avg_image)
    if head == "ArcFace":
        logits = arcface_forward(feats, head_w, label, s, m)
    else:
        logits = cosface_forward(feats, head_w, label, s, m)
    loss = focal_loss(logits, label, gamma)
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names] + [head_w])
    grads = dict(zip(names + ["head.weight"], gs))
    return feats, logits, loss, grads
