"""Training-loop helpers with the reference's names (``from util.utils import ...``, reference train.py:11-12).

Hot-path helpers (they shape the optimizer and the step) are re-stated here and pinned by
tests/golden/g8_structure.json:
  separate_irse_bn_paras   util/utils.py:118-139   (class-name based BN / non-BN split for weight decay)
  warm_up_lr / schedule_lr util/utils.py:184-196   (linear warm-up; divide by 1.5, "temporarily hardcoded")
  accuracy                 util/utils.py:343-358   (top-k precision; HIP rank kernel on device tensors)
  _initialize_weights      util/utils.py:24-44     (kaiming-normal fan_out init used by pSp)
  AverageMeter, collate_fn_ignore_none, get_time, l2_norm

The evaluation half of the reference file (bcolz validation sets, TTA ``perform_val``, wandb ``buffer_val``) is
outside the accelerated path; the names exist so the reference's import line works, and
they degrade explicitly: no bcolz / no data -> ``get_val_data`` returns ``None`` entries, which the training loop
already treats as "skip this benchmark".
"""
import datetime
import os

import torch
import torch.nn as nn


def _initialize_weights(model):
    """He-normal (fan_out) for every Conv2d / Linear, zeros for their biases, (1, 0) for BatchNorm2d."""
    for mod in model.modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            if mod.bias is not None:
                nn.init.zeros_(mod.bias)
        elif isinstance(mod, nn.BatchNorm2d):
            nn.init.ones_(mod.weight)
            nn.init.zeros_(mod.bias)


def get_time():
    return str(datetime.datetime.now())[:-10].replace(" ", "-").replace(":", "-")


def l2_norm(input, axis=1):
    return input / torch.norm(input, 2, axis, True)


# ---------------------------------------------------------------------------------------------- param groups

_SKIP_CLASS_TOKENS = ("model", "container", "backbone")


def separate_irse_bn_paras(modules):
    """Split parameters into (batch-norm, everything else) by the *class string* of each sub-module.

    Semantics kept from the reference (hard parts): a module whose ``str(type(m))``
    contains 'model', 'container' or 'backbone' is skipped (its children are visited on their own); a module
    whose class string contains 'batchnorm' contributes all its parameters to the first list; any other
    module contributes ``m.parameters()`` (recursively!) to the second.  Leaf parameter holders must
    therefore keep torch-like class names living outside a ``backbone.*`` module path -- which is why the
    layers of this repo are plain ``torch.nn`` leaves even though their forward is fused elsewhere.
    """
    named = modules if isinstance(modules, list) else list(modules.named_modules())
    only_bn, without_bn = [], []
    for _name, layer in named:
        cls = str(layer.__class__).lower()
        if any(tok in cls for tok in _SKIP_CLASS_TOKENS):
            continue
        (only_bn if "batchnorm" in cls else without_bn).extend(layer.parameters())
    return only_bn, without_bn


def separate_resnet_bn_paras(modules):
    """torchvision-style ResNets: parameters whose *name* contains 'bn' vs the rest (util/utils.py:169-181)."""
    bn = [p for n, p in modules.named_parameters() if "bn" in n]
    ids = {id(p) for p in bn}
    return bn, [p for p in modules.parameters() if id(p) not in ids]


def warm_up_lr(batch, num_batch_warm_up, init_lr, optimizer):
    for group in optimizer.param_groups:
        group["lr"] = batch * init_lr / num_batch_warm_up


def schedule_lr(optimizer):
    for group in optimizer.param_groups:
        group["lr"] /= 1.5  # the reference's "temporarily hardcoded" decay factor (utils.py:193-194)
    print(optimizer)


# ---------------------------------------------------------------------------------------------- metrics


class AverageMeter(object):
    """Running value / sum / count / average."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """precision@k in percent, one 0-d tensor per k.

    The reference sorts with ``topk`` over all N classes; here one HIP pass counts, per row, the classes that
    score strictly above the label (its rank) and precision@k = mean(rank < k) -- identical unless two logits
    tie exactly at the k-th place.
    """
    from frhip import functional as FRF
    rank = FRF.topk_ranks(output, target)
    n = target.size(0)
    return [(rank < k).float().sum().mul_(100.0 / n) for k in topk]


def collate_fn_ignore_none(batch):
    """Drop samples the dataset failed to load (returned None) and refill by repeating survivors."""
    want = len(batch)
    batch = [b for b in batch if b is not None]
    missing = want - len(batch)
    if missing > 0:
        print("[collate] len_batch", want, "len(batch)", len(batch))
        for _ in range(missing):
            batch = batch + batch[:missing]
    return torch.utils.data.dataloader.default_collate(batch)


# ---------------------------------------------------------------------------------------------- evaluation (8f)

_VAL_SLOTS = 14  # lfw, cfp_ff, cfp_fp, agedb, calfw, cplfw, vgg2_fp and their *_issame lists


def get_val_data(data_path):
    """Returns the reference's 16-tuple.  RFW subsets are read from ``<data_path>/RFW_<eth>.npy`` +
    ``RFW_<eth>_list.npy`` (an .npy fallback for the bcolz carrays of scripts/pack_RFW_in_bcolz.py);
    anything absent is None."""
    import numpy as np
    rfw, rfw_issame = {}, {}
    for eth in ("African", "Asian", "Caucasian", "Indian"):
        arr = os.path.join(str(data_path), "RFW_%s.npy" % eth)
        lst = os.path.join(str(data_path), "RFW_%s_list.npy" % eth)
        if os.path.isfile(arr) and os.path.isfile(lst):
            rfw[eth] = np.load(arr, mmap_mode="r")
            rfw_issame[eth] = np.load(lst)
    if not rfw:
        rfw = rfw_issame = None
    return (None,) * _VAL_SLOTS + (rfw, rfw_issame)


def perform_val(multi_gpu, device, embedding_size, batch_size, backbone, carray, issame, nrof_folds=10, tta=True,
                dset_name="", ccrop=True):
    raise NotImplementedError("perform_val (flip-TTA verification, reference util/utils.py:254-307) is the next "
                              "scope row (SURVEY.md 8f rank 2) and is not part of this build")


def buffer_val(writer, db_name, acc, best_threshold, roc_curve_tensor, epoch, n_samples_passed=None):
    stats = {"{}_Accuracy".format(db_name): acc, "{}_Best_Threshold".format(db_name): best_threshold,
             "epoch": epoch}
    if n_samples_passed is not None:
        stats["step"] = n_samples_passed
    writer.log(stats)
