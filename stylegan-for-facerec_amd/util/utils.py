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
    if n == 0 or len(topk) > 4 or not rank.is_cuda:
        return [(rank < k).float().sum().mul_(100.0 / n) for k in topk]
    return list(FRF.topk_precision(rank, topk).unbind(0))  # one launch: count(rank < k) * float32(100 / n) per k


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


def de_preprocess(tensor):
    return tensor * 0.5 + 0.5


def _to_uint8(batch):
    """torchvision ToPILImage on a float CHW tensor: ``pic.mul(255).byte()`` (truncation) -- utils.py:204-228."""
    return de_preprocess(batch).mul(255).to(torch.uint8)


_U8_LUT = {}


def _from_uint8(u8):
    """ToTensor + Normalize(0.5, 0.5): byte / 255, then (v - 0.5) / 0.5, in float32 on the host.  Device tensors go
    through the 256-entry table of those host results (the device's float division is not correctly rounded: 1 ulp
    off for a third of the byte values), so both paths give the same bits."""
    if not u8.is_cuda:
        return u8.to(torch.float32).div(255).sub_(0.5).div_(0.5)
    lut = _U8_LUT.get(u8.device)
    if lut is None:
        lut = _U8_LUT[u8.device] = torch.arange(256, dtype=torch.float32).div(255).sub_(0.5).div_(0.5).to(u8.device)
    return lut[u8.long()]


def hflip_batch(imgs_tensor):
    """Horizontal flip through the reference's uint8 round trip (utils.py:204-218): the flipped copy is quantised
    to 1/255 steps exactly as ToPILImage / ToTensor do, without leaving torch (works on device tensors)."""
    return _from_uint8(torch.flip(_to_uint8(imgs_tensor), dims=[-1]))


_CCROP_GPU = {}


def ccrop_batch(imgs_tensor):
    """Resize([128, 128]) + CenterCrop([112, 112]) through the reference's uint8 round trip (utils.py:221-236).

    Device tensors: ONE launch of the input-transform kernel for the whole batch (frhip/input_pipeline.py; Pillow's 8-bit
    bilinear resample restated bit-exactly, crop offset ``int(round((128 - 112) / 2.))`` = 8, no flip).  Host tensors: the
    PIL calls torchvision makes (``Image.resize(size, BILINEAR)``, centre crop box), per image, like the reference."""
    top = int(round((128 - 112) / 2.0))
    if imgs_tensor.is_cuda:
        from frhip.input_pipeline import GpuTrainTransform
        tf = _CCROP_GPU.get("tf")
        if tf is None:
            tf = _CCROP_GPU["tf"] = GpuTrainTransform(112, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        n = imgs_tensor.shape[0]
        u8 = _to_uint8(imgs_tensor.detach()).permute(0, 2, 3, 1).contiguous()
        key = (imgs_tensor.device, n)
        if key not in _CCROP_GPU:  # constant offsets, kept on the device: no per-batch host-to-device copy / sync
            _CCROP_GPU[key] = (torch.full((n, 2), top, dtype=torch.int32, device=imgs_tensor.device),
                               torch.zeros(n, dtype=torch.uint8, device=imgs_tensor.device))
        crop, flip = _CCROP_GPU[key]
        return tf(u8, crop, flip, validate=False)
    import numpy as np
    from PIL import Image
    u8 = _to_uint8(imgs_tensor.detach().cpu()).permute(0, 2, 3, 1).contiguous().numpy()
    out = torch.empty(u8.shape[0], u8.shape[3], 112, 112, dtype=torch.float32)
    for i in range(u8.shape[0]):
        img = Image.fromarray(u8[i]).resize((128, 128), Image.BILINEAR)
        img = img.crop((top, top, top + 112, top + 112))
        out[i] = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1)
    return _from_uint8(out.to(torch.uint8))


def gen_plot(fpr, tpr):
    """ROC curve as a JPEG in memory (utils.py:239-251); None when matplotlib is unavailable."""
    try:
        import io
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001
        return None
    plt.figure()
    plt.xlabel("FPR", fontsize=14)
    plt.ylabel("TPR", fontsize=14)
    plt.title("ROC Curve", fontsize=14)
    plt.plot(fpr, tpr, linewidth=2)
    buf = io.BytesIO()
    plt.savefig(buf, format="jpeg")
    buf.seek(0)
    plt.close()
    return buf


def perform_val(multi_gpu, device, embedding_size, batch_size, backbone, carray, issame, nrof_folds=10, tta=True,
                dset_name=None, ccrop=True, rank=0, world=1, group=None):
    """Flip-TTA verification of ``backbone`` on interleaved image pairs (reference util/utils.py:254-307; SURVEY 8f
    rank 2).  Same protocol: eval mode, batches of ``batch_size`` plus the remainder, optional centre-crop, embedding =
    f(img) + f(hflip(img)) summed on the host, ``l2_norm``, then the k-fold metrics of util/verification.py.  Returns
    (mean accuracy, mean best threshold, ROC-curve image tensor or None).  ``carray``: [2P, 3, 112, 112] (or NHWC)
    float array in [-1, 1] (bcolz carray or numpy).

    ``world > 1`` (one process per GPU; collective -- every rank calls it): batch k of the SAME batch partition goes to
    rank k % world, the embedding sums are combined with one all-reduce (every row is written by exactly one rank and is
    zero elsewhere, so the sum is exact) and every rank computes the same metrics.  Data-parallel training keeps the
    BatchNorm running statistics per rank (as nn.DataParallel keeps them per replica and the reference evaluates replica 0,
    train.py:219-222), so the ranks hold slightly different eval models: for the evaluation every rank takes rank 0's
    buffers (one broadcast per buffer) and gets its own back afterwards.  The model evaluated is then the one rank 0 saves
    as the checkpoint, the batches are the ones a single rank would run, and the result equals rank 0 evaluating alone bit
    for bit."""
    import numpy as np
    from util.verification import evaluate
    if multi_gpu:
        backbone = backbone.module
    backbone = backbone.to(device)
    backbone.eval()
    own_buffers = None
    if world > 1:
        import torch.distributed as dist
        own_buffers = [b.detach().clone() for b in backbone.buffers()]
        # rank 0's buffers in ONE broadcast per dtype (a flat copy; 150+ small collectives per validation set otherwise)
        by_dtype = {}
        for b in backbone.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, src=0, group=group)
            off = 0
            with torch.no_grad():
                for b in bufs:
                    b.data.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
    try:
        n = len(carray)
        is_dev = torch.device(device).type == "cuda"
        sums = torch.zeros(n, embedding_size, device=device, dtype=torch.float32)  # f(img) [+ f(hflip(img))], on the device
        # two pinned staging buffers: the host prepares batch i+1 while the GPU works on batch i, and nothing in the loop waits
        # for the device (the reference copies every batch's embeddings back before it reads the next batch)
        stage, free = [None, None], [None, None]
        with torch.no_grad():
            for k, idx in enumerate(range(rank * batch_size, n, batch_size * world)):
                host = torch.from_numpy(np.ascontiguousarray(carray[idx:idx + batch_size], dtype=np.float32))
                if host.shape[-1] == 3:
                    host = host.permute(0, 3, 1, 2)
                if is_dev:
                    slot = k & 1
                    if stage[slot] is None or stage[slot].shape != host.shape:
                        stage[slot] = torch.empty(host.shape, dtype=torch.float32).pin_memory()
                    if free[slot] is not None:
                        free[slot].synchronize()  # the copy that last read this buffer is done
                    stage[slot].copy_(host)
                    batch = stage[slot].to(device, non_blocking=True)
                    free[slot] = torch.cuda.Event()
                    free[slot].record()
                else:
                    batch = host.contiguous().to(device)
                cropped = ccrop_batch(batch) if ccrop else batch  # crop and flip run on the device (one launch each)
                emb = backbone(cropped)
                if tta:
                    emb = emb + backbone(hflip_batch(cropped))  # fp32 add: same bits as on the host
                sums[idx:idx + batch.shape[0]] = emb
            if world > 1:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    finally:
        if own_buffers is not None:  # training continues with this rank's own running statistics, also after an exception
            with torch.no_grad():
                for b, mine in zip(backbone.buffers(), own_buffers):
                    b.data.copy_(mine)
    embeddings = np.zeros([n, embedding_size])
    embeddings[:] = l2_norm(sums.cpu()).numpy()  # one copy back; normalisation on the host, as in the reference
    tpr, fpr, acc, best_thresholds = evaluate(embeddings, issame, nrof_folds)
    roc = None
    buf = gen_plot(fpr, tpr)
    if buf is not None:
        from PIL import Image
        img = np.asarray(Image.open(buf).convert("RGB"), dtype=np.float32) / 255.0
        roc = torch.from_numpy(img).permute(2, 0, 1).contiguous()
    return acc.mean(), best_thresholds.mean(), roc


def buffer_val(writer, db_name, acc, best_threshold, roc_curve_tensor, epoch, n_samples_passed=None):
    stats = {"{}_Accuracy".format(db_name): acc, "{}_Best_Threshold".format(db_name): best_threshold,
             "epoch": epoch}
    if n_samples_passed is not None:
        stats["step"] = n_samples_passed
    writer.log(stats)
