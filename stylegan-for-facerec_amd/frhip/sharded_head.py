"""Class-sharded margin head + focal loss: one process per GPU, every rank owns a contiguous range of the classes.

SURVEY.md 8(f) rank 1.  The reference's way to spread a large head is the class-dimension split of
head/metrics.py:104-113 (ArcFace) / :170-179 (CosFace): ONE process chunks ``weight`` over ``device_id`` GPUs, copies
the features to each of them and concatenates the logit chunks back on GPU 0.  The process-per-GPU form of the same
math, with nothing of size [B, N] ever crossing xGMI:

  forward   all-gather features [B,512] and labels -> every rank has the global batch (Bg = world * B rows)
            local logits [Bg, N/world] = margin head on the rank's classes (same HIP kernels as the replicated head;
            rows whose label lives elsewhere select nothing)
            per-row (max, sum exp, label logit) -> ONE all-gather of [3, Bg] floats -> combined in rank order
            -> log-sum-exp, cross entropy; top-k rank of the label = sum over ranks of "local logits above it"
            focal loss on the GLOBAL batch mean, as the reference computes it (one head over the whole DataParallel batch)
  backward  d logits locally (softmax with the global lse) -> weight-shard gradient complete on its owner (NO all-reduce
            of the N x 512 gradient: 57 MB at N = 28 000) and feature gradients summed by reduce-scatter ([Bg,512] in,
            the rank's own [B,512] out) before the normalisation backward.

Against the replicated head this removes the largest gradient bucket from the all-reduce, 1 - 1/world of the head's
weight / momentum / logit memory and FLOPs per GPU, and makes the loss the global-batch focal loss (the replicated
data-parallel step weights each rank's batch-mean separately).

The device arithmetic sits behind ``HipKernels`` (the only implementation in the package: HIP through the C ABI; no
CPU fallback).  tests/ substitutes an oracle-backed stand-in to run the collective choreography at world_size 2 on
``gloo``.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn import Parameter

from . import functional as FRF
from . import ops

KINDS = {"ArcFace": 0, "CosFace": 1}


def class_range(num_classes, world, rank):
    """[lo, hi) of the classes rank ``rank`` owns: contiguous, the first ``num_classes % world`` ranks hold one more."""
    if not 0 <= rank < world:
        raise ValueError("class_range: rank %d outside world of %d" % (rank, world))
    if num_classes < world:
        raise ValueError("class_range: %d classes cannot be split over %d ranks" % (num_classes, world))
    base, rem = divmod(num_classes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def localize_labels(label_all, lo, hi):
    """label - lo where this rank owns the label, -1 elsewhere."""
    own = (label_all >= lo) & (label_all < hi)
    return torch.where(own, label_all - lo, torch.full_like(label_all, -1))


class Comm(object):
    """The four exchanges of the sharded head; identity when there is one rank."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.native = self.on and dist.get_backend(group) == "nccl"  # RCCL: reduce-scatter available

    def all_gather(self, t):
        """[n, ...] per rank -> [world * n, ...] in rank order."""
        if not self.on:
            return t
        t = t.contiguous()
        out = t.new_empty((self.world * t.shape[0],) + tuple(t.shape[1:]))
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def reduce_scatter_rows(self, t):
        """[world * n, D] per rank -> sum over ranks of this rank's [n, D] row block."""
        if not self.on:
            return t
        n = t.shape[0] // self.world
        if self.native:
            out = t.new_empty((n,) + tuple(t.shape[1:]))
            dist.reduce_scatter_tensor(out, t.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
            return out
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)  # gloo has no reduce-scatter
        return t[self.rank * n:(self.rank + 1) * n].contiguous()

    def all_reduce_sum(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_ragged_rows(self, t, sizes):
        """Row blocks of different heights (the weight shards) -> the stacked tensor on every rank."""
        if not self.on:
            return t
        top = max(sizes)
        pad = t.new_zeros((top,) + tuple(t.shape[1:]))
        pad[:t.shape[0]] = t
        out = self.all_gather(pad).view((self.world, top) + tuple(t.shape[1:]))
        return torch.cat([out[r, :sizes[r]] for r in range(self.world)], 0)


class HipKernels(object):
    """Device arithmetic of the sharded head on the HIP kernels (current stream).  Host tensors raise in ``ops``."""

    def logits(self, x_all, w, label_local, kind, s, m, easy_margin):
        return FRF.margin_forward(x_all, w, label_local, kind, s, m, easy_margin)

    def row_stats(self, logits, label_local):
        rows, n = logits.shape
        stats = torch.empty(3, rows, device=logits.device)
        ops.call("fr_shard_row_stats", logits, label_local, stats, rows, n, logits.stride(0),
                 ops.current_stream_ptr())()
        return stats

    def combine(self, stats_all, world, rows):
        dev = stats_all.device
        lse, ce, tl = (torch.empty(rows, device=dev) for _ in range(3))
        ops.call("fr_shard_combine", stats_all, world, rows, lse, ce, tl, ops.current_stream_ptr())()
        return lse, ce, tl

    def shard_rank(self, logits, tlogit):
        rows, n = logits.shape
        rank = torch.empty(rows, device=logits.device, dtype=torch.int32)
        ops.call("fr_shard_rank_rows", logits, tlogit, rank, rows, n, logits.stride(0), ops.current_stream_ptr())()
        return rank

    def focal(self, ce, rank, gamma):
        scalars = torch.empty(8, device=ce.device)
        ops.call("fr_focal_finalize", ce, rank, ce.shape[0], float(gamma), scalars, ops.current_stream_ptr())()
        return scalars

    def dlogits(self, logits, label_local, lse, scalars, gup):
        rows, n = logits.shape
        ld = logits.stride(0)  # the logits keep a 16-byte row pitch; the gradient uses the same one
        store = torch.empty(rows, ld, device=logits.device)
        gup = gup.contiguous().float().reshape(1)
        ops.call("fr_focal_bwd", logits, label_local, lse, scalars, gup, store, rows, n, ld,
                 ops.current_stream_ptr())()
        return store if ld == n else store[:, :n]

    def head_bwd(self, saved, cfg, g, need_x, need_w):
        """(G = d loss / d normalize(x_all) from this rank's classes, gradient of the weight shard)."""
        return FRF.margin_backward(saved, cfg, g, need_x, need_w, raw_x_grad=True)

    def normalize_bwd(self, G, x, inv_x):
        gx = torch.empty_like(G)
        ops.call("fr_normalize_bwd", G, x, inv_x, gx, G.shape[0], G.shape[1], ops.current_stream_ptr())()
        return gx


class ShardedHeadLossFn(torch.autograd.Function):
    """(loss, prec@1, prec@5) of the global batch from this rank's features, labels and weight shard."""

    @staticmethod
    def forward(ctx, x, w, label, head):
        K, C = head.kernels, head.comm
        B = x.shape[0]
        x_loc = x.detach().contiguous().float()
        x_all = C.all_gather(x_loc)
        lab_all = C.all_gather(label.contiguous().long())
        rows = x_all.shape[0]
        lab_loc = localize_labels(lab_all, head.lo, head.hi)
        logits, saved, cfg = K.logits(x_all, w.detach(), lab_loc, head.kind, head.s, head.m, head.easy_margin)
        stats_all = C.all_gather(K.row_stats(logits, lab_loc).view(1, 3, rows))
        lse, ce, tlogit = K.combine(stats_all, C.world, rows)
        rank = C.all_reduce_sum(K.shard_rank(logits, tlogit))
        scalars = K.focal(ce, rank, head.gamma)
        ctx.head, ctx.saved, ctx.cfg = head, saved, cfg
        ctx.rows_of_rank = (C.rank * B, (C.rank + 1) * B)
        ctx.save_for_backward(logits, lab_loc, lse, scalars, x_loc)
        loss, prec1, prec5 = scalars[0].clone(), scalars[2].clone(), scalars[3].clone()
        ctx.mark_non_differentiable(prec1, prec5)
        return loss, prec1, prec5

    @staticmethod
    def backward(ctx, gloss, _g1, _g5):
        head = ctx.head
        K, C = head.kernels, head.comm
        logits, lab_loc, lse, scalars, x_loc = ctx.saved_tensors
        grad = K.dlogits(logits, lab_loc, lse, scalars, gloss)
        G_all, gw = K.head_bwd(ctx.saved, ctx.cfg, grad, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        gx = None
        if G_all is not None:
            G = C.reduce_scatter_rows(G_all)
            lo, hi = ctx.rows_of_rank
            inv_x = ctx.saved[5][lo:hi].contiguous()
            gx = K.normalize_bwd(G, x_loc, inv_x)
            if head.grad_scale != 1.0:
                gx = gx * head.grad_scale
        return gx, gw, None, None


class ShardedMarginLoss(nn.Module):
    """ArcFace / CosFace + FocalLoss + accuracy with the class dimension sharded over the ranks of ``group``.

        crit = ShardedMarginLoss(512, 28000, "ArcFace", s=64.0, m=0.5, gamma=2)      # after init_process_group
        loss, prec1, prec5 = crit(features, labels)                                   # per-rank features / labels
        loss.backward()                        # crit.weight.grad is final (no all-reduce), features get their gradient

    ``weight`` is this rank's [hi - lo, in_features] slice of the head weight: the slice of the tensor a replicated
    ``ArcFace`` would have drawn from the same RNG state (``full_weight`` given: of that tensor), so the two are
    interchangeable; ``gather_weight()`` returns the reference's ``weight`` ([N, 512], checkpoint key ``weight``).

    The loss is the focal loss of the GLOBAL batch (head/metrics.py + loss/focal.py applied to the concatenated batch,
    what the reference's single head sees under nn.DataParallel).  ``average_over_ranks=True`` (default) scales the
    feature gradient by the world size so that a backbone whose gradients are AVERAGED over ranks
    (frhip.parallel.DataParallel) ends up with the gradient of that global loss.
    """

    def __init__(self, in_features, out_features, head="ArcFace", s=64.0, m=0.50, easy_margin=False, gamma=2.0,
                 group=None, full_weight=None, average_over_ranks=True, kernels=None):
        super().__init__()
        if head not in KINDS:
            raise ValueError("ShardedMarginLoss: head must be 'ArcFace' or 'CosFace', got %r" % (head,))
        self.in_features, self.out_features = in_features, out_features
        self.head_name, self.kind = head, KINDS[head]
        self.s, self.m, self.easy_margin, self.gamma = float(s), float(m), bool(easy_margin), float(gamma)
        self.comm = Comm(group)
        self.kernels = kernels if kernels is not None else HipKernels()
        self.lo, self.hi = class_range(out_features, self.comm.world, self.comm.rank)
        self.grad_scale = float(self.comm.world) if average_over_ranks else 1.0
        if full_weight is None:
            full_weight = torch.empty(out_features, in_features)
            nn.init.xavier_uniform_(full_weight)  # head/metrics.py:87-88: same draw as the replicated head
        if tuple(full_weight.shape) != (out_features, in_features):
            raise ValueError("ShardedMarginLoss: full_weight must be [%d, %d]" % (out_features, in_features))
        self.weight = Parameter(full_weight.detach()[self.lo:self.hi].clone().float())

    @classmethod
    def from_head(cls, head, gamma=2.0, group=None, **kw):
        """Shard an existing ``head.metrics.ArcFace`` / ``CosFace`` (same weights on every rank)."""
        name = head.__class__.__name__
        return cls(head.in_features, head.out_features, name, s=head.s, m=head.m,
                   easy_margin=getattr(head, "easy_margin", False), gamma=gamma, group=group,
                   full_weight=head.weight.detach().cpu(), **kw)

    def shard_sizes(self):
        return [hi - lo for lo, hi in (class_range(self.out_features, self.comm.world, r)
                                       for r in range(self.comm.world))]

    def gather_weight(self):
        """The full [out_features, in_features] weight on every rank (checkpoints keep the reference's layout)."""
        return self.comm.gather_ragged_rows(self.weight.detach(), self.shard_sizes())

    def load_full_weight(self, full_weight):
        with torch.no_grad():
            self.weight.copy_(full_weight[self.lo:self.hi])

    def forward(self, input, label):
        if self.weight.device != input.device:
            self.to(input.device)
        label = label.to(input.device)
        if FRF.CHECK_LABELS and (label.min() < 0 or label.max() >= self.out_features):  # metrics.py:134 (scatter_)
            raise RuntimeError("index %d is out of bounds for dimension 1 with size %d"
                               % (int(label.max()), self.out_features))
        return ShardedHeadLossFn.apply(input, self.weight, label, self)

    def __repr__(self):
        return "ShardedMarginLoss(%s, in_features = %d, classes [%d, %d) of %d, s = %s, m = %s, gamma = %s)" % (
            self.head_name, self.in_features, self.lo, self.hi, self.out_features, self.s, self.m, self.gamma)
