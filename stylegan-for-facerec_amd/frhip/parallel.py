"""Data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI, overlapped with backward.

The reference is single-process ``nn.DataParallel`` (train.py:219-222): every step it re-broadcasts all parameters,
scatters the *global* batch, and reduce-adds gradients to GPU 0.  Here every rank owns a full replica and its own
batch (weak scaling: BATCH_SIZE is per GPU), BatchNorm statistics stay per rank (as they are per replica under
DataParallel), and the only exchange is the gradient average:

  * the engine writes gradients into one flat fp32 arena ordered by *readiness* (output layer first, stem last),
    so a bucket is a contiguous arena slice -- no gather/scatter copies around the collective;
  * as soon as backward has produced every gradient of a bucket the bucket's all-reduce is enqueued
    (``async_op=True``: RCCL runs it on its own stream while the remaining backward kernels keep the CUs busy);
  * the head weight (N x 512, the largest single gradient, ready first) is reduced from its own autograd hook;
  * ``synchronize()`` before the optimizer step makes the compute stream wait for the outstanding collectives.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): IR-50's 174 MB of fp32 gradients need ~2 ms even on a single
ring, far below the backward time, so bucket size (32 MB default) is chosen for launch overhead, not bandwidth.

The class is engine-agnostic (any flat arena + readiness callbacks), which is what the world_size-2 ``gloo`` tests
on CPU exercise.
"""
import os

import torch
import torch.distributed as dist


class BucketedAllReduce(object):
    def __init__(self, arena, slices, group=None, bucket_bytes=32 << 20):
        """arena: flat tensor; slices: [(param, offset, numel)] in readiness order, offsets increasing."""
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = dist.is_initialized() and os.environ.get("FRHIP_FORCE_DP", "0") == "1"  # 1-rank self test
        self.avg_native = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # FRHIP_DP_OVERLAP=0 (A/B switch for multi-GPU runs): no collective during the backward pass -- everything is enqueued
        # by synchronize().  RCCL's kernels need CUs of their own; while one is resident a 256-workgroup strip launch takes two
        # rounds, so on some fabrics the un-overlapped exchange may be the faster one.  Unmeasured (no multi-GPU box so far).
        self.overlap = os.environ.get("FRHIP_DP_OVERLAP", "1") != "0"
        self.buckets = []  # (start, end, [param ids])
        start, ids, nbytes = 0, [], 0
        esz = arena.element_size()
        for i, (p, off, n) in enumerate(slices):
            ids.append(id(p))
            nbytes += n * esz
            last = i + 1 == len(slices)
            if nbytes >= bucket_bytes or last:
                end = arena.numel() if last else slices[i + 1][1]
                self.buckets.append((start, end, ids))
                start, ids, nbytes = end, [], 0
        self.owner = {}
        for b, (_s, _e, ids) in enumerate(self.buckets):
            for pid in ids:
                self.owner[pid] = b
        self.reset()

    def reset(self):
        self.pending = [len(ids) for (_s, _e, ids) in self.buckets]
        self.next_bucket = 0
        self.works = []

    def _launch(self, t):
        if self.world == 1 and not self.force:
            return
        if self.avg_native:
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t))

    def on_ready(self, params):
        """Called by the backward pass with parameters whose gradients are final.  Frozen parameters never show
        up; their arena slots stay zero and are reduced along with their bucket."""
        for p in params:
            b = self.owner.get(id(p))
            if b is not None:
                self.pending[b] -= 1
        # buckets complete in order because the arena is in readiness order; a bucket whose remaining members
        # are all frozen is flushed by flush_frozen()
        while self.overlap and self.next_bucket < len(self.buckets) and self.pending[self.next_bucket] <= 0:
            s, e, _ = self.buckets[self.next_bucket]
            self._launch(self.arena[s:e])
            self.next_bucket += 1

    def add_tensor(self, t):
        """Reduce a stand-alone gradient (the margin head's weight) right now."""
        self._launch(t)

    def synchronize(self):
        """Enqueue whatever is left (buckets holding frozen parameters), then wait for every collective."""
        while self.next_bucket < len(self.buckets):
            s, e, _ = self.buckets[self.next_bucket]
            self._launch(self.arena[s:e])
            self.next_bucket += 1
        for work, post in self.works:
            work.wait()
            if post is not None:
                post.div_(self.world)
        self.reset()


def _dense_flat(t):
    """1-D view of a dense tensor's memory in storage order.  The conv weights are channels-last behind an OIHW shape
    (not ``is_contiguous()``); collectives want contiguous tensors, and every rank has the same layout, so the flat
    memory is what gets exchanged."""
    if t.is_contiguous():
        return t.view(-1)
    sizes_strides = sorted((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1)
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            raise ValueError("frhip.parallel: cannot broadcast a non-dense tensor (shape %s, strides %s)"
                             % (tuple(t.shape), t.stride()))
        expect *= sz
    return torch.as_strided(t, (t.numel(),), (1,), t.storage_offset())


class DataParallel(object):
    """Glue between a frhip backbone (+ head) and ``BucketedAllReduce``.

        dp = DataParallel(backbone, head)         # after dist.init_process_group("nccl")
        loss.backward(); dp.synchronize(); optimizer.step()

    ``backbone`` is a model_irse.Backbone or a restyle_psp.pSp (the reference's ``.module`` indirection is not
    needed; ``dp.module`` returns the backbone for code written against nn.DataParallel, train.py:266-274,415).
    """

    def __init__(self, backbone, head=None, group=None, bucket_bytes=32 << 20):
        self.module, self.head, self.group, self.bucket_bytes = backbone, head, group, bucket_bytes
        inner = backbone.encoder if hasattr(backbone, "encoder") else backbone
        self.runner = inner._runner[0]
        self.runner.on_grads_ready = self._on_ready
        self.reducer, self.plan = None, None
        self.extra = BucketedAllReduce(torch.zeros(0), [], group) if dist.is_initialized() else None
        if head is not None:
            for p in head.parameters():
                p.register_post_accumulate_grad_hook(self._head_hook)
        self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        mods = [self.module] + ([self.head] if self.head is not None else [])
        for m in mods:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(_dense_flat(t.data), src, group=self.group)

    def _on_ready(self, params):
        plan = self.runner.plan
        if plan is not self.plan:
            self.plan = plan
            self.reducer = BucketedAllReduce(plan.arena, plan.arena_slices, self.group, self.bucket_bytes)
        self.reducer.on_ready(params)

    def _head_hook(self, p):
        if self.extra is not None:
            self.extra.add_tensor(p.grad)

    def synchronize(self):
        if self.extra is not None:
            self.extra.synchronize()
        if self.reducer is not None:
            self.reducer.synchronize()

    def __call__(self, x):
        return self.module(x)
