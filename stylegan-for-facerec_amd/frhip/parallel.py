"""Data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI, overlapped with backward.

The reference is single-process ``nn.DataParallel`` (train.py:219-222): every step it re-broadcasts all parameters,
scatters the *global* batch, and reduce-adds gradients to GPU 0.  Here every rank owns a full replica and its own
batch (weak scaling: BATCH_SIZE is per GPU), BatchNorm statistics stay per rank (as they are per replica under
DataParallel), and the only exchange is the gradient average:

  * the engine writes gradients into one flat fp32 arena ordered by *readiness* (output layer first, stem last),
    so a bucket is a contiguous arena slice -- no gather/scatter copies around the collective;
  * a bucket's all-reduce is enqueued (``async_op=True``: RCCL runs it on its own stream beside the remaining backward
    kernels) once backward has produced every gradient of the bucket AND the gate is open: the 7x7 / 14x14 layers launch
    exactly one workgroup per compute unit, and a collective kernel holding even four CUs makes every such launch take two
    rounds (+3.3 ms per step if it stays for the whole backward pass: tools/cu_hog.py, profiles/r05_hog_matrix.txt).  The
    plan names the point of the backward pass behind which every launch has thousands of workgroups (the 28x28 layers
    onward, ``plan.comm_gate``); complete buckets wait until then and are enqueued together -- 3 ms of backward are left at
    that point for ~1 ms of exchange.  FRHIP_DP_OVERLAP=1 enqueues every bucket as soon as it is complete, =0 only in
    ``synchronize()``;
  * the head weight (N x 512, the largest single gradient, ready first) is announced from its own autograd hook and
    waits for the same gate;
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
    def __init__(self, arena, slices, group=None, bucket_bytes=32 << 20, gate=0):
        """arena: flat tensor; slices: [(param, offset, numel)] in readiness order, offsets increasing.  gate: number of
        announced parameters (arena order) from which on collectives may be enqueued (0: from the start)."""
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = dist.is_initialized() and os.environ.get("FRHIP_FORCE_DP", "0") == "1"  # 1-rank self test
        self.avg_native = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # FRHIP_DP_OVERLAP (A/B switch for multi-GPU runs): 2 (default) overlapped with the backward pass, but only behind the
        # plan's gate (module docstring); 1 from the first complete bucket; 0 no collective during the backward pass --
        # everything is enqueued by synchronize().  Emulated on one GPU (tools/cu_hog.py --comm, profiles/r05_comm_policy.txt):
        # 14.45 / 14.73 / 15.43 ms per step for 2 / 1 / 0 with 1.26 ms of collectives per step; no multi-GPU box so far.
        self.policy = int(os.environ.get("FRHIP_DP_OVERLAP", "2") or 2)
        self.overlap = self.policy != 0
        self.gate = gate if self.policy == 2 else 0
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
        self.announced = 0
        self.gate_open = self.overlap and self.gate <= 0
        self.held = []  # stand-alone tensors announced before the gate opened

    def _launch(self, t):
        if self.world == 1 and not self.force:
            return
        if self.avg_native:
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t))

    def on_ready(self, params, fence=None):
        """Called by the backward pass with parameters whose gradients are final.  Frozen parameters never show
        up; their arena slots stay zero and are reduced along with their bucket.  fence: called once before anything is
        enqueued; returns the stream to enqueue on (ordered behind the announced gradients) or None for the current one."""
        for p in params:
            b = self.owner.get(id(p))
            if b is not None:
                self.pending[b] -= 1
                self.announced += 1
        if self.overlap and not self.gate_open and self.announced >= self.gate:
            self.gate_open = True
        if not self.gate_open:
            return
        # buckets complete in order because the arena is in readiness order; a bucket whose remaining members
        # are all frozen is flushed by synchronize()
        out, self.held = self.held, []
        while self.next_bucket < len(self.buckets) and self.pending[self.next_bucket] <= 0:
            s, e, _ = self.buckets[self.next_bucket]
            out.append(self.arena[s:e])
            self.next_bucket += 1
        self._enqueue(out, fence)

    def _enqueue(self, tensors, fence):
        if not tensors:
            return
        stream = fence() if fence is not None else None
        if stream is None:
            for t in tensors:
                self._launch(t)
        else:
            with torch.cuda.stream(stream):
                for t in tensors:
                    self._launch(t)

    def open_gate(self, fence=None):
        self.gate_open = True
        out, self.held = self.held, []
        self._enqueue(out, fence)

    def add_tensor(self, t):
        """Reduce a stand-alone gradient (the margin head's weight): now if the gate is open, else when it opens."""
        if self.gate_open:
            self._launch(t)
        else:
            self.held.append(t)

    def synchronize(self):
        """Enqueue whatever is left (buckets holding frozen parameters), then wait for every collective."""
        for t in self.held:
            self._launch(t)
        self.held = []
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
        # the head's gradients: no arena, opened together with the backbone reducer's gate (_on_ready)
        self.extra = BucketedAllReduce(torch.zeros(0), [], group, gate=1) if dist.is_initialized() else None
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
            self.reducer = BucketedAllReduce(plan.arena, plan.arena_slices, self.group, self.bucket_bytes,
                                             gate=getattr(plan, "comm_gate", 0))
        self.reducer.on_ready(params, plan.comm_fence)
        if self.extra is not None and self.reducer.gate_open and not self.extra.gate_open and self.extra.overlap:
            self.extra.open_gate(plan.comm_fence)

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
