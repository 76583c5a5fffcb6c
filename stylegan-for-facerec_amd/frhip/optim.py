"""SGD with momentum as ONE multi-tensor HIP launch per step (reference: optim.SGD at train.py:196).

Same constructor shape as ``torch.optim.SGD`` (param groups with per-group ``weight_decay`` / ``lr``), same
update:  d = g + wd*p ; buf = momentum*buf + d ; p -= lr*buf, with buf starting at zero so
the first step yields buf = d.  Parameters whose ``.grad`` is None (frozen body, train.py:263-268) are
skipped entirely.  ``param_groups`` stay plain dicts so warm_up_lr / schedule_lr work unchanged.
"""
import ctypes

import torch

from . import _lib, ops


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False, *,
                 maximize=False):
        # the defaults carry every key torch.optim.SGD reads in step(): an Optimizer_*.pth written here loads into
        # the reference's torch.optim.SGD (train.py:196, :227-230) and steps there, and the other way round
        if dampening != 0 or nesterov or maximize:
            raise NotImplementedError("frhip.optim.SGD implements the reference's update only (train.py:196: dampening 0, "
                                      "no nesterov, minimise)")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov, maximize=maximize, foreach=None, differentiable=False,
                                      fused=None))
        self._sig = None
        self._tables = []  # per group: (table_dev, chunks_dev, nchunks)

    def zero_grad(self, set_to_none=False):
        """Keeps gradient buffers alive (the engine writes into them).  Gradients that are views of an engine
        gradient arena are cleared with ONE fill of the arena instead of one launch per parameter.
        (Round 6, measured and taken back: dropping the gradients that are not arena views -- the head's -- so that autograd
        installs the new tensor instead of adding it onto zeros saves a fill and an add of N x 512 floats per step, 14 us at
        7000 classes; but the optimizer's launch tables are keyed on the gradient addresses, and a step whose allocator hands
        out another block rebuilds them: not worth the exposure.)"""
        arenas = {}
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if set_to_none:
                    p.grad = None
                elif g._is_view() and getattr(g._base, "_frhip_grad_arena", False):
                    arenas[id(g._base)] = g._base
                else:
                    g.zero_()
        for a in arenas.values():
            a.zero_()
            a._frhip_zeroed = True  # the engine's next backward skips its own fill (engine.run_backward)

    def _build(self):
        for group in self.param_groups:  # a loaded state dict may carry settings this kernel does not implement
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise NotImplementedError("frhip.optim.SGD: dampening / nesterov / maximize are not implemented")
        chunk = _lib.lib.fr_sgd_chunk_elems()
        self._tables = []
        keep = []
        for group in self.param_groups:
            recs, chunks = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise _lib.FrhipError("frhip.optim.SGD: parameter on %s -- the HIP optimizer needs ROCm tensors"
                                          % p.device)
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                buf, g = st["momentum_buffer"], p.grad
                if buf is None:  # torch.optim.SGD writes None before its first step
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if buf.shape == p.shape and not _dense_same(p, buf):
                    # a buffer loaded from a checkpoint written with another memory layout (the reference's
                    # torch.optim.SGD keeps OIHW-contiguous buffers; the conv weights here are channels-last)
                    relaid = torch.empty_like(p, memory_format=torch.preserve_format)
                    relaid.copy_(buf)
                    buf = st["momentum_buffer"] = relaid
                if not (_dense_same(p, g) and _dense_same(p, buf)):
                    raise _lib.FrhipError("frhip.optim.SGD: param / grad / momentum buffer must share one dense layout")
                n = p.numel()
                t = len(recs)
                recs.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, float(group["weight_decay"])))
                chunks.extend((t, c) for c in range((n + chunk - 1) // chunk))
            if not recs:
                self._tables.append(None)
                continue
            arr = (_lib.FrSgdTensor * len(recs))()
            for i, (pp, gp, bp, n, wd) in enumerate(recs):
                arr[i].p, arr[i].g, arr[i].buf, arr[i].n, arr[i].wd = pp, gp, bp, n, wd
            dev = group["params"][0].device
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            ch = torch.tensor(chunks, dtype=torch.int32).reshape(-1).to(dev)
            self._tables.append((raw, ch, len(chunks)))
            keep.append(arr)

    def load_state_dict(self, state_dict):
        """As torch.optim.Optimizer.load_state_dict (works before the first step, like the reference's resume at
        train.py:227-230); the launch tables are rebuilt because the momentum buffers are new tensors."""
        super().load_state_dict(state_dict)
        self._sig = None

    def _signature(self):
        return tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr(), float(g["weight_decay"]))
                     for g in self.param_groups for p in g["params"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        sig = self._signature()
        if sig != self._sig:
            self._build()
            self._sig = sig
        st = ops.current_stream_ptr()
        for group, tab in zip(self.param_groups, self._tables):
            if tab is None:
                continue
            raw, ch, n = tab
            table = ctypes.cast(ctypes.c_void_p(raw.data_ptr()), ctypes.POINTER(_lib.FrSgdTensor))  # device array
            ops.Launch("fr_sgd_step", [table, ctypes.c_void_p(ch.data_ptr()), n, float(group["lr"]),
                                       float(group["momentum"]), st])()
        return loss


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr)`` with its defaults (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) as one
    multi-tensor HIP launch per parameter group -- the reference's ``OPTIMIZER_NAME == 'Adam'`` branch (train.py:197-198).
    State keys and layout follow torch (``step`` as a host float tensor, ``exp_avg``, ``exp_avg_sq``), so state dicts
    are interchangeable with ``torch.optim.Adam``.  Parameters whose ``.grad`` is None are skipped and keep their step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._sig = None
        self._tables = []  # (group index, step value before this update, table_dev, chunks_dev, nchunks, params)

    zero_grad = SGD.zero_grad

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._sig = None

    def _relaid(self, p, t):
        if t.shape == p.shape and not _dense_same(p, t):
            r = torch.empty_like(p, memory_format=torch.preserve_format)
            r.copy_(t)
            return r
        return t

    def _build(self):
        for group in self.param_groups:  # a loaded state dict may carry settings this kernel does not implement
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise NotImplementedError("frhip.optim.Adam: dampening / nesterov / maximize are not implemented")
        chunk = _lib.lib.fr_sgd_chunk_elems()
        self._tables = []
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise _lib.FrhipError("frhip.optim.Adam: parameter on %s -- the HIP optimizer needs ROCm tensors"
                                          % p.device)
                st = self.state[p]
                if "step" not in st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg"], st["exp_avg_sq"] = self._relaid(p, st["exp_avg"]), self._relaid(p, st["exp_avg_sq"])
                if not (_dense_same(p, p.grad) and _dense_same(p, st["exp_avg"]) and _dense_same(p, st["exp_avg_sq"])):
                    raise _lib.FrhipError("frhip.optim.Adam: param / grad / moments must share one dense layout")
                by_step.setdefault(float(st["step"]), []).append(p)
            for step0, plist in by_step.items():
                arr = (_lib.FrAdamTensor * len(plist))()
                chunks = []
                for i, p in enumerate(plist):
                    st = self.state[p]
                    arr[i].p, arr[i].g = p.data_ptr(), p.grad.data_ptr()
                    arr[i].m, arr[i].v, arr[i].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                    chunks.extend((i, c) for c in range((p.numel() + chunk - 1) // chunk))
                dev = plist[0].device
                raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
                ch = torch.tensor(chunks, dtype=torch.int32).reshape(-1).to(dev)
                self._tables.append([gi, step0, raw, ch, len(chunks), plist])

    def _signature(self):
        return tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr())
                     for g in self.param_groups for p in g["params"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        sig = self._signature()
        if sig != self._sig:
            self._build()
            self._sig = sig
        st = ops.current_stream_ptr()
        for rec in self._tables:
            gi, step0, raw, ch, n, plist = rec
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            step = step0 + 1.0
            # the scalars torch computes in double on the host (torch/optim/adam.py, _single_tensor_adam)
            step_size = group["lr"] / (1.0 - b1 ** step)
            bc2_sqrt = (1.0 - b2 ** step) ** 0.5
            table = ctypes.cast(ctypes.c_void_p(raw.data_ptr()), ctypes.POINTER(_lib.FrAdamTensor))
            ops.Launch("fr_adam_step", [table, ctypes.c_void_p(ch.data_ptr()), n, float(step_size), float(1.0 - b1),
                                        float(b2), float(1.0 - b2), float(group["eps"]), float(bc2_sqrt), st])()
            for p in plist:
                self.state[p]["step"] += 1.0
            rec[1] = step
        return loss


def _dense_same(a, b):
    return a.shape == b.shape and a.stride() == b.stride() and _is_dense(a)


def _is_dense(t):
    """True when the tensor's elements occupy one contiguous run of memory (any permutation of a dense layout)."""
    if t.numel() == 0:
        return True
    dims = sorted((s, n) for s, n in zip(t.stride(), t.shape) if n > 1)
    expect = 1
    for s, n in dims:
        if s != expect:
            return False
        expect *= n
    return True
