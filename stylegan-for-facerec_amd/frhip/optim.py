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
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._sig = None
        self._tables = []  # per group: (table_dev, chunks_dev, nchunks)

    def zero_grad(self, set_to_none=False):
        """Keeps gradient buffers alive (the engine writes into them).  Gradients that are views of an engine
        gradient arena are cleared with ONE fill of the arena instead of one launch per parameter."""
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

    def _build(self):
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
