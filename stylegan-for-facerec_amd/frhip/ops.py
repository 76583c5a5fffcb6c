"""Launch records for the C-ABI entry points of libfrhip.so.

Every builder returns a ``Launch``: the C function plus a fully marshalled argument list (device pointers,
geometry, stream).  The training step is static -- same shapes, same buffers every iteration -- so the
engine marshals once and replays lists of launches; nothing here touches tensors element-wise and nothing
synchronises.  All tensors must live on a ROCm device: there is no CPU path behind these calls.
"""
import ctypes

import torch

from . import _lib
from ._lib import (EPI_ATOMIC, EPI_BIAS_RES, EPI_BNBWD, EPI_MARGIN, EPI_PRELU_BWD, EPI_SLAB, EPI_STATS, EPI_STATS_X, EPI_STORE,  # noqa: F401
                   FR_BF16,
                   FR_F32,
                   PRO_BN, PRO_NONE, PRO_PRELU, PRO_RESBN, PRO_RESBN_SE, lib)

TORCH_DTYPE = {FR_F32: torch.float32, FR_BF16: torch.bfloat16}


def fr_dtype(t):
    if t.dtype == torch.float32:
        return FR_F32
    if t.dtype == torch.bfloat16:
        return FR_BF16
    raise _lib.FrhipError("frhip: unsupported tensor dtype %s" % t.dtype)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the product path has no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.FrhipError("frhip: expected a ROCm device tensor, got a %s tensor -- the HIP path has no CPU "
                              "fallback (the CPU restatement lives in oracle/ for tests only)" % t.device)
    return ctypes.c_void_p(t.data_ptr())


def current_stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_ptr(stream):
    """hipStream_t of a torch.cuda.Stream as the void* the C ABI takes."""
    return ctypes.c_void_p(stream.cuda_stream)


class Launch(object):
    __slots__ = ("fn", "args", "name", "keep", "tstream", "stop_event", "stop_handle", "stop_stream")

    def __init__(self, name, args, keep=None):
        self.fn = getattr(lib, name)
        self.args = list(args)
        self.name = name
        self.keep = keep
        self.tstream = None  # torch stream the launch targets when it is not the current one
        self.stop_event = None  # torch.cuda.Event this launch signals when its (last) kernel completes, see arm()

    def arm(self, event, stream_ptr):
        """Make ``event`` the completion signal of this launch (fr_arm_stop_event / fr_finish_stop_event, ABI v6): what an
        ``event.record(stream)`` right behind the launch would capture, without the marker packet the next kernel of the
        stream would wait for (tools/edge_probe.hip: +1.6 instead of +5.1 us per dependency edge).  The event must have been
        recorded once before (torch creates the hipEvent_t lazily); ``stream_ptr``: the stream the launch is enqueued on."""
        if not event.cuda_event:
            raise _lib.FrhipError("frhip: arm() needs an event that exists (record it once first)")
        self.stop_event, self.stop_handle, self.stop_stream = event, ctypes.c_void_p(event.cuda_event), stream_ptr

    def __call__(self):
        if self.stop_event is not None:
            lib.fr_arm_stop_event(self.stop_handle)
            rc = self.fn(*self.args)
            n = lib.fr_finish_stop_event(self.stop_stream)  # records the ordinary way unless exactly one kernel took the event
            if n < 0:
                _lib.check(n, "fr_finish_stop_event")
        else:
            rc = self.fn(*self.args)
        if rc:
            _lib.check(rc, self.name)


def run(launches):
    for l in launches:
        l()


def _fill(struct, **kw):
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = ptr(v)
        setattr(struct, k, v)
    return struct


def conv(stream, dtype, **kw):
    """fr_conv_igemm.  kw = FrConvArgs fields (tensors allowed for pointer fields)."""
    a = _fill(_lib.FrConvArgs(), **kw)
    return Launch("fr_conv_igemm", [ctypes.byref(a), dtype, stream], keep=(a, kw))


def conv_strip(stream, **kw):
    """fr_conv3x3_strip (bf16, stride-1 3x3, input strip resident in LDS).  Same FrConvArgs fields as conv()."""
    a = _fill(_lib.FrConvArgs(), **kw)
    return Launch("fr_conv3x3_strip", [ctypes.byref(a), stream], keep=(a, kw))


def conv1x1_stream(stream, **kw):
    """fr_conv1x1_stream (bf16 1x1 convolution as a row-streaming GEMM).  Same FrConvArgs fields as conv()."""
    a = _fill(_lib.FrConvArgs(), **kw)
    return Launch("fr_conv1x1_stream", [ctypes.byref(a), stream], keep=(a, kw))


def conv1x1_stream_parts(B, RH, RW, K, N):
    """Partial-sum rows of fr_conv1x1_stream for a shape; 0 when it is not served."""
    return int(lib.fr_conv1x1_stream_parts(int(B), int(RH), int(RW), int(K), int(N)))


def strip_parts(B, cin, cout, w, epi=EPI_STORE):
    """Workgroups (= partial-sum rows) of the strip kernel for a shape + epilogue; 0 when it is not served."""
    return int(lib.fr_conv3x3_strip_parts(int(B), int(cin), int(cout), int(w), int(epi)))


def conv_s2_strip(stream, **kw):
    """fr_conv3x3_s2_strip (bf16 stride-2 3x3, Cin == Cout): mode 0 forward, mode 2 (par -1) data gradient."""
    a = _fill(_lib.FrConvArgs(), **kw)
    return Launch("fr_conv3x3_s2_strip", [ctypes.byref(a), stream], keep=(a, kw))


def s2_strip_parts(B, cin, cout, wl, mode):
    """Partial-sum rows of the stride-2 strip kernel for a shape; 0 when it is not served."""
    return int(lib.fr_conv3x3_s2_strip_parts(int(B), int(cin), int(cout), int(wl), int(mode)))


def wgrad(stream, dtype, **kw):
    a = _fill(_lib.FrWgradArgs(), **kw)
    return Launch("fr_conv_wgrad", [ctypes.byref(a), dtype, stream], keep=(a, kw))


def wgrad_strip(stream, **kw):
    """fr_conv_wgrad_strip (bf16 stride-1 3x3; needs kw['slab'] and kw['nsplit'] = strip groups)."""
    a = _fill(_lib.FrWgradArgs(), **kw)
    return Launch("fr_conv_wgrad_strip", [ctypes.byref(a), stream], keep=(a, kw))


def wgrad_strip_supported(cout, cin, w):
    return bool(lib.fr_conv_wgrad_strip_supported(int(cout), int(cin), int(w)))


def bn_apply(stream, dtype, **kw):
    a = _fill(_lib.FrApplyArgs(), **kw)
    return Launch("fr_bn_apply", [ctypes.byref(a), dtype, stream], keep=(a, kw))


def bn_bwd_reduce(stream, dtype, **kw):
    a = _fill(_lib.FrBnBwdArgs(), **kw)
    return Launch("fr_bn_bwd_reduce", [ctypes.byref(a), dtype, stream], keep=(a, kw))


def bn_bwd_apply(stream, dtype, **kw):
    a = _fill(_lib.FrBnBwdArgs(), **kw)
    return Launch("fr_bn_bwd_apply", [ctypes.byref(a), dtype, stream], keep=(a, kw))


def bn_fin(count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, invstd, scale, shift):
    """FrBnFinArgs: the BatchNorm arguments of fr_bn_finalize as a struct (fr_bn_finalize_res takes two)."""
    t = _lib.FrBnFinArgs()
    _fill(t, gamma=gamma, beta=beta, running_mean=running_mean, running_var=running_var, nbt=nbt, mean=mean, invstd=invstd,
          scale=scale, shift=shift)
    t.count, t.eps, t.momentum = float(count), float(eps), float(momentum)
    t._keep = (gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift)
    return t


def set_option(name, value):
    """Override a run-time switch of the library (fr_set_option); returns the previous value."""
    return int(lib.fr_set_option(name.encode(), int(value)))


# Switches the LIBRARY reads (kernel-family A/B switches and test hooks; README "Switches").  The library caches a switch at
# its first use, so the environment is pushed again whenever a plan is built (and by tests that flip one in-process).
LIB_SWITCHES = {"FRHIP_ROLL64": 1, "FRHIP_WGRAD_ROLL": 1, "FRHIP_WGRAD_DEFER": 1, "FRHIP_SPLIT_STRIPS": 0, "FRHIP_XCD_ORDER": 1,
                "FRHIP_IGEMM_BN": 0, "FRHIP_ROLL_NSEG": -1, "FRHIP_S2ROLL_NSEG": -1}


def sync_switches():
    import os
    for name, dflt in LIB_SWITCHES.items():
        set_option(name, int(os.environ.get(name, dflt)))


def call(name, *args):
    """Generic positional launch: tensors -> pointers, everything else passed through."""
    keep = args
    conv_args = [ptr(a) if isinstance(a, torch.Tensor) else
                 (ctypes.byref(a) if isinstance(a, ctypes.Structure) else a) for a in args]
    return Launch(name, conv_args, keep=keep)


def grid_blocks(rows, C, dtype, cap=1024):
    """Grid size for the [row-thread][channel-chunk] kernels: enough blocks to fill 256 CUs a few times."""
    vec = 4 if dtype == FR_F32 else 8
    rows_per_block = max(1, 256 // (C // vec))
    need = (rows + rows_per_block - 1) // rows_per_block
    return int(max(1, min(cap, need)))
