"""frhip -- the HIP engine behind the drop-in modules (backbone/, head/, loss/, util/).

Importing the package asks the HIP runtime for 8 hardware queues (``GPU_MAX_HW_QUEUES``, default 4) unless the caller
chose a value: the engine puts the weight gradients on a second stream, streams are multiplexed onto hardware queues
round-robin, and once RCCL has created its own streams the side stream can land on the SAME queue as the main stream,
which silently serialises the two (measured: +12 % step time with one rank under ``torch.distributed``).  The variable
only takes effect if it is set before the first HIP call of the process, so entry points (bench.py, train.py) also set
it themselves before importing torch.
"""
import os


def _warn_if_too_late():
    """``GPU_MAX_HW_QUEUES`` is read once, when the HIP runtime initialises.  If the process has already touched the GPU
    with fewer than 8 queues, the two-stream schedule may serialise silently: say so instead."""
    import sys
    import warnings
    torch = sys.modules.get("torch")
    try:
        late = torch is not None and torch.cuda.is_initialized()
    except Exception:  # noqa: BLE001
        late = False
    if late and _requested_before_import is None:
        warnings.warn("frhip: the HIP runtime was initialised before frhip was imported, so GPU_MAX_HW_QUEUES=8 cannot take "
                      "effect any more; with RCCL streams present the weight-gradient stream may share a hardware queue "
                      "with the main stream (+10 % step time).  Set GPU_MAX_HW_QUEUES=8 in the environment or import "
                      "frhip before the first CUDA/HIP call.", RuntimeWarning, stacklevel=3)


_requested_before_import = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_warn_if_too_late()


_DTYPE_NAMES = {"bf16": "bfloat16", "bfloat16": "bfloat16", "fp32": "float32", "float32": "float32", "f32": "float32"}


def set_compute_dtype(model, dtype):
    """Select the numerics of a backbone from a config value (``COMPUTE_DTYPE`` in ``configurations[1]``; the reference's
    train.py:41-90 has no such key, so drivers read it with ``cfg.get``): 'bf16' = bf16 storage / fp32 accumulate on the
    MFMA bf16 kernels (the path bench.py times), 'fp32' = the parity path.  ``None`` leaves the process default
    (``FRHIP_COMPUTE_DTYPE``, fp32).  pSp keeps its trunk in ``.encoder``; the plan is keyed on the dtype, so switching
    between steps rebuilds it.  Returns the torch dtype selected (or None)."""
    if dtype is None:
        return None
    import torch
    if isinstance(dtype, str):
        name = _DTYPE_NAMES.get(dtype.lower())
        if name is None:
            raise ValueError("COMPUTE_DTYPE %r: expected 'bf16' or 'fp32'" % (dtype,))
        dtype = getattr(torch, name)
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("COMPUTE_DTYPE %r: expected torch.bfloat16 or torch.float32" % (dtype,))
    inner = model.module if hasattr(model, "module") and not hasattr(model, "_runner") else model
    inner = inner.encoder if hasattr(inner, "encoder") else inner
    inner.compute_dtype = dtype
    return dtype
