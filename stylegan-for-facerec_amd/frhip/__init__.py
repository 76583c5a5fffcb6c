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
