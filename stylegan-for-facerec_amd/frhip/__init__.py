"""frhip -- the HIP engine behind the drop-in modules (backbone/, head/, loss/, util/).

Importing the package asks the HIP runtime for 8 hardware queues (``GPU_MAX_HW_QUEUES``, default 4) unless the caller
chose a value: the engine puts the weight gradients on a second stream, streams are multiplexed onto hardware queues
round-robin, and once RCCL has created its own streams the side stream can land on the SAME queue as the main stream,
which silently serialises the two (measured: +12 % step time with one rank under ``torch.distributed``).  The variable
only takes effect if it is set before the first HIP call of the process, so entry points (bench.py, train.py) also set
it themselves before importing torch.
"""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
