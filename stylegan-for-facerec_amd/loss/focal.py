"""``from loss.focal import FocalLoss`` (reference train.py:10, loss/focal.py:9-21).

Focal modulation of the *batch-mean* cross entropy, returned as ``(loss, None)`` -- the tuple is relied upon
by the training loop (reference train.py:304).  Runs on the HIP row-softmax kernels; logits must be on a
ROCm device.
"""
import torch.nn as nn

from frhip import functional as FRF


class FocalLoss(nn.Module):
    def __init__(self, gamma=2, eps=1e-7, use_weights=False):
        super().__init__()
        self.gamma = gamma
        self.eps = eps  # unused, as in the reference
        self.use_weights = use_weights  # unused, as in the reference

    def forward(self, input, target):
        return FRF.focal_loss(input, target, float(self.gamma)), None
