"""Baseline Stage-3 run: IR_50_ReStyle trunk trained from scratch on BUPT-BalancedFace (no Stage-2 encoder).
Same keys/values as the reference's configs/config_BUPT_IR_50_baseline.py."""
from configs._common import stage3

EXP_NAME = "BUPT_IR_50_baseline"

# COMPUTE_DTYPE (not a reference key, read with cfg.get): 'bf16' = bf16 storage / fp32 accumulate on the MFMA kernels --
# the path bench.py measures; 'fp32' = the parity path (logits within 1e-3 of the reference's CPU run).
configurations = {1: stage3(EXP_NAME, ENCODER_CHECKPOINT=None, COMPUTE_DTYPE="bf16")}
