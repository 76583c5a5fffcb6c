"""Baseline Stage-3 run: IR_50_ReStyle trunk trained from scratch on BUPT-BalancedFace (no Stage-2 encoder).
Same keys/values as the reference's configs/config_BUPT_IR_50_baseline.py."""
from configs._common import stage3

EXP_NAME = "BUPT_IR_50_baseline"

configurations = {1: stage3(EXP_NAME, ENCODER_CHECKPOINT=None)}
