"""Stage-3 run initialised from a Stage-2 ReStyle-pSp encoder trained on the AfrAsian StyleGAN prior.
Differs from the baseline config only in EXP_NAME, ENCODER_CHECKPOINT and ENCODER_AVG_IMAGE, as in the reference."""
from configs._common import stage3

EXP_NAME = "BUPT_IR_50_AfrAsian"

configurations = {1: stage3(
    EXP_NAME,
    ENCODER_CHECKPOINT="<path to the Stage-2 encoder checkpoint (.pt) trained with the AfrAsian prior>",
    ENCODER_AVG_IMAGE="<path to the avg_image.jpg written by Stage-2 training>",
    COMPUTE_DTYPE="bf16",  # not a reference key (cfg.get): the throughput path bench.py measures; 'fp32' = parity path
)}
