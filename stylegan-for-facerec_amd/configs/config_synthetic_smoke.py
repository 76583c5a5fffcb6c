"""BASELINE.json configs[0] made runnable without data: the baseline config on a synthetic identity folder.
``train.py --config configs/config_synthetic_smoke.py --synthetic 100x12`` generates 100 identities x 12 samples of
seeded 112x112 tensors instead of reading DATA_ROOT."""
from configs._common import stage3
from frhip import synth

EXP_NAME = "synthetic_smoke"

configurations = {1: stage3(EXP_NAME, ENCODER_AVG_IMAGE=synth.uniform(900, 'avg_image', (3, 112, 112)), NUM_EPOCH=1,
                            BATCH_SIZE=100, NUM_WORKERS=0,
                            FREEZE_BACKBONE_EPOCHS=None, DATA_ROOT="", STAGES=[],
                            COMPUTE_DTYPE="fp32")}  # the parity path: the smoke run is compared with the CPU reference
