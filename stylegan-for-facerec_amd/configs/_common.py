"""Shared defaults of the Stage-3 configs.  Keys and meanings are the reference's (train.py:41-90):
a config module exposes ``configurations = {1: dict(...)}`` and the driver reads ``configurations[1]``."""
import os

import numpy as np
import torch


def stage3(exp_name, **overrides):
    cfg = dict(
        SEED=900,
        EXP_NAME=exp_name,
        DATA_ROOT="<path to the folder containing BUPT-BalancedFace and test datasets as subfolders>",
        TRAIN_IMAGES_FOLDER="bupt-balancedface",
        MODEL_ROOT=os.path.join("exps/model/", exp_name),
        LOG_ROOT=os.path.join("exps/log", exp_name),
        BACKBONE_RESUME_ROOT="",
        HEAD_RESUME_ROOT="",
        OPTIMIZER_RESUME_ROOT="",
        BACKBONE_NAME="IR_50_ReStyle",   # pSp IR-SE-50 trunk, with or without a Stage-2 encoder checkpoint
        HEAD_NAME="ArcFace",             # ArcFace | CosFace | SphereFace | Am_softmax
        LOSS_NAME="Focal",               # Focal | Softmax
        ENCODER_CHECKPOINT=None,
        ENCODER_AVG_IMAGE="<an arbitrary 112x112 image here>",
        ENCODER_INPUT_SIZE=112,
        ENCODER_ADDITIONAL_DROPOUT=0.15,
        INPUT_SIZE=[112, 112],
        RGB_MEAN=[0.5, 0.5, 0.5],
        RGB_STD=[0.5, 0.5, 0.5],
        EMBEDDING_SIZE=512,
        BATCH_SIZE=100,                  # reference: global batch of nn.DataParallel; here: per process (= per GPU)
        DROP_LAST=True,
        FREEZE_BACKBONE_EPOCHS=3,
        LR=0.03,
        NUM_EPOCH=100,
        WEIGHT_DECAY=2e-3,               # not applied to batch-norm parameters
        MOMENTUM=0.9,
        STAGES=np.arange(10, 125, 5) + 5,
        WARMUP=False,
        LAYER_DECAY=None,
        DEVICE=torch.device("cuda:0" if torch.cuda.is_available() else "cpu"),
        MULTI_GPU=True,
        GPU_ID=[0],
        PIN_MEMORY=True,
        NUM_WORKERS=8,
    )
    cfg.update(overrides)
    return cfg
