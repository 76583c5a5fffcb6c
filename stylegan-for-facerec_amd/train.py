#!/usr/bin/env python3
"""Stage-3 face-recognition training driver -- counterpart of the reference's train.py on the frhip HIP engine.

    python train.py --config configs/config_BUPT_IR_50_baseline.py
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config configs/...
    python train.py --config configs/config_synthetic_smoke.py --synthetic 100x12 --max-steps 20

Kept from the reference (train.py:41-421): every config key, module construction order (all four heads are built
eagerly), the BN / non-BN parameter groups, LR stages (/1.5), optional warm-up, the freeze -> unfreeze schedule on
``.module.encoder.body``, per-step top-1/5, per-epoch checkpoint file names.  Changed on purpose (SURVEY.md section 7,
hard parts): one process per GPU with RCCL gradient all-reduce instead of nn.DataParallel (BATCH_SIZE is per GPU),
the head lives on the GPU, metrics are read back every DISP_FREQ steps instead of three ``.item()`` syncs per step,
wandb / bcolz validation are optional.  Resume: the reference restores weights and optimizer state
but restarts the batch counter (warm-up, logging), the shuffle and the dropout stream; here a ``State_*`` file written
next to every checkpoint carries epoch, batch counter, dropout stream and the host RNG states, the per-epoch shuffle and
the GPU crop / flip stream are functions of (SEED, epoch), and ``STATE_RESUME_ROOT`` restores them -- training resumed
at an epoch boundary continues bit for bit (tests/test_gpu_model.py::test_resume_continues_bit_for_bit; with the HOST
transform the workers' python ``random`` streams are re-seeded from the restored torch generator, which reproduces
them only for the same NUM_WORKERS).  An epoch cut short by --max-steps is recorded as unfinished and repeated.
"""
import argparse
import importlib
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # main / side / RCCL streams on separate hardware queues (frhip/__init__.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from backbone.model_irse import IR_50, IR_101, IR_152, IR_SE_50, IR_SE_101, IR_SE_152
from backbone.model_resnet import ResNet_50, ResNet_101, ResNet_152  # noqa: F401  (import parity with the reference)
from backbone.restyle_psp import pSp
from dataset import FacesDataset, StageTransform, SyntheticFaces, TrainTransform
from frhip import functional as FRF
from frhip import set_compute_dtype
from frhip.optim import SGD, Adam
from frhip.parallel import DataParallel
from head.metrics import Am_softmax, ArcFace, CosFace, SphereFace
from loss.focal import FocalLoss
from util.utils import (AverageMeter, accuracy, buffer_val, collate_fn_ignore_none, get_time, get_val_data, perform_val,
                        schedule_lr, separate_irse_bn_paras, warm_up_lr)

IRSE = {"IR_50": IR_50, "IR_101": IR_101, "IR_152": IR_152, "IR_SE_50": IR_SE_50, "IR_SE_101": IR_SE_101,
        "IR_SE_152": IR_SE_152}
RESTYLE = {"IR_34_ReStyle": "BackboneEncoder34", "IR_50_ReStyle": "BackboneEncoder", "IR_100_ReStyle": "BackboneEncoder100"}


class _ConsoleLog(object):
    def log(self, stats):
        print("[log]", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in stats.items()})


def make_logger(cfg, rank):
    if rank != 0:
        return None
    try:
        import wandb
        wandb.init(project=cfg.get("PROJECT_NAME", "face-evolve"), config=cfg)
        wandb.run.name = cfg["EXP_NAME"]
        return wandb
    except Exception:  # noqa: BLE001 -- wandb is optional here
        return _ConsoleLog()


def build_backbone(cfg):
    name = cfg["BACKBONE_NAME"]
    if name in IRSE:
        return IRSE[name](cfg["INPUT_SIZE"])
    if name in RESTYLE:
        return pSp(encoder_type=RESTYLE[name], size=cfg.get("ENCODER_INPUT_SIZE", 112),
                   checkpoint_path=cfg.get("ENCODER_CHECKPOINT"), avg_image=cfg.get("ENCODER_AVG_IMAGE"),
                   include_dropout=False)
    raise ValueError("unsupported BACKBONE_NAME %r" % name)


def _head_state_index(optimizer, crit):
    """Index of the sharded head weight in the optimizer's flat parameter numbering (state_dict keys)."""
    i = 0
    for group in optimizer.param_groups:
        for p in group["params"]:
            if p is crit.weight:
                return i
            i += 1
    raise RuntimeError("the sharded head weight is not in the optimizer")


def optimizer_state_for_checkpoint(optimizer, crit):
    """``optimizer.state_dict()`` in the reference's layout.  With a class-sharded head the momentum buffer of the head
    weight is gathered to [classes, 512] (collective: every rank calls this), so optimizer checkpoints are
    interchangeable between replicated and sharded runs and between world sizes."""
    sd = optimizer.state_dict()
    if crit is None:
        return sd
    idx = _head_state_index(optimizer, crit)
    st = sd["state"].get(idx)
    if st is not None:
        full = {k: crit.comm.gather_ragged_rows(v, crit.shard_sizes()) for k, v in st.items()
                if torch.is_tensor(v) and v.shape == crit.weight.shape}  # momentum_buffer / exp_avg / exp_avg_sq
        sd = dict(sd, state=dict(sd["state"]))
        sd["state"][idx] = dict(st, **full)
    return sd


def _np_state_plain(st):
    """numpy's RNG state with the key array as a plain list: State_* files stay loadable with torch.load's default
    ``weights_only=True`` (no numpy globals in the pickle)."""
    return (str(st[0]), [int(v) for v in st[1]], int(st[2]), int(st[3]), float(st[4]))


def load_optimizer_checkpoint(optimizer, crit, sd):
    """Inverse of ``optimizer_state_for_checkpoint``: this rank keeps its class range of the head's optimizer state."""
    if crit is not None:
        idx = _head_state_index(optimizer, crit)
        st = sd["state"].get(idx)
        if st is not None:
            part = {k: v[crit.lo:crit.hi].clone() for k, v in st.items()
                    if torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == crit.out_features}
            sd = dict(sd, state=dict(sd["state"]))
            sd["state"][idx] = dict(st, **part)
    optimizer.load_state_dict(sd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config.py")
    ap.add_argument("--synthetic", default="", help="IDSxPER: train on seeded synthetic identities, no DATA_ROOT")
    ap.add_argument("--max-steps", type=int, default=0)
    args = ap.parse_args()
    cfg = importlib.import_module(args.config.replace(".py", "").replace("/", ".")).configurations[1]

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("train.py: the frhip engine needs a ROCm GPU; the CPU restatement in oracle/ is for tests only")
    if os.environ.get("FRHIP_TRAIN_ONE_DEVICE") == "1":
        local = 0  # test hook: several ranks share GPU 0 (with FRHIP_DIST_BACKEND=gloo; RCCL refuses duplicate devices)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FRHIP_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        # Generous watchdog: the per-epoch RFW verification (4 x 12 k images with flip-TTA, sharded over the ranks) and the
        # checkpoint writes of rank 0 sit between collectives.
        import datetime
        patience = datetime.timedelta(hours=2)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=patience)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)

    np.random.seed(cfg["SEED"])
    torch.manual_seed(cfg["SEED"])  # identical initial weights on every rank (and broadcast below)
    os.makedirs(cfg["MODEL_ROOT"], exist_ok=True)
    os.makedirs(cfg["LOG_ROOT"], exist_ok=True)
    logger = make_logger(cfg, rank)

    gpu_tf = None
    if cfg.get("GPU_INPUT_PIPELINE", False):
        # workers decode only; resize / crop / flip / normalise run on the GPU per batch (frhip/input_pipeline.py)
        from frhip.input_pipeline import GpuTrainTransform
        gpu_tf = GpuTrainTransform(cfg["INPUT_SIZE"][0], cfg["RGB_MEAN"], cfg["RGB_STD"])
        aug_rng = torch.Generator().manual_seed(cfg["SEED"] + 7919 * rank)
    if args.synthetic:
        ids, per = (int(v) for v in args.synthetic.split("x"))
        dataset = SyntheticFaces(ids, per, cfg["INPUT_SIZE"][0], cfg["SEED"], staged=gpu_tf is not None)
    else:
        dataset = FacesDataset(os.path.join(cfg["DATA_ROOT"], cfg["TRAIN_IMAGES_FOLDER"]),
                               StageTransform() if gpu_tf is not None else
                               TrainTransform(cfg["INPUT_SIZE"][0], cfg["RGB_MEAN"], cfg["RGB_STD"]))
    num_class = len(dataset.classes)
    # shuffle = f(SEED, epoch) on every world size (set_epoch below), so a resumed run sees the epochs it would have seen
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True, seed=cfg["SEED"])
    loader = torch.utils.data.DataLoader(dataset, batch_size=cfg["BATCH_SIZE"], sampler=sampler,
                                         shuffle=False, pin_memory=cfg["PIN_MEMORY"],
                                         num_workers=cfg["NUM_WORKERS"], drop_last=cfg["DROP_LAST"],
                                         collate_fn=collate_fn_ignore_none)
    print("Number of Training Classes: {}".format(num_class))
    val = get_val_data(cfg["DATA_ROOT"]) if not args.synthetic else None

    backbone = build_backbone(cfg)
    # COMPUTE_DTYPE ('bf16' | 'fp32', optional -- the reference's configs have no such key): the numerics of the backbone.
    # The shipped BUPT configs select 'bf16', the path bench.py times; absent -> FRHIP_COMPUTE_DTYPE (default fp32).
    cdt = set_compute_dtype(backbone, cfg.get("COMPUTE_DTYPE"))
    if rank == 0:
        print("Backbone compute dtype: {}".format(cdt if cdt is not None else "process default (FRHIP_COMPUTE_DTYPE)"))
    emb, s = cfg["EMBEDDING_SIZE"], cfg.get("ARCFACE_S", 64.0)
    heads = {"ArcFace": ArcFace(emb, num_class, None, s=s), "CosFace": CosFace(emb, num_class, None),
             "SphereFace": SphereFace(emb, num_class, None), "Am_softmax": Am_softmax(emb, num_class, None)}
    head = heads[cfg["HEAD_NAME"]]
    bn_params, other_params = separate_irse_bn_paras(backbone)
    _, head_params = separate_irse_bn_paras(head)

    if cfg["BACKBONE_RESUME_ROOT"] and cfg["HEAD_RESUME_ROOT"]:
        if os.path.isfile(cfg["BACKBONE_RESUME_ROOT"]) and os.path.isfile(cfg["HEAD_RESUME_ROOT"]):
            backbone.load_state_dict(torch.load(cfg["BACKBONE_RESUME_ROOT"], map_location="cpu"))
            head.load_state_dict(torch.load(cfg["HEAD_RESUME_ROOT"], map_location="cpu"))
        else:
            print("No Checkpoint Found at '{}' and '{}'".format(cfg["BACKBONE_RESUME_ROOT"], cfg["HEAD_RESUME_ROOT"]))
    backbone, head = backbone.to(device), head.to(device)
    def make_optimizer(head_weights):
        name = cfg.get("OPTIMIZER_NAME", "SGD")
        if name == "SGD":  # reference train.py:195-196
            return SGD([{"params": other_params + head_weights, "weight_decay": cfg["WEIGHT_DECAY"]},
                        {"params": bn_params}], lr=cfg["LR"], momentum=cfg["MOMENTUM"])
        if name == "Adam":  # reference train.py:197-198: one group, torch defaults, no weight decay
            return Adam([{"params": bn_params + other_params + head_weights}], lr=cfg["LR"])
        raise NotImplementedError("OPTIMIZER_NAME %r: the reference builds 'SGD' or 'Adam'" % (name,))

    optimizer = make_optimizer(head_params)
    opt_resume = cfg.get("OPTIMIZER_RESUME_ROOT")
    loss_fn = FocalLoss() if cfg["LOSS_NAME"] == "Focal" else None
    crit = None
    if cfg.get("SHARDED_HEAD", False):
        # class-sharded head + focal loss over the ranks (frhip/sharded_head.py): the N x 512 weight, its momentum and the
        # [B, N] logits are split by class range; the loss is the focal loss of the GLOBAL batch, as under the
        # reference's nn.DataParallel.  Checkpoints keep the reference's layout (gathered ``weight``).
        if cfg["HEAD_NAME"] not in ("ArcFace", "CosFace") or loss_fn is None:
            raise NotImplementedError("SHARDED_HEAD needs HEAD_NAME ArcFace/CosFace and LOSS_NAME 'Focal'")
        from frhip.sharded_head import ShardedMarginLoss
        crit = ShardedMarginLoss.from_head(head, gamma=loss_fn.gamma).to(device)
        optimizer = make_optimizer([crit.weight])
    # exposes .module like nn.DataParallel; all-reduce only when world > 1 (a weight shard is complete on its owner)
    BACKBONE = DataParallel(backbone, None if crit is not None else head)
    ce = torch.nn.CrossEntropyLoss()

    optimizer_loaded = False
    if opt_resume:  # reference train.py:227-232: before the first step (momentum buffers and the groups' LR come back)
        if os.path.isfile(opt_resume):
            print("Loading Optimizer Checkpoint '{}'".format(opt_resume))
            load_optimizer_checkpoint(optimizer, crit, torch.load(opt_resume, map_location=device))
            optimizer_loaded = True
        else:
            print("No Checkpoint Found at '{}'. Please Have a Check or Continue to Train from Scratch".format(opt_resume))
    runner = (backbone.encoder if hasattr(backbone, "encoder") else backbone)._runner[0]
    start_epoch, batch, lr_stage_done = cfg.get("START_EPOCH", 0), 0, None
    state_resume = cfg.get("STATE_RESUME_ROOT")
    if state_resume and os.path.isfile(state_resume):
        state = torch.load(state_resume, map_location="cpu")
        start_epoch, batch, runner.step_seed = int(state["epoch"]), int(state["batch"]), int(state["dropout_stream"])
        # an unfinished epoch whose LR stage is already in the optimizer file -- honoured only if that file was really
        # loaded: a State_* file alone carries no learning rate, and skipping schedule_lr then would lose the division
        lr_stage_done = state.get("lr_stage_applied") if optimizer_loaded else None
        if state.get("lr_stage_applied") is not None and not optimizer_loaded:
            print("State file marks the LR stage of epoch {} as applied, but no optimizer checkpoint was loaded: "
                  "the stage is applied again".format(state.get("lr_stage_applied")))
        if "torch_rng" in state:
            torch.set_rng_state(state["torch_rng"])
            n = state["numpy_rng"]
            np.random.set_state((n[0], np.asarray(n[1], dtype=np.uint32), n[2], n[3], n[4]))
        print("Resuming at epoch {} batch {}".format(start_epoch, batch))

    disp_freq = max(1, len(loader) // 10)  # the reference divides by zero below 10 batches/epoch
    lts = cfg.get("LIMIT_TRAIN_SAMPLES", None)
    limit_batches = None if lts is None else max(1, lts // cfg["BATCH_SIZE"])
    warm_epochs = cfg["NUM_EPOCH"] // 25
    warm_batches = len(loader) * warm_epochs
    freeze = cfg.get("FREEZE_BACKBONE_EPOCHS")
    FRF.CHECK_LABELS = False  # labels come from the dataset's own class index
    for epoch in range(start_epoch, cfg["NUM_EPOCH"]):
        epoch_first_batch = batch
        if epoch in cfg["STAGES"] and epoch != lr_stage_done:
            schedule_lr(optimizer)
        backbone.train()
        head.train()
        if freeze is not None and hasattr(BACKBONE.module, "encoder"):
            enc = BACKBONE.module.encoder
            enc.input_layer.requires_grad_(True)
            enc.body.requires_grad_(epoch > freeze)
            enc.output_layer.requires_grad_(True)
        sampler.set_epoch(epoch)
        if gpu_tf is not None:  # crops / flips are a function of (SEED, epoch, rank): a resumed run redraws the same ones
            aug_rng.manual_seed(cfg["SEED"] + 7919 * rank + 104729 * (epoch + 1))
        losses, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter()
        pending = []

        def flush():  # one host sync per display interval; EVERY step reaches the meters (reference train.py:308-310)
            for l, p1, p5, n in pending:
                losses.update(float(l), n)
                top1.update(float(p1), n)
                top5.update(float(p5), n)
            del pending[:]

        for inputs, labels in loader:
            if cfg.get("WARMUP", True) and epoch + 1 <= warm_epochs and batch + 1 <= warm_batches:
                warm_up_lr(batch, warm_batches, cfg["LR"], optimizer)
            inputs = inputs.to(device, non_blocking=True)
            labels = labels.to(device, non_blocking=True).long()
            if gpu_tf is not None:
                inputs = gpu_tf(inputs, generator=aug_rng)
            if crit is not None:
                loss, prec1, prec5 = crit(BACKBONE(inputs), labels)
            else:
                outputs = head(BACKBONE(inputs), labels)
                loss = loss_fn(outputs, labels)[0] if loss_fn is not None else ce(outputs, labels)
                prec1, prec5 = accuracy(outputs.data, labels, topk=(1, 5))
            pending.append((loss.detach(), prec1, prec5, inputs.size(0)))
            optimizer.zero_grad()
            loss.backward()
            BACKBONE.synchronize()
            optimizer.step()
            if (batch + 1) % disp_freq == 0 or (args.max_steps and batch + 1 >= args.max_steps):
                flush()
                if rank == 0:
                    print("Epoch {}/{} Batch {}\tTraining Loss {:.4f} ({:.4f})\tPrec@1 {:.3f} ({:.3f})\tPrec@5 {:.3f} "
                          "({:.3f})".format(epoch + 1, cfg["NUM_EPOCH"], batch + 1, losses.val, losses.avg, top1.val,
                                            top1.avg, top5.val, top5.avg))
                    if logger is not None:
                        logger.log({"train_loss": losses.val, "step": batch * cfg["BATCH_SIZE"] * world})
            batch += 1
            if args.max_steps and batch >= args.max_steps:
                break
            if limit_batches is not None and batch - epoch_first_batch >= limit_batches:
                break  # LIMIT_TRAIN_SAMPLES: a shorter epoch, e.g. to validate more often (reference train.py:68-69,344)
        flush()  # the trailing steps of the epoch
        samples_seen = batch * cfg["BATCH_SIZE"] * world  # ONE x axis (global samples) for every log call
        if rank == 0:  # per-epoch summary (reference train.py:347-357)
            print("=" * 60)
            print("Epoch: {}/{}\tTraining Loss {:.4f}\tTraining Prec@1 {:.3f}\tTraining Prec@5 {:.3f}".format(
                epoch + 1, cfg["NUM_EPOCH"], losses.avg, top1.avg, top5.avg))
            print("=" * 60)
            if logger is not None:
                logger.log({"train_loss_ep": losses.avg, "train_acc_ep": top1.avg, "train_acc_top5_ep": top5.avg,
                            "epoch": epoch + 1, "step": samples_seen})
        if val is not None and val[-2] is not None:
            # per-epoch verification on the RFW subsets (reference train.py:403-410); flip-TTA, k-fold accuracy.  Every
            # rank embeds its share of the batches (collective), rank 0 logs.
            rfw, rfw_issame = val[-2], val[-1]
            if rank == 0:
                print("=" * 60)
            for eth in ("African", "Asian", "Caucasian", "Indian"):
                if eth not in rfw:
                    continue
                acc, thr, roc = perform_val(True, device, cfg["EMBEDDING_SIZE"], cfg["BATCH_SIZE"], BACKBONE, rfw[eth],
                                            rfw_issame[eth], dset_name="RFW_" + eth,
                                            ccrop=cfg.get("CCROP_AT_VAL", True), rank=rank, world=world)
                if rank == 0:
                    if logger is not None:
                        buffer_val(logger, "RFW_" + eth, acc, thr, roc, epoch + 1, samples_seen)
                    print("Evaluation: RFW {} Acc: {}".format(eth, acc))
            if rank == 0:
                print("=" * 60)
            BACKBONE.module.train()
        if crit is not None:
            with torch.no_grad():
                head.weight.copy_(crit.gather_weight())  # collective: every rank takes part, rank 0 writes the file
        opt_state = optimizer_state_for_checkpoint(optimizer, crit)  # collective with a sharded head
        if rank == 0:
            tag = "Epoch_{}_Batch_{}_Time_{}_checkpoint.pth".format(epoch + 1, batch, get_time())
            root = cfg["MODEL_ROOT"]
            torch.save(BACKBONE.module.state_dict(), os.path.join(root, "Backbone_{}_{}".format(cfg["BACKBONE_NAME"], tag)))
            torch.save(head.state_dict(), os.path.join(root, "Head_{}_{}".format(cfg["HEAD_NAME"], tag)))
            torch.save(opt_state, os.path.join(root, "Optimizer_{}_{}".format(cfg["HEAD_NAME"], tag)))
            # what the reference forgets on resume (train.py:206-232 restores weights and optimizer only): the batch
            # counter (warm-up, logging), the dropout stream, the host RNG (DataLoader worker seeds -> python `random` of
            # the host transform).  The per-epoch shuffle and the GPU crop / flip stream are functions of (SEED, epoch).
            # An epoch cut short by --max-steps is recorded as NOT finished: resuming repeats it from its first batch
            # with the weights of the checkpoint (mid-epoch positions are not restored).
            epoch_len = len(loader) if limit_batches is None else min(len(loader), limit_batches)
            finished = not (args.max_steps and batch >= args.max_steps and (batch - epoch_first_batch) < epoch_len)
            # the repeated epoch restarts the batch counter at its first batch (warm-up, logging axis) and must not divide
            # the LR a second time if it is a stage epoch (the Optimizer_* file already holds the divided LR)
            torch.save({"epoch": epoch + 1 if finished else epoch, "batch": batch if finished else epoch_first_batch,
                        "dropout_stream": runner.step_seed, "epoch_finished": finished,
                        "lr_stage_applied": int(epoch) if (not finished and epoch in cfg["STAGES"]) else None,
                        "torch_rng": torch.get_rng_state(),
                        "numpy_rng": _np_state_plain(np.random.get_state())},
                       os.path.join(root, "State_{}_{}".format(cfg["HEAD_NAME"], tag)))
        if args.max_steps and batch >= args.max_steps:
            break
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
