"""Class-sharded margin head on the GPU (frhip/sharded_head.py): the three sharded-softmax kernels
against torch, the one-rank module against the replicated HIP head and the oracle, and two ranks sharing GPU 0."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from frhip import ops, synth  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("rows,N,world", [(7, 33, 1), (64, 1000, 3), (256, 3500, 8), (5, 1, 2)])
def test_sharded_softmax_kernels(rows, N, world):
    """Per-shard statistics -> combine == the unsharded ``fr_ce_rows`` on the concatenated classes; the per-shard
    counts add up to its rank exactly."""
    _need_gpu()
    dev = torch.device("cuda", 0)
    st = ops.current_stream_ptr()
    z = (synth.uniform(3, "shard.z", (rows, N * world), -30.0, 30.0)).to(dev)
    z[0, :] = z[0, 0]  # a row of ties
    lab = synth.labels(4, "shard.lab", rows, N * world).to(dev)
    lab[1], lab[2 % rows] = 0, N * world - 1
    stats_all = torch.empty(world, 3, rows, device=dev)
    shards = []
    for w in range(world):
        zs = z[:, w * N:(w + 1) * N].contiguous()
        ll = torch.where((lab >= w * N) & (lab < (w + 1) * N), lab - w * N, torch.full_like(lab, -1))
        ops.call("fr_shard_row_stats", zs, ll, stats_all[w], rows, N, N, st)()
        shards.append(zs)
        # against torch
        torch.testing.assert_close(stats_all[w, 0], zs.max(1).values, rtol=0, atol=0)
        torch.testing.assert_close(stats_all[w, 1], torch.exp(zs - zs.max(1, keepdim=True).values).sum(1), rtol=2e-6,
                                   atol=0)
        own = ll >= 0
        assert torch.equal(stats_all[w, 2][own], zs[own, ll[own]]) and (stats_all[w, 2][~own] == 0).all()
    lse, ce, tl = (torch.empty(rows, device=dev) for _ in range(3))
    ops.call("fr_shard_combine", stats_all, world, rows, lse, ce, tl, st)()
    lse_f, ce_f = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    rank_f = torch.empty(rows, device=dev, dtype=torch.int32)
    ops.call("fr_ce_rows", z, lab, lse_f, ce_f, rank_f, rows, N * world, N * world, st)()
    assert torch.equal(tl, z[torch.arange(rows, device=dev), lab])
    torch.testing.assert_close(lse, lse_f, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(ce, ce_f, rtol=2e-6, atol=4e-6)
    torch.testing.assert_close(lse, torch.logsumexp(z.double(), 1).float(), rtol=2e-6, atol=2e-6)
    total = torch.zeros(rows, device=dev, dtype=torch.int32)
    for zs in shards:
        r = torch.empty(rows, device=dev, dtype=torch.int32)
        ops.call("fr_shard_rank_rows", zs, tl, r, rows, N, N, st)()
        total += r
    assert torch.equal(total, rank_f)


def test_empty_shard_is_refused():
    _need_gpu()
    from frhip._lib import FrhipError
    dev = torch.device("cuda", 0)
    z = torch.zeros(4, 4, device=dev)
    with pytest.raises(FrhipError):
        ops.call("fr_shard_row_stats", z, torch.zeros(4, dtype=torch.int64, device=dev), torch.zeros(3, 4, device=dev),
                 4, 0, 4, ops.current_stream_ptr())()


# (ArcFace, 28000, 256) and (CosFace, 28000, 128): the BUPT-Balancedface head sizes of BASELINE.json configs[2] / configs[3]
# (head/metrics.py:77-140 at N = 28 000), replicated and class-sharded at world size 1, against the oracle (~22 GFLOP on CPU)
@pytest.mark.parametrize("kind,N,B", [("ArcFace", 100, 8), ("CosFace", 1001, 16), ("ArcFace", 7000, 64), ("ArcFace", 875, 24),
                                      ("ArcFace", 28000, 256), ("CosFace", 28000, 128)])
def test_one_rank_equals_replicated_head(kind, N, B):
    """Without a process group the sharded module is the whole head: loss, accuracy and both gradients equal the
    replicated HIP head + FocalLoss + accuracy, and the oracle within the north-star bar."""
    _need_gpu()
    from frhip.sharded_head import ShardedMarginLoss
    from head.metrics import ArcFace, CosFace
    from loss.focal import FocalLoss
    from util.utils import accuracy
    from oracle import irse_ref as O
    D = 512
    w = synth.uniform(11, "sh1.w", (N, D), -0.1, 0.1)
    x0 = synth.uniform(12, "sh1.x", (B, D), -1.0, 1.0)
    y = synth.labels(13, "sh1.y", B, N)
    y[0], y[1] = 0, N - 1
    head = (ArcFace if kind == "ArcFace" else CosFace)(D, N, None).cuda()
    with torch.no_grad():
        head.weight.copy_(w)
    crit = ShardedMarginLoss.from_head(head, gamma=2.0).cuda()
    assert (crit.lo, crit.hi) == (0, N) and crit.grad_scale == 1.0
    x = x0.cuda().requires_grad_(True)
    loss, p1, p5 = crit(x, y.cuda())
    loss.backward()
    xr = x0.cuda().requires_grad_(True)
    logits = head(xr, y.cuda())
    floss, _ = FocalLoss()(logits, y.cuda())
    floss.backward()
    e1, e5 = accuracy(logits.detach(), y.cuda(), topk=(1, 5))
    assert abs(float(loss.detach()) - float(floss)) <= 2e-6 * max(1.0, abs(float(floss)))
    assert float(p1) == float(e1) and float(p5) == float(e5)
    assert (x.grad - xr.grad).norm() <= 1e-5 * xr.grad.norm()
    assert (crit.weight.grad - head.weight.grad).norm() <= 1e-5 * head.weight.grad.norm()
    xo, wo = x0.clone().requires_grad_(True), w.clone().requires_grad_(True)
    fwd = O.arcface_forward if kind == "ArcFace" else O.cosface_forward
    lo = O.focal_loss(fwd(xo, wo, y, s=64.0, m=0.5), y, 2)
    ogx, ogw = torch.autograd.grad(lo, [xo, wo])
    assert abs(float(loss.detach()) - float(lo)) < 1e-3
    assert (x.grad.cpu() - ogx).norm() < 1e-3 * ogx.norm()
    assert (crit.weight.grad.cpu() - ogw).norm() < 1e-3 * ogw.norm()
    with pytest.raises(RuntimeError):
        crit(x.detach(), torch.full((B,), N, device="cuda"))
    with pytest.raises(Exception):
        crit(x0, y)  # host tensors: no CPU fallback


def test_one_rank_through_rccl(tmp_path):
    """The RCCL code path of every exchange (all_gather_into_tensor, reduce_scatter_tensor, all_reduce) at world size 1."""
    _need_gpu()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FRHIP_SHARD_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29561", os.path.join(here, "shard_worker.py")]
    out = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SHARD_WORKER_OK" in out.stdout, out.stdout[-4000:] + out.stderr[-1500:]


def test_two_ranks_sharing_one_gpu():
    _need_gpu()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29563", os.path.join(here, "shard_worker.py")]
    out = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SHARD_WORKER_OK" in out.stdout, out.stdout[-4000:] + out.stderr[-1500:]


def test_bench_with_sharded_head_two_ranks(tmp_path):
    """bench.py --sharded-head as the driver would launch it for N = 2 (both ranks on GPU 0, gloo): the full training step
    with the backbone gradients averaged by frhip.parallel and the head sharded; one JSON line, finite loss."""
    _need_gpu()
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FRHIP_BENCH_ONE_DEVICE="1", FRHIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29567", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "16", "--classes", "1001", "--no-cpu-baseline", "--no-roofline", "--sharded-head"]
    out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp2+class-sharded head"
    assert rec["config"]["final_loss"] == rec["config"]["final_loss"]


def test_train_driver_with_sharded_head(tmp_path):
    """train.py with SHARDED_HEAD=True (one rank): the step runs through ShardedMarginLoss and the head checkpoint keeps
    the reference's layout (key ``weight``, [classes, 512]) with the trained shard gathered into it."""
    _need_gpu()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylegan-for-facerec_amd")
    env = dict(os.environ, PYTHONPATH=root)
    argv = ["train.py", "--config", "configs/config_synthetic_smoke.py", "--synthetic", "12x10", "--max-steps", "3"]
    cfg_patch = ("import configs.config_synthetic_smoke as c; c.configurations[1].update(BATCH_SIZE=20, "
                 "SHARDED_HEAD=True, MODEL_ROOT=r'%s', LOG_ROOT=r'%s')" % (tmp_path / "model", tmp_path / "log"))
    code = "import sys, runpy; sys.argv=%r; %s; runpy.run_path('train.py', run_name='__main__')" % (argv, cfg_patch)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Training Loss" in out.stdout
    files = sorted(os.listdir(tmp_path / "model"))
    heads = [f for f in files if f.startswith("Head_ArcFace_Epoch_1_")]
    assert heads, files
    sd = torch.load(os.path.join(tmp_path / "model", heads[0]), map_location="cpu")
    assert list(sd.keys()) == ["weight"] and tuple(sd["weight"].shape) == (12, 512)
    # three SGD steps moved the weight away from its xavier draw (bound sqrt(6/(12+512)) ~ 0.107)
    assert torch.isfinite(sd["weight"]).all() and float(sd["weight"].abs().max()) > 0


def test_train_driver_two_ranks_sharded_head_and_gpu_input(tmp_path):
    """train.py as launched for two GPUs (torch.distributed.run), both ranks on GPU 0 with gloo: DistributedSampler,
    parameter broadcast, averaged backbone gradients, class-sharded head (12 classes over 2 ranks), GPU-side input
    transform, the collective weight gather before rank 0 writes the reference-layout head checkpoint."""
    _need_gpu()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylegan-for-facerec_amd")
    env = dict(os.environ, PYTHONPATH=root, FRHIP_TRAIN_ONE_DEVICE="1", FRHIP_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    patch = tmp_path / "run_train.py"
    patch.write_text(
        "import sys, runpy\n"
        "import configs.config_synthetic_smoke as c\n"
        "c.configurations[1].update(BATCH_SIZE=10, SHARDED_HEAD=True, GPU_INPUT_PIPELINE=True, MODEL_ROOT=r'%s', "
        "LOG_ROOT=r'%s')\n"
        "sys.argv = ['train.py', '--config', 'configs/config_synthetic_smoke.py', '--synthetic', '12x10', "
        "'--max-steps', '3']\n"
        "runpy.run_path('train.py', run_name='__main__')\n" % (tmp_path / "model", tmp_path / "log"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", str(patch)]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2500:] + out.stderr[-2500:]
    assert "Training Loss" in out.stdout
    files = sorted(os.listdir(tmp_path / "model"))
    heads = [f for f in files if f.startswith("Head_ArcFace_Epoch_1_")]
    assert len(heads) == 1, files  # rank 0 only
    sd = torch.load(os.path.join(tmp_path / "model", heads[0]), map_location="cpu")
    assert tuple(sd["weight"].shape) == (12, 512) and torch.isfinite(sd["weight"]).all()
    # the optimizer checkpoint keeps the reference layout too: the head momentum is gathered to [classes, 512]
    opts = [f for f in files if f.startswith("Optimizer_ArcFace_Epoch_1_")]
    osd = torch.load(os.path.join(tmp_path / "model", opts[0]), map_location="cpu")
    shapes = [tuple(v["momentum_buffer"].shape) for v in osd["state"].values()]
    assert (12, 512) in shapes and (6, 512) not in shapes
    # ... and both ranks resume from those files, each taking its class range of weight and momentum back
    state = [f for f in files if f.startswith("State_ArcFace_Epoch_1_")][0]
    bb = [f for f in files if f.startswith("Backbone_")][0]
    patch2 = tmp_path / "resume_train.py"
    m = tmp_path / "model"
    patch2.write_text(
        "import sys, runpy\n"
        "import configs.config_synthetic_smoke as c\n"
        "c.configurations[1].update(BATCH_SIZE=10, NUM_EPOCH=2, SHARDED_HEAD=True, GPU_INPUT_PIPELINE=True, "
        "MODEL_ROOT=r'%s', LOG_ROOT=r'%s', BACKBONE_RESUME_ROOT=r'%s', HEAD_RESUME_ROOT=r'%s', "
        "OPTIMIZER_RESUME_ROOT=r'%s', STATE_RESUME_ROOT=r'%s')\n"
        "sys.argv = ['train.py', '--config', 'configs/config_synthetic_smoke.py', '--synthetic', '12x10', "
        "'--max-steps', '5']\n"
        "runpy.run_path('train.py', run_name='__main__')\n"
        % (tmp_path / "model2", tmp_path / "log", m / bb, m / heads[0], m / opts[0], m / state))
    cmd[-1] = str(patch2)
    cmd[cmd.index("--master-port") + 1] = "29573"
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2500:] + out.stderr[-2500:]
    # the first run stopped after 3 of its 6 batches: the State file marks epoch 1 as unfinished, the resume repeats it
    # from its first batch (batch counter of that point: warm-up and the log axis do not shift)
    assert "Resuming at epoch 0 batch 0" in out.stdout and "Loading Optimizer Checkpoint" in out.stdout
    assert any(f.startswith("Head_ArcFace_Epoch_1_Batch_5_") for f in os.listdir(tmp_path / "model2"))
