"""Worker for tests/test_gpu_sharded_head.py::test_two_ranks_sharing_one_gpu (torch.distributed.run, two ranks on GPU 0,
``gloo`` collectives on device tensors).  Each rank feeds its own features / labels to frhip.sharded_head.ShardedMarginLoss
(HIP kernels); the expected values are the REPLICATED HIP head + focal loss over the concatenated batch with the full
weight on the same GPU, and the oracle (CPU) on the same numbers."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("FRHIP_SHARD_BACKEND", "gloo"))  # "nccl" = RCCL, one rank only on one GPU
    rank, world = dist.get_rank(), dist.get_world_size()
    from frhip import synth
    from frhip.sharded_head import ShardedMarginLoss, class_range
    from head.metrics import ArcFace, CosFace
    from loss.focal import FocalLoss
    from util.utils import accuracy
    from oracle import irse_ref as O
    for kind, N, B in (("ArcFace", 1001, 12), ("CosFace", 7000, 32)):
        D = 512
        w_full = synth.uniform(31, "shard.w." + kind, (N, D), -0.1, 0.1)
        xs = [synth.uniform(40 + r, "shard.x", (B, D), -1.0, 1.0) for r in range(world)]
        labs = [synth.labels(40 + r, "shard.y", B, N) for r in range(world)]
        labs[0][0], labs[-1][-1] = 0, N - 1
        crit = ShardedMarginLoss(D, N, kind, s=64.0, m=0.5, gamma=2.0, full_weight=w_full).cuda()
        lo, hi = class_range(N, world, rank)
        x = xs[rank].cuda().requires_grad_(True)
        loss, p1, p5 = crit(x, labs[rank].cuda())
        loss.backward()
        # replicated head over the concatenated batch (HIP)
        head = (ArcFace if kind == "ArcFace" else CosFace)(D, N, None).cuda()
        with torch.no_grad():
            head.weight.copy_(w_full)
        xc = torch.cat(xs).cuda().requires_grad_(True)
        yc = torch.cat(labs).cuda()
        logits = head(xc, yc)
        floss, _ = FocalLoss()(logits, yc)
        floss.backward()
        e1, e5 = accuracy(logits.detach(), yc, topk=(1, 5))
        assert abs(float(loss.detach()) - float(floss)) <= 2e-6 * max(1.0, abs(float(floss))), (float(loss.detach()), float(floss))
        assert float(p1) == float(e1) and float(p5) == float(e5), (float(p1), float(e1), float(p5), float(e5))
        gx_e = xc.grad[rank * B:(rank + 1) * B] * world
        err = (x.grad - gx_e).norm() / gx_e.norm()
        assert err < 1e-5, ("gx", float(err))
        gw_e = head.weight.grad[lo:hi]
        err = (crit.weight.grad - gw_e).norm() / gw_e.norm()
        assert err < 1e-5, ("gw", float(err))
        # oracle on the same numbers (fp32 CPU): the 1e-3 bar of the north star, gradients norm-wise
        xo = torch.cat(xs).clone().requires_grad_(True)
        wo = w_full.clone().requires_grad_(True)
        fwd = O.arcface_forward if kind == "ArcFace" else O.cosface_forward
        lo_ = O.focal_loss(fwd(xo, wo, torch.cat(labs), s=64.0, m=0.5), torch.cat(labs), 2)
        ogx, ogw = torch.autograd.grad(lo_, [xo, wo])
        assert abs(float(loss.detach()) - float(lo_)) < 1e-3, (float(loss.detach()), float(lo_))
        err = (x.grad.cpu() / world - ogx[rank * B:(rank + 1) * B]).norm() / ogx[rank * B:(rank + 1) * B].norm()
        assert err < 1e-3, ("oracle gx", float(err))
        err = (crit.weight.grad.cpu() - ogw[lo:hi]).norm() / ogw[lo:hi].norm()
        assert err < 1e-3, ("oracle gw", float(err))
        # identical loss bits on every rank; ragged gather restores the full weight
        both = [torch.zeros((), device="cuda") for _ in range(world)]
        dist.all_gather(both, loss.detach())
        assert all(torch.equal(both[0], t) for t in both)
        assert torch.equal(crit.gather_weight().cpu(), w_full)
    dist.barrier()
    if rank == 0:
        print("SHARD_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception:  # noqa: BLE001 -- the launcher's summary hides the traceback
        import traceback
        print("SHARD_WORKER_FAILED\n" + traceback.format_exc(), flush=True)
        raise
