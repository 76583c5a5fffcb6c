"""Worker for tests/test_gpu_model.py::test_two_rank_gradients_are_the_rank_average (run under torch.distributed.run,
two ranks sharing GPU 0, collectives through gloo).  Each rank trains on its own batch through frhip.parallel.DataParallel;
the averaged gradients must equal the mean of the two single-rank gradients, which every rank recomputes locally, and the
parameters must stay identical across ranks after optimizer steps."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "stylegan-for-facerec_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(seed_w=15):
    from backbone.model_irse import IR_50
    from frhip import synth
    from head.metrics import ArcFace
    m = IR_50([112, 112])
    synth.fill_state_dict(m.state_dict(), seed_w)
    m.output_layer[1].p = 0.0
    m.compute_dtype = torch.bfloat16
    m = m.cuda().train()
    head = ArcFace(512, 100, None).cuda()
    with torch.no_grad():
        head.weight.copy_(synth.uniform(16, "dp.head", (100, 512), -0.1, 0.1))
    return m, head


def grads_for(rank_data, dp=None, models=None):
    from frhip import synth
    from loss.focal import FocalLoss
    m, head = models if models is not None else build()
    x = synth.uniform(50 + rank_data, "dp.x", (6, 3, 112, 112)).cuda()
    y = synth.labels(50 + rank_data, "dp.y", 6, 100).cuda()
    for p in list(m.parameters()) + list(head.parameters()):
        p.grad = None
    loss, _ = FocalLoss()(head(m(x), y), y)
    loss.backward()
    if dp is not None:
        dp.synchronize()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    named["head.weight"] = head.weight
    return {n: p.grad.detach().clone() for n, p in named.items()}, (m, head)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frhip.optim import SGD
    from frhip.parallel import DataParallel
    from util.utils import separate_irse_bn_paras
    # single-rank gradients of BOTH batches, computed locally (fresh models, identical weights)
    g0, _ = grads_for(0)
    g1, _ = grads_for(1)
    # the data-parallel run: this rank's batch, gradients averaged over the two ranks
    m, head = build(seed_w=15 + rank)          # different weights per rank: the constructor broadcast must fix that
    dp = DataParallel(m, head)
    gd, _ = grads_for(rank, dp=dp, models=(m, head))
    bad = [n for n in gd if not torch.equal(gd[n], (g0[n] + g1[n]) / 2)]
    assert not bad, "rank %d: averaged gradients differ from the mean of the single-rank gradients: %s" % (rank, bad[:5])
    # two optimizer steps keep the replicas identical
    bn, wo = separate_irse_bn_paras(m)
    opt = SGD([{"params": wo + list(head.parameters()), "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
    for _ in range(2):
        grads_for(rank, dp=dp, models=(m, head))
        opt.step()
    torch.cuda.synchronize()
    for n, p in list(m.named_parameters()) + [("head.weight", head.weight)]:
        flat = p.detach().float().reshape(-1).cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "parameter %s diverged between ranks" % n
    if rank == 0:
        print("DP_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
