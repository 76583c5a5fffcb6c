"""Class-sharded margin head (frhip/sharded_head.py) on CPU: the class partition, label localisation, and -- at
world_size 2 and 3 over ``gloo`` -- the whole collective choreography (feature/label all-gather, statistics all-gather,
rank-count all-reduce, feature-gradient reduce-scatter, ragged weight gather) with the oracle standing in for the HIP
kernels (tests/shard_ref.py).  Expected values: the oracle's head + focal loss + accuracy on the concatenated batch with
the full weight, gradients by autograd."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_class_range_partitions_the_classes():
    from frhip.sharded_head import class_range
    for n, w in ((100, 1), (101, 2), (7000, 8), (28000, 8), (10, 3), (8, 8)):
        spans = [class_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        class_range(3, 4, 0)
    with pytest.raises(ValueError):
        class_range(10, 2, 2)


def test_localize_labels():
    from frhip.sharded_head import localize_labels
    lab = torch.tensor([0, 49, 50, 99, 51])
    assert localize_labels(lab, 0, 50).tolist() == [0, 49, -1, -1, -1]
    assert localize_labels(lab, 50, 100).tolist() == [-1, -1, 0, 49, 1]


def _expected(xs, labs, w_full, kind, s, m, gamma):
    from oracle import irse_ref as O
    x = torch.cat(xs).clone().requires_grad_(True)
    w = w_full.clone().requires_grad_(True)
    lab = torch.cat(labs)
    fwd = O.arcface_forward if kind == "ArcFace" else O.cosface_forward
    logits = fwd(x, w, lab, s=s, m=m)
    loss = O.focal_loss(logits, lab, gamma)
    gx, gw = torch.autograd.grad(loss, [x, w])
    p1, p5 = O.topk_accuracy(logits.detach(), lab)
    return loss.detach(), gx, gw, float(p1), float(p5)


def _worker(rank, world, port, kind, n_classes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frhip import functional as FRF
        from frhip.sharded_head import ShardedMarginLoss, class_range
        from shard_ref import OracleKernels
        B, D, s, m, gamma = 5, 32, 16.0, 0.5, 2.0
        g = torch.Generator().manual_seed(7)
        w_full = torch.randn(n_classes, D, generator=g) * 0.3
        xs = [torch.randn(B, D, generator=g) for _ in range(world)]
        labs = [torch.randint(0, n_classes, (B,), generator=g) for _ in range(world)]
        labs[0][0], labs[-1][-1] = 0, n_classes - 1  # first / last class, owned by the first / last rank
        labs[0][1] = labs[0][0]                      # a repeated label
        crit = ShardedMarginLoss(D, n_classes, kind, s=s, m=m, gamma=gamma, full_weight=w_full,
                                 kernels=OracleKernels())
        lo, hi = class_range(n_classes, world, rank)
        assert (crit.lo, crit.hi) == (lo, hi) and crit.weight.shape == (hi - lo, D)
        x = xs[rank].clone().requires_grad_(True)
        loss, p1, p5 = crit(x, labs[rank])
        loss.backward()
        e_loss, e_gx, e_gw, e_p1, e_p5 = _expected(xs, labs, w_full, kind, s, m, gamma)
        assert abs(float(loss.detach()) - float(e_loss)) < 1e-5 * max(1.0, abs(float(e_loss))), (float(loss.detach()), float(e_loss))
        assert float(p1) == pytest.approx(e_p1) and float(p5) == pytest.approx(e_p5)
        # feature gradient: the global-loss gradient of this rank's rows, times world (consumed by an averaging reducer)
        torch.testing.assert_close(x.grad / world, e_gx[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(crit.weight.grad, e_gw[lo:hi], rtol=1e-4, atol=1e-6)
        # every rank sees the same loss bits (combined in rank order)
        losses = [torch.zeros(()) for _ in range(world)]
        dist.all_gather(losses, loss.detach())
        assert all(torch.equal(losses[0], t) for t in losses)
        # ragged weight gather restores the reference layout
        assert torch.equal(crit.gather_weight(), w_full)
        # out-of-range label: the reference's scatter_ error
        with pytest.raises(RuntimeError):
            crit(x.detach(), torch.full((B,), n_classes))
        assert FRF.CHECK_LABELS
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,n_classes", [(2, "ArcFace", 101), (3, "CosFace", 20), (2, "CosFace", 64)])
def test_sharded_head_matches_full_batch_head(world, kind, n_classes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_classes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
