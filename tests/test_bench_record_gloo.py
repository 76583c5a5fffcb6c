"""world_size-2 ``gloo`` test (CPU) of the part of bench.py's JSON line that makes an N > 1 record self-verifying
(reference data parallelism: train.py:219-222): ``bench.dp_record`` / ``bench.SyncTimer`` -- the code
the benchmark itself runs -- over two host processes that exchange a bucketed arena through frhip.parallel."""
import json
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostDP(object):
    """What bench.dp_record reads of frhip.parallel.DataParallel, over a host arena."""

    def __init__(self, reducer, bucket_bytes):
        self.reducer, self.bucket_bytes = reducer, bucket_bytes

    def synchronize(self):
        self.reducer.synchronize()


def _worker(rank, world, port, q, same_device):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)  # bench.py spawns ranks itself when it is imported as __main__ without it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        repo = os.path.dirname(here)
        sys.path.insert(0, os.path.join(repo, "stylegan-for-facerec_amd"))
        sys.path.insert(0, repo)
        import bench
        from frhip.parallel import BucketedAllReduce
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(n)) for n in (700, 3000, 64, 5000)]
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]
        arena = torch.zeros(sum(sizes))
        slices, off = [], 0
        for p, sz in zip(params, sizes):
            slices.append((p, off, p.numel()))
            off += sz
        dp = _HostDP(BucketedAllReduce(arena, slices, bucket_bytes=8192, gate=2), 8192)
        timer = bench.SyncTimer(dp, use_events=False)
        if same_device:
            bench.device_identity = lambda device: "0000:05:00.0"  # two ranks that report ONE PCI address
        steps = 3
        import time
        t0 = time.perf_counter()
        for step in range(steps + 1):
            timer.on = step > 0  # one warm-up step outside the record, as in bench.timed_loop
            arena.fill_(float(rank + 1))
            for i in range(0, len(params), 2):
                dp.reducer.on_ready(params[i:i + 2])
            timer.synchronize()
            assert torch.allclose(arena, torch.full_like(arena, (1 + world) * 0.5))
        rec = bench.dp_record(rank, world, "cpu", dp, timer, time.perf_counter() - t0, steps)
        assert (rec is None) == (rank != 0)
        q.put((rank, "ok", json.dumps(rec) if rec is not None else ""))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", "%s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _run(same_device):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, same_device)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, msg in res:
        assert status == "ok", "rank %d: %s" % (rank, msg)
    return json.loads([m for r, _s, m in res if r == 0][0])


def test_bench_record_names_its_ranks_devices_library_and_policy():
    rec = _run(same_device=False)
    assert [r[0] for r in rec["ranks"]] == [0, 1] and len(set(r[1] for r in rec["ranks"])) == 2
    assert rec["ranks_distinct_devices"] is True and len(set(rec["rank_pids"])) == 2
    assert rec["collective_backend"] == "gloo" and rec["rccl_version"] is None  # RCCL's version only behind "nccl"
    pol = rec["dp_policy"]
    assert pol["FRHIP_DP_OVERLAP"] == 2 and pol["gate_gradients"] == 2 and pol["buckets"] >= 2
    assert pol["bucket_mb"] == 8192 / 2.0 ** 20 and "one-workgroup-per-CU" in pol["meaning"]
    assert rec["comm_exposed_ms"] is not None and 0 <= rec["comm_exposed_ms"] <= rec["comm_exposed_ms_max"]
    assert len(rec["ms_per_step_by_rank"]) == 2
    assert rec["ms_per_step_min"] == min(rec["ms_per_step_by_rank"]) <= rec["ms_per_step_max"] == max(rec["ms_per_step_by_rank"])


def test_bench_record_flags_ranks_that_share_a_device():
    """Two ranks on one GPU (the FRHIP_BENCH_ONE_DEVICE test hook, or a mis-launched job) cannot pass for a 2-GPU record."""
    rec = _run(same_device=True)
    assert rec["ranks_distinct_devices"] is False and len(rec["ranks"]) == 2
