"""Worker of test_gradient_allreduce_overlaps_backward (run under torch.distributed.run, one rank, RCCL).

Proves on the GPU timeline, with HIP events, that the bucketed gradient all-reduce of frhip.parallel runs WHILE the
backward pass is still producing gradients, not after it:

  * E_ready[k]  recorded on the communication stream at the moment bucket k's collective is enqueued (that stream is
                ordered behind the main stream up to the bucket's last gradient and behind the unit's side-stream weight
                gradients: the event fires when the bucket's inputs are final);
  * E_done[k]   recorded on a probe stream that waits for the collective's work handle: the all-reduce has finished;
  * E_end       recorded on the main stream behind the last launch of the backward pass.

FRHIP_DP_OVERLAP=1 enqueues a bucket the moment it is complete; the default (2) holds complete buckets until the backward pass
has left the 7x7 / 14x14 layers (plan.comm_gate) and enqueues them together there.

Asserted: every bucket but the last was ENQUEUED before the host finished enqueuing the backward pass (host order), the
early buckets' inputs were final and their collectives complete on the GPU before E_end, and the gradients equal those of a
run without data parallelism (one rank: AVG over one rank is the identity).
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["FRHIP_FORCE_DP"] = "1"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
    from backbone.model_irse import IR_50
    from frhip import parallel, synth
    from head.metrics import ArcFace
    from loss.focal import FocalLoss
    B, N = 64, 1000
    m = IR_50([112, 112])
    synth.fill_state_dict(m.state_dict(), 15)
    m.output_layer[1].p = 0.0
    m.compute_dtype = torch.bfloat16
    m = m.cuda().train()
    head = ArcFace(512, N, None).cuda()
    x = synth.uniform(16, "ov.x", (B, 3, 112, 112)).cuda()
    y = synth.labels(16, "ov.y", B, N).cuda()

    def step():
        loss, _ = FocalLoss()(head(m(x), y), y)
        for p in list(m.parameters()) + list(head.parameters()):
            p.grad = None
        loss.backward()

    step()  # reference gradients without data parallelism (plan + kernels warm)
    torch.cuda.synchronize()
    ref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    dp = parallel.DataParallel(m, head, bucket_bytes=16 << 20)
    probe = torch.cuda.Stream()
    log = []  # (tag, numel, E_ready, E_done, backward launches enqueued so far)
    plan_holder = {}
    orig = parallel.BucketedAllReduce._launch
    counter = {"n": 0}
    from frhip import ops
    orig_call = ops.Launch.__call__

    def counting_call(self):
        counter["n"] += 1
        orig_call(self)

    def traced(self, t):
        e_ready, e_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e_ready.record(torch.cuda.current_stream())
        before = len(self.works)
        orig(self, t)
        assert len(self.works) == before + 1, "the collective was not enqueued (world 1 without FRHIP_FORCE_DP?)"
        work = self.works[-1][0]
        with torch.cuda.stream(probe):
            work.wait()
            e_done.record(probe)
        log.append(("head" if self is dp.extra else "bucket", t.numel(), e_ready, e_done, counter["n"]))

    parallel.BucketedAllReduce._launch = traced
    ops.Launch.__call__ = counting_call
    try:
        for _ in range(2):  # second pass: plan, reducer and events are reused
            log.clear()
            counter["n"] = 0
            e_start, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            loss, _ = FocalLoss()(head(m(x), y), y)
            for p in list(m.parameters()) + list(head.parameters()):
                p.grad = None
            counter["n"] = 0
            e_start.record(torch.cuda.current_stream())
            loss.backward()
            total_launches = counter["n"]
            e_end.record(torch.cuda.current_stream())
            n_before_sync = len(log)
            dp.synchronize()
            torch.cuda.synchronize()
    finally:
        parallel.BucketedAllReduce._launch = orig
        ops.Launch.__call__ = orig_call

    buckets = [r for r in log if r[0] == "bucket"]
    assert len(buckets) >= 4, "expected several gradient buckets, got %d" % len(buckets)
    assert any(r[0] == "head" for r in log), "the head weight's own all-reduce is missing"
    bwd_ms = e_start.elapsed_time(e_end)
    # host order: all buckets but the last are enqueued while backward launches are still being enqueued
    early = [r for r in buckets if r[4] < total_launches]
    assert len(early) >= len(buckets) - 1 and n_before_sync == len(log), (len(early), len(buckets), n_before_sync, len(log))
    # GPU timeline: inputs final / collective complete before the backward pass ends
    ready_lead = [r[2].elapsed_time(e_end) for r in buckets]   # ms by which E_ready precedes E_end
    done_lead = [r[3].elapsed_time(e_end) for r in buckets]
    print("backward %.2f ms; buckets: %s" % (bwd_ms, ", ".join(
        "%.1fMB ready %.2f / done %.2f ms before the end" % (r[1] * 4 / 1e6, a, b) for r, a, b in zip(buckets, ready_lead, done_lead))))
    policy = int(os.environ.get("FRHIP_DP_OVERLAP", "2"))
    gate = dp.runner.plan.comm_gate
    assert 0 < gate < len(dp.runner.plan.arena_slices), gate
    if policy == 1:
        assert ready_lead[0] > 0.5 * bwd_ms, "the first bucket (output layer) must be ready in the first half of backward"
    else:
        # gated: nothing is enqueued beside the one-workgroup-per-CU layers; the buckets complete by then start together
        assert 0.1 * bwd_ms < ready_lead[0] < 0.6 * bwd_ms, "first bucket enqueued %.2f of %.2f ms before the end" % (ready_lead[0], bwd_ms)
        held = [r for r in log if r[4] == log[0][4]]
        assert len(held) >= 3 and any(r[0] == "head" for r in held), "the gate released %d collectives at once" % len(held)
    overlapped = sum(1 for d in done_lead[:-1] if d > 0)
    assert overlapped >= len(buckets) - 2, "collectives did not complete under the backward pass: %s" % done_lead
    assert all(a > b for a, b in zip(ready_lead, done_lead))
    bad = [n for n, p in m.named_parameters() if not torch.equal(p.grad, ref[n])]
    assert not bad, "gradients changed under the one-rank all-reduce: %s" % bad[:5]
    print("OVERLAP_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
