#!/usr/bin/env python3
"""Step time of the headline workload while K compute units are held by another kernel.

At batch 256 the 14x14 / 28x28 strip launches put exactly one workgroup on every CU; a co-resident RCCL kernel (data-parallel
all-reduce during the backward pass) takes some CUs away, and such a launch then needs a second round of workgroups.  There is
no multi-GPU box in this project's pool, so this tool emulates the collective on ONE GPU: tools/cu_hog.hip parks K
workgroups (96 KB of LDS each: nothing LDS-heavy fits beside them) on a third stream for the length of the timed loop.

    python tools/cu_hog.py --hog 8 --steps 20                      # one measurement, prints ms per step
    FRHIP_SPLIT_STRIPS=1 python tools/cu_hog.py --hog 8            # the half-channel strip instances (proportional cost)
    bash tools/hog_matrix.sh                                       # K x variant table -> profiles/

    python tools/cu_hog.py --comm 16 --comm-gbps 150               # every gradient collective of frhip.parallel replaced by a
                                                                   # 16-CU hold of bytes / 150 GB/s on the communication
                                                                   # stream, under FRHIP_DP_OVERLAP = 2 (gate, default) / 1 / 0

Kernel-selection switches are read when the library / plan is first used, so every variant is its own process.
"""
import argparse
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (sets GPU_MAX_HW_QUEUES before torch starts the HIP runtime)
import torch  # noqa: E402


def hog_lib():
    so = os.path.join(REPO, "gpurun_out", "libcuhog.so")
    src = os.path.join(REPO, "tools", "cu_hog.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hog", type=int, default=0, help="CUs held during the timed loop")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--comm", type=int, default=0, help="CUs held per emulated collective (frhip.parallel's launch points)")
    ap.add_argument("--comm-gbps", type=float, default=150.0, help="emulated all-reduce rate: a collective holds bytes / rate")
    a = ap.parse_args()
    device = torch.device("cuda:0")
    args = argparse.Namespace(dtype="bf16", sharded_head=False, resident_batches=8, model="IR_50", head="ArcFace",
                              classes=7000, batch=a.batch)
    model, head, loss_fn, opt, xs, ys = bench.build_job(args, device, 0)
    dp, held_ms = None, [0.0, 0]
    if a.comm > 0:
        # the data-parallel wrapper of a multi-GPU run on ONE GPU: same buckets, same launch points and streams; the
        # collective itself is a hold of a.comm CUs for bytes / rate
        from frhip import parallel
        lib = hog_lib()

        class Held(object):
            def __init__(self, ev):
                self.ev = ev

            def wait(self):
                torch.cuda.current_stream().wait_event(self.ev)

        def launch(self, t):
            sec = t.numel() * t.element_size() / (a.comm_gbps * 1e9)
            held_ms[0] += sec * 1e3
            held_ms[1] += 1
            st = torch.cuda.current_stream()
            assert lib.cu_hog_launch(a.comm, sec, ctypes.c_void_p(st.cuda_stream)) == 0
            ev = torch.cuda.Event()
            ev.record(st)
            self.works.append((Held(ev), None))

        parallel.BucketedAllReduce._launch = launch
        dp = parallel.DataParallel(model, head)
        dp.extra = parallel.BucketedAllReduce(torch.zeros(0), [], None, gate=1)
    step = bench.make_step(model, head, loss_fn, opt, dp)
    for i in range(a.warmup):
        step(xs[i % len(xs)], ys[i % len(ys)])
    torch.cuda.synchronize()
    # a rough step time sizes the hog: it must outlast the timed loop, bounded
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step(xs[0], ys[0])
    e1.record()
    torch.cuda.synchronize()
    est = e0.elapsed_time(e1) * 1e-3
    hog_stream = torch.cuda.Stream(priority=-1)
    if a.hog > 0:
        lib = hog_lib()
        rc = lib.cu_hog_launch(a.hog, est * a.steps * 2.0 + 0.05, ctypes.c_void_p(hog_stream.cuda_stream))
        assert rc == 0, rc
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(a.steps):
        step(xs[i % len(xs)], ys[i % len(ys)])
    t1.record()
    t1.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    torch.cuda.synchronize()
    sw = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items())
                  if k.startswith("FRHIP_") and k not in ("FRHIP_COMPUTE_DTYPE",)) or "default"
    if a.comm > 0:
        n = a.steps + a.warmup + 1
        print("collectives = %d CUs held for bytes / %.0f GB/s (%d per step, %.2f ms per step)  %-24s %.3f ms per step" % (
            a.comm, a.comm_gbps, held_ms[1] // n, held_ms[0] / n, sw, ms))
    else:
        print("hog %3d CUs  %-40s %.3f ms per step" % (a.hog, sw, ms))


if __name__ == "__main__":
    main()
