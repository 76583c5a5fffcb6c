#!/usr/bin/env python3
"""What one main -> side stream dependency edge costs the MAIN stream (round 6).  The backward pass sets three edges per
residual unit (engine.py: record on the main stream, wait on the weight-gradient stream), and the two-stream timelines
show a ~6-us gap on the main stream at each of them.  Variants timed here, N kernels of ~20 us on the main stream:
  none        no edges (the floor)
  event       torch.cuda.Event record on main + wait on side + a tiny side kernel (what the engine does)
  event_nowait  record only, nobody waits
  value       hipStreamWriteValue32 on main + hipStreamWaitValue32 on side (stream memory operations, no signal object)
  waitback    event recorded on the SIDE stream, waited for by the main stream (the buffer-reuse edge)
"""
import ctypes
import os
import sys
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda")
    x = torch.randn(64 * 1024 * 1024 // 4, device=dev)   # 64 MB: an elementwise pass of ~20-25 us
    y = torch.empty_like(x)
    small = torch.zeros(1024, device=dev)
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream(priority=1)
    flag = torch.zeros(64, device=dev, dtype=torch.int32)
    sig = ctypes.c_void_p()
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    rc = hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2)  # hipMallocSignalMemory
    have_sig = rc == 0
    if not have_sig:
        hip.hipGetLastError()  # clear the sticky error

    def k0():
        torch.mul(x, 1.0001, out=y)

    def run(variant):
        evs = [torch.cuda.Event() for _ in range(n)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for i in range(n):
            k0()
            if variant == "event":
                evs[i].record(s0)
                s1.wait_event(evs[i])
                with torch.cuda.stream(s1):
                    small.add_(1.0)
            elif variant == "event_nowait":
                evs[i].record(s0)
            elif variant == "value":
                p = sig if have_sig else ctypes.c_void_p(flag.data_ptr())
                hip.hipStreamWriteValue32(ctypes.c_void_p(s0.cuda_stream), p, ctypes.c_uint32(i + 1), 0)
                hip.hipStreamWaitValue32(ctypes.c_void_p(s1.cuda_stream), p, ctypes.c_uint32(i + 1), 0,
                                         ctypes.c_uint32(0xFFFFFFFF))  # flags 0 = hipStreamWaitValueGte
                with torch.cuda.stream(s1):
                    small.add_(1.0)
            elif variant == "waitback":
                with torch.cuda.stream(s1):
                    small.add_(1.0)
                    evs[i].record(s1)
                s0.wait_event(evs[i])
        e1.record(s0)
        torch.cuda.synchronize()
        host = time.perf_counter() - t0
        return e0.elapsed_time(e1) * 1e3 / n, host * 1e6 / n

    for v in ("none", "event", "event_nowait", "value", "waitback", "none", "event", "value"):
        if v == "value" and os.environ.get("NO_VALUE"):
            continue
        try:
            run(v)
            gpu, host = run(v)
            print("%-13s main stream %.2f us per kernel (host enqueue %.2f us)   signal memory: %s" % (v, gpu, host, have_sig), flush=True)
        except Exception as e:  # noqa: BLE001
            print(v, "failed:", e)


if __name__ == "__main__":
    main()
