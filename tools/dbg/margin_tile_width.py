#!/usr/bin/env python3
"""ArcFace head forward + backward at a large class count with the logits GEMM on the 64-wide (always chosen for
FR_EPI_MARGIN) and on the 128-wide tile instance (FRHIP_IGEMM_BN=128), advisor r4: what the fixed narrow tile costs.

    python tools/dbg/margin_tile_width.py [classes=85742] [batch=256]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch  # noqa: E402

from frhip import ops  # noqa: E402
from head.metrics import ArcFace  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 85742
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    torch.manual_seed(3)
    head = ArcFace(512, N, None).cuda()
    x = torch.randn(B, 512, device="cuda", requires_grad=True)
    y = torch.randint(0, N, (B,), device="cuda")
    outs = {}
    for width in (64, 128, 64, 128):
        ops.set_option("FRHIP_IGEMM_BN", width)
        for phase in ("fwd", "fwd+bwd"):
            def run():
                out = head(x, y)
                if phase != "fwd":
                    head.weight.grad = None
                    x.grad = None
                    out.backward(torch.ones_like(out) * 1e-3)
                return out
            for _ in range(3):
                out = run()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                out = run()
            t1.record()
            t1.synchronize()
            print("classes %d batch %d tile %3d %-8s %.3f ms" % (N, B, width, phase, t0.elapsed_time(t1) / 20), flush=True)
        outs[width] = out.detach().clone()
    d = (outs[64] - outs[128]).abs().max().item()
    print("max |logit(64) - logit(128)| = %.3g (scale 64)" % d)
    ops.set_option("FRHIP_IGEMM_BN", 0)


if __name__ == "__main__":
    main()
