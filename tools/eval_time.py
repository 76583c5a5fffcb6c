#!/usr/bin/env python3
"""Eval-mode (inference) throughput of the backbone at an RFW-sized workload: one ethnicity subset = 6000 pairs = 12 000
images, flip-TTA doubles the forwards (reference util/utils.py:254-307).  Compares the BN-folded forward-only plan with
the unfolded one (FRHIP_NO_FOLD=1) on the same box.

    python tools/eval_time.py [--batch 256] [--images 24000] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
import torch  # noqa: E402

from backbone.model_irse import IR_50  # noqa: E402
from frhip import synth  # noqa: E402


def run(model, x, n_batches):
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_batches):
            model(x)
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--images", type=int, default=24000)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    m = IR_50([112, 112])
    synth.fill_state_dict(m.state_dict(), 15)
    m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    m = m.cuda().eval()
    x = synth.uniform(3, "eval.x", (a.batch, 3, 112, 112)).cuda()
    nb = (a.images + a.batch - 1) // a.batch
    out = {"workload": "IR-50 eval forward, %d images (one RFW subset with flip-TTA), batch %d, %s" % (nb * a.batch, a.batch, a.dtype)}
    for tag, env in (("folded", None), ("unfolded", "1")):
        if env:
            os.environ["FRHIP_NO_FOLD"] = env
        m._runner[0].plans = {}
        dt = run(m, x, nb)
        launches = len(m._runner[0].plan.pack_list) + len(m._runner[0].plan.fwd_list)
        out[tag] = {"images_per_sec": round(nb * a.batch / dt, 1), "ms_per_batch": round(dt / nb * 1e3, 3),
                    "launches_per_forward": launches, "fold": bool(m._runner[0].plan.fold)}
        os.environ.pop("FRHIP_NO_FOLD", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
