"""How many dependency edges of a step ride on their producer's completion signal (fr_finish_stop_event == 1)?"""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "stylegan-for-facerec_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from frhip import _lib, ops  # noqa: E402

args = bench.argparse.Namespace(model="IR_50", head="ArcFace", classes=7000, batch=int(os.environ.get("B", "256")), dtype="bf16",
                                sharded_head=False, resident_batches=2)
model, head, loss_fn, opt, xs, ys = bench.build_job(args, torch.device("cuda"), 0)
step = bench.make_step(model, head, loss_fn, opt, None)
step(xs[0], ys[0])
hist = collections.Counter()
orig = ops.Launch.__call__


def counted(self):
    if self.stop_event is not None:
        _lib.lib.fr_arm_stop_event(self.stop_handle)
        rc = self.fn(*self.args)
        n = _lib.lib.fr_finish_stop_event(self.stop_stream)
        hist[(self.name, n)] += 1
        assert rc == 0
    else:
        orig(self)


ops.Launch.__call__ = counted
step(xs[1], ys[1])
torch.cuda.synchronize()
for k, v in sorted(hist.items()):
    print(k, v)
plan = model._runner[0].plan
print("event records left in the backward list:", sum(1 for l in plan.bwd_list if l.__class__.__name__ == "_EvRecord"),
      "waits:", sum(1 for l in plan.bwd_list if l.__class__.__name__ == "_EvWait"))
