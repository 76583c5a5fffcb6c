import argparse, os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import bench, torch
mode = sys.argv[1]
device = torch.device("cuda:0")
args = argparse.Namespace(dtype="bf16", sharded_head=False, resident_batches=8, model="IR_50", head="ArcFace", classes=7000, batch=256)
model, head, loss_fn, opt, xs, ys = bench.build_job(args, device, 0)
dp = None
runner = model._runner[0]
if mode == "noop_cb":
    runner.on_grads_ready = lambda params: None
elif mode == "dp_nolaunch":
    from frhip import parallel
    parallel.BucketedAllReduce._launch = lambda self, t: None
    dp = parallel.DataParallel(model, head)
    dp.extra = parallel.BucketedAllReduce(torch.zeros(0), [], None, gate=1)
elif mode == "dp_fence_only":
    from frhip import parallel
    def launch(self, t): pass
    parallel.BucketedAllReduce._launch = launch
    dp = parallel.DataParallel(model, head)
step = bench.make_step(model, head, loss_fn, opt, dp)
for i in range(8): step(xs[i % 8], ys[i % 8])
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(30): step(xs[i % 8], ys[i % 8])
    t1.record(); t1.synchronize()
    best.append(t0.elapsed_time(t1) / 30)
print("%-14s %s ms per step" % (mode, " ".join("%.3f" % b for b in best)))
