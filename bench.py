#!/usr/bin/env python3
"""Headline benchmark: Stage-3 training step throughput (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): IR-50 + ArcFace(512 -> 7000 ids, s=64, m=0.5) + FocalLoss(gamma=2) +
SGD(lr 0.03, momentum 0.9, wd 2e-3 on the non-BN group), synthetic 112x112x3 batches resident in HBM, 256 images per
GPU, bf16 storage / fp32 accumulate.  A step = forward, margin head, loss, top-1/5 rank pass, backward, gradient
all-reduce (N > 1), optimizer step -- everything the reference loop does per iteration (train.py:287-316) except
its three host syncs.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      dominant kernel FAMILY (by time; prologue variants of one template counted together) of one
                event-instrumented step after the timed region: algorithmic FLOPs of its launches / their summed
                duration, against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md); ``all_mfma`` = the same
                quotient over EVERY MFMA launch of the step (convolutions, weight gradients, GEMMs), each timed ALONE on
                one stream (the weight-gradient launches cover half of the CUs by design: half rate alone); ``step_frac`` =
                the whole step incl. the HBM-bound channel-wise passes; ``traffic`` = HBM bytes per launch of the
                dominant family from the committed rocprofv3 --pmc passes (the newest profiles/rNN_pmc_kernels.json)
  cpu_baseline  the CPU oracle (oracle/irse_ref.py, a port of the reference's PyTorch path) timed on the host
                cores of this box on a bounded sample of the same workload (N = 1, rank 0 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "stylegan-for-facerec_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# The weight gradients run on a second HIP stream.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware
# queues round-robin; once RCCL has created its own streams the side stream can land on the SAME queue as the main
# stream and the two serialise (measured: 19.7 instead of 17.8 ms per step with one rank).  Must be set before the HIP
# runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--classes", type=int, default=7000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="IR_50", choices=["IR_50", "IR_101", "IR_152", "IR_SE_50", "IR_SE_101", "IR_SE_152",
                                                         "pSp"],
                    help="IR_50 is the BASELINE.json workload; the others are for kernel tables of the SE / deep variants")
    ap.add_argument("--head", default="ArcFace", choices=["ArcFace", "CosFace"])
    ap.add_argument("--resident-batches", type=int, default=16,
                    help="distinct synthetic batches kept in HBM and rotated through the timed loop (one batch alone is "
                         "memorised within the warm-up: the loss, and with it every gradient, goes to ~0)")
    ap.add_argument("--sharded-head", action="store_true",
                    help="class-sharded ArcFace + focal loss over the ranks (frhip/sharded_head.py) instead of the "
                         "replicated head; off by default: the BASELINE configs replicate the head")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short timings of the other BASELINE.json configs after the headline")
    ap.add_argument("--kernel-table", default="", help="write the per-launch timing table of the instrumented step")
    ap.add_argument("--clock-log", default="", help="write every shader-clock / power sample of the timed loop (rank 0)")
    return ap.parse_args()


def spawn_ranks(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks as a CHILD
    ``torch.distributed.run`` (one process per GPU, RCCL) and relay rank 0's JSON line and the exit code.  Runs before
    torch is imported, i.e. before anything in this process has touched the GPU, and never re-execs."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=REPO)
    for line in proc.stdout:  # rank 0's contract line goes to our stdout, anything else the ranks print to stderr
        if line.startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _early = parse()
    if _early.gpus > 1:
        sys.exit(spawn_ranks(_early))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA == vector peak
IR50_FLOPS_PER_IMG = 37.7356e9   # conv + Linear, fwd+bwd, 2 flop/MAC (measured on the reference)
FLOPS_PER_IMG = {"IR_50": IR50_FLOPS_PER_IMG, "IR_SE_50": 37.7356e9, "IR_SE_101": 72.4194e9, "IR_101": 72.4194e9,
                 "pSp": 37.8236e9}


def synthetic_batches(n, batch, classes, device, rank):
    """n distinct synthetic batches resident in HBM: x ~ U(-1, 1) [batch, 3, 112, 112] fp32, labels uniform.  Batch 0 is
    the counter-based host stream (frhip.synth, regenerates bit-identically anywhere); the others come from a seeded
    device generator (a host stream of 16 x 9.6 M floats would cost seconds of start-up per rank)."""
    from frhip import synth
    xs = [synth.uniform(1000 + rank, "bench.x", (batch, 3, 112, 112)).to(device)]
    ys = [synth.labels(1000 + rank, "bench.y", batch, classes).to(device)]
    g = torch.Generator(device=device)
    g.manual_seed(900 + 7919 * rank)
    for _ in range(1, n):
        xs.append(torch.rand(batch, 3, 112, 112, device=device, generator=g) * 2.0 - 1.0)
        ys.append(torch.randint(0, classes, (batch,), device=device, generator=g))
    return xs, ys


def build_job(args, device, rank):
    """args: model, head, classes, batch, dtype, sharded_head, resident_batches."""
    import backbone.model_irse as irse
    import head.metrics as metrics
    from frhip import synth
    from frhip.optim import SGD
    from loss.focal import FocalLoss
    from util.utils import separate_irse_bn_paras
    torch.manual_seed(900)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.model == "pSp":  # BASELINE configs[0] / [4]: IR-SE-50 trunk, 6-channel stem fed with the average image
        from backbone.restyle_psp import pSp
        model = pSp(size=112, encoder_type="BackboneEncoder", avg_image=synth.uniform(2, "bench.avg", (3, 112, 112)))
        synth.fill_state_dict(model.state_dict(), 15)
        model.encoder.compute_dtype = cdt
    else:
        model = getattr(irse, args.model)([112, 112])  # BASELINE metric: IR_50
        synth.fill_state_dict(model.state_dict(), 15)   # identical "trained-looking" weights on every rank
        model.compute_dtype = cdt
    model = model.to(device).train()
    head = getattr(metrics, getattr(args, "head", "ArcFace"))(512, args.classes, None, s=64.0)
    with torch.no_grad():
        head.weight.copy_(synth.uniform(16, "bench.head", (args.classes, 512), -0.05, 0.05))
    head = head.to(device).train()
    if getattr(args, "sharded_head", False):
        from frhip.sharded_head import ShardedMarginLoss
        head = ShardedMarginLoss.from_head(head, gamma=2.0).to(device)  # this rank's class range of the same weight
    bn, wo = separate_irse_bn_paras(model)
    _, hwo = separate_irse_bn_paras(head)
    opt = SGD([{"params": wo + hwo, "weight_decay": 2e-3}, {"params": bn}], lr=0.03, momentum=0.9)
    xs, ys = synthetic_batches(max(1, getattr(args, "resident_batches", 16)), args.batch, args.classes, device, rank)
    return model, head, FocalLoss(), opt, xs, ys


def make_step(model, head, loss_fn, opt, dp):
    from frhip import functional as FRF
    from util.utils import accuracy
    FRF.CHECK_LABELS = False  # synthetic labels are in range by construction; the check is a host sync

    def sharded_step(x, y):
        loss, p1, p5 = head(model(x), y)  # global-batch loss / accuracy; the weight-shard gradient needs no exchange
        opt.zero_grad()
        loss.backward()
        if dp is not None:
            dp.synchronize()
        opt.step()
        return loss, (p1, p5)

    if head.__class__.__name__ == "ShardedMarginLoss":
        return sharded_step

    def step(x, y):
        feats = model(x)
        logits = head(feats, y)
        loss, _ = loss_fn(logits, y)
        prec = accuracy(logits.data, y, topk=(1, 5))
        opt.zero_grad()
        loss.backward()
        if dp is not None:
            dp.synchronize()
        opt.step()
        return loss, prec
    return step


def conv_flops(launch):
    """Algorithmic FLOPs of one fr_conv_igemm / fr_conv3x3_strip launch (2 per MAC actually needed)."""
    a = launch.keep[0]
    if a.mode == 2:  # one parity class of a stride-2 data gradient: (1|2) x (1|2) of the 9 taps, a quarter of the rows
        taps = 9 if a.par_h < 0 else (2 if a.par_h else 1) * (2 if a.par_w else 1)  # par -1: all four classes
        return 2.0 * a.B * (a.RH // 2) * (a.RW // 2) * a.N * taps * a.SC
    M = a.B * a.RH * a.RW
    return 2.0 * M * a.N * a.KH * a.KW * a.SC


def wgrad_flops(launch):
    a = launch.keep[0]
    return 2.0 * a.B * a.GH * a.GW * a.Cout * a.SC * a.KH * a.KW


def instrumented_step(step, x, y, dtype_name):
    """One extra step with a HIP event pair around every launch of the C ABI (same stream as the launches)."""
    from frhip import ops
    records = []
    orig_call = ops.Launch.__call__

    def timed_call(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(self)
        e1.record()
        records.append((self, e0, e1))

    ops.Launch.__call__ = timed_call
    try:
        step(x, y)
        torch.cuda.synchronize()
    finally:
        ops.Launch.__call__ = orig_call
    fams = {}
    detail = []
    for launch, e0, e1 in records:
        ms = e0.elapsed_time(e1)
        name, flops = launch.name, 0.0
        if launch.name in ("fr_conv_igemm", "fr_conv_wgrad"):
            a = launch.keep[0]
            if launch.name == "fr_conv_igemm":
                detail.append(("igemm M=%d N=%d K=%dx%d s%d mode%d pro%d epi%d splitk%d" % (
                    a.B * a.RH * a.RW, a.N, a.KH * a.KW, a.SC, a.stride, a.mode, a.pro, a.epi, a.splitk), round(ms, 4)))
            else:
                detail.append(("wgrad P=%d Cout=%d taps%d Cin=%d s%d pro%d nsplit%d" % (
                    a.B * a.GH * a.GW, a.Cout, a.KH * a.KW, a.SC, a.stride, a.pro, a.nsplit), round(ms, 4)))
        if launch.name in ("fr_bn_apply", "fr_bn_bwd_reduce", "fr_bn_bwd_apply"):
            a = launch.keep[0]
            elt = 2 if launch.args[1] == 1 else 4
            if launch.name == "fr_bn_apply":
                rows, passes = a.B * a.H * a.W, 2 + (1 if a.res_kind else 0)
            else:
                rows = a.rows
                passes = 2 if launch.name == "fr_bn_bwd_reduce" else 3 + (1 if a.add_kind == 1 else 0)
            gbytes = rows * a.C * elt * passes / 1e9
            detail.append(("%s rows=%d C=%d passes=%d blocks=%d %.1fMB" % (launch.name[3:], rows, a.C, passes, a.nblocks,
                                                                          gbytes * 1e3),
                           round(ms, 4), round(gbytes / (ms * 1e-3), 1)))
        if launch.name == "fr_conv_igemm":
            a = launch.keep[0]
            name = "conv_igemm<%s,BN=%d,PRO=%d>" % (dtype_name if launch.args[1] == 1 else "f32",
                                                    64 if a.N <= 64 else 128, a.pro)
            flops = conv_flops(launch)
        elif launch.name == "fr_conv3x3_strip":
            a = launch.keep[0]
            name = "conv3x3_strip<%d,%d,%d,PRO=%d>" % (a.SC, a.N, a.SW, a.pro)
            flops = conv_flops(launch)
        elif launch.name == "fr_conv3x3_s2_strip":
            a = launch.keep[0]
            name = "conv3x3_s2_strip<%d,%s,PRO=%d>" % (a.SC, "fwd%d" % a.RW if a.mode == 0 else "dgrad%d" % a.SW, a.pro)
            flops = conv_flops(launch)
        elif launch.name == "fr_conv_wgrad_strip":
            a = launch.keep[0]
            name = "conv_wgrad_strip<%d,PRO=%d>(%dx%d)" % (a.SW, a.pro, a.Cout, a.SC)
            flops = wgrad_flops(launch)
        elif launch.name == "fr_conv_wgrad":
            a = launch.keep[0]
            name = "conv_wgrad<%s,%d,%d,PRO=%d>" % (dtype_name if launch.args[1] == 1 else "f32",
                                                    128 if a.Cout >= 128 else 64,
                                                    128 if a.SC >= 128 else (64 if a.SC >= 64 else 32), a.pro)
            flops = wgrad_flops(launch)
        f = fams.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += ms
        f[2] += flops
    instrumented_step.detail = detail
    return fams


def host_cores():
    """Cores this process may actually run on (cgroup/affinity aware), not the machine total."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(classes, seconds_budget=15.0, model="IR_50", B=32):
    """The oracle (CPU port of the reference path) on this box's host cores: backbone + ArcFace + focal + SGD, fp32
    (IR-50 at batch 32, and BASELINE configs[0] -- pSp IR-SE-50 with the 6-channel stem at batch 100):
    one warm-up step, then as many timed steps as fit the budget (at least 1)."""
    from frhip import synth
    from oracle import irse_ref as O
    cores = host_cores()
    torch.set_num_threads(cores)
    kw = {}
    if model == "pSp":
        from backbone.restyle_psp import pSp
        m = pSp(size=112, encoder_type="BackboneEncoder", avg_image=None)
        kw = dict(se=True, prefix="encoder.", avg_image=synth.uniform(2, "bench.avg", (3, 112, 112)))
    else:
        from backbone.model_irse import IR_50
        m = IR_50([112, 112])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    del m
    synth.fill_state_dict(sd, 15)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in names:
        sd[k].requires_grad_(True)
    hw = synth.uniform(16, "bench.head", (classes, 512), -0.05, 0.05).requires_grad_(True)
    x = synth.uniform(7, "cpu.x", (B, 3, 112, 112))
    y = synth.labels(7, "cpu.y", B, classes)
    bufs = {n: None for n in names + ["head"]}

    def one():
        _f, logits, _loss, grads = O.train_step(sd, x, y, hw, **kw)
        O.topk_accuracy(logits.detach(), y)
        with torch.no_grad():
            for n in names:
                bufs[n] = O.sgd_step([sd[n]], [grads[n]], [bufs[n]], 0.03, 0.9, 0.0 if O.is_bn_key(n) else 2e-3)[0]
            bufs["head"] = O.sgd_step([hw], [grads["head.weight"]], [bufs["head"]], 0.03, 0.9, 2e-3)[0]

    one()
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < seconds_budget and n < 200):
        one()
        n += 1
    dt = time.time() - t0
    what = "IR-50" if model == "IR_50" else "pSp(IR-SE-50, 6-ch stem, avg image)"
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "oracle/irse_ref.py train_step+sgd, %s+ArcFace(%d)+Focal, fp32, batch %d, %d steps in %.1fs"
                      % (what, classes, B, n, dt)}


class ClockSampler(threading.Thread):
    """Shader clock / board power of this rank's GPU every 100 ms while the timed loop runs ("the clocks
    observed").  Reads the amdgpu sysfs nodes of the device (hwmon freq1_input = current sclk in Hz, power1_average /
    power1_input in uW, else the starred line of pp_dpm_sclk); no sysfs access -> one ``rocm-smi --showclocks`` child
    per sample.  The MFMA peak scales with the clock the chip holds under load (MI355X_MICROARCH.md, DVFS give-back), so
    the roofline fraction is also quoted against peak x clock / 2400 MHz.  The in-kernel clock can read up to ~10 % below
    these nodes (same guide); tools/stamps.py measures that one for single kernels."""

    def __init__(self, device_index, period=0.1):
        super().__init__(daemon=True)
        self.period, self.samples, self.power, self.source = period, [], [], None
        self.series, self.t0 = [], time.perf_counter()
        self._stop_ev = threading.Event()
        self.freq_file = self.power_file = self.dpm_file = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            base = "/sys/bus/pci/devices/" + bdf
            import glob
            for h in glob.glob(base + "/hwmon/hwmon*"):
                if os.path.exists(h + "/freq1_input"):
                    self.freq_file = h + "/freq1_input"
                for n in ("power1_average", "power1_input"):
                    if os.path.exists(h + "/" + n):
                        self.power_file = h + "/" + n
                        break
            if os.path.exists(base + "/pp_dpm_sclk"):
                self.dpm_file = base + "/pp_dpm_sclk"
        except Exception:  # noqa: BLE001
            pass
        self.smi = None
        if self.freq_file is None and self.dpm_file is None:
            import shutil
            self.smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi")
                                                    else None)
            self.device_index = device_index

    def _read(self):
        import re
        try:
            if self.freq_file:
                self.source = "sysfs hwmon freq1_input"
                mhz = int(open(self.freq_file).read()) / 1e6
            elif self.dpm_file:
                self.source = "sysfs pp_dpm_sclk"
                m = re.search(r"(\d+)Mhz \*", open(self.dpm_file).read())
                mhz = float(m.group(1)) if m else None
            elif self.smi:
                self.source = "rocm-smi --showclocks"
                txt = subprocess.run([self.smi, "-d", str(self.device_index), "--showclocks"], capture_output=True,
                                     text=True, timeout=5).stdout
                m = re.search(r"sclk clock level:? *\d*:? *\((\d+)Mhz\)", txt)
                mhz = float(m.group(1)) if m else None
            else:
                return
            watts = int(open(self.power_file).read()) / 1e6 if self.power_file else None
            if mhz:
                self.samples.append(mhz)
                self.series.append((time.perf_counter() - self.t0, mhz, watts))
            if watts is not None:
                self.power.append(watts)
        except Exception:  # noqa: BLE001
            pass

    def run(self):
        self.t0 = time.perf_counter()
        if self.freq_file is None and self.dpm_file is None:
            # no sysfs node: the only source is a `rocm-smi` child, and forking one every 100 ms from this multi-GB process
            # would perturb the host-side launch loop being timed -- one snapshot when the loop starts, one when it stops
            self._read()
            self._stop_ev.wait()
            self._read()
            return
        while not self._stop_ev.is_set():
            self._read()
            self._stop_ev.wait(self.period)

    def stop(self):
        self._stop_ev.set()
        self.join(timeout=10)
        if not self.samples:
            return None
        v = sorted(self.samples)
        out = {"mhz_median": round(v[len(v) // 2], 1), "mhz_min": round(v[0], 1), "mhz_max": round(v[-1], 1),
               "samples": len(v), "source": self.source}
        if self.power:
            w = sorted(self.power)
            out["watts_median"] = round(w[len(w) // 2], 1)
        return out


class SyncTimer(object):
    """``dp.synchronize()`` of the timed steps bracketed by a HIP event pair on the compute stream: the time the main stream
    spends waiting for collectives that the backward pass did not hide (RCCL: a stream wait; gloo: the host blocks).  Host
    tensors (the CPU ``gloo`` test of this record): the host clock."""

    def __init__(self, dp, use_events):
        self.dp, self.use_events, self.on, self.marks = dp, use_events, False, []

    def synchronize(self):
        if not self.on:
            return self.dp.synchronize()
        if self.use_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.dp.synchronize()
            e1.record()
            self.marks.append((e0, e1))
        else:
            t0 = time.perf_counter()
            self.dp.synchronize()
            self.marks.append(time.perf_counter() - t0)

    def mean_ms(self):
        """Call after a device synchronize."""
        if not self.marks:
            return None
        if self.use_events:
            return sum(e0.elapsed_time(e1) for e0, e1 in self.marks) / len(self.marks)
        return sum(self.marks) / len(self.marks) * 1e3


def device_identity(device):
    """What tells two ranks' devices apart: the PCI address of a GPU (domain:bus:device), host + pid for host tensors."""
    device = torch.device(device)
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        try:
            return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except AttributeError:
            return "cuda:%d uuid %s" % (device.index, getattr(pr, "uuid", "?"))
    return "cpu %s pid %d" % (socket.gethostname(), os.getpid())


def collective_library():
    """Backend of the default process group and the version of the library behind it ("nccl" IS RCCL on ROCm)."""
    backend = dist.get_backend()
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            ver = "unknown"
    return backend, ver


def dp_record(rank, world, device, dp, timer, dt_local, steps):
    """The part of the JSON line that lets a reader verify an N > 1 record against itself (collective: every rank calls
    it): which process ran on which device (all-gathered, must be N distinct PCI addresses), the library and version the
    gradients went through, the exchange policy and its buckets, the time the compute stream waited for collectives, and
    every rank's own step time (the headline is the maximum)."""
    mine = {"rank": rank, "device": device_identity(device), "host": socket.gethostname(), "pid": os.getpid(),
            "ms_per_step": round(dt_local / steps * 1e3, 3),
            "comm_exposed_ms": None if timer is None or timer.mean_ms() is None else round(timer.mean_ms(), 4)}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank != 0:
        return None
    backend, ver = collective_library()
    red = getattr(dp, "reducer", None)
    policy = {"FRHIP_DP_OVERLAP": getattr(red, "policy", None), "gate_gradients": getattr(red, "gate", None),
              "bucket_mb": getattr(dp, "bucket_bytes", 0) / 2.0 ** 20,
              "buckets": len(red.buckets) if red is not None else None,
              "arena_mb": round(red.arena.numel() * red.arena.element_size() / 2.0 ** 20, 1) if red is not None else None,
              "meaning": {0: "after backward", 1: "as soon as a bucket is complete",
                          2: "complete buckets, once backward has left the one-workgroup-per-CU layers"}.get(
                              getattr(red, "policy", None))}
    ms = [r["ms_per_step"] for r in everyone]
    exposed = [r["comm_exposed_ms"] for r in everyone if r["comm_exposed_ms"] is not None]
    return {"ranks": [[r["rank"], r["device"]] for r in everyone],
            "ranks_distinct_devices": len(set(r["device"] for r in everyone)) == world and
            sorted(r["rank"] for r in everyone) == list(range(world)),
            "rank_pids": [r["pid"] for r in everyone],
            "collective_backend": backend, "rccl_version": ver, "dp_policy": policy,
            "comm_exposed_ms": round(sum(exposed) / len(exposed), 4) if exposed else None,
            "comm_exposed_ms_max": round(max(exposed), 4) if exposed else None,
            "ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "ms_per_step_by_rank": ms}


def timed_loop(step, xs, ys, warmup, steps, world, device, sampler=None, sync_timer=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
    nb = len(xs)
    for i in range(warmup):
        step(xs[i % nb], ys[i % nb])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if sampler is not None:
        sampler.start()
    if sync_timer is not None:
        sync_timer.on = True
    t0 = time.perf_counter()
    for i in range(steps):
        loss, _prec = step(xs[(warmup + i) % nb], ys[(warmup + i) % nb])
    torch.cuda.synchronize()
    timed_loop.dt_local = time.perf_counter() - t0  # this rank's own time, before it waits for the slowest one
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if sync_timer is not None:
        sync_timer.on = False
    clocks = sampler.stop() if sampler is not None else None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, float(loss.detach()), clocks


# The other BASELINE.json configs, timed briefly behind the headline (same step function, same data parallelism when
# world > 1).  FLOPs per image: SURVEY.md 8(d).
OTHER_CONFIGS = (
    ("configs[2] IR-50 + ArcFace(28000), bs=256/GPU", dict(model="IR_50", head="ArcFace", classes=28000, batch=256)),
    ("configs[3] IR-SE-101 + CosFace(28000), bs=128/GPU", dict(model="IR_SE_101", head="CosFace", classes=28000, batch=128)),
    ("configs[4] pSp IR-SE-50 6-ch stem + ArcFace(28000), bs=256/GPU", dict(model="pSp", head="ArcFace", classes=28000,
                                                                           batch=256)),
    # "for the fp32 parity mode report against the fp32 matrix peak separately" -- the headline workload on the
    # fp32 path (exact-fma f32 MFMA, the mode the parity tests run and COMPUTE_DTYPE='fp32' selects); step_frac is against
    # PEAK_F32_TFLOPS
    ("configs[1] on the fp32 parity path: IR-50 + ArcFace(7000), bs=256/GPU", dict(model="IR_50", head="ArcFace",
                                                                                  classes=7000, batch=256, dtype="fp32",
                                                                                  steps=6, warmup=2)),
)


def run_other_config(label, spec, args, device, rank, world, steps=10, warmup=4):
    import gc
    spec = dict(spec)
    steps, warmup = spec.pop("steps", steps), spec.pop("warmup", warmup)
    spec.setdefault("dtype", args.dtype)
    peak = PEAK_BF16_TFLOPS if spec["dtype"] == "bf16" else PEAK_F32_TFLOPS
    a = argparse.Namespace(sharded_head=False, resident_batches=warmup + steps, **spec)  # a new batch every step
    model, head, loss_fn, opt, xs, ys = build_job(a, device, rank)
    dp = None
    if world > 1:
        from frhip.parallel import DataParallel
        dp = DataParallel(model, head)
    step = make_step(model, head, loss_fn, opt, dp)
    dt, loss_val, _ = timed_loop(step, xs, ys, warmup, steps, world, device)
    ips = a.batch * world * steps / dt
    flops_img = FLOPS_PER_IMG[a.model] + 6.0 * 512 * a.classes
    rec = {"config": label, "value": round(ips, 1), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "n_gpus": world, "dtype": a.dtype, "final_loss": float("%.3e" % loss_val),
           "step_achieved": round(flops_img * ips / world / 1e12, 2), "peak": peak,
           "step_frac": round(flops_img * ips / world / 1e12 / peak, 4)}
    inner = model.encoder if hasattr(model, "encoder") else model
    inner._runner[0].plans.clear()
    inner._runner[0].plan = None
    del model, head, opt, xs, ys, step, dp, inner
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def main():
    args = parse()
    json_out, sys.stdout = sys.stdout, sys.stderr  # the modules print while they build; stdout carries the ONE JSON line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world, ("bench.py --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it spawns the "
                                "ranks itself) or under torch.distributed.run with --nproc-per-node N" % (args.gpus, world))
    if os.environ.get("FRHIP_BENCH_ONE_DEVICE") == "1":
        local = 0  # test hook: several ranks share GPU 0 (with FRHIP_DIST_BACKEND=gloo; RCCL refuses duplicate devices)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dp = os.environ.get("FRHIP_FORCE_DP", "0") == "1"  # exercise the RCCL path with a single rank (tests)
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("FRHIP_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for the shared-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    model, head, loss_fn, opt, xs, ys = build_job(args, device, rank)
    dp = None
    if world > 1 or force_dp:
        from frhip.parallel import DataParallel
        dp = DataParallel(model, None if args.sharded_head else head)  # a weight shard is complete on its owner
    sync_timer = SyncTimer(dp, True) if dp is not None else None
    step = make_step(model, head, loss_fn, opt, sync_timer if dp is not None else None)
    x, y = xs[0], ys[0]

    sampler = ClockSampler(local) if rank == 0 else None
    dt, loss_val, clocks = timed_loop(step, xs, ys, args.warmup, args.steps, world, device, sampler, sync_timer)
    dp_rec = dp_record(rank, world, device, dp, sync_timer, timed_loop.dt_local, args.steps) if dp is not None else None
    assert loss_val == loss_val, "loss is NaN"
    if sampler is not None and args.clock_log:
        with open(args.clock_log, "w") as f:
            f.write("# shader clock / board power of GPU %d during the timed loop of bench.py (%d steps, %.3f ms per step); "
                    "source: %s\n# t_s mhz watts\n" % (local, args.steps, dt / args.steps * 1e3, sampler.source))
            for t, mhz, w in sampler.series:
                f.write("%.3f %.0f %s\n" % (t, mhz, "%.0f" % w if w is not None else "-"))

    ms = dt / args.steps * 1e3
    ips = args.batch * world * args.steps / dt
    mname = "pSp(IR-SE-50, 6-ch stem)" if args.model == "pSp" else args.model.replace("IR_", "IR-")
    out = {
        "metric": "images/sec/GPU IR-50+ArcFace 112x112 bs=256; 1->8 GPU scaling eff.",
        "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s + %s(%d ids) + Focal + SGD train step, synthetic 112x112x3, bs=%d/GPU"
                               % (mname, args.head, args.classes, args.batch),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world + ("+class-sharded head" if args.sharded_head else ""),
                   "images_per_sec_per_gpu": round(ips / world, 1), "final_loss": float("%.3e" % loss_val),
                   "resident_batches": len(xs)},
    }
    if dp_rec is not None:
        # N > 1 (or FRHIP_FORCE_DP): the record verifies itself -- N ranks on N distinct devices through this library
        for k in ("ranks", "ranks_distinct_devices", "rank_pids", "collective_backend", "rccl_version", "dp_policy"):
            out["config"][k] = dp_rec[k]
        for k in ("comm_exposed_ms", "comm_exposed_ms_max", "ms_per_step_min", "ms_per_step_max", "ms_per_step_by_rank"):
            out[k] = dp_rec[k]
    fams = None
    inner = model.encoder if hasattr(model, "encoder") else model
    if not args.no_roofline:
        # per-kernel timing needs the weight gradients back on the main stream (no co-running kernels).  Every rank
        # runs the same two extra steps (they contain the gradient all-reduce); only rank 0 wraps its launches in events.
        inner._runner[0].single_stream = True
        step(x, y)
        torch.cuda.synchronize()
        if rank == 0:
            fams = instrumented_step(step, x, y, args.dtype)
        else:
            step(x, y)
            torch.cuda.synchronize()
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    if rank == 0:
        flops_img = FLOPS_PER_IMG.get(args.model, IR50_FLOPS_PER_IMG) + 6.0 * 512 * args.classes
        step_tflops = flops_img * ips / world / 1e12
        if fams is not None:
            table = sorted(fams.items(), key=lambda kv: -kv[1][1])
            total_ms = sum(v[1] for v in fams.values())
            if args.kernel_table:
                with open(args.kernel_table, "w") as f:
                    d = {k: {"launches": v[0], "ms": round(v[1], 4), "tflops": round(v[2] / 1e12, 4)} for k, v in table}
                    d["_launch_detail"] = getattr(instrumented_step, "detail", [])
                    json.dump(d, f, indent=1)
            # families: the prologue is a template argument of the same kernel -- count its variants together
            import re
            groups = {}
            for k, v in fams.items():
                if v[2] <= 0:
                    continue
                base = re.sub(r",PRO=\d", "", k)
                g = groups.setdefault(base, [0, 0.0, 0.0, {}])
                g[0] += v[0]
                g[1] += v[1]
                g[2] += v[2]
                g[3][k] = v[0]
            mf_ms = sum(g[1] for g in groups.values())
            mf_flops = sum(g[2] for g in groups.values())
            if groups:
                name, (cnt, kms, flops, members) = max(groups.items(), key=lambda kv: kv[1][1])
                ach = flops / (kms * 1e-3) / 1e12
                all_ach = mf_flops / (mf_ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": None, "kernel": name, "launches": cnt,
                                   "avg_launch_ms": round(kms / cnt, 4),
                                   "share_of_step": round(kms / total_ms, 3),
                                   "all_mfma": {"achieved": round(all_ach, 2), "frac": round(all_ach / peak, 4),
                                                "ms": round(mf_ms, 3), "share_of_step": round(mf_ms / total_ms, 3),
                                                "launches": sum(g[0] for g in groups.values()),
                                                "note": "every launch timed alone on one stream; the weight-gradient "
                                                        "launches cover half of the CUs by design (they run beside the "
                                                        "data gradients in the step), so their stand-alone rate is half"},
                                   "channelwise_ms": round(sum(v[1] for k, v in fams.items() if k.startswith(
                                       ("fr_bn_", "fr_reduce_parts", "fr_channel_stats", "fr_se_"))), 3),
                                   "launches_per_step": sum(v[0] for v in fams.values()),
                                   "step_kernel_ms_single_stream": round(total_ms, 3),
                                   "step_achieved": round(step_tflops, 2), "step_frac": round(step_tflops / peak, 4)}
                if clocks is not None:
                    # the MFMA rate is per clock: peak at the clock the chip held during the timed loop
                    out["roofline"]["clock_mhz"] = clocks["mhz_median"]
                    out["roofline"]["clock"] = clocks
                    scale = clocks["mhz_median"] / 2400.0
                    out["roofline"]["frac_at_clock"] = round(ach / (peak * scale), 4)
                    out["roofline"]["step_frac_at_clock"] = round(step_tflops / (peak * scale), 4)
                # HBM bytes per launch cannot be counted from inside this process; they come from the committed
                # rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections) over the same
                # kernel instances (tools/pmc_round.sh + tools/pmc_summary.py), averaged over this family's launches
                import glob
                cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_kernels.json")))
                pmc = cands[-1] if cands else None  # the newest round's table
                m = re.match(r"conv3x3_strip<(\d+),(\d+),(\d+)>", name)
                if pmc and m:
                    with open(pmc) as f:
                        recs = {r["kernel"]: r for r in json.load(f)["kernels"]}
                    pre = "strip_%s_%s_%s_" % m.groups()
                    by_pro = {0: pre + "dgrad", 1: pre + "fwd_bn", 2: pre + "fwd_prelu", 4: pre + "fwd_resbn"}
                    num = den = alg = 0.0
                    for member, n in members.items():
                        r = recs.get(by_pro.get(int(member[-2]), ""))
                        if r and "hbm_bytes" in r:
                            num += n * r["hbm_bytes"]
                            alg += n * r["algorithmic_bytes"]
                            den += n
                    if den:
                        out["roofline"]["traffic"] = int(num / den)
                        out["roofline"]["traffic_algorithmic"] = int(alg / den)
                        out["roofline"]["traffic_source"] = os.path.relpath(pmc, REPO)
        elif clocks is not None:
            out["clock"] = clocks
    # ---- the other BASELINE configs (short), then the CPU legs
    default_workload = (args.model, args.head, args.classes, args.batch, args.dtype) == ("IR_50", "ArcFace", 7000, 256, "bf16")
    if default_workload and not args.no_other_configs and not args.sharded_head and \
            os.environ.get("FRHIP_BENCH_OTHER", "1") != "0":
        import gc
        inner._runner[0].plans.clear()
        inner._runner[0].plan = None
        del model, head, opt, xs, ys, x, y, step, dp, inner
        gc.collect()
        torch.cuda.empty_cache()
        others = []
        for label, spec in OTHER_CONFIGS:
            others.append(run_other_config(label, spec, args, device, rank, world))
        if rank == 0:
            out["other_configs"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.classes)
            if default_workload:
                # BASELINE configs[0] (the reference's own CPU-runnable case): pSp IR-SE-50, batch 100, 100 identities
                out["cpu_baseline_config0"] = cpu_baseline(100, seconds_budget=8.0, model="pSp", B=100)
        print(json.dumps(out), file=json_out, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
