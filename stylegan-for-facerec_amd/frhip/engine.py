"""Static execution plan of the IR / IR-SE backbone training step on one MI355X.

The reference runs ``Backbone.forward`` (backbone/model_irse.py:167-172) / ``BackboneEncoderDiffHead.forward``
(backbone/restyle_psp.py:193-216) as ~300 separate torch.nn calls and lets autograd replay them.  Here the
module tree only *holds* the parameters (same names, same state-dict keys); the arithmetic is a fixed list of
HIP launches over buffers that are laid out once per (batch, dtype):

  forward, per residual unit (x = unit input, NHWC, statistics of x already reduced by its producer)
    conv1  : y1  = conv3x3( BN1(x) )            BN apply fused into the A-operand gather      (model_irse.py:57)
    conv2  : y2  = conv3x3_s( PReLU(y1) )       PReLU fused into the gather; epilogue emits sum / sumsq of y2
    [convS : yS  = conv1x1_s(x)                 + statistics]                                  (model_irse.py:55)
    [SE    : pooled = mean_hw BN2(y2); s = sigmoid(W2 relu(W1 pooled))]                        (model_irse.py:23-46)
    apply  : out = BN2(y2) [* s] + (x[::s, ::s] | BNS(yS))   + statistics of out for the next BN1
  backward walks the same units in reverse with: BN-backward reduce/apply, conv2 data-gradient with the
  PReLU-backward + slope-gradient epilogue, conv1 data-gradient with the BN-backward-sums epilogue, the two
  weight gradients (split over pixel slices, fp32 atomics), and one pass that forms the unit's input gradient.

Activations are stored once (y1, y2, [yS], out per unit); nothing is recomputed and no BN/PReLU output is ever
materialised.  Parameter gradients are written straight into one flat fp32 arena whose order is the order in
which they become ready (output layer first, stem last) so that data-parallel buckets are contiguous slices.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import FR_BF16, FR_F32

_EPS, _MOM = 1e-5, 0.1


def _switch(name, default):
    """An engine-level switch (README "Switches"): an integer environment variable, read when a plan is built."""
    return int(os.environ.get(name, default))


def compute_dtype_default():
    v = os.environ.get("FRHIP_COMPUTE_DTYPE", "fp32").lower()
    return torch.bfloat16 if v in ("bf16", "bfloat16") else torch.float32


# ------------------------------------------------------------------------------------------------ structure


class _Unit(object):
    __slots__ = ("idx", "cin", "depth", "stride", "H", "Ho", "bn1", "conv1", "prelu", "conv2", "bn2", "se",
                 "sc_conv", "sc_bn")


def _children_of(container):
    return list(container.children()) if not isinstance(container, (list, tuple)) else list(container)


def describe(module):
    """Walk the parameter-holding module tree (either Backbone of model_irse.py or BackboneEncoderDiffHead of
    restyle_psp.py -- Sequential or ModuleList children) and return (stem, units, out) layer handles."""
    body_only = getattr(module, "input_layer", None) is None  # a bare stack of residual units (_UnitStack)
    stem = None if body_only else tuple(_children_of(module.input_layer)[:3])
    units = []
    H = int(module.input_size if isinstance(module.input_size, int) else module.input_size[0])
    for i, blk in enumerate(module.body):
        u = _Unit()
        u.idx = i
        res = _children_of(blk.res_layer)
        for c in res:
            if isinstance(c, nn.Dropout) and c.p > 0 and c.training:
                raise NotImplementedError("frhip: Dropout inside residual units (pSp include_dropout) is not "
                                          "on the accelerated path")
        bns = [c for c in res if isinstance(c, nn.BatchNorm2d)]
        convs = [c for c in res if isinstance(c, nn.Conv2d)]
        prelus = [c for c in res if isinstance(c, nn.PReLU)]
        ses = [c for c in res if hasattr(c, "fc1") and hasattr(c, "fc2")]
        u.bn1, u.bn2 = bns
        u.conv1, u.conv2 = convs
        (u.prelu,) = prelus
        u.se = ses[0] if ses else None
        u.cin, u.depth = u.conv1.in_channels, u.conv1.out_channels
        u.stride = u.conv2.stride[0]
        u.H, u.Ho = H, H // u.stride
        sc = blk.shortcut_layer
        if isinstance(sc, nn.MaxPool2d):
            u.sc_conv = u.sc_bn = None
        else:
            ch = _children_of(sc)
            u.sc_conv = [c for c in ch if isinstance(c, nn.Conv2d)][0]
            u.sc_bn = [c for c in ch if isinstance(c, nn.BatchNorm2d)][0]
        units.append(u)
        H = u.Ho
    if body_only:
        return None, units, None
    out = _children_of(module.output_layer)
    out_bn = [c for c in out if isinstance(c, nn.BatchNorm2d)][0]
    out_drop = [c for c in out if isinstance(c, nn.Dropout)][0]
    out_lin = [c for c in out if isinstance(c, nn.Linear)][0]
    out_bn1d = [c for c in out if isinstance(c, nn.BatchNorm1d)][0]
    return stem, units, (out_bn, out_drop, out_lin, out_bn1d)


def ready_order_params(module):
    """Parameters in the order their gradients complete during backward (for bucketing)."""
    stem, units, out = describe(module)
    order = []
    if out is not None:
        ob, _od, ol, ob1 = out
        order = [ob1.weight, ob1.bias, ol.weight, ol.bias, ob.weight, ob.bias]
    for u in reversed(units):
        order += [u.bn2.weight, u.bn2.bias]
        if u.se is not None:
            order += [u.se.fc1.weight, u.se.fc2.weight]
        if u.sc_conv is not None:
            order += [u.sc_bn.weight, u.sc_bn.bias, u.sc_conv.weight]
        order += [u.prelu.weight, u.conv2.weight, u.bn1.weight, u.bn1.bias, u.conv1.weight]
    if stem is not None:
        sc, sb, sp = stem
        order += [sb.weight, sb.bias, sp.weight, sc.weight]
    return order


# ------------------------------------------------------------------------------------------------ helpers


class _BN(object):
    """Device-side coefficients of one BatchNorm: mean, invstd, scale (= gamma*invstd), shift."""
    __slots__ = ("mod", "C", "mean", "invstd", "scale", "shift")

    def __init__(self, mod, pool):
        self.mod, self.C = mod, mod.num_features
        self.mean, self.invstd, self.scale, self.shift = (pool.take(self.C) for _ in range(4))


class _Pool(object):
    """Bump allocator over one fp32 device tensor (small per-channel vectors)."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, device=device)
        self.off = 0

    def take(self, n):
        n_pad = (n + 63) // 64 * 64
        if self.off + n_pad > self.buf.numel():
            raise _lib.FrhipError("frhip: coefficient pool exhausted")
        t = self.buf[self.off:self.off + n]
        self.off += n_pad
        return t


def _cl_weight_ok(w):
    """Conv weight stored as [Cout][kh][kw][Cin] (channels-last memory of the OIHW Parameter)?"""
    return w.permute(0, 2, 3, 1).is_contiguous()


def _dense(t):
    dims = sorted((s, n) for s, n in zip(t.stride(), t.shape) if n > 1)
    expect = 1
    for s, n in dims:
        if s != expect:
            return False
        expect *= n
    return True


def _wgrad_slices(pixels, tiles):
    """Pixel slices for the weight-gradient kernel: aim at >= 512 blocks, slices >= 256 pixels, <= 256 slices."""
    want = max(1, (768 + tiles - 1) // tiles)
    return int(max(1, min(want, 256, pixels // 256 if pixels >= 256 else 1)))


class _ArgProxy(object):
    """Lets the slab-buffer bookkeeping re-point a positional-argument launch (fr_reduce_slabs) like a struct field."""

    def __init__(self, launch, index):
        self.launch, self.index = launch, index
        self.keep = (self,)

    def __setattr__(self, name, value):
        if name in ("launch", "index", "keep"):
            object.__setattr__(self, name, value)
        else:
            self.launch.args[self.index] = value


class _EvRecord(object):
    """List entry: record an event on a stream (weight-gradient side stream bookkeeping)."""
    __slots__ = ("ev", "stream")

    def __init__(self, ev, stream):
        self.ev, self.stream = ev, stream

    def __call__(self):
        self.ev.record(self.stream)


class _EvWait(object):
    """List entry: make a stream wait for an event."""
    __slots__ = ("ev", "stream")

    def __init__(self, stream, ev):
        self.ev, self.stream = ev, stream

    def __call__(self):
        self.stream.wait_event(self.ev)


_SIDE_STREAMS = {}


def _side_stream(device, priority):
    """ONE weight-gradient stream per device for every plan of the process.  HIP maps streams onto a handful of
    hardware queues round-robin; a new stream per plan (new batch size, new model) would sooner or later land on the
    main stream's queue and silently serialise with it."""
    key = (torch.device(device).index, priority)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return _SIDE_STREAMS[key]


# ------------------------------------------------------------------------------------------------ the plan


class BackbonePlan(object):
    def __init__(self, module, B, dtype, device, in_channels, avg_channels, single_stream=False, infer=False):
        self.module, self.B, self.device = module, B, device
        ops.sync_switches()  # the library's own switches follow the environment as of now
        # infer: forward only (no gradient arena, no backward scratch, no backward list).  fold: additionally every
        # BatchNorm runs on its running statistics, so BN2 / the shortcut BN are folded into the packed conv weights and
        # the residual add moves into conv2's epilogue (reference: util/utils.py:254-307 evaluates in eval mode).
        self.infer = infer
        self.fold = infer and all(not m.training for m in module.modules()
                                  if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))) and \
            not _switch("FRHIP_NO_FOLD", 0)
        self.tdtype = dtype
        self.fr = FR_F32 if dtype == torch.float32 else FR_BF16
        self.esz = 4 if dtype == torch.float32 else 2
        self.stem, self.units, self.out = describe(module)
        self.body_only = self.stem is None  # residual units only: NHWC activation in, NHWC activation (and dL/dx) out
        self.in_channels, self.avg_channels = in_channels, avg_channels
        self.stream = ops.current_stream_ptr()
        self.stream1_t = torch.cuda.current_stream()
        self.stream_id = self.stream1_t.cuda_stream
        # Weight gradients feed nothing downstream in backward, so they run on a side stream: an MFMA-bound wgrad
        # kernel and the HBM-bound BN-backward / residual passes of the next unit then share the CUs (measured with
        # tools/overlap_probe.py: ~2/3 of an elementwise pass hides under a wgrad strip kernel).
        self.dual = not _switch("FRHIP_SINGLE_STREAM", 0) and not single_stream
        # the side stream is low priority: weight gradients fill the CUs the main chain (dgrads, BN/PReLU backward)
        # leaves idle instead of splitting the machine with it (HIP: 1 low .. -1 high; either way measured the same)
        self.stream2_t = _side_stream(device, 1) if self.dual else self.stream1_t
        # Workgroups of a weight-gradient launch (dW tiles x image groups): 128, HALF the chip.  A weight-gradient workgroup
        # holds 87 KB of LDS and a strip workgroup of the main stream 142 KB: the two never share a CU, so a launch that covers
        # (nearly) every CU does not run BESIDE the data gradients, it alternates with them (the timeline of the 224-workgroup
        # default of rounds 3-4: every strip launch of the backward pass started on the 32 free CUs and waited for the rest).
        # With 128 the side stream owns half of the CUs and the main stream the other half, and the image groups divide the
        # batch exactly (256 images / 8 groups; 14 groups left 19 images to some workgroups and 18 to others).  Round 5, same-box
        # alternations (profiles/r05_ab_wgrad_wgs.txt): IR-50 bs 256 14.26-14.33 against 14.43-14.50 ms per step (224), 96:
        # 15.18, 160 / 192 / 208 / 256: 14.72-14.77 / 14.53; IR-SE-101 bs 128 16.94-16.99 against 18.14-18.22; pSp bs 256
        # 16.31 against 16.72.  Per-width / per-role overrides (temporary switches, removed) found nothing better than one
        # value for all.
        self.wgrad_wgs = min(256, max(64, _switch("FRHIP_WGRAD_WGS", 128)))  # the slab sum takes <= 256 groups
        # (Round 6, measured again with the lighter tail of the main stream: 256 / 192 workgroups for the two HBM-bound weight
        # gradients of the first unit, 112x112 -- 14.055-14.078 / 14.085-14.094 against 14.021-14.064 ms per step: nothing.)
        self.stream2 = ctypes.c_void_p(self.stream2_t.cuda_stream)
        self.edge_signal = bool(_switch("FRHIP_EDGE_SIGNAL", 1))  # dependency edges as completion signals (_side_after_main)
        self.S = int(module.input_size if isinstance(module.input_size, int) else module.input_size[0])
        self.generation = 0
        # (FRHIP_GRAPH, rounds 2-3: the two launch lists captured into HIP graphs.  Replay was SLOWER than eager launches at
        # every batch size on this stack -- ms per step eager / graph: B = 8: 6.05 / 6.86, 64: 7.67 / 8.11, 256: 16.0 / 16.4 --
        # and a packed native list executor left both the enqueue time and the step time unchanged; removed in round 4.)
        # (Round 4, built, bit-identical, measured and removed in round 5 -- DESIGN.md section 8, git tag r05-before-prune:
        # in-launch reductions of the partial rows by arrival ticket, FRHIP_TAIL, +0.35-0.75 ms per step; the backward of BN2
        # inside conv2's data gradient, FRHIP_FUSE_BN2, -0.19 ms on one stream and nothing on two.)
        self.use_strip = not _switch("FRHIP_NO_STRIP", 0)  # 1: every LDS-strip family back on the generic GEMM
        # Residual-sum statistics from moments (round 4).  In the FORWARD pass nothing runs beside the channel-wise passes, and
        # per identity unit the pass `out = BN2(y2) + x` (fr_bn_apply: 75 MB of traffic at 14x14) existed for two reasons: the
        # next unit's conv1 reads `out`, and its train-mode BN1 needs the batch statistics of `out` first.  The statistics do
        # not need the pass: mean / variance of a*y2 + b + x follow from the moments of y2 (conv2's rows), the statistics of
        # x (this unit's BN1) and one cross moment sum(y2*x) that conv2's epilogue adds while it has y2 in registers
        # (FR_EPI_STATS_X, fr_bn_finalize_res: BN2's and the next BN1's coefficients from one launch).  And `out` is formed by
        # its consumer: the next conv1 loads y2 and x, writes out = a*y2 + b + x (same bits as fr_bn_apply) on the way and
        # applies BN1 to it (FR_PRO_RESBN).  Per fused edge: bn_apply + one finalize launch gone.  Identity units without SE
        # whose conv2 and whose successor's conv1 run on LDS-strip instances (IR-50: 17 of 24 units).  FRHIP_RES_MOMENTS=0:
        # A/B switch.
        self.res_moments = bool(_switch("FRHIP_RES_MOMENTS", 1)) and self.fr == FR_BF16 and self.use_strip and not self.fold
        # ... and behind squeeze-excite units (out = gate[image][c] * BN2(y2) + x): the same moments PER IMAGE, combined with
        # the gates by the launch that computes them (fr_se_pool_parts_mlp_fwd_res); the moments of x are carried from unit to
        # unit (one fr_image_moments pass at the head of a stage).
        self.res_moments_se = self.res_moments
        self.use_stem_gemm = self.fr == FR_BF16 and not _switch("FRHIP_NO_STEM_GEMM", 0) and not self.body_only
        self.use_s2 = self.use_strip  # the stride-2 strip kernels follow FRHIP_NO_STRIP
        self.slab, self._slab_users = None, []
        # deferred slab sums (FrWgradArgs.defer / prev_*): a weight-gradient launch that supports it leaves the sum of its
        # slabs to the NEXT such launch of the side stream (two slab buffers alternate); fr_reduce_slabs flushes the last
        self.defer_slabs = True  # (FRHIP_WGRAD_DEFER=0 turns the library's answer off)
        self.slab2, self._slab2_users, self._slab_flip = None, [], 0
        self.slab3, self._slab3_users = None, []  # slabs of the launches that sum their own (the deferring chain skips them)
        self._pending = None  # (launch that wrote the slabs, groups, n, dw tensor, parameter)
        self.part_slope = None  # per-buffer-set partial rows of the PReLU slope gradient (side-stream reduction)
        self.comm_stream_t, self.comm_events = None, None  # readiness callbacks (run_backward)
        self._normalize_params()
        self._alloc()
        self._bind_params()
        # weight layouts (round 6): the packed 3x3 copies the LDS-strip kernels read are kept in MFMA-fragment order
        self.w_frag = self.fr == FR_BF16 and bool(_switch("FRHIP_W_FRAG", 1))
        self._weight_layout = {}
        self._pack_targets = set(id(d[k]) for d in self.ubuf for k in ("wp1", "wp2", "wt1", "wt2") if k in d)
        self._build_forward()
        if not self.infer:
            self._build_backward()
        self._finish_pack()

    # ---- buffers -----------------------------------------------------------------------------------
    def _act(self, rows, C):
        return torch.empty(rows, C, device=self.device, dtype=self.tdtype)

    def _alloc(self):
        B, S, dev = self.B, self.S, self.device
        self.pool = _Pool(64 * 1024 + 16 * 512 * (len(self.units) * 4 + 8), dev)
        ct = self.in_channels + self.avg_channels
        self.K0 = 32 if 9 * ct <= 32 else 64
        M0 = B * S * S
        self.M0 = M0
        C0 = self.units[0].cin if self.body_only else 64  # channels of the first unit's input
        self.z0 = self._act(M0, C0)
        if not self.body_only:
            # (Round 4 also carried the stem GEMMs on implicit im2col rows, FRHIP_STEM_IMPLICIT: 205 MB less memory, +0.05-0.09
            # ms per step, profiles/r04_ab_stem_implicit.txt; removed in round 5.)
            self.X0 = self._act(M0, self.K0)
            # Round 4: the stem forward as two passes over the rows (statistics, then GEMM + BN + PReLU in one kernel) and its
            # backward on a RECOMPUTED y0: the GEMM is 13 GFLOP at batch 256, its output 411 MB -- y0 is never written or
            # read, the fr_bn_apply pass over it is gone (-0.06 ... -0.09 ms per step, profiles/r04_ab_stem_two_pass.txt).
            self.stem_two_pass = self.use_stem_gemm and not self.fold
            self.stem_recompute = self.stem_two_pass
            self.y0 = None if (self.stem_recompute or (self.stem_two_pass and self.infer)) else self._act(M0, 64)
            self.W0p = torch.empty(64, self.K0, device=dev, dtype=self.tdtype)
            self.gW0p = torch.zeros(64, self.K0, device=dev)
            self.bn0 = _BN(self.stem[1], self.pool)
        self.ubuf = []
        max_in = M0 * C0
        max_mid = 0
        max_out = 0
        max_xs = 0
        for u in self.units:
            rin, rout = B * u.H * u.H, B * u.Ho * u.Ho
            d = {
                "y1": self._act(rin, u.depth), "y2": self._act(rout, u.depth), "out": self._act(rout, u.depth),
                "bn1": _BN(u.bn1, self.pool), "bn2": _BN(u.bn2, self.pool),
                "wt1": torch.empty(u.cin, 9, u.depth, device=dev, dtype=self.tdtype),
                "wt2": torch.empty(u.depth, 9, u.depth, device=dev, dtype=self.tdtype),
            }
            if self.fr == FR_BF16:
                d["wp1"] = torch.empty(u.depth, 9, u.cin, device=dev, dtype=self.tdtype)
                d["wp2"] = torch.empty(u.depth, 9, u.depth, device=dev, dtype=self.tdtype)
            if u.sc_conv is not None:
                d["yS"] = self._act(rout, u.depth)
                d["bnS"] = _BN(u.sc_bn, self.pool)
                d["wtS"] = torch.empty(u.cin, 1, u.depth, device=dev, dtype=self.tdtype)
                if self.fr == FR_BF16:
                    d["wpS"] = torch.empty(u.depth, 1, u.cin, device=dev, dtype=self.tdtype)
                max_xs = max(max_xs, rin * u.cin)
            if self.fold and u.se is None:
                d["wf2"] = torch.empty(u.depth, 9, u.depth, device=dev, dtype=self.tdtype)
                if u.sc_conv is not None:
                    d["wfS"] = torch.empty(u.depth, 1, u.cin, device=dev, dtype=self.tdtype)
            if u.se is not None:
                R = u.se.fc1.out_channels
                for k, n in (("pooled", u.depth), ("s", u.depth), ("gs", u.depth), ("gpooled", u.depth), ("hidden", R),
                             ("gz", u.depth), ("gh", R)):
                    d[k] = torch.zeros(B, n, device=dev)
            self.ubuf.append(d)
            max_in = max(max_in, rin * u.cin)
            max_mid = max(max_mid, rin * u.depth)
            max_out = max(max_out, rout * u.depth)
        last = self.units[-1]
        self.HWo = last.Ho * last.Ho
        self.feat_in = last.depth * self.HWo
        if not self.body_only:
            self.bn_out = _BN(self.out[0], self.pool)
            self.bn1d = _BN(self.out[3], self.pool)
            self.a = self._act(B, self.feat_in)
            self.f = torch.empty(B, 512, device=dev)
            self.feat = torch.empty(B, 512, device=dev)
            # Linear(25088, 512) on the master weight in its own (reference Flatten) layout: the activation `a` is written
            # c-major instead (csrc/linear_gemm.hip; bf16 path; fp32 and odd shapes: per-step permuted copies)
            ol = self.out[2]
            self.lin_cm = (self.fr == FR_BF16 and last.depth % 64 == 0 and
                           self.feat_in % 128 == 0 and ol.weight.is_contiguous() and
                           _lib.lib.fr_linear_slices(512, self.feat_in) > 0)
            if self.lin_cm:
                self.g_cm = self._act(B, self.feat_in)
            else:
                self.Wlin = torch.empty(512, self.feat_in, device=dev, dtype=self.tdtype)
                self.WlinT = torch.empty(self.feat_in, 512, device=dev, dtype=self.tdtype)
                self.gWlin = torch.zeros(512, self.feat_in, device=dev)
        self.zeros_c = torch.zeros(512, device=dev)
        if self.infer:
            self.part = torch.zeros(max(4 * 1024 * 1024, ((M0 + 127) // 128) * 2 * 64 + 4096), device=dev)
            return
        # backward scratch (sized for the largest unit)
        self.g_pp = [self._act(max_in, 1).view(-1), self._act(max_in, 1).view(-1)]   # unit input/output gradients
        # gradients consumed by the side-stream wgrads are double-buffered by unit parity (the next unit must not
        # overwrite what a still-running weight gradient reads)
        # two sets: the main stream may run one unit ahead of the side stream's weight gradients (three: measured the same)
        # (Round 4, measured and removed: the two-stream timeline shows a ~6-us gap on the main stream at every event edge,
        # three per residual unit; ONE main -> side edge per unit behind the second data gradient and four buffer sets with
        # the main stream waiting every second unit ran 14.92-15.02 ms per step against 14.66-14.67 with the round-3 edges on
        # the same box -- profiles/r04_ab_edges.txt.  The early edge behind conv2's data gradient is worth more than its gap:
        # the side stream's kernels are queued while the first data gradient still holds the CUs and take them as its
        # workgroups retire.)
        # Round 6: ONE buffer set per unit, each of its unit's own size (FRHIP_BWD_SETS=0, default; IR-50 at batch 256: 2.5 GB
        # instead of 1.2 GB for two sets of the largest unit's size).  With two alternating sets the main stream waits at the
        # head of every unit for the side stream's event of the unit two back -- a wait that (almost) never blocks, but costs
        # the main stream a barrier packet its next kernel waits for: +3.7 us per unit (tools/edge_probe.hip); with a set per
        # unit nothing is ever reused inside a step and the main stream waits for the side stream once, at the end.
        sets = _switch("FRHIP_BWD_SETS", 0)
        self.per_unit_sets = self.dual and sets <= 0
        nset = (len(self.units) if self.per_unit_sets else max(2, sets)) if self.dual else 1
        self.nset = nset
        if self.per_unit_sets:
            self.g_y2s, self.g_ySs, self.g_y1s = [], [], []
            for u in self.units:
                rin, rout = B * u.H * u.H, B * u.Ho * u.Ho
                self.g_y2s.append(self._act(rout * u.depth, 1).view(-1))
                self.g_ySs.append(self._act(rout * u.depth, 1).view(-1) if u.sc_conv is not None else None)
                self.g_y1s.append(self._act(rin * u.depth, 1).view(-1))
        else:
            self.g_y2s = [self._act(max_out, 1).view(-1) for _ in range(nset)]
            self.g_ySs = [self._act(max_out, 1).view(-1) if max_xs else None for _ in range(nset)]
            self.g_y1s = [self._act(max(max_mid, M0 * C0), 1).view(-1) for _ in range(nset)]
        self.g_y1 = self.g_y1s[0]
        self.g_xh = self._act(max_in, 1).view(-1)
        self.g_xS = self._act(max_xs, 1).view(-1) if max_xs else None
        if not self.body_only:
            self.g_f32 = torch.empty(B, 512, device=dev)     # BN1d backward output (fp32)
            self.g_fT = self._act(B, 512)
        # partial rows of every epilogue / channel-wise reduction.  Largest users: the stride-2 data gradient at 56x56
        # ([4 classes][B * 28 strips][2][64] floats), the stem ([M0/128][2][64]), 64-channel strips at 112 ([B * 56][2][64]);
        # _check_part() verifies every launch against the allocation when the plan is built.
        self.part = torch.zeros(max(4 * 1024 * 1024, ((M0 + 127) // 128) * 2 * 64 + 4096, B * 4 * 28 * 2 * 64 + 4096),
                                device=dev)
        self.side_slope = self.dual  # the PReLU-slope partial sums are added on the side stream (on the main one: +0.1 ms)
        if self.side_slope:
            # (per-unit sets: sized when the backward list is built, from the rows the unit's data gradient writes)
            self.part_slope = [None if self.per_unit_sets else torch.zeros_like(self.part) for _ in range(nset)]
        self.se_scratch = torch.zeros(2, 512 * 64, device=dev)  # dW1/dW2 sink while the SE weights are frozen
        self.se_gs_part = torch.empty(B * 8 * 4 * 512, device=dev)  # row-slice partials of the squeeze-excite gradient squeeze (x 4 sums)
        self.sums = torch.zeros(3, 512, device=dev)      # scratch reduce target for frozen parameters
        self.nbt_dummy = None

    # ---- parameters --------------------------------------------------------------------------------
    def _normalize_params(self):
        """Conv weights of the residual units live as [Cout][kh][kw][Cin] (channels-last memory behind the OIHW
        Parameter) so the fp32 master is already the packed GEMM operand; convert once if a caller replaced them."""
        for u in self.units:
            for conv in (u.conv1, u.conv2, u.sc_conv):
                if conv is None:
                    continue
                w = conv.weight
                if w.dtype != torch.float32:
                    raise _lib.FrhipError("frhip: parameters must be fp32 master weights")
                if not _cl_weight_ok(w):
                    with torch.no_grad():
                        w.data = w.data.contiguous(memory_format=torch.channels_last)
                    if w.grad is not None:
                        w.grad = None

    def _param_list(self):
        return ready_order_params(self.module)

    def _bind_params(self):
        """(Re)create the flat gradient arena and remember parameter storage addresses."""
        params = self._param_list()
        if self.infer:
            self.arena, self.gviews, self.arena_slices = None, {}, []
            self.param_sig = self._signature()
            return
        total = sum((p.numel() + 63) // 64 * 64 for p in params)
        self.arena = torch.zeros(total, device=self.device)
        self.arena._frhip_grad_arena = True  # lets frhip.optim.SGD.zero_grad clear all views with one fill
        self.gviews = {}
        self.arena_slices = []
        off = 0
        for p in params:
            n = p.numel()
            flat = self.arena[off:off + n]
            if not p.is_contiguous() and not _dense(p):
                raise _lib.FrhipError("frhip: parameter storage must be dense")
            v = flat.as_strided(tuple(p.shape), tuple(p.stride()))  # gradient shares the parameter's layout
            self.gviews[id(p)] = v
            self.arena_slices.append((p, off, n))
            off += (n + 63) // 64 * 64
        self._grad_pairs = [(p, self.gviews[id(p)]) for p, _o, _n in self.arena_slices]
        self.param_sig = self._signature()

    def _signature(self):
        return (tuple((p.data_ptr(), p.requires_grad) for p in self._param_list()),
                tuple(b.data_ptr() for b in self.module.buffers()),
                tuple(m.training for m in self.module.modules()))

    def grad_of(self, p):
        """Gradient target of a parameter: its arena view when it trains, else None."""
        return self.gviews[id(p)] if p.requires_grad else None

    def _conv_master(self, conv):
        return conv.weight

    # ---- conv dispatch ----------------------------------------------------------------------------
    def _check_part(self, n, kw):
        """Fail loudly (instead of a GPU memory fault) if an epilogue's partial rows would not fit their buffer."""
        part = kw.get("part")
        nv = 3 if kw.get("epi") == ops.EPI_STATS_X else 2
        if part is not None and n * nv * kw["N"] > part.numel():
            raise _lib.FrhipError("frhip: %d partial rows x 2 x %d channels exceed the %d-float partial-sum buffer "
                                  "(batch %d)" % (n, kw["N"], part.numel(), kw["B"]))
        return n

    def _conv(self, L, **kw):
        """Append a convolution launch; returns the number of partial rows its epilogue writes.  bf16 stride-1
        3x3 layers whose shape is in the strip table run with the input strip resident in LDS."""
        return self._check_part(self._conv_launch(L, **kw), kw)

    def _conv_launch(self, L, **kw):
        if (self.fr == FR_BF16 and self.use_strip and kw["KH"] == 3 and kw["stride"] == 1 and kw["RH"] == kw["SH"]):
            n = ops.strip_parts(kw["B"], kw["SC"], kw["N"], kw["SW"], kw.get("epi", 0))
            # (Round 5, measured and removed -- VERDICT r4 item 8, profiles/r05_hog_matrix.txt: the 256 -> 256 @14x14 launches as
            # B - T whole images + a second launch of the last T images on the two-workgroups-per-image instance: T = 16 / 32 /
            # 64 cost +1.0 ... 1.1 ms per step with no CU held (launches of one stream do not overlap) and 19.0 / 17.6 / 17.3
            # against 17.8 ms with 4-32 CUs held for the whole step.  What data-parallel runs do instead: frhip.parallel holds
            # the collectives back until the backward pass has left these layers, comm_gate below.)
            if n:
                # Round 6: weights of the LDS-strip instances in MFMA-fragment order (FrConvArgs.w_frag): a weight load reads
                # 1024 contiguous bytes instead of half of each of 16 lines (-5 ... -9 % per launch; FRHIP_W_FRAG=0: A/B)
                if self._note_weight(kw["w"], self.w_frag and kw["SC"] % 64 == 0 and kw["N"] % 64 == 0 and bool(
                        _lib.lib.fr_conv3x3_strip_takes_frag(kw["B"], kw["SC"], kw["N"], kw["SW"]))):
                    kw = dict(kw, w_frag=1)
                L.append(ops.conv_strip(self.stream, **kw))
                self._last_conv_strips = n  # partial rows = strips, image-major
                return n
        if (self.fr == FR_BF16 and self.use_strip and self.use_s2 and kw["KH"] == 3 and kw["stride"] == 2 and
                kw.get("mode", 0) in (0, 2)):
            mode = kw.get("mode", 0)
            n = ops.s2_strip_parts(kw["B"], kw["SC"], kw["N"], kw["RW"] if mode == 0 else kw["SW"], mode)
            if n:
                wl = kw["RW"] if mode == 0 else kw["SW"]
                if self._note_weight(kw["w"], self.w_frag and bool(
                        _lib.lib.fr_conv3x3_s2_strip_takes_frag(kw["B"], kw["SC"], wl, mode))):
                    kw = dict(kw, w_frag=1)
                L.append(ops.conv_s2_strip(self.stream, **kw))
                self._last_conv_strips = n if mode == 0 else 0
                return n
        self._last_conv_strips = 0
        self._note_weight(kw["w"], False)
        L.append(ops.conv(self.stream, self.fr, **kw))
        if kw.get("mode", 0) == 2:  # all four parity classes in one launch: [class][M tile] partial rows
            return 4 * ((kw["B"] * (kw["RH"] // 2) * (kw["RW"] // 2) + 127) // 128)
        return (kw["B"] * kw["RH"] * kw["RW"] + 127) // 128

    def _note_weight(self, w, frag):
        """Book-keeping of the weight layouts: a packed weight tensor has ONE layout, so every launch that reads it must agree.
        Returns frag for tensors the plan packs itself (others -- a caller's fp32 master on the parity path -- stay plain)."""
        if self._pack_targets is None or id(w) not in self._pack_targets:
            return False
        frag = bool(frag)
        if self._weight_layout.setdefault(id(w), frag) != frag:
            raise _lib.FrhipError("frhip: two launches want different layouts of one packed weight tensor")
        return frag

    def _side_after_main(self, L):
        """Order the side stream behind everything enqueued on the main stream so far.  Round 6: the edge is the completion
        signal of the launch in front of it (ops.Launch.arm) where that is a launch of the main stream -- an event recorded
        behind it costs the main stream a marker packet its next kernel waits for: +5.1 us per edge against +1.6 (three edges
        per residual unit; tools/edge_probe.hip, profiles/r06_edge_probe.txt).  FRHIP_EDGE_SIGNAL=0: A/B switch."""
        if self.dual:
            ev = torch.cuda.Event()
            last = L[-1] if L else None
            if (self.edge_signal and isinstance(last, ops.Launch) and last.tstream is None and last.stop_event is None):
                ev.record(self.stream1_t)  # creates the hipEvent_t (torch allocates it at the first record)
                last.arm(ev, self.stream)
            else:
                L.append(_EvRecord(ev, self.stream1_t))
            L.append(_EvWait(self.stream2_t, ev))

    def _wgrad(self, L, param=None, **kw):
        """Append a weight-gradient launch: LDS-strip kernel for bf16 stride-1 3x3 layers, generic otherwise.  Returns the
        parameters whose gradients are final once this launch has run (a deferring launch completes its predecessor's)."""
        if (self.fr == FR_BF16 and self.use_strip and kw["KH"] == 3 and kw["stride"] == 1 and
                ops.wgrad_strip_supported(kw["Cout"], kw["SC"], kw["SW"])):
            tiles = (kw["Cout"] // 64) * (kw["SC"] // 64)
            rows = {112: 2, 56: 4, 28: 7, 14: 14, 7: 7}[kw["SW"]]
            fills = kw["B"] * (kw["SW"] // rows) // (4 if kw["SW"] == 7 else 1)
            groups = int(max(1, min(fills, self.wgrad_wgs // tiles if tiles <= self.wgrad_wgs else 1)))
            return self._slab_launch(L, dict(kw, nsplit=groups), groups * kw["Cout"] * 9 * kw["SC"], param=param)
        if (self.fr == FR_BF16 and self.use_strip and self.use_s2 and kw["KH"] == 3 and kw["stride"] == 2 and
                kw["GW"] in (56, 28, 14, 7) and kw["SW"] == 2 * kw["GW"] and kw["Cout"] % 64 == 0 and kw["SC"] % 64 == 0):
            # stride-2 layers: the same kernel on the four parity planes of the input (conv_wgrad_strip.hip, S2)
            tiles = (kw["Cout"] // 64) * (kw["SC"] // 64)
            rows, nimg = {56: (2, 1), 28: (4, 1), 14: (7, 1), 7: (7, 2)}[kw["GW"]]
            fills = (kw["B"] * (kw["GW"] // rows) + nimg - 1) // nimg
            groups = int(max(1, min(fills, self.wgrad_wgs // tiles if tiles <= self.wgrad_wgs else 1)))
            return self._slab_launch(L, dict(kw, nsplit=groups), groups * kw["Cout"] * 9 * kw["SC"], param=param)
        if kw.get("nsplit", 1) > 1:  # pixel slices go to slabs and are added in a fixed order (no float atomics)
            return self._slab_launch(L, kw, kw["nsplit"] * kw["Cout"] * kw["KH"] * kw["KW"] * kw["SC"], strip=False,
                                     param=param)
        l = ops.wgrad(self.stream2, self.fr, **kw)
        l.tstream = self.stream2_t
        L.append(l)
        return [param]

    def _slab_buffer(self, which, need):
        """One of the two slab buffers, grown on demand (earlier launches are re-pointed at the new allocation)."""
        name, users = (("slab", self._slab_users), ("slab2", self._slab2_users), ("slab3", self._slab3_users))[which]
        buf = getattr(self, name)
        if buf is None or buf.numel() < need:
            buf = torch.empty(need, device=self.device)
            setattr(self, name, buf)
            for l, field in users:
                setattr(l.keep[0], field, ops.ptr(buf))
        return buf, users

    def _flush_pending(self, L):
        """Sum the slabs a deferring launch left behind with a launch of its own (end of the list, or in front of a
        launch that cannot fold them).  Returns the parameter whose gradient became final, or None."""
        if self._pending is None:
            return None
        l0, groups, n, dw, param, which = self._pending
        self._pending = None
        buf, users = self._slab_buffer(which, 0)
        r = ops.call("fr_reduce_slabs", buf, groups, n, dw, self.stream2)
        r.tstream = self.stream2_t
        users.append((_ArgProxy(r, 0), "ptr"))
        L.append(r)
        return param

    def _slab_launch(self, L, kw, need, strip=True, param=None):
        """Append a slab-mode weight-gradient launch.  Returns the parameters whose gradients THIS launch completes:
        [param] for a launch that sums its own slabs, the previous deferring launch's parameter for one that folds it,
        [] for the first deferring launch."""
        done = []
        if strip and self.defer_slabs:
            probe = ops._fill(_lib.FrWgradArgs(), **dict(kw, slab=self.part))
            if _lib.lib.fr_conv_wgrad_strip_defers(ctypes.byref(probe)):
                which = self._slab_flip
                self._slab_flip ^= 1
                buf, users = self._slab_buffer(which, need)
                kw = dict(kw, slab=buf, defer=1)
                if self._pending is not None:
                    _l0, pg, pn, pdw, pparam, pwhich = self._pending
                    pbuf, pusers = self._slab_buffer(pwhich, 0)
                    kw.update(prev_slab=pbuf, prev_dw=pdw, prev_n=pn, prev_groups=pg)
                    done.append(pparam)
                l = ops.wgrad_strip(self.stream2, **kw)
                l.tstream = self.stream2_t
                users.append((l, "slab"))
                if self._pending is not None:
                    pusers.append((l, "prev_slab"))
                self._pending = (l, kw["nsplit"], kw["Cout"] * kw["KH"] * kw["KW"] * kw["SC"], kw["dw"], param, which)
                L.append(l)
                return done
        # A launch that sums its own slabs (1x1 shortcut weight gradients, shapes off the strip tables).  Rounds 3-5 first
        # flushed the slabs a deferring launch had left behind (fr_reduce_slabs: 34-53 us on the weight-gradient stream, three
        # times per IR-50 step) so that both could share buffer 0; with a buffer of its own the launch goes out at once and
        # the deferred sum stays with the next deferring launch (FRHIP_WGRAD_SKIP_FLUSH=0: the old order).
        if _switch("FRHIP_WGRAD_SKIP_FLUSH", 1) and self.defer_slabs:
            buf, users = self._slab_buffer(2, need)
        else:
            flushed = self._flush_pending(L)  # keep the side stream's sums in order
            if flushed is not None:
                done.append(flushed)
            buf, users = self._slab_buffer(0, need)
        kw = dict(kw, slab=buf)
        l = ops.wgrad_strip(self.stream2, **kw) if strip else ops.wgrad(self.stream2, self.fr, **kw)
        l.tstream = self.stream2_t
        users.append((l, "slab"))
        L.append(l)
        done.append(param)
        return done

    # ---- forward -----------------------------------------------------------------------------------
    def _bn_fields(self, bn, count):
        """The BatchNorm arguments of fr_bn_finalize as a struct (fr_bn_finalize_res takes two)."""
        m = bn.mod
        return ops.bn_fin(count, m.weight, m.bias, m.eps, m.momentum if m.momentum is not None else 0.1,
                          m.running_mean if m.track_running_stats else None,
                          m.running_var if m.track_running_stats else None,
                          m.num_batches_tracked if m.track_running_stats else None, bn.mean, bn.invstd, bn.scale, bn.shift)

    def _res_edge(self, i):
        """Non-zero when unit i's output is formed by unit i+1's conv1 and its statistics come from moments: 1 = plain unit
        (FR_PRO_RESBN, fr_bn_finalize_res), 2 = squeeze-excite unit (FR_PRO_RESBN_SE, per-image moments through
        fr_se_pool_parts_mlp_fwd_res)."""
        if not self.res_moments or i < 0 or i + 1 >= len(self.units):
            return 0
        u, n = self.units[i], self.units[i + 1]
        if u.sc_conv is not None or u.stride != 1 or n.cin != u.depth:
            return 0
        bns = (self.ubuf[i]["bn1"], self.ubuf[i]["bn2"], self.ubuf[i + 1]["bn1"])
        if not all(b.mod.training for b in bns):
            return 0
        # conv2 of unit i and conv1 of unit i + 1 on LDS-strip instances (the 64 -> 64 rolling-window kernel takes neither.
        # Measured and removed: the two 56x56 64-channel edges on the strip instance instead -- 14.83-15.07 against 14.87-15.04
        # ms per step, nothing: what the apply pass costs there the slower convolution instance gives back)
        if not _lib.lib.fr_conv3x3_strip_serves_resbn(self.B, u.depth, u.Ho):
            return 0
        if n.cin != n.depth and ops.strip_parts(self.B, n.cin, n.depth, n.H, ops.EPI_STORE) <= 0:
            return 0
        if u.se is None:
            return 1
        # squeeze-excite: the gates weigh every image differently, so the moments are kept per image -- conv2's partial rows
        # must be strips of single images (not the multi-image 7x7 workgroups) and so must the consumer's workgroups
        if not self.res_moments_se or self.B > 65535:
            return 0
        rows2 = ops.strip_parts(self.B, u.depth, u.depth, u.Ho, ops.EPI_STATS_X)
        rows1 = ops.strip_parts(self.B, n.cin, n.depth, n.H, ops.EPI_STORE) if n.cin != n.depth else rows2
        if rows2 <= 0 or rows2 % self.B or rows1 <= 0 or rows1 % self.B or (u.depth // 8) > 256 or 256 % (u.depth // 8):
            return 0
        return 2

    def _c1_parts(self, B, Ho, K, N, stride, H):
        """Partial rows of fr_conv1x1_stream for a 1x1 convolution K -> N on a Ho x Ho output grid, 0 = use the generic GEMM."""
        if self.fr != FR_BF16 or not self.use_strip or H != Ho * stride:
            return 0
        if B < 32:
            # small batches keep the tiled GEMM: nothing to gain there, and the batch-4 bf16 golden fixture sits within the
            # noise of the summation order of these statistics (profiles/r04_bf16_g6b_kernel_selection_noise.txt)
            return 0
        return ops.conv1x1_stream_parts(B, Ho, Ho, K, N)

    def _bn_train_launches(self, L, bn, part, nparts, count):
        m = bn.mod
        st = self.stream
        if self.fold:  # one fr_bn_eval_coeffs_multi launch in front of the weight packing covers every BatchNorm
            self._fold_bns.append(bn)
            return
        if m.training:
            L.append(ops.call("fr_bn_finalize", part, nparts, bn.C, float(count), m.weight, m.bias, float(m.eps),
                              float(m.momentum if m.momentum is not None else 0.1),
                              m.running_mean if m.track_running_stats else None,
                              m.running_var if m.track_running_stats else None,
                              m.num_batches_tracked if m.track_running_stats else None, bn.mean, bn.invstd, bn.scale,
                              bn.shift, st))
        else:
            L.append(ops.call("fr_bn_eval_coeffs", m.running_mean, m.running_var, m.weight, m.bias, float(m.eps),
                              bn.C, bn.mean, bn.invstd, bn.scale, bn.shift, st))

    def _build_forward(self):
        B, S, st, fr = self.B, self.S, self.stream, self.fr
        P = []  # weight packing (runs every step: master weights change)
        L = []
        self._pack_reqs = []  # (master, wp|None, wt|None, Cout, taps, Cin[, oscale]) -> one multi-tensor launch
        self._fold_bns = []   # fold mode: every BatchNorm whose eval coefficients the multi launch computes
        fold = self.fold
        stats_epi = ops.EPI_STORE if fold else ops.EPI_STATS  # eval mode: nobody reads the partial sums
        stats_part = None if fold else self.part
        first_bn = self.ubuf[0]["bn1"]
        if self.body_only:
            # a bare stack of residual units: the caller's activation sits in z0; its statistics for the first BN1
            C0 = self.units[0].cin
            nb = ops.grid_blocks(self.M0, C0, fr)
            if not fold:
                L.append(ops.call("fr_channel_stats", self.z0, self.M0, C0, self.part, nb, fr, st))
            self._bn_train_launches(L, first_bn, self.part, nb, self.M0)
        else:
            sc, sb, sp = self.stem
            # ---- stem: im2col -> GEMM(+stats) -> BN+PReLU apply (+stats for unit 0's BN1)
            w0 = sc.weight
            P.append(ops.call("fr_pack_stem", w0, w0.stride(0), w0.stride(1), w0.stride(2), w0.stride(3), self.W0p, 64,
                              w0.shape[1], self.K0, fr, st))
            self.l_im2col = None  # bound per call (input pointer changes)
            two_pass = self.stem_two_pass
            if two_pass:
                # Round 4: the GEMM is 13 GFLOP, its output 411 MB.  Pass 1 leaves only the statistics of y0; pass 2 recomputes
                # y0 and writes z0 = PReLU(BN0(y0)) (+ y0 for the backward pass) with the statistics of z0 in its epilogue:
                # the fr_bn_apply pass over the stem output (822 MB of traffic, 161 us at batch 256) is gone.
                mt0 = int(min(2048, (self.M0 + 63) // 64))
                if self.bn0.mod.training:
                    L.append(ops.call("fr_stem_gemm", self.X0, self.W0p, None, self.part, self.M0, self.K0, mt0, st))
                self._bn_train_launches(L, self.bn0, self.part, mt0, self.M0)
                L.append(ops.call("fr_stem_gemm_bn_prelu", self.X0, self.W0p, self.bn0.scale, self.bn0.shift, sp.weight,
                                  self.y0, self.z0, stats_part, self.M0, self.K0, mt0, st))
                self._bn_train_launches(L, first_bn, self.part, mt0, self.M0)
            elif self.use_stem_gemm:  # 3.2 M rows x 64 columns x K0: the row-streaming kernels of stem_gemm.hip
                mt0 = int(min(2048, (self.M0 + 63) // 64))
                L.append(ops.call("fr_stem_gemm", self.X0, self.W0p, self.y0, self.part, self.M0, self.K0, mt0, st))
            else:
                mt0 = (self.M0 + 127) // 128
                L.append(ops.conv(st, fr, src=self.X0, w=self.W0p, out=self.y0, B=self.M0, RH=1, RW=1, SH=1, SW=1,
                                  SC=self.K0, N=64, KH=1, KW=1, stride=1, pad=0, mode=0, lda=self.K0, ldc=64, pro=0,
                                  epi=ops.EPI_STATS, part=self.part))
            if not two_pass:
                self._bn_train_launches(L, self.bn0, self.part, mt0, self.M0)
                nb = ops.grid_blocks(self.M0, 64, fr)
                L.append(ops.bn_apply(st, fr, x=self.y0, out=self.z0, scale=self.bn0.scale, shift=self.bn0.shift,
                                      slope=sp.weight, part=stats_part, B=B, H=S, W=S, C=64, res_kind=0, res_stride=1,
                                      nblocks=nb))
                self._bn_train_launches(L, first_bn, self.part, nb, self.M0)
        x = self.z0
        # Round 6: the packing of the 3x3 weights (350 MB of traffic, 60 us) runs on the weight-gradient stream, idle in the
        # forward pass, beside the stem (im2col rows + two GEMM passes, HBM-bound but below the HBM rate on their own); the main
        # stream waits for it in front of the first residual unit.  Training plans with two streams; FRHIP_PACK_SIDE=0: A/B
        self.pack_side = (self.dual and not fold and not self.infer and not self.body_only and
                          bool(_switch("FRHIP_PACK_SIDE", 1)))
        if self.pack_side:
            self._pack_ev0, self._pack_ev1 = torch.cuda.Event(), torch.cuda.Event()
            L.append(_EvWait(self.stream1_t, self._pack_ev1))
        for i, u in enumerate(self.units):
            d = self.ubuf[i]
            rin, rout = B * u.H * u.H, B * u.Ho * u.Ho
            w1, w2 = self._conv_master(u.conv1), self._conv_master(u.conv2)
            if fr == FR_BF16:
                self._pack_reqs.append((w1, d["wp1"], d["wt1"], u.depth, 9, u.cin))
                self._pack_reqs.append((w2, d["wp2"], d["wt2"], u.depth, 9, u.depth))
                wp1, wp2 = d["wp1"], d["wp2"]
            else:
                self._pack_reqs.append((w1, None, d["wt1"], u.depth, 9, u.cin))
                self._pack_reqs.append((w2, None, d["wt2"], u.depth, 9, u.depth))
                wp1, wp2 = w1, w2
            bn1, bn2 = d["bn1"], d["bn2"]
            folded = fold and u.se is None and (u.sc_conv is not None or u.stride == 1)
            edge_in, edge_out = self._res_edge(i - 1), self._res_edge(i)
            if edge_in:
                # x (the previous unit's output) does not exist yet: this launch forms it from that unit's y2 and input,
                # stores it, and applies BN1 to it
                pd = self.ubuf[i - 1]
                se_kw = dict(pro=ops.PRO_RESBN_SE, pro_g=pd["s"]) if edge_in == 2 else dict(pro=ops.PRO_RESBN)
                self._conv(L, src=pd["y2"], src2=x_in, pro_out=x, w=wp1, out=d["y1"], B=B, RH=u.H, RW=u.H, SH=u.H, SW=u.H,
                           SC=u.cin, N=u.depth, KH=3, KW=3, stride=1, pad=1, mode=0, lda=u.cin, ldc=u.depth,
                           pro_a=pd["bn2"].scale, pro_b=pd["bn2"].shift, pro_c=bn1.scale, pro_d=bn1.shift,
                           epi=ops.EPI_STORE, **se_kw)
            else:
                self._conv(L, src=x, w=wp1, out=d["y1"], B=B, RH=u.H, RW=u.H, SH=u.H, SW=u.H, SC=u.cin,
                           N=u.depth, KH=3, KW=3, stride=1, pad=1, mode=0, lda=u.cin, ldc=u.depth,
                           pro=ops.PRO_BN, pro_a=bn1.scale, pro_b=bn1.shift, epi=ops.EPI_STORE)
            x_in = x  # this unit's input (the residual term of its output)
            # (Round 4, measured and removed: the convolved shortcut of a stage entry + its statistics launch on the side stream
            # beside conv2 -- the side stream idles during the forward pass and the three launches take 24-51 us each on the
            # generic GEMM: 15.01-15.05 against 14.98-15.00 ms per step, profiles/r04_ab_shortcut_on_side_stream.txt: the two
            # event edges cost what the overlap saves.)
            if folded:
                # inference with BN2 (and the shortcut BN) folded into the packed weights: out = conv2'(PReLU(y1)) +
                # shift2 [+ shiftS] + shortcut straight from conv2's epilogue -- y2 is never written, no BN-apply pass
                self._fold_bns += [bn2] + ([d["bnS"]] if u.sc_conv is not None else [])
                self._pack_reqs.append((w2, d["wf2"], None, u.depth, 9, u.depth, bn2.scale))
                if u.sc_conv is not None:
                    ws = self._conv_master(u.sc_conv)
                    self._pack_reqs.append((ws, d["wfS"], None, u.depth, 1, u.cin, d["bnS"].scale))
                    L.append(ops.conv(st, fr, src=x, w=d["wfS"], out=d["yS"], B=B, RH=u.Ho, RW=u.Ho, SH=u.H, SW=u.H,
                                      SC=u.cin, N=u.depth, KH=1, KW=1, stride=u.stride, pad=0, mode=0, lda=u.cin,
                                      ldc=u.depth, pro=0, epi=ops.EPI_STORE))
                    res, shift_s = d["yS"], d["bnS"].shift
                else:
                    res, shift_s = x, self.zeros_c[:u.depth]
                self._conv(L, src=d["y1"], w=d["wf2"], out=d["out"], B=B, RH=u.Ho, RW=u.Ho, SH=u.H, SW=u.H, SC=u.depth,
                           N=u.depth, KH=3, KW=3, stride=u.stride, pad=1, mode=0, lda=u.depth, ldc=u.depth,
                           ldaux=u.depth, pro=ops.PRO_PRELU, pro_a=u.prelu.weight, epi=ops.EPI_BIAS_RES, epi_a=bn2.shift,
                           epi_b=shift_s, aux=res)
                nxt = self.ubuf[i + 1]["bn1"] if i + 1 < len(self.units) else (None if self.body_only else self.bn_out)
                if nxt is not None:
                    self._bn_train_launches(L, nxt, None, 0, rout)
                x = d["out"]
                continue
            if edge_out:
                if edge_out == 2 and edge_in != 2:  # head of a squeeze-excite chain: per-image moments of its input
                    d["xm"] = torch.empty(B, 2, u.depth, device=self.device)
                    L.append(ops.call("fr_image_moments", x_in, B, u.H * u.H, u.depth, d["xm"], st))
                np2 = self._conv(L, src=d["y1"], w=wp2, out=d["y2"], B=B, RH=u.Ho, RW=u.Ho, SH=u.H, SW=u.H, SC=u.depth,
                                 N=u.depth, KH=3, KW=3, stride=1, pad=1, mode=0, lda=u.depth, ldc=u.depth, ldaux=u.depth,
                                 pro=ops.PRO_PRELU, pro_a=u.prelu.weight, epi=ops.EPI_STATS_X, aux=x_in, part=self.part)
                nxt = self.ubuf[i + 1]["bn1"]
                if edge_out == 1:
                    L.append(ops.call("fr_bn_finalize_res", self.part, np2, u.depth, self._bn_fields(bn2, rout), bn1.mean,
                                      bn1.invstd, float(bn1.mod.eps), self._bn_fields(nxt, rout), st))
                else:
                    L.append(ops.call("fr_bn_finalize_res", self.part, np2, u.depth, self._bn_fields(bn2, rout), None, None,
                                      0.0, None, st))
                    d["om"] = torch.empty(B, 2, u.depth, device=self.device)
                    xm = self.ubuf[i - 1]["om"] if edge_in == 2 else d["xm"]
                    L.append(ops.call("fr_se_pool_parts_mlp_fwd_res", self.part, np2 // B, 3, bn2.scale, bn2.shift,
                                      u.se.fc1.weight, u.se.fc2.weight, d["pooled"], d["hidden"], d["s"], B, u.Ho * u.Ho,
                                      u.depth, u.se.fc1.out_channels, xm, d["om"], st))
                    self._bn_train_launches(L, nxt, d["om"], B, rout)
                x = d["out"]  # written by the next unit's conv1
                continue
            np2 = self._conv(L, src=d["y1"], w=wp2, out=d["y2"], B=B, RH=u.Ho, RW=u.Ho, SH=u.H, SW=u.H,
                             SC=u.depth, N=u.depth, KH=3, KW=3, stride=u.stride, pad=1, mode=0, lda=u.depth,
                             ldc=u.depth, pro=ops.PRO_PRELU, pro_a=u.prelu.weight, epi=stats_epi, part=stats_part)
            self._bn_train_launches(L, bn2, self.part, np2, rout)
            strips2 = self._last_conv_strips if not fold else 0  # conv2's partial rows, if they are whole strips of single images
            if u.sc_conv is not None:
                ws = self._conv_master(u.sc_conv)
                if fr == FR_BF16:
                    self._pack_reqs.append((ws, d["wpS"], d["wtS"], u.depth, 1, u.cin))
                    wps = d["wpS"]
                else:
                    self._pack_reqs.append((ws, None, d["wtS"], u.depth, 1, u.cin))
                    wps = ws
                kws = dict(src=x, w=wps, out=d["yS"], B=B, RH=u.Ho, RW=u.Ho, SH=u.H, SW=u.H, SC=u.cin, N=u.depth, KH=1, KW=1,
                           stride=u.stride, pad=0, mode=0, lda=u.cin, ldc=u.depth, pro=0, epi=stats_epi, part=stats_part)
                nps = self._c1_parts(B, u.Ho, u.cin, u.depth, u.stride, u.H)
                if nps:
                    # Round 4: the convolved shortcut as a row-streaming GEMM with its weights in registers (conv1x1_stream.hip)
                    L.append(ops.conv1x1_stream(st, **kws))
                    self._bn_train_launches(L, d["bnS"], self.part, nps, rout)
                else:
                    L.append(ops.conv(st, fr, **kws))
                    self._bn_train_launches(L, d["bnS"], self.part, (rout + 127) // 128, rout)
            if u.se is not None:
                R = u.se.fc1.out_channels
                if strips2 and strips2 % B == 0 and u.sc_conv is None:
                    # the squeeze from conv2's per-strip column sums (still in self.part): no pass over y2, and in the
                    # same launch as the MLP
                    L.append(ops.call("fr_se_pool_parts_mlp_fwd", self.part, strips2 // B, bn2.scale, bn2.shift,
                                      u.se.fc1.weight, u.se.fc2.weight, d["pooled"], d["hidden"], d["s"], B,
                                      u.Ho * u.Ho, u.depth, R, st))
                else:
                    L.append(ops.call("fr_se_pool", d["y2"], bn2.scale, bn2.shift, d["pooled"], B, u.Ho * u.Ho,
                                      u.depth, fr, st))
                    L.append(ops.call("fr_se_mlp_fwd", d["pooled"], u.se.fc1.weight, u.se.fc2.weight, d["hidden"],
                                      d["s"], B, u.depth, R, st))
            nb = ops.grid_blocks(rout, u.depth, fr)
            kw = dict(x=d["y2"], out=d["out"], scale=bn2.scale, shift=bn2.shift, part=stats_part, B=B, H=u.Ho, W=u.Ho,
                      C=u.depth, nblocks=nb)
            if u.se is not None:
                kw["se"] = d["s"]
            if u.sc_conv is None:
                kw.update(res=x, res_kind=1, res_stride=u.stride)
            else:
                kw.update(res=d["yS"], res_kind=2, res_stride=1, rscale=d["bnS"].scale, rshift=d["bnS"].shift)
            nxt = self.ubuf[i + 1]["bn1"] if i + 1 < len(self.units) else (None if self.body_only else self.bn_out)
            if nxt is None:
                kw["part"] = None  # nobody consumes the statistics of a bare stack's output
            L.append(ops.bn_apply(st, fr, **kw))
            if nxt is not None:
                self._bn_train_launches(L, nxt, self.part, nb, rout)
            x = d["out"]
        if self.body_only:
            self.feat = x
            P.append(self._pack_launch())
            if fold:
                P.insert(0, self._eval_coeffs_launch())
            self.pack_list, self.fwd_list = P, L
            return
        # ---- output layer: BN -> Dropout -> Flatten -> Linear(+bias) -> BN1d
        ob, od, ol, ob1 = self.out
        last = self.units[-1]
        C = last.depth
        if self.lin_cm:
            drop = ops.call("fr_bn_dropout_cm", x, self.a, self.bn_out.scale, self.bn_out.shift, B, C, self.HWo, 0.0, 0, fr,
                            st)
            L.append(drop)
            self.l_drop_fwd = drop
            self.lin_splitk = int(_lib.lib.fr_linear_slices(512, self.feat_in))
            self.lin_slab = torch.empty(self.lin_splitk * B * 512, device=self.device)
            L.append(ops.call("fr_linear_fwd", self.a, ol.weight, ol.bias, self.lin_slab, B, 512, self.feat_in,
                              self.lin_splitk, st))
        else:
            P.append(ops.call("fr_permute_linear", ol.weight, self.Wlin, self.WlinT, 512, C, self.HWo, 0, fr, st))
            drop = ops.call("fr_bn_dropout", x, self.a, self.bn_out.scale, self.bn_out.shift, B * self.HWo, C, self.HWo,
                            0.0, 0, fr, st)
            L.append(drop)
            self.l_drop_fwd = drop
            # split-K over 25088: every K slice stores its [B][512] partial (+ bias in slice 0) to its own slab and the
            # slabs are added in a fixed order -- reproducible, unlike atomics, and the sums are formed in double
            nk = self.feat_in // 32
            self.lin_splitk = max(1, min(64, nk // 16))
            self.lin_slab = torch.empty(self.lin_splitk * B * 512, device=self.device)
            L.append(ops.conv(st, fr, src=self.a, w=self.Wlin, out=self.lin_slab, B=B, RH=1, RW=1, SH=1, SW=1,
                              SC=self.feat_in, N=512, KH=1, KW=1, stride=1, pad=0, mode=0, lda=self.feat_in, ldc=512, pro=0,
                              epi=ops.EPI_SLAB, out_f32=1, splitk=self.lin_splitk, bias=ol.bias))
        L.append(ops.call("fr_reduce_parts", self.lin_slab, self.lin_splitk, 1, B * 512, self.f, None, None, st))
        nbf = ops.grid_blocks(B, 512, FR_F32)
        if not fold:
            L.append(ops.call("fr_channel_stats", self.f, B, 512, self.part, nbf, FR_F32, st))
        self._bn_train_launches(L, self.bn1d, self.part, nbf, B)
        L.append(ops.bn_apply(st, FR_F32, x=self.f, out=self.feat, scale=self.bn1d.scale, shift=self.bn1d.shift, B=B,
                              H=1, W=1, C=512, res_kind=0, res_stride=1, nblocks=nbf))
        if self.pack_side:
            pl = self._pack_launch(self.stream2)
            pl.tstream = self.stream2_t
            self.pack_side_list = [pl]
        else:
            P.append(self._pack_launch())
        if fold:  # the coefficients feed the weight folding: first launch of the step
            P.insert(0, self._eval_coeffs_launch())
        self.pack_list, self.fwd_list = P, L

    def _finish_pack(self):
        """Behind the forward AND backward lists: tell the packing launch which copies are read in fragment order."""
        arr, dirty = self._pack_arr, False
        for i, req in enumerate(self._pack_reqs):
            bits = sum(b for b, t in ((1, req[1]), (2, req[2])) if t is not None and self._weight_layout.get(id(t), False))
            if bits != arr[i].frag:
                arr[i].frag, dirty = bits, True
        if dirty:
            self._pack_table.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))

    def _pack_launch(self, stream=None):
        """One launch that writes the compute-dtype and transposed copies of every conv weight of the network."""
        n = len(self._pack_reqs)
        arr = self._pack_arr = (_lib.FrPackTensor * n)()
        chunks = []
        for i, req in enumerate(self._pack_reqs):
            w, wp, wt, cout, taps, cin = req[:6]
            arr[i].w = w.data_ptr()
            arr[i].wp = wp.data_ptr() if wp is not None else None
            arr[i].wt = wt.data_ptr() if wt is not None else None
            arr[i].oscale = req[6].data_ptr() if len(req) > 6 else None
            arr[i].Cout, arr[i].taps, arr[i].Cin = cout, taps, cin
            if self.fr == FR_BF16 and cout % 64 == 0 and cin % 64 == 0:
                # 64 x 64 tiles with 16-byte accesses (chunk index -(tile + 1)): every 3x3 / shortcut convolution of the IR nets
                chunks.extend((i, -(t + 1)) for t in range(taps * (cout // 64) * (cin // 64)))
                continue
            tiles = taps * ((cout + 31) // 32) * ((cin + 31) // 32)
            chunks.extend((i, t) for t in range(tiles))
        self._pack_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self._pack_chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1).to(self.device)
        table = ctypes.cast(ctypes.c_void_p(self._pack_table.data_ptr()), ctypes.POINTER(_lib.FrPackTensor))
        return ops.Launch("fr_pack_weights_multi", [table, ops.ptr(self._pack_chunks), len(chunks), self.fr,
                                                    stream if stream is not None else self.stream], keep=(self._pack_reqs,))

    def _eval_coeffs_launch(self):
        """One launch that turns the running statistics of every BatchNorm into (mean, invstd, scale, shift)."""
        seen, bns = set(), []
        for bn in self._fold_bns:
            if id(bn) not in seen:
                seen.add(id(bn))
                bns.append(bn)
        arr = (_lib.FrBnEvalEntry * len(bns))()
        for i, bn in enumerate(bns):
            m = bn.mod
            arr[i].rm, arr[i].rv = m.running_mean.data_ptr(), m.running_var.data_ptr()
            arr[i].gamma, arr[i].beta = m.weight.data_ptr(), m.bias.data_ptr()
            arr[i].mean, arr[i].invstd = bn.mean.data_ptr(), bn.invstd.data_ptr()
            arr[i].scale, arr[i].shift = bn.scale.data_ptr(), bn.shift.data_ptr()
            arr[i].C, arr[i].eps = bn.C, float(m.eps)
        self._bn_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        table = ctypes.cast(ctypes.c_void_p(self._bn_table.data_ptr()), ctypes.POINTER(_lib.FrBnEvalEntry))
        return ops.Launch("fr_bn_eval_coeffs_multi", [table, len(bns), self.stream], keep=(bns,))

    # ---- backward ----------------------------------------------------------------------------------
    def _reduce(self, L, nparts, K, C, o0, o1, o2=None):
        """Add the partial rows the last launch wrote."""
        L.append(ops.call("fr_reduce_parts", self.part, nparts, K, C, o0, o1, o2, self.stream))

    def _bn_grads(self, bn):
        """(dbeta target, dgamma target): the parameter gradients when they train, scratch otherwise."""
        m = bn.mod
        gb = self.grad_of(m.bias)
        gg = self.grad_of(m.weight)
        return (gb if gb is not None else self.sums[0, :bn.C]), (gg if gg is not None else self.sums[1, :bn.C])

    def _s01(self, bn, s0, s1):
        """BN in eval mode has no batch-statistics terms in its backward."""
        if bn.mod.training:
            return s0, s1
        return self.zeros_c[:bn.C], self.zeros_c[:bn.C]

    def _build_backward(self):
        B, st, fr = self.B, self.stream, self.fr
        L = []
        self.ready_marks = []  # (index into L after which a group of params is complete, [params])
        last = self.units[-1]
        C = last.depth
        rows_o = B * self.HWo
        if self.body_only:
            return self._build_backward_units(L, self.g_pp[0][:rows_o * C], 0)
        ob, od, ol, ob1 = self.out
        # ---- BN1d backward (fp32 tensors): g_in arrives in self.g_feat_in (bound per call)
        self.g_feat_in = torch.empty(B, 512, device=self.device)
        nbf = ops.grid_blocks(B, 512, FR_F32)
        db, dg = self._bn_grads(self.bn1d)
        common = dict(g=self.g_feat_in, x=self.f, mean=self.bn1d.mean, invstd=self.bn1d.invstd, rows=B, C=512,
                      rows_per_image=1, nblocks=nbf)
        L.append(ops.bn_bwd_reduce(st, FR_F32, part=self.part, **common))
        self._reduce(L, nbf, 3, 512, db, dg)
        s0, s1 = self._s01(self.bn1d, db, dg)
        L.append(ops.bn_bwd_apply(st, FR_F32, gx=self.g_f32, gamma=ob1.weight, s0=s0, s1=s1, inv_count=1.0 / B,
                                  **common))
        # Linear bias gradient = column sums of g_f
        gbias = self.grad_of(ol.bias)
        if gbias is not None:
            L.append(ops.call("fr_channel_stats", self.g_f32, B, 512, self.part, nbf, FR_F32, st))
            self._reduce(L, nbf, 2, 512, gbias, None)
        if fr == FR_BF16:
            L.append(ops.call("fr_cast", self.g_f32, self.g_fT, B * 512, FR_F32, FR_BF16, st))
            gfT = self.g_fT
        else:
            gfT = self.g_f32
        # Linear weight gradient (packed layout) -> torch layout
        glw = self.grad_of(ol.weight)
        g_a = self.g_xh[:B * self.feat_in].view(B, self.feat_in)
        head_done = None
        if self.lin_cm:
            if glw is not None:  # straight into the master's layout (one workgroup per element: the add is onto the zeroed arena)
                # Round 6: on the weight-gradient stream, which has nothing to do until the last unit's data gradients exist
                # (70 us of a 51-MB write that feeds nothing downstream, off a main stream that is a chain of 5-us launches
                # here; FRHIP_LINEAR_WGRAD_SIDE=0: A/B switch)
                side = self.dual and bool(_switch("FRHIP_LINEAR_WGRAD_SIDE", 1))
                if side:
                    self._side_after_main(L)
                lw = ops.wgrad(self.stream2 if side else st, fr, g=gfT, src=self.a, dw=glw, B=B, GH=1, GW=1, Cout=512, SH=1,
                               SW=1, SC=self.feat_in, KH=1, KW=1, stride=1, pad=0, ldg=512, lda=self.feat_in, pro=0, nsplit=1)
                L.append(lw)
                if side:
                    lw.tstream = self.stream2_t
                    head_done = torch.cuda.Event()
                    L.append(_EvRecord(head_done, self.stream2_t))
            # Linear data gradient (c-major) -> dropout backward + back to NHWC -> BN(out) backward
            L.append(ops.call("fr_linear_dgrad", gfT, ol.weight, self.g_cm, B, 512, self.feat_in, st))
            dropb = ops.call("fr_dropout_bwd_cm", self.g_cm, g_a, B, C, self.HWo, 0.0, 0, fr, st)
            self._drop_bwd_idx = (5, 6)
        else:
            if glw is not None:
                L.append(ops.call("fr_fill_rows", self.gWlin, None, 512, self.feat_in, st))
                L.append(ops.wgrad(st, fr, g=gfT, src=self.a, dw=self.gWlin, B=B, GH=1, GW=1, Cout=512, SH=1, SW=1,
                                   SC=self.feat_in, KH=1, KW=1, stride=1, pad=0, ldg=512, lda=self.feat_in, pro=0,
                                   nsplit=1))
                L.append(ops.call("fr_permute_linear", self.gWlin, glw, None, 512, C, self.HWo, 1, FR_F32, st))
            # Linear data gradient -> dropout backward -> BN(out) backward
            L.append(ops.conv(st, fr, src=gfT, w=self.WlinT, out=g_a, B=B, RH=1, RW=1, SH=1, SW=1, SC=512, N=self.feat_in,
                              KH=1, KW=1, stride=1, pad=0, mode=0, lda=512, ldc=self.feat_in, pro=0, epi=ops.EPI_STORE))
            dropb = ops.call("fr_dropout_bwd", g_a, rows_o, C, self.HWo, 0.0, 0, fr, st)
            self._drop_bwd_idx = (4, 5)
        L.append(dropb)
        self.l_drop_bwd = dropb
        x_last = self.ubuf[-1]["out"]
        nb = ops.grid_blocks(rows_o, C, fr)
        db, dg = self._bn_grads(self.bn_out)
        common = dict(g=g_a, x=x_last, mean=self.bn_out.mean, invstd=self.bn_out.invstd, rows=rows_o, C=C,
                      rows_per_image=self.HWo, nblocks=nb)
        L.append(ops.bn_bwd_reduce(st, fr, part=self.part, **common))
        self._reduce(L, nb, 3, C, db, dg)
        s0, s1 = self._s01(self.bn_out, db, dg)
        cur = 0
        g_out = self.g_pp[cur][:rows_o * C]
        L.append(ops.bn_bwd_apply(st, fr, gx=g_out, gamma=ob.weight, s0=s0, s1=s1, inv_count=1.0 / rows_o, **common))
        self.ready_marks.append((len(L), [ob1.weight, ob1.bias, ol.weight, ol.bias, ob.weight, ob.bias], head_done))
        self._build_backward_units(L, g_out, cur)

    def _build_backward_units(self, L, g_out, cur):
        B, st, fr = self.B, self.stream, self.fr
        # ---- residual units in reverse
        unit_done = {}  # unit index -> event recorded on the side stream after its weight gradients
        sums_left = 0  # rows of BN2-backward sums the previous launch left in self.part (0: none)
        for i in range(len(self.units) - 1, -1, -1):
            u, d = self.units[i], self.ubuf[i]
            x = self.ubuf[i - 1]["out"] if i > 0 else self.z0
            rin, rout = B * u.H * u.H, B * u.Ho * u.Ho
            HWo = u.Ho * u.Ho
            bn1, bn2 = d["bn1"], d["bn2"]
            par = i % self.nset if self.dual else 0
            g_y2 = self.g_y2s[par][:rout * u.depth]
            if self.dual:
                # unit i + nset's weight gradients read this buffer set
                if i + self.nset in unit_done:
                    L.append(_EvWait(self.stream1_t, unit_done[i + self.nset]))
            # (Round 4, schedule experiments measured and removed again -- profiles/r04_ab_edges.txt, r04_ab_edge2_wgrad_order.txt:
            # ONE main -> side edge per unit with both weight gradients behind it: +0.26 ms (the first weight gradient then
            # starts a data gradient later); the main stream waiting for the side stream every second unit with four buffer
            # sets: +0.05; no second edge in front of conv1's weight gradient: nothing; conv1's weight gradient before
            # conv2's: +0.39 ms.  Round 5, 128-workgroup weight gradients: conv2's weight gradient launched BEFORE its data
            # gradient -- it needs g_y2 and y1 only --: +0.3 ms at bs 256, +0.9 ms IR-SE-101 bs 128, bit-identical.)
            S = S2 = L

            def edge():
                self._side_after_main(L)
            nb = ops.grid_blocks(rout, u.depth, fr)
            ready = [u.bn2.weight, u.bn2.bias]
            se_kw = {}
            if u.se is not None:
                R = u.se.fc1.out_channels
                g1, g2 = self.grad_of(u.se.fc1.weight), self.grad_of(u.se.fc2.weight)
                if g1 is None:
                    g1 = self.se_scratch[0, :R * u.depth]
                if g2 is None:
                    g2 = self.se_scratch[1, :R * u.depth]
                # gradient wrt the excite scale (a pass over g and y2) + the MLP backward of the same image in one launch.
                # Round 5: with fewer images than CUs the squeeze runs over row slices of an image in 256-thread workgroups
                # (IR-SE-101 + CosFace(28000), bs 128: 17.57 against 17.66-17.81 ms per step; at bs 256 -- pSp -- the one
                # 1024-thread workgroup per image is the faster one: 16.05 against 16.17; profiles/r05_ab_se_slices.txt).
                # (Round 4, measured and removed: the same pass also leaving the per-image sums BN2's backward needs, so that
                # fr_bn_bwd_reduce disappears from the IR-SE units -- IR-SE-101 + CosFace(28000), bs 128: 19.50-19.52 against
                # 19.22-19.30 ms per step; pSp bs 256: 17.31 against 17.22; profiles/r04_ab_se_sums.txt: the pass it deleted ran
                # beside the weight gradients of the side stream, the extra work sat in a 128-workgroup launch.)
                # Round 6: the weight gradients of the two 1x1 convolutions of the MLP (one launch, 6-13 us, 4 workgroups) feed
                # nothing downstream: with two streams they run on the weight-gradient stream behind conv2's data gradient
                # (IR-SE-101 bs 128: 49 launches off a main stream whose channel-wise chain is longer than its convolutions)
                se_side = self.dual and bool(_switch("FRHIP_SE_WGRAD_SIDE", 1))
                # Round 6: the squeeze pass also takes the per-image sums from which BN2's backward sums follow (the gate and
                # its gradient are constant over an image): fr_bn_bwd_reduce -- a third pass over (g, y2) behind the MLP,
                # 14-19 us per unit on the main stream's chain -- is gone (bf16 path; FRHIP_SE_BN_SUMS=0: A/B switch)
                se_sums = (fr == FR_BF16 and bool(_switch("FRHIP_SE_BN_SUMS", 1)) and u.depth % 8 == 0 and
                           (u.depth // 8) <= 256 and 256 % (u.depth // 8) == 0)
                if se_sums:
                    L.append(ops.call("fr_se_gscale_mlp_bwd_sums", g_out, d["y2"], bn2.scale, bn2.shift, bn2.mean, bn2.invstd,
                                      d["s"], d["hidden"], u.se.fc1.weight, u.se.fc2.weight, d["gpooled"], d["gz"], d["gh"],
                                      self.se_gs_part, self.part, B, u.depth, R, HWo, fr, st))
                    if not se_side:
                        L.append(ops.call("fr_se_mlp_wgrad", d["gz"], d["gh"], d["hidden"], d["pooled"], g1, g2, B, u.depth,
                                          R, st))
                else:
                    L.append(ops.call("fr_se_gscale_mlp_bwd", g_out, d["y2"], bn2.scale, bn2.shift, d["s"], d["hidden"],
                                      d["pooled"], u.se.fc1.weight, u.se.fc2.weight, d["gpooled"], None if se_side else g1,
                                      None if se_side else g2, d["gz"], d["gh"], self.se_gs_part if B <= 160 else None, B,
                                      u.depth, R, HWo, fr, st))
                se_wgrad = None
                if se_side:
                    se_wgrad = ops.call("fr_se_mlp_wgrad", d["gz"], d["gh"], d["hidden"], d["pooled"], g1, g2, B, u.depth, R,
                                        self.stream2)
                    se_wgrad.tstream = self.stream2_t
                se_kw = dict(se=d["s"], gse=d["gpooled"])
                ready += [u.se.fc1.weight, u.se.fc2.weight]
            db, dg = self._bn_grads(bn2)
            common = dict(g=g_out, x=d["y2"], mean=bn2.mean, invstd=bn2.invstd, rows=rout, C=u.depth,
                          rows_per_image=HWo, nblocks=nb, **se_kw)
            if u.se is not None and se_sums:
                self._reduce(L, B, 2, u.depth, db, dg)  # one row pair per image, left by fr_se_gscale_mlp_bwd_sums
            elif sums_left:
                self._reduce(L, sums_left, 2, u.depth, db, dg)  # left by the launch that formed g_out (nx below)
            else:
                L.append(ops.bn_bwd_reduce(st, fr, part=self.part, **common))
                self._reduce(L, nb, 3, u.depth, db, dg)
            s0, s1 = self._s01(bn2, db, dg)
            L.append(ops.bn_bwd_apply(st, fr, gx=g_y2, gamma=u.bn2.weight, s0=s0, s1=s1, inv_count=1.0 / rout, **common))
            g_xS = None
            if u.sc_conv is not None:
                bnS = d["bnS"]
                g_yS = self.g_ySs[par][:rout * u.depth]
                db, dg = self._bn_grads(bnS)
                common = dict(g=g_out, x=d["yS"], mean=bnS.mean, invstd=bnS.invstd, rows=rout, C=u.depth,
                              rows_per_image=HWo, nblocks=nb)
                L.append(ops.bn_bwd_reduce(st, fr, part=self.part, **common))
                self._reduce(L, nb, 3, u.depth, db, dg)
                s0, s1 = self._s01(bnS, db, dg)
                L.append(ops.bn_bwd_apply(st, fr, gx=g_yS, gamma=u.sc_bn.weight, s0=s0, s1=s1,
                                          inv_count=1.0 / rout, **common))
                # 1x1 stride-s data gradient = dense GEMM on the Ho x Ho grid; it lands on the pixels (s*i, s*j) of
                # the unit input, which the final bn_bwd_apply adds as a strided scatter (add_kind 2)
                g_xS = self.g_xS[:rout * u.cin]
                kws = dict(src=g_yS, w=d["wtS"], out=g_xS, B=B, RH=u.Ho, RW=u.Ho, SH=u.Ho, SW=u.Ho, SC=u.depth, N=u.cin, KH=1,
                           KW=1, stride=1, pad=0, mode=0, lda=u.depth, ldc=u.cin, pro=0, epi=ops.EPI_STORE)
                if self._c1_parts(B, u.Ho, u.depth, u.cin, 1, u.Ho):
                    L.append(ops.conv1x1_stream(st, **kws))
                else:
                    L.append(ops.conv(st, fr, **kws))
                gws = self.grad_of(u.sc_conv.weight)
                if gws is not None:
                    tiles = ((u.depth + 127) // 128) * ((u.cin + 127) // 128)
                    edge()
                    ready += self._wgrad(S, param=u.sc_conv.weight, g=g_yS, src=x, dw=gws, B=B, GH=u.Ho, GW=u.Ho,
                                         Cout=u.depth, SH=u.H, SW=u.H, SC=u.cin, KH=1, KW=1, stride=u.stride, pad=0,
                                         ldg=u.depth, lda=u.cin, pro=0, nsplit=_wgrad_slices(rout, tiles))
                else:
                    ready.append(u.sc_conv.weight)  # frozen: never reported anyway (on_ready filters requires_grad)
                ready += [u.sc_bn.weight, u.sc_bn.bias]
            # conv2: data gradient with the PReLU backward epilogue, then the weight gradient
            g_y1 = self.g_y1s[par][:rin * u.depth]
            c2 = dict(src=g_y2, w=d["wt2"], out=g_y1, B=B, RH=u.H, RW=u.H, SH=u.Ho, SW=u.Ho, SC=u.depth, N=u.depth,
                      KH=3, KW=3, stride=u.stride, pad=1, lda=u.depth, ldc=u.depth, ldaux=u.depth, pro=0,
                      epi=ops.EPI_PRELU_BWD, aux=d["y1"], epi_a=u.prelu.weight)
            # The PReLU-slope partial sums of this data gradient feed nothing downstream (a parameter gradient): with the
            # side stream they go to a buffer of their own (one per buffer set) and are added there, off the main chain.
            gsl = self.grad_of(u.prelu.weight)
            gsl = gsl if gsl is not None else self.sums[2, :u.depth]
            if self.side_slope and self.part_slope[par] is None:  # this unit's own rows: [<= 4 classes x strips][2][depth] floats
                probe = []
                rows = self._conv_launch(probe, mode=2 if (u.stride == 2 and u.H % 2 == 0) else 1, par_h=-1, par_w=-1,
                                         part=self.part, **c2)
                self.part_slope[par] = torch.zeros(rows * 2 * u.depth + 4096, device=self.device)
            part2 = self.part_slope[par] if self.side_slope else self.part
            if u.stride == 2 and u.H % 2 == 0:
                # one launch per output-pixel parity class: 9/4 taps per pixel instead of 9 (3/4 of them misses)
                # (all four classes in one launch: par = -1; partial rows come back as [class][M tile])
                mt = self._conv(L, mode=2, par_h=-1, par_w=-1, part=part2, **c2)
            else:
                mt = self._conv(L, mode=1, part=part2, **c2)
            gw2 = self.grad_of(u.conv2.weight)
            if self.side_slope:
                edge()  # g_y1 / the slope partials, g_y2 (BN2 backward) and y1 are final
                r = ops.call("fr_reduce_parts", part2, mt, 2, u.depth, gsl, None, None, self.stream2)
                r.tstream = self.stream2_t
                S2.append(r)
                if u.se is not None and se_wgrad is not None:
                    S2.append(se_wgrad)
            else:
                self._reduce(L, mt, 2, u.depth, gsl, None)
            if gw2 is not None:
                tiles = ((u.depth + 127) // 128) ** 2 * 9
                if not self.side_slope:
                    edge()
                kw2 = dict(param=u.conv2.weight, g=g_y2, src=d["y1"], dw=gw2, B=B, GH=u.Ho, GW=u.Ho,
                           Cout=u.depth, SH=u.H, SW=u.H, SC=u.depth, KH=3, KW=3, stride=u.stride, pad=1,
                           ldg=u.depth, lda=u.depth, pro=ops.PRO_PRELU, pro_a=u.prelu.weight,
                           nsplit=_wgrad_slices(rout, tiles))
                ready += self._wgrad(S2, **kw2)
            else:
                ready.append(u.conv2.weight)
            # conv1: data gradient with the BN1-backward sums epilogue, then the weight gradient
            g_xh = self.g_xh[:rin * u.cin]
            db, dg = self._bn_grads(bn1)
            mt = self._conv(L, src=g_y1, w=d["wt1"], out=g_xh, B=B, RH=u.H, RW=u.H, SH=u.H, SW=u.H,
                            SC=u.depth, N=u.cin, KH=3, KW=3, stride=1, pad=1, mode=1, lda=u.depth, ldc=u.cin,
                            ldaux=u.cin, pro=0, epi=ops.EPI_BNBWD, aux=x, epi_a=bn1.mean, epi_b=bn1.invstd,
                            part=self.part)
            self._reduce(L, mt, 2, u.cin, db, dg)
            gw1 = self.grad_of(u.conv1.weight)
            if gw1 is not None:
                tiles = ((u.depth + 127) // 128) * ((u.cin + 127) // 128) * 9
                edge()  # g_y1 (conv2 data gradient) is final on the main stream
                ready += self._wgrad(S, param=u.conv1.weight, g=g_y1, src=x, dw=gw1, B=B, GH=u.H, GW=u.H, Cout=u.depth,
                                     SH=u.H, SW=u.H, SC=u.cin, KH=3, KW=3, stride=1, pad=1, ldg=u.depth, lda=u.cin,
                                     pro=ops.PRO_BN, pro_a=bn1.scale, pro_b=bn1.shift, nsplit=_wgrad_slices(rin, tiles))
            else:
                ready.append(u.conv1.weight)
            ready += [u.prelu.weight, u.bn1.weight, u.bn1.bias]
            # unit input gradient = BN1 backward of g_xh + shortcut gradient
            nxt = 1 - cur
            g_x = self.g_pp[nxt][:rin * u.cin]
            s0, s1 = self._s01(bn1, db, dg)
            kw = dict(g=g_xh, x=x, gx=g_x, mean=bn1.mean, invstd=bn1.invstd, gamma=u.bn1.weight, s0=s0, s1=s1,
                      rows=rin, inv_count=1.0 / rin, C=u.cin, rows_per_image=u.H * u.H,
                      nblocks=ops.grid_blocks(rin, u.cin, fr))
            if u.sc_conv is not None:
                kw.update(add=g_xS, add_kind=2, H=u.H, W=u.H, add_stride=u.stride)
            elif u.stride == 1:
                kw.update(add=g_out, add_kind=1)
            else:
                kw.update(add=g_out, add_kind=2, H=u.H, W=u.H, add_stride=u.stride)
            # Round 6: the first unit's input gradient is read by exactly one kernel, the stem's backward sums, and both are
            # HBM-bound passes over 3.2 M rows x 64 channels: fr_stem_bwd_sums_from forms it on the way (bit-identical gx and
            # sums, 411 MB less traffic at batch 256; FRHIP_STEM_FROM_UNIT=0: the two launches)
            self._unit0_apply = None
            if (i == 0 and not self.body_only and self.stem_recompute and fr == FR_BF16 and u.cin == 64 and
                    self.M0 < (1 << 24) and self.bn0.mod.training and _switch("FRHIP_STEM_FROM_UNIT", 1)):
                # ... and x, the unit's input, is the stem's own output: recomputed from the rows instead of read (x = NULL)
                kw0 = dict(kw, x=None) if (x is self.z0 and _switch("FRHIP_STEM_RECOMPUTE_X", 1)) else kw
                self._unit0_apply = ops._fill(_lib.FrBnBwdArgs(), **kw0)
                self._unit0_apply_keep = kw0
            else:
                # Round 6: where the unit in front has no gate and the tensor only streams through HBM, this launch also takes
                # the sums BN2's backward of that unit needs (FrBnBwdArgs.nx) -- fr_bn_bwd_reduce, a pass over (g_x, y2) right
                # behind this one, is gone there.  FRHIP_BN_NEXT_MB: smallest tensor (MB) it is done for, 0 = never
                sums_left = 0
                lim = _switch("FRHIP_BN_NEXT_MB", 20)
                if (i > 0 and fr == FR_BF16 and lim > 0 and self.units[i - 1].se is None and
                        rin * u.cin * 2 >= lim << 20):
                    pb = self.ubuf[i - 1]
                    kw.update(nx=pb["y2"], nmean=pb["bn2"].mean, ninvstd=pb["bn2"].invstd, npart=self.part)
                    sums_left = kw["nblocks"]
                L.append(ops.bn_bwd_apply(st, fr, **kw))
            done = None
            if self.dual:
                done = torch.cuda.Event()
                L.append(_EvRecord(done, self.stream2_t))
                unit_done[i] = done
            self.ready_marks.append((len(L), ready, done))
            if u.Ho < _switch("FRHIP_DP_GATE_HO", 28):  # 14x14 / 7x7 layers: one workgroup per compute unit and launch (see comm_gate below)
                self._gate_end = len(L)
            g_out, cur = g_x, nxt
        # the slabs of the last deferring weight-gradient launch are summed by a launch of their own
        flushed = self._flush_pending(L)
        tail_ready = [flushed] if flushed is not None else []
        if flushed is not None and self.dual:
            ev = torch.cuda.Event()
            L.append(_EvRecord(ev, self.stream2_t))
            unit_done[-1] = ev  # the side stream is FIFO: this event is behind every unit's weight gradients
        if self.body_only:  # the gradient with respect to the stack's input is the result
            self.g_input = g_out
            if tail_ready:
                self.ready_marks.append((len(L), tail_ready, unit_done.get(-1)))
            self._order_ready_marks()
            if self.dual and unit_done:
                L.append(_EvWait(self.stream1_t, unit_done[min(unit_done)]))  # join: the side stream is FIFO
            self.bwd_list = L
            return
        # ---- stem: z0 = PReLU(BN0(y0)); y0 = X0 * W0p^T
        sc, sb, sp = self.stem
        nb = ops.grid_blocks(self.M0, 64, fr)
        db, dg = self._bn_grads(self.bn0)
        gsl = self.grad_of(sp.weight)
        common = dict(g=g_out, x=self.y0, mean=self.bn0.mean, invstd=self.bn0.invstd, scale=self.bn0.scale,
                      shift=self.bn0.shift, slope=sp.weight, rows=self.M0, C=64, rows_per_image=self.S * self.S,
                      nblocks=nb)
        if self.stem_recompute:  # the sums of fr_bn_bwd_reduce over (g, y0) with y0 recomputed from the rows
            nb = int(min(2048, (self.M0 + 63) // 64))
            if getattr(self, "_unit0_apply", None) is not None:
                L.append(ops.call("fr_stem_bwd_sums_from", self._unit0_apply, self.X0, self.W0p, self.bn0.mean,
                                  self.bn0.invstd, self.bn0.scale, self.bn0.shift, sp.weight, self.part, self.M0, self.K0,
                                  nb, st))
            else:
                L.append(ops.call("fr_stem_bwd_sums", self.X0, self.W0p, g_out, self.bn0.mean, self.bn0.invstd,
                                  self.bn0.scale, self.bn0.shift, sp.weight, self.part, self.M0, self.K0, nb, st))
            self._reduce(L, nb, 3, 64, db, dg, gsl if gsl is not None else self.sums[2, :64])
        else:
            L.append(ops.bn_bwd_reduce(st, fr, part=self.part, **common))
            self._reduce(L, nb, 3, 64, db, dg, gsl if gsl is not None else self.sums[2, :64])
        s0, s1 = self._s01(self.bn0, db, dg)
        gw0 = self.grad_of(sc.weight)
        fuse = self.use_stem_gemm  # BN0 / PReLU backward applied while the weight gradient stages its rows
        g_y0 = None
        if gw0 is not None and not fuse:  # the materialised stem-output gradient (fp32 path)
            if self.per_unit_sets:
                g_y0 = self._act(self.M0 * 64, 1).view(-1)
            else:
                g_y0 = self.g_y1s[1 if self.dual else 0][:self.M0 * 64]  # unit 0's wgrads (side stream) still read set 0
                if self.dual and 1 in unit_done:
                    L.append(_EvWait(self.stream1_t, unit_done[1]))  # set 1 was last read by unit 1's weight gradients
        if gw0 is not None and not fuse:
            L.append(ops.bn_bwd_apply(st, fr, gx=g_y0, gamma=sb.weight, s0=s0, s1=s1, inv_count=1.0 / self.M0,
                                      **common))
        if gw0 is not None:
            if self.use_stem_gemm and not fuse:
                nsl = int(min(1024 if self.K0 == 32 else 512, (self.M0 + 63) // 64))
                L.append(ops.call("fr_stem_wgrad", g_y0, self.X0, self.part, self.M0, self.K0, nsl, st))
                L.append(ops.call("fr_reduce_parts", self.part, nsl, 1, 64 * self.K0, self.gW0p, None, None, st))
            elif fuse:
                # BN0 backward is applied while the gradient rows are staged: g_y0 is never materialised
                nsl = int(min(1024 if self.K0 == 32 else 512, (self.M0 + 63) // 64))  # partials live in self.part
                if self.stem_recompute:
                    L.append(ops.call("fr_stem_wgrad_bn_r", g_out, self.X0, self.W0p, self.bn0.mean, self.bn0.invstd,
                                      self.bn0.scale, self.bn0.shift, sp.weight, sb.weight, s0, s1, 1.0 / self.M0,
                                      self.part, self.M0, self.K0, nsl, st))
                else:
                    L.append(ops.call("fr_stem_wgrad_bn", g_out, self.y0, self.X0, self.bn0.mean, self.bn0.invstd,
                                      self.bn0.scale, self.bn0.shift, sp.weight, sb.weight, s0, s1, 1.0 / self.M0,
                                      self.part, self.M0, self.K0, nsl, st))
                L.append(ops.call("fr_reduce_parts", self.part, nsl, 1, 64 * self.K0, self.gW0p, None, None, st))
            else:
                nsl = int(min(_wgrad_slices(self.M0, 1), self.part.numel() // (64 * self.K0)))
                if nsl == 1:  # a single slice adds onto the buffer: clear it first
                    L.append(ops.call("fr_fill_rows", self.gW0p, None, 64, self.K0, st))
                L.append(ops.wgrad(st, fr, g=g_y0, src=self.X0, dw=self.gW0p, B=self.M0, GH=1, GW=1, Cout=64, SH=1,
                                   SW=1, SC=self.K0, KH=1, KW=1, stride=1, pad=0, ldg=64, lda=self.K0, pro=0,
                                   nsplit=nsl, slab=self.part if nsl > 1 else None))
            L.append(ops.call("fr_unpack_stem_grad", self.gW0p, gw0, gw0.stride(0), gw0.stride(1), gw0.stride(2),
                              gw0.stride(3), 64, sc.weight.shape[1], self.K0, st))
        if self.dual and unit_done:
            L.append(_EvWait(self.stream1_t, unit_done[min(unit_done)]))  # join: the side stream is FIFO
        self.ready_marks.append((len(L), [sb.weight, sb.bias, sp.weight, sc.weight] + tail_ready, None))
        self._order_ready_marks()
        self.bwd_list = L

    def _order_ready_marks(self):
        """Announce gradients in ARENA order (data-parallel buckets are contiguous arena slices that complete front to
        back).  A deferring weight-gradient launch completes its predecessor's gradient one launch late, i.e. possibly in
        the next unit's mark: inside a mark the parameters are sorted by arena position, and a parameter is held back
        until everything in front of it in the arena has been announced."""
        pos = {id(p): k for k, (p, _o, _n) in enumerate(self.arena_slices)}
        total, nxt, pool, out = len(pos), 0, [], []
        for end, params, done in self.ready_marks:
            pool = sorted(pool + [p for p in params if id(p) in pos], key=lambda p: pos[id(p)])
            emit = []
            while pool and pos[id(pool[0])] == nxt:
                emit.append(pool.pop(0))
                nxt += 1
            out.append((end, emit, done))
        if pool or nxt != total:
            raise _lib.FrhipError("frhip: %d gradients were never announced (readiness bookkeeping)" % (total - nxt))
        self.ready_marks = out
        # Data parallelism (frhip.parallel): the number of gradients announced by the time the backward pass has left the
        # layers that launch ONE workgroup per compute unit (output grids below 28x28).  A collective kernel resident
        # beside those launches doubles each of them (tools/cu_hog.py); behind this point every launch has thousands of
        # workgroups and pays for held CUs in proportion.
        gate_end = getattr(self, "_gate_end", 0)
        self.comm_gate = sum(len([p for p in emit if p.requires_grad]) for end, emit, _d in out if end <= gate_end)

    # ---- execution ---------------------------------------------------------------------------------
    def check_current(self):
        if torch.cuda.current_stream().cuda_stream != self.stream_id:
            return False
        return self._signature() == self.param_sig

    def run_body_forward(self, x):
        """Bare stack of residual units: x [B, C, H, H] (any float dtype, NCHW) -> [B, depth, Ho, Ho] fp32 NCHW."""
        B, S, C0 = self.B, self.S, self.units[0].cin
        self.z0.view(B, S, S, C0).copy_(x.permute(0, 2, 3, 1))
        ops.run(self.pack_list)
        ops.run(self.fwd_list)
        self.generation += 1
        last = self.units[-1]
        # an OWNED tensor: on the fp32 path .float() would be a view of the plan's static buffer, which the next call of
        # the same unit overwrites
        return self.feat.view(B, last.Ho, last.Ho, last.depth).permute(0, 3, 1, 2).to(torch.float32, copy=True)

    def run_body_backward(self, g):
        """dL/d(output) [B, depth, Ho, Ho] -> dL/d(input) fp32 NCHW; parameter gradients land in the arena views."""
        B, last, first = self.B, self.units[-1], self.units[0]
        self.g_pp[0][:g.numel()].view(B, last.Ho, last.Ho, last.depth).copy_(g.permute(0, 2, 3, 1))
        if not getattr(self.arena, "_frhip_zeroed", False):
            self.arena.zero_()
        self.arena._frhip_zeroed = False
        for p, v in self._grad_pairs:
            if p.grad is not v and p.requires_grad:
                p.grad = v
        ops.run(self.bwd_list)
        return self.g_input.view(B, first.H, first.H, first.cin).permute(0, 3, 1, 2).to(torch.float32, copy=True)

    def run_forward(self, x, avg_image, seed):
        B, S = self.B, self.S
        st = self.stream
        avg = avg_image
        if self.pack_side:  # behind the previous step on the main stream (the optimizer's update, the last data gradient)
            self._pack_ev0.record(self.stream1_t)
            self.stream2_t.wait_event(self._pack_ev0)
            ops.run(self.pack_side_list)
            self._pack_ev1.record(self.stream2_t)
        ops.call("fr_stem_im2col", x, avg, self.X0, B, S, S, self.in_channels, self.avg_channels, self.K0, self.fr, st)()
        od = self.out[1]
        p = float(od.p) if od.training else 0.0
        self.l_drop_fwd.args[7] = p
        self.l_drop_fwd.args[8] = seed
        if not self.infer:
            self.l_drop_bwd.args[self._drop_bwd_idx[0]] = p
            self.l_drop_bwd.args[self._drop_bwd_idx[1]] = seed
        ops.run(self.pack_list)
        ops.run(self.fwd_list)
        self.generation += 1
        return self.feat

    def run_backward(self, g_feat, on_ready=None):
        self.g_feat_in.copy_(g_feat)
        # frhip.optim's zero_grad() has just cleared the arena with one fill and says so: do not fill 174 MB twice
        if not getattr(self.arena, "_frhip_zeroed", False):
            self.arena.zero_()
        self.arena._frhip_zeroed = False
        # (re)bind .grad to the arena views; an identity test per parameter -- this loop runs while the GPU drains the
        # short head / loss kernels and was a visible idle gap in the trace when it called data_ptr() 188 times
        for p, v in self._grad_pairs:
            if p.grad is not v and p.requires_grad:
                p.grad = v
        if on_ready is None:
            ops.run(self.bwd_list)
            return
        # Readiness callbacks (gradient all-reduce): host bookkeeping only.  A callee that enqueues a collective first calls
        # comm_fence(), which orders the communication stream behind the main stream up to this point and behind the side
        # stream's weight gradients of the units announced so far -- the main stream itself never waits for the side stream
        # here, so the two-stream overlap survives data-parallel runs.  (Round 5: the fence used to be set at every mark --
        # 26 event records between the backward kernels of the main stream cost 0.2-0.3 ms per step with nothing to
        # exchange; tools/cu_hog.py --comm, profiles/r05_comm_policy.txt.)
        if self.comm_stream_t is None:
            self.comm_stream_t = _side_stream(self.device, -1)  # one communication stream per device, too
            self.comm_events = [torch.cuda.Event() for _ in self.ready_marks]
        pos, last_done = 0, None
        for k, (end, params, done) in enumerate(self.ready_marks):
            ops.run(self.bwd_list[pos:end])
            pos = end
            if done is not None:
                last_done = done  # the side stream is FIFO: the latest event is behind every earlier unit's weight gradients
            self._fence_at = (k, last_done)
            on_ready([p for p in params if p.requires_grad])
        ops.run(self.bwd_list[pos:])
        self.stream1_t.wait_stream(self.comm_stream_t)  # whatever the callbacks enqueued themselves (not the async collectives)

    def comm_fence(self):
        """Called from inside a readiness callback: orders the communication stream behind every gradient announced so far
        and returns it (collectives are enqueued under ``torch.cuda.stream(plan.comm_fence())``)."""
        k, done = self._fence_at
        comm = self.comm_stream_t
        ev = self.comm_events[k]
        ev.record(self.stream1_t)
        comm.wait_event(ev)
        if done is not None:
            comm.wait_event(done)
        return comm


# ------------------------------------------------------------------------------------------------ autograd glue


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, x, *params):
        feats = runner._forward_impl(x)
        ctx.runner = runner
        ctx.plan = runner.plan
        ctx.generation = runner.plan.generation
        return feats.clone()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("frhip: the backbone ran another forward before this backward; the activation "
                               "buffers of the static plan hold one step at a time")
        plan.run_backward(g.contiguous().float(), ctx.runner.on_grads_ready)
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class BackboneRunner(object):
    """Owns the plan cache of one backbone module and exposes ``__call__(x) -> features [B,512]``."""

    def __init__(self, module, in_channels=3):
        self.module = module
        self.in_channels = in_channels
        self.plan = None
        self.plans = {}
        self.compute_dtype = None
        self.on_grads_ready = None
        self.single_stream = False  # True: weight gradients stay on the main stream (per-kernel profiling)
        self.step_seed = 0x5EED

    def _get_plan(self, x, avg_channels, infer=False):
        dtype = self.compute_dtype or getattr(self.module, "compute_dtype", None) or compute_dtype_default()
        key = (x.shape[0], dtype, x.device, avg_channels, self.single_stream, infer)
        plan = self.plans.get(key)
        if plan is None or not plan.check_current():
            plan = BackbonePlan(self.module, x.shape[0], dtype, x.device, self.in_channels - avg_channels,
                                avg_channels, single_stream=self.single_stream, infer=infer)
            self.plans = {key: plan}  # one live plan: activations of a 256-batch are several GB
        return plan

    def _forward_impl(self, x, infer=False):
        avg = self._avg
        self.plan = self._get_plan(x, 0 if avg is None else avg.shape[0], infer)
        self.step_seed = (self.step_seed * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return self.plan.run_forward(x, avg, self.step_seed)

    def _device_avg(self, avg_image, device):
        """The constant average image on the device.  pSp keeps it as a plain (host) tensor attribute
        (restyle_psp.py:381-389); copying it every step is a synchronous pageable H2D transfer that stops the host
        from running ahead of the GPU (30.7 instead of 18.9 ms per step), so the device copy is cached per
        (tensor, version)."""
        key = (id(avg_image), avg_image._version, str(device))
        if getattr(self, "_avg_key", None) != key:
            self._avg_dev = avg_image.detach().to(device).contiguous().float()
            self._avg_key, self._avg_src = key, avg_image  # keep the source alive: id() stays unique
        return self._avg_dev

    def __call__(self, x, avg_image=None):
        if not x.is_cuda:
            raise _lib.FrhipError("frhip: the backbone runs on the HIP path only -- got a %s tensor. Move the "
                                  "model and batch to a ROCm device (the CPU restatement is oracle/, for tests)."
                                  % x.device)
        if x.requires_grad:
            raise NotImplementedError("frhip: gradients with respect to the input images are not implemented")
        x = x.contiguous().float()
        S = self.module.input_size if isinstance(self.module.input_size, int) else self.module.input_size[0]
        if x.shape[2] != S or x.shape[3] != S:
            raise _lib.FrhipError("frhip: expected %dx%d inputs, got %s" % (S, S, tuple(x.shape)))
        if x.shape[0] == 0:  # the reference fails in Flatten (model_irse.py:146) in both modes
            raise RuntimeError("cannot reshape tensor of 0 elements into shape [0, -1] because the unspecified dimension "
                               "size -1 can be any value and is ambiguous")
        bn1d = self.module.output_layer[-1]
        if x.shape[0] == 1 and bn1d.training:  # BatchNorm1d(512) on one row (model_irse.py:148), as torch raises it
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             "torch.Size([1, %d])" % bn1d.num_features)
        self._avg = None if avg_image is None else self._device_avg(avg_image, x.device)
        have = x.shape[1] + (0 if self._avg is None else self._avg.shape[0])
        if have != self.in_channels:
            raise RuntimeError("frhip: the stem expects %d input channels, got %d from the batch%s" % (
                self.in_channels, x.shape[1], "" if self._avg is None else " + %d from avg_image" % self._avg.shape[0]))
        params = [p for p in self.module.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _BackboneFn.apply(self, x, *params)
        return self._forward_impl(x, infer=True).clone()  # forward-only plan (BatchNorm folded when in eval mode)


# ------------------------------------------------------------------------------------------------ bare unit stacks


class _UnitStack(object):
    """What the plan needs from a module tree, for a bare list of residual units (no stem, no output layer)."""
    input_layer = None

    def __init__(self, blocks, size):
        self.body, self.input_size = list(blocks), int(size)

    def modules(self):
        for b in self.body:
            for m in b.modules():
                yield m

    def parameters(self):
        for b in self.body:
            for p in b.parameters():
                yield p

    def buffers(self):
        for b in self.body:
            for t in b.buffers():
                yield t


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, x, *params):
        y = runner._forward_impl(x, infer=False)
        ctx.plan, ctx.generation = runner.plan, runner.plan.generation
        return y

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("frhip: the unit ran another forward before this backward; the activation buffers of the "
                               "static plan hold one step at a time")
        gx = plan.run_body_backward(g.contiguous().float())
        return (None, gx) + (None,) * (len(ctx.needs_input_grad) - 2)


class UnitStackRunner(object):
    """Runs ``bottleneck_IR`` / ``bottleneck_IR_SE`` modules (reference backbone/model_irse.py:49-91,
    restyle_psp_helpers.py:119-199) on their own: ``x [B, C, H, H] -> [B, depth, H/stride, H/stride]`` with gradients
    for the input and every parameter, through the same launch lists a whole backbone uses (statistics of the input
    for the first BatchNorm included).  This is what makes the reference's block classes callable outside a backbone,
    and what the block-level golden fixtures (g3 / g4) run through on the GPU."""

    def __init__(self, blocks):
        self.blocks = list(blocks)
        self.plan, self.plans, self.compute_dtype = None, {}, None
        self.single_stream = False

    def _forward_impl(self, x, infer):
        dtype = self.compute_dtype or getattr(self.blocks[0], "compute_dtype", None) or compute_dtype_default()
        key = (x.shape[0], x.shape[2], dtype, x.device, infer)
        plan = self.plans.get(key)
        if plan is None or not plan.check_current():
            plan = BackbonePlan(_UnitStack(self.blocks, x.shape[2]), x.shape[0], dtype, x.device, 0, 0,
                                single_stream=self.single_stream, infer=infer)
            self.plans = {key: plan}
        self.plan = plan
        return plan.run_body_forward(x)

    def __call__(self, x):
        if not x.is_cuda:
            raise _lib.FrhipError("frhip: residual units run on the HIP path only -- got a %s tensor (the CPU restatement "
                                  "is oracle/, for tests)" % x.device)
        cin = self.blocks[0].res_layer[0].num_features
        if x.dim() != 4 or x.shape[1] != cin or x.shape[2] != x.shape[3]:
            raise RuntimeError("frhip: expected a [B, %d, H, H] activation, got %s" % (cin, tuple(x.shape)))
        params = [p for b in self.blocks for p in b.parameters()]
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _StackFn.apply(self, x, *params)
        return self._forward_impl(x, infer=True)


def run_unit(block, x):
    """forward() of a residual-unit module called on its own."""
    r = block.__dict__.get("_frhip_runner")
    if r is None:
        r = block.__dict__["_frhip_runner"] = [UnitStackRunner([block])]
    return r[0](x)
