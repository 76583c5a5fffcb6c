"""CPU-only checks: the C-ABI library loads and exports every symbol include/frhip.h declares (no compute calls),
and the host-side mirror of the reference interface behaves like the reference (pinned by g8/g9 fixtures)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def structure(golden_dir):
    with open(os.path.join(golden_dir, "g8_structure.json")) as f:
        return json.load(f)


def test_abi_exports_every_declared_symbol():
    from frhip import _lib
    assert os.path.exists(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = sorted(_lib.protos)
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "include/frhip.h declares %s but libfrhip.so does not export it" % name
    assert _lib.self_check()
    assert _lib.lib.fr_last_error_string() is not None


def test_abi_rejects_bad_arguments_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on a GPU-less host."""
    from frhip import _lib
    a = _lib.FrConvArgs()
    a.SC, a.stride, a.B, a.RH, a.RW = 48, 1, 1, 1, 1  # 48 is not a multiple of 32
    rc = _lib.lib.fr_conv_igemm(ctypes.byref(a), 0, None)
    assert rc < 0 and b"multiple of 32" in _lib.lib.fr_last_error_string()
    a.SC, a.stride = 64, 3
    assert _lib.lib.fr_conv_igemm(ctypes.byref(a), 0, None) < 0
    with pytest.raises(_lib.FrhipError):
        _lib.check(-1, "x")


def test_run_time_switches_are_named_integers(monkeypatch):
    """fr_set_option / fr_get_option (ABI v5): a switch is first read from the environment variable of its name, cached, and
    overridable in-process; frhip.ops.sync_switches pushes the environment again (what a plan build does), so that no
    launcher of the library calls getenv."""
    from frhip import _lib, ops
    lib = _lib.lib
    assert _lib.lib.fr_abi_version() == 7
    monkeypatch.setenv("FRHIP_TEST_SWITCH_A", "7")
    assert lib.fr_get_option(b"FRHIP_TEST_SWITCH_A", 3) == 7          # from the environment
    assert lib.fr_get_option(b"FRHIP_TEST_SWITCH_B", 3) == 3          # unset: the first reader's default
    monkeypatch.setenv("FRHIP_TEST_SWITCH_B", "9")
    assert lib.fr_get_option(b"FRHIP_TEST_SWITCH_B", 3) == 3          # cached for the life of the process ...
    assert lib.fr_set_option(b"FRHIP_TEST_SWITCH_B", 11) == 3 and lib.fr_get_option(b"FRHIP_TEST_SWITCH_B", 0) == 11  # ... until set
    monkeypatch.setenv("FRHIP_ROLL64", "0")
    ops.sync_switches()
    assert lib.fr_get_option(b"FRHIP_ROLL64", 1) == 0
    monkeypatch.delenv("FRHIP_ROLL64")
    ops.sync_switches()
    assert lib.fr_get_option(b"FRHIP_ROLL64", 1) == 1
    # no launcher reads the environment: the only getenv of the library is the registry's
    csrc = os.path.join(os.path.dirname(_lib.__file__), "csrc")
    hits = [(f, l.strip()) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))
            for l in open(os.path.join(csrc, f)) if "getenv(" in l and not l.lstrip().startswith("//")]
    assert [f for f, _ in hits] == ["api.hip"], hits


@pytest.mark.parametrize("name", ["IR_50", "IR_SE_50", "IR_SE_101", "IR_101", "pSp", "pSp34"])
def test_state_dict_layout_and_param_split(structure, name):
    from backbone import model_irse as M
    from backbone.restyle_psp import pSp
    from util.utils import separate_irse_bn_paras
    ctor = {"IR_50": lambda: M.IR_50([112, 112]), "IR_SE_50": lambda: M.IR_SE_50([112, 112]),
            "IR_SE_101": lambda: M.IR_SE_101([112, 112]), "IR_101": lambda: M.IR_101([112, 112]),
            "pSp": lambda: pSp(size=112), "pSp34": lambda: pSp(size=112, encoder_type="BackboneEncoder34")}[name]
    m = ctor()
    want = structure[name]
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == want["keys"]
    assert [n for n, _ in m.named_parameters()] == want["param_names"]
    bn, wo = separate_irse_bn_paras(m)
    assert (len(bn), len(wo)) == (want["n_bn"], want["n_wo"])
    assert sum(p.numel() for p in bn) == want["bn_numel"] and sum(p.numel() for p in wo) == want["wo_numel"]


def test_heads_construct_like_the_reference(structure):
    from head import metrics as H
    from util.utils import separate_irse_bn_paras
    for name in ("ArcFace", "CosFace", "SphereFace", "Am_softmax"):
        h = getattr(H, name)(512, 100, None)
        assert [[k, list(v.shape)] for k, v in h.state_dict().items()] == structure[name]["keys"]
        bn, wo = separate_irse_bn_paras(h)
        assert (len(bn), len(wo)) == (structure[name]["n_bn"], structure[name]["n_wo"])
    a = H.ArcFace(512, 10, None)
    assert abs(a.th - (-0.8775825618903726)) < 1e-15 and abs(a.mm - 0.23971276930210156) < 1e-15
    assert H.CosFace(512, 10, None).m == 0.50  # the reference's default, not the paper's 0.35


def test_psp_dropout_insertion_and_errors(structure):
    from backbone.restyle_psp import pSp
    m = pSp(size=112, include_dropout=0.15)
    assert [type(c).__name__ for c in m.encoder.body[0].res_layer] == structure["pSp.dropout.body0.res_layer"]
    assert [type(c).__name__ for c in m.encoder.body[3].shortcut_layer] == structure["pSp.dropout.body3.shortcut_layer"]
    with pytest.raises(Exception) as e:
        pSp(size=112, encoder_type="nope")
    assert str(e.value) == structure["pSp.bad_encoder_error"]


def test_lr_helpers(structure):
    from util.utils import AverageMeter, schedule_lr, warm_up_lr
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.03)
    warm_up_lr(3, 10, 0.03, opt)
    assert opt.param_groups[0]["lr"] == structure["warm_up_lr_3_10_0.03"]
    opt.param_groups[0]["lr"] = 0.03
    schedule_lr(opt)
    assert opt.param_groups[0]["lr"] == structure["schedule_lr_0.03"]
    am = AverageMeter()
    am.update(2.0, 3)
    am.update(4.0, 1)
    assert am.avg == 2.5 and am.val == 4.0


def test_stage2_checkpoint_import(golden_dir, tmp_path):
    """g9: only encoder.input_layer.* / encoder.body.* are read from a Stage-2 checkpoint; output head untouched."""
    from backbone.restyle_psp import pSp
    from frhip import synth
    with open(os.path.join(golden_dir, "g9_stage2.json")) as f:
        want = json.load(f)
    src = pSp(size=112)
    sd = {k: v.clone() for k, v in src.state_dict().items()
          if k.startswith("encoder.input_layer") or k.startswith("encoder.body")}
    synth.fill_state_dict(sd, 19)
    ck = dict(sd)
    ck["encoder.styles.0.convs.0.weight"] = torch.ones(2, 2)
    ck["decoder.style.1.weight"] = torch.ones(3)
    path = str(tmp_path / "stage2.pt")
    torch.save({"state_dict": ck, "latent_avg": torch.zeros(18, 512), "opts": {"x": 1}}, path)
    m = pSp(size=112, checkpoint_path=path)
    got = m.state_dict()
    assert len(sd) == want["n_ckpt_encoder_keys"]
    assert sum(torch.equal(got[k], sd[k]) for k in sd) == want["n_loaded_equal"]
    assert [k for k in got if k.startswith("encoder.output_layer")] == want["output_layer_keys"]
    for k in ("encoder.input_layer.0.weight", "encoder.body.23.res_layer.3.weight"):
        assert abs(float(got[k].double().sum()) - want["sum." + k]) < 1e-6


def test_conv_weights_are_stored_packed():
    """3x3 weights live as [Cout][kh][kw][Cin] behind the OIHW Parameter; state-dict round trips keep values."""
    from backbone.model_irse import IR_50
    m = IR_50([112, 112])
    w = m.body[0].res_layer[1].weight
    assert w.shape == (64, 64, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.clone().contiguous() for k, v in m.state_dict().items()}
    m2 = IR_50([112, 112])
    m2.load_state_dict(sd)
    assert torch.equal(m2.body[0].res_layer[1].weight, w)
    assert m2.body[0].res_layer[1].weight.permute(0, 2, 3, 1).is_contiguous()


def test_synth_is_deterministic():
    from frhip import synth
    a = synth.uniform(1, "x", (5,))
    b = synth.uniform(1, "x", (5,))
    assert torch.equal(a, b) and not torch.equal(a, synth.uniform(2, "x", (5,)))
    assert abs(float(synth.normal(1, "n", (20000,)).std()) - 1.0) < 0.03
    lab = synth.labels(1, "l", 1000, 7)
    assert int(lab.min()) >= 0 and int(lab.max()) == 6


# ------------------------------------------------------------------------------------------------ evaluation (8f rank 2)


def _verification_inputs(n_pairs, dim, seed, tag):
    from frhip import synth
    base = synth.normal(seed, tag + ".a", (n_pairs, dim))
    noise = synth.normal(seed, tag + ".n", (n_pairs, dim))
    other = synth.normal(seed, tag + ".o", (n_pairs, dim))
    same = (synth.uniform(seed, tag + ".s", (n_pairs,), 0.0, 1.0) < 0.5)
    second = torch.where(same.view(-1, 1), base + 0.9 * noise, other)
    emb = torch.stack([base, second], 1).reshape(2 * n_pairs, dim)
    emb = emb / emb.norm(dim=1, keepdim=True)
    return emb.double().numpy(), same.numpy()


def test_verification_matches_reference_golden(golden_dir):
    """util/verification.evaluate == the reference's (tests/golden/g10_verification.npz): ROC curves, per-fold
    accuracy and best thresholds, incl. pair counts that do not divide into the folds."""
    from util import verification as V
    g = np.load(os.path.join(golden_dir, "g10_verification.npz"))
    for tag, n_pairs, folds in (("a", 600, 10), ("b", 203, 10), ("c", 57, 5)):
        emb, same = _verification_inputs(n_pairs, 32, 77, "ver." + tag)
        tpr, fpr, acc, best = V.evaluate(emb, same, nrof_folds=folds)
        np.testing.assert_array_equal(tpr, g[tag + "_tpr"])
        np.testing.assert_array_equal(fpr, g[tag + "_fpr"])
        np.testing.assert_array_equal(acc, g[tag + "_acc"])
        np.testing.assert_array_equal(best, g[tag + "_best"])
    # single-threshold helper agrees with the curves
    emb, same = _verification_inputs(57, 32, 77, "ver.c")
    d = ((emb[0::2] - emb[1::2]) ** 2).sum(1)
    tpr1, fpr1, acc1 = V.calculate_accuracy(1.0, d, same)
    assert abs(acc1 - ((d < 1.0) == same).mean()) < 1e-12 and 0.0 <= fpr1 <= tpr1 <= 1.0
    with pytest.raises(NotImplementedError):
        V.evaluate(emb, same, nrof_folds=5, pca=8)


def test_tta_transforms_follow_the_uint8_round_trip():
    """hflip_batch / ccrop_batch reproduce the reference's ToPILImage -> (flip | resize+crop) -> ToTensor -> Normalize
    pipeline (util/utils.py:204-236): values land on the 1/255 grid, a double flip stays within one grid step of the
    quantised image, the crop is the centre of the 128x128 bilinear up-sampling."""
    from frhip import synth
    from util.utils import ccrop_batch, hflip_batch
    x = synth.uniform(5, "tta.x", (3, 3, 112, 112), -1.0, 1.0)
    u8 = (x * 0.5 + 0.5).mul(255).to(torch.uint8)          # what ToPILImage stores (truncation)
    quant = (u8.float() / 255 - 0.5) / 0.5                   # what ToTensor + Normalize give back
    f = hflip_batch(x)
    assert torch.equal(f, torch.flip(quant, dims=[-1]))
    # (the truncating round trip is NOT idempotent -- v/255 can come back as v-1 -- which the reference shares)
    assert float((hflip_batch(f) - quant).abs().max()) <= 2.0 / 255 + 1e-6
    c = ccrop_batch(x)
    assert c.shape == (3, 3, 112, 112)
    grid = (c * 0.5 + 0.5) * 255
    assert float((grid - grid.round()).abs().max()) < 1e-3
    # a constant image stays constant through resize + crop
    k = torch.full((1, 3, 112, 112), 0.25)
    ck = ccrop_batch(k)
    assert float((ck - ck[0, 0, 0, 0]).abs().max()) == 0.0 and abs(float(ck[0, 0, 0, 0]) - 0.25) < 1.0 / 255


def test_initial_weight_distributions_follow_the_reference():
    """``Backbone`` draws xavier-uniform Conv2d / Linear weights (bound sqrt(6 / (fan_in + fan_out)),
    model_irse.py:174-189); ``pSp`` re-initialises its encoder with kaiming-normal(fan_out) (std sqrt(2 / (Cout * kh * kw)),
    Linear std sqrt(2 / 512), util/utils.py:24-44, incl. the SE 1x1 convolutions); biases 0, BatchNorm (1, 0)."""
    import math
    import torch.nn as nn
    from backbone.model_irse import IR_50
    from backbone.restyle_psp import pSp
    torch.manual_seed(3)
    m = IR_50([112, 112])
    for name, mod in m.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            w = mod.weight.detach()
            rf = w[0][0].numel() if w.dim() > 2 else 1
            bound = math.sqrt(6.0 / (w.shape[1] * rf + w.shape[0] * rf))
            assert float(w.abs().max()) <= bound + 1e-7, name
            if w.numel() >= 4096:
                assert abs(float(w.std()) / (bound / math.sqrt(3.0)) - 1.0) < 0.05, name  # uniform: std = bound / sqrt(3)
                assert float(w.abs().max()) > 0.98 * bound, name
            if mod.bias is not None:
                assert float(mod.bias.detach().abs().max()) == 0.0
        elif isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), name
    p = pSp(size=112, encoder_type="BackboneEncoder", avg_image=None)
    seen_se = False
    for name, mod in p.encoder.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            w = mod.weight.detach()
            rf = w[0][0].numel() if w.dim() > 2 else 1
            std = math.sqrt(2.0 / (w.shape[0] * rf))  # fan_out
            if w.numel() >= 4096:
                assert abs(float(w.std()) / std - 1.0) < 0.05, (name, float(w.std()), std)
                assert abs(float(w.mean())) < 0.05 * std, name
                assert float(w.abs().max()) > 3.0 * std, name  # a normal draw has tails a uniform one has not
            seen_se = seen_se or (w.dim() == 4 and rf == 1 and "fc" in name)
            if mod.bias is not None:
                assert float(mod.bias.detach().abs().max()) == 0.0
        elif isinstance(mod, nn.BatchNorm2d):
            assert bool((mod.weight == 1).all()) and bool((mod.bias == 0).all()), name
    assert seen_se  # the SE squeeze / excite 1x1 convolutions are re-initialised too


def test_get_val_data_layout_and_buffer_val(tmp_path):
    """``get_val_data`` returns the reference's 16-tuple (util/utils.py:89-114: seven benchmark sets, their issame lists,
    then the RFW dicts) with the RFW subsets read from the .npy fallback; ``buffer_val`` logs the reference's keys
    (:310-321)."""
    from util.utils import buffer_val, get_val_data
    out = get_val_data(str(tmp_path))
    assert len(out) == 16 and all(v is None for v in out)
    pairs = np.zeros((6, 3, 112, 112), np.float32)
    np.save(tmp_path / "RFW_Asian.npy", pairs)
    np.save(tmp_path / "RFW_Asian_list.npy", np.array([True, False, True]))
    np.save(tmp_path / "RFW_Indian.npy", pairs)  # no list file: ignored
    out = get_val_data(str(tmp_path))
    assert all(v is None for v in out[:14])
    rfw, rfw_issame = out[14], out[15]
    assert list(rfw) == ["Asian"] and rfw["Asian"].shape == (6, 3, 112, 112) and rfw_issame["Asian"].tolist() == [True, False, True]

    class Log(object):
        def __init__(self):
            self.rows = []

        def log(self, stats):
            self.rows.append(stats)

    w = Log()
    buffer_val(w, "RFW_Asian", 0.9, 1.25, None, 3)
    buffer_val(w, "RFW_Asian", 0.91, 1.2, None, 4, n_samples_passed=1000)
    assert w.rows[0] == {"RFW_Asian_Accuracy": 0.9, "RFW_Asian_Best_Threshold": 1.25, "epoch": 3}
    assert w.rows[1]["step"] == 1000 and w.rows[1]["epoch"] == 4


def test_faces_dataset_matches_reference_golden(golden_dir, tmp_path):
    """dataset.FacesDataset on a tiny tree with ``Race^id`` directory names (reference dataset.py:38-58, :68-91): file
    order, classes = sorted set of the BARE ids (the same id under two ethnicity prefixes is ONE class), id2label and
    the label of every sample equal what the reference produced (tests/golden/g11_dataset.json, make_golden.py g11)."""
    import numpy as np
    from PIL import Image
    from dataset import FacesDataset
    g = json.load(open(os.path.join(golden_dir, "g11_dataset.json")))
    for d, files in g["tree"].items():
        os.makedirs(tmp_path / d, exist_ok=True)
        for k, f in enumerate(files):
            if f.endswith(".txt"):
                (tmp_path / d / f).write_text("x")
            else:
                Image.fromarray(np.full((8, 8, 3), 10 * k + len(d), np.uint8)).save(tmp_path / d / f)
    ds = FacesDataset(str(tmp_path))
    assert [os.path.relpath(f, str(tmp_path)) for f in ds.filenames] == g["filenames"]
    assert ds.classes == g["classes"] and ds.id_list == g["classes"] and ds.id2label == g["id2label"]
    assert len(ds) == g["len"] and ds.n_identities == g["n_identities"] and ds.orig_n_samples == g["orig_n_samples"]
    assert list(ds.dims) == g["dims"]
    items = [ds[i] for i in range(len(ds))]
    assert [it[1] for it in items] == g["labels"]
    assert type(items[0][0]).__module__.split(".")[0] == g["item0_type"] == "PIL"  # transform=None: the PIL image
    assert FacesDataset.class2race["Caucasian"] == 2 and FacesDataset.race2class[3] == "Indian"


def test_resnet_structure_matches_reference_golden(golden_dir):
    """backbone.model_resnet (plain PyTorch, off the accelerated path; reference model_resnet.py:91-188): ordered
    state-dict keys, shapes, parameter count, output shape and the zero-initialised last BN of every residual branch."""
    from backbone.model_resnet import ResNet_50, ResNet_101, ResNet_152
    g = json.load(open(os.path.join(golden_dir, "g12_resnet_structure.json")))
    for name, ctor in (("ResNet_50", ResNet_50), ("ResNet_101", ResNet_101), ("ResNet_152", ResNet_152)):
        m = ctor([g[name]["input"]] * 2)
        sd = m.state_dict()
        assert list(sd.keys()) == g[name]["keys"]
        assert [list(v.shape) for v in sd.values()] == g[name]["shapes"]
        assert sum(p.numel() for p in m.parameters()) == g[name]["n_params"]
    m = ResNet_50([112, 112]).eval()
    with torch.no_grad():
        assert list(m(torch.zeros(2, 3, 112, 112)).shape) == g["ResNet_50"]["out_shape"]
    assert (float(m.layer1[0].bn3.weight.abs().sum()) == 0.0) == g["ResNet_50"]["bn3_weight_zero"]
    with pytest.raises(AssertionError):
        ResNet_50([96, 96])


def test_sgd_state_dict_round_trips_with_torch_sgd():
    """An Optimizer_*.pth written by frhip.optim.SGD must load into the reference's torch.optim.SGD AND step there
    (train.py:196, :227-230): param_groups carry dampening / nesterov / maximize like torch's own."""
    from frhip.optim import SGD
    p = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    mine = SGD([{"params": [p[0]], "weight_decay": 2e-3}, {"params": [p[1]]}], lr=0.03, momentum=0.9)
    ref_defaults = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.03, momentum=0.9).param_groups[0]
    for k in ("dampening", "nesterov", "maximize", "weight_decay", "momentum", "lr"):
        assert k in mine.param_groups[0] and k in mine.param_groups[1], k
    assert mine.param_groups[1]["dampening"] == ref_defaults["dampening"] == 0
    assert mine.param_groups[1]["nesterov"] is False and mine.param_groups[0]["weight_decay"] == 2e-3
    q = [torch.nn.Parameter(t.detach().clone()) for t in p]
    theirs = torch.optim.SGD([{"params": [q[0]], "weight_decay": 0.5}, {"params": [q[1]]}], lr=1.0, momentum=0.0)
    theirs.load_state_dict(mine.state_dict())
    for t in q:
        t.grad = torch.ones_like(t)
    theirs.step()  # KeyError('dampening') before the defaults were completed
    assert theirs.param_groups[0]["lr"] == 0.03 and theirs.param_groups[0]["weight_decay"] == 2e-3
    torch.testing.assert_close(q[1].detach(), p[1].detach() - 0.03)
    with pytest.raises(NotImplementedError):
        SGD(p, lr=0.1, momentum=0.9, nesterov=True)
    with pytest.raises(NotImplementedError):
        SGD(p, lr=0.1, momentum=0.9, dampening=0.1)


def test_bench_gpus_flag_starts_that_many_ranks():
    """``python bench.py --gpus 2`` with no launcher around it spawns two ranks (child torch.distributed.run, before torch
    is imported in the parent).  Without a GPU each rank stops at the product path's "needs a ROCm GPU" assertion -- which
    is the observable proof here that two rank processes were started with WORLD_SIZE=2 and that the exit code propagates.
    The GPU half (n_gpus == 2 in the JSON line) is tests/test_gpu_model.py::test_bench_spawns_its_own_ranks."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check; the GPU box runs test_bench_spawns_its_own_ranks")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=repo, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a ROCm GPU") >= 2, out.stderr[-3000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_readiness_marks_follow_the_arena_order():
    """engine.BackbonePlan._order_ready_marks (round 3): a deferring weight-gradient launch completes its predecessor's
    gradient one launch late, possibly in the NEXT unit's mark.  Announcements must still walk the gradient arena front to
    back (data-parallel buckets are contiguous arena slices): inside a mark parameters are sorted by arena position and a
    parameter is held back until everything in front of it has been announced; a gradient that is never announced is an
    error, not a silent hang of its bucket."""
    from types import SimpleNamespace
    from frhip.engine import BackbonePlan
    from frhip._lib import FrhipError
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(8)]
    fake = SimpleNamespace(arena_slices=[(p, 64 * k, 1) for k, p in enumerate(ps)])
    # unit A announces 0, 1, 3 (its conv1 gradient, slot 2, is still a pending slab sum); unit B's first launch completes
    # slot 2 and B announces 4, 5, 2 in launch order; the tail announces 7, 6
    fake.ready_marks = [(10, [ps[0], ps[1], ps[3]], "evA"), (20, [ps[4], ps[5], ps[2]], "evB"), (30, [ps[7], ps[6]], None)]
    BackbonePlan._order_ready_marks(fake)
    where = {id(p): k for k, p in enumerate(ps)}
    got = [[where[id(p)] for p in params] for _end, params, _ev in fake.ready_marks]
    assert got == [[0, 1], [2, 3, 4, 5], [6, 7]], got
    assert [m[0] for m in fake.ready_marks] == [10, 20, 30] and [m[2] for m in fake.ready_marks] == ["evA", "evB", None]
    fake.ready_marks = [(10, [ps[0], ps[1]], None), (20, [ps[3]], None)]
    with pytest.raises(FrhipError):
        BackbonePlan._order_ready_marks(fake)


def test_dispatch_geometry_of_the_strip_tables():
    """The capability / geometry queries of the C ABI are host arithmetic (no GPU call): the number of partial-sum rows a
    strip launch writes IS the dispatch decision the engine sizes its buffers from.  Pins the round-3 tables -- 7x7: four
    images per workgroup above the small-batch threshold, two below, one for odd batches; stride 2: whole images at 256
    channels forward, image pairs at 512; the shapes no strip kernel serves answer 0 (the generic kernel runs them)."""
    from frhip import _lib
    L = _lib.lib
    STORE, STATS, BNBWD = _lib.EPI_STORE, _lib.EPI_STATS, _lib.EPI_BNBWD
    sp = L.fr_conv3x3_strip_parts
    assert sp(256, 256, 256, 14, STATS) == 256                      # one workgroup per image
    assert sp(128, 256, 256, 14, STATS) == 128                      # small batch: channels split, still one ROW per image
    assert sp(256, 512, 512, 7, STATS) == 64                        # four images per workgroup
    assert sp(128, 512, 512, 7, STATS) == 64 and sp(6, 512, 512, 7, STATS) == 3   # small batch: two images
    assert sp(3, 512, 512, 7, STATS) == 3                           # odd batch: one image
    assert sp(256, 128, 128, 28, STATS) == 1024                     # 7-row strips
    assert sp(256, 512, 256, 14, BNBWD) == 256                      # stage-entry data gradient: whole images
    assert sp(256, 64, 128, 56, STORE) == 256 * 14                  # 4-row strips
    assert sp(256, 256, 512, 14, STATS) == 0 and sp(256, 256, 512, 14, STORE) > 0   # two-pass instance: plain store only
    assert sp(256, 96, 96, 14, STORE) == 0 and sp(256, 256, 256, 20, STORE) == 0    # not in the table
    s2 = L.fr_conv3x3_s2_strip_parts
    assert s2(256, 128, 128, 28, 0) == 1024 and s2(256, 128, 128, 28, 2) == 4096    # mode 2: four parity classes
    assert s2(256, 256, 256, 14, 0) == 256 and s2(256, 256, 256, 14, 2) == 4 * 512  # forward whole images, gradient 7-row strips
    assert s2(256, 512, 512, 7, 0) == 64 and s2(256, 512, 512, 7, 2) == 4 * 128     # forward (round 6, warp-specialised): four
    assert s2(6, 512, 512, 7, 0) == 3                                               # images per workgroup; else image pairs
    assert s2(3, 512, 512, 7, 0) == 3                                               # odd batch: one image
    assert s2(256, 128, 256, 28, 0) == 0                                            # Cin != Cout: generic kernel
    ws = L.fr_conv_wgrad_strip_supported
    assert all(ws(c, c, w) == 1 for c, w in ((64, 112), (64, 56), (128, 28), (256, 14), (512, 7)))
    assert ws(96, 64, 56) == 0 and ws(64, 64, 20) == 0


def test_configs_select_the_compute_dtype():
    """COMPUTE_DTYPE is the config key that selects the numerics of the backbone (reference configs: train.py:41-90 have
    none, so train.py reads it with cfg.get): the two BUPT configs run the bf16 path bench.py times, the synthetic smoke
    config the fp32 parity path; frhip.set_compute_dtype puts it where the runner looks (pSp: on ``.encoder``)."""
    import importlib
    from frhip import set_compute_dtype
    for name, want in (("config_BUPT_IR_50_baseline", "bf16"), ("config_BUPT_IR_50_AfrAsian", "bf16"),
                       ("config_synthetic_smoke", "fp32")):
        cfg = importlib.import_module("configs." + name).configurations[1]
        assert cfg.get("COMPUTE_DTYPE") == want, name
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylegan-for-facerec_amd",
                            "train.py")).read()
    assert 'cfg.get("COMPUTE_DTYPE")' in src  # optional key: a reference config without it still loads

    class Trunk(object):
        pass

    class Psp(object):
        def __init__(self):
            self.encoder = Trunk()

    class Wrapped(object):  # nn.DataParallel-style wrapper
        def __init__(self, m):
            self.module = m

    t, p = Trunk(), Psp()
    assert set_compute_dtype(t, "bf16") is torch.bfloat16 and t.compute_dtype is torch.bfloat16
    assert set_compute_dtype(p, "FP32") is torch.float32 and p.encoder.compute_dtype is torch.float32
    assert not hasattr(p, "compute_dtype")
    assert set_compute_dtype(Wrapped(p), torch.bfloat16) is torch.bfloat16 and p.encoder.compute_dtype is torch.bfloat16
    assert set_compute_dtype(t, None) is None and t.compute_dtype is torch.bfloat16  # absent key: nothing changes
    with pytest.raises(ValueError):
        set_compute_dtype(t, "fp16")
    with pytest.raises(ValueError):
        set_compute_dtype(t, torch.float16)
