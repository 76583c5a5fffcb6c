"""Training-input transform.  CPU part: the oracle's restatement of Pillow's 8-bit bilinear resample
is pinned against the installed Pillow (the third-party library the reference's torchvision Resize calls), the product's
per-axis tables equal the oracle's, and the ToTensor/Normalize table equals torch's float32 arithmetic.  GPU part: the
HIP kernel against the oracle and against Pillow directly, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import input_ref as R

SIZES = [(112, 112, 128, 128), (250, 250, 128, 128), (128, 128, 128, 128), (97, 131, 128, 128), (300, 200, 146, 146),
         (64, 64, 128, 128), (113, 112, 128, 128), (1, 1, 128, 128), (512, 32, 128, 128), (32, 512, 128, 128),
         (1000, 64, 146, 146)]  # up to the 16:1 aspect ratio the product accepts (beyond ~100:1 Pillow reorders its passes)


def _img(seed, h, w, kind="random"):
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "white":
        return np.full((h, w, 3), 255, np.uint8)
    if kind == "black":
        return np.zeros((h, w, 3), np.uint8)
    ramp = (np.arange(h * w * 3) % 256).astype(np.uint8).reshape(h, w, 3)  # strong gradients
    return ramp


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
@pytest.mark.parametrize("kind", ["random", "white", "black", "ramp"])
def test_oracle_resize_is_pillow_bit_for_bit(h, w, oh, ow, kind):
    from PIL import Image
    a = _img(h * 1000 + w, h, w, kind)
    want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(R.resize_u8(a, oh, ow), want)


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_product_tables_equal_oracle_tables(h, w, oh, ow):
    from frhip.input_pipeline import resize_tables
    xt, yt = resize_tables(h, w, oh, ow)
    for tab, (bounds, coeffs) in ((xt, R.resize_tables(w, ow)), (yt, R.resize_tables(h, oh))):
        assert np.array_equal(tab[:, :2], bounds) and np.array_equal(tab[:, 2:], coeffs)
        assert (tab[:, 2:].sum(1) > 0).all()


def test_identity_axis_is_one_tap():
    from frhip.input_pipeline import resize_tables
    xt, _ = resize_tables(128, 128, 128, 128)
    assert (xt[:, 0] == np.arange(128)).all() or (xt[:, 1] >= 1).all()
    a = _img(5, 128, 128)
    assert np.array_equal(R.resize_u8(a, 128, 128), a)
    # applying the identity table (instead of skipping the pass, as Pillow does) changes nothing
    assert np.array_equal(R._pass(a, *R.resize_tables(128, 128), axis=1), a)


def test_normalize_lut_is_torch_float32_arithmetic():
    from frhip.input_pipeline import normalize_lut
    for mean, std in (((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)), ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))):
        lut = normalize_lut(mean, std)
        v = torch.arange(256, dtype=torch.uint8).view(256, 1).repeat(1, 3)
        t = v.to(torch.float32).div(255)                                     # ToTensor
        t = t.sub(torch.tensor(mean, dtype=torch.float32)).div(torch.tensor(std, dtype=torch.float32))  # Normalize
        assert np.array_equal(lut, t.numpy()) and np.array_equal(lut, R.normalize_lut(mean, std))
    assert lut.dtype == np.float32


def test_oracle_train_transform_composition():
    from PIL import Image
    a = _img(9, 112, 112)
    out = R.train_transform(a, 112, (5, 11), True)
    pil = np.asarray(Image.fromarray(a).resize((128, 128), Image.BILINEAR).crop((5, 11, 117, 123)))[:, ::-1]
    want = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
    assert np.array_equal(out, want.numpy())


def test_draws_cover_the_crop_range():
    from frhip.input_pipeline import GpuTrainTransform
    tf = GpuTrainTransform(112)
    crop, flip = tf.draw(4000, torch.Generator().manual_seed(1))
    assert crop.min() == 0 and crop.max() == 16 and 0.4 < flip.float().mean() < 0.6
    with pytest.raises(ValueError):
        tf(torch.zeros(2, 112, 112, 3))  # not uint8


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,size", [(112, 112, 112), (250, 250, 112), (128, 128, 112), (97, 131, 112), (64, 64, 56),
                                      (300, 200, 224)])
def test_gpu_transform_is_bit_exact(h, w, size):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    from frhip.input_pipeline import GpuTrainTransform
    B = 9
    imgs = np.stack([_img(100 + i, h, w, ("random", "ramp", "white")[i % 3]) for i in range(B)])
    tf = GpuTrainTransform(size)
    crop, flip = tf.draw(B, torch.Generator().manual_seed(3))
    span = tf.big - size
    crop[0], crop[1] = torch.tensor([0, 0]), torch.tensor([span, span])  # the corners
    flip[0], flip[1] = 0, 1
    out = tf(torch.from_numpy(imgs).cuda(), crop, flip).cpu().numpy()
    assert out.shape == (B, 3, size, size) and out.dtype == np.float32
    for i in range(B):
        x0, y0 = int(crop[i, 0]), int(crop[i, 1])
        want = R.train_transform(imgs[i], size, (x0, y0), bool(flip[i]))
        assert np.array_equal(out[i], want), (i, float(np.abs(out[i] - want).max()))
        pil = np.asarray(Image.fromarray(imgs[i]).resize((tf.big, tf.big), Image.BILINEAR).crop(
            (x0, y0, x0 + size, y0 + size)))
        if flip[i]:
            pil = pil[:, ::-1]
        direct = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
        assert np.array_equal(out[i], direct.numpy())


@pytest.mark.gpu
def test_gpu_transform_refuses_bad_input():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from frhip.input_pipeline import GpuTrainTransform
    tf = GpuTrainTransform(112)
    u8 = torch.zeros(2, 112, 112, 3, dtype=torch.uint8).cuda()
    with pytest.raises(ValueError):
        tf(u8, torch.tensor([[0, 17], [0, 0]], dtype=torch.int32), torch.zeros(2, dtype=torch.uint8))
    with pytest.raises(Exception):
        tf(u8.cpu())  # host tensors: no CPU fallback
    with pytest.raises(ValueError):
        tf(torch.zeros(1, 500, 3, 3, dtype=torch.uint8).cuda())  # aspect ratio beyond the verified range
    assert tf(u8[:0]).shape == (0, 3, 112, 112)


def test_stage_transform_hands_over_decoded_pixels(tmp_path):
    """FacesDataset + StageTransform: the worker's whole job is the decode; the batch collates to uint8 [B,H,W,3]."""
    from PIL import Image
    from dataset import FacesDataset, StageTransform
    from util.utils import collate_fn_ignore_none
    want = {}
    for ident in ("id_b", "id_a"):
        os.makedirs(tmp_path / ident)
        for k in range(3):
            a = _img(hash((ident, k)) % 1000, 112, 112)
            Image.fromarray(a).save(tmp_path / ident / ("%d.png" % k))
            want[(ident, k)] = a
    (tmp_path / "id_a" / "broken.png").write_bytes(b"not an image")
    ds = FacesDataset(str(tmp_path), StageTransform(), extensions=(".png",))  # lossless fixtures
    assert ds.classes == ["id_a", "id_b"] and len(ds) == 7
    items = [ds[i] for i in range(len(ds))]
    assert sum(it is None for it in items) == 1  # the broken file is skipped, as in the reference (dataset.py:77-81)
    good = [it for it in items if it is not None]
    assert all(x.dtype == torch.uint8 and tuple(x.shape) == (112, 112, 3) for x, _ in good)
    assert np.array_equal(good[0][0].numpy(), want[("id_a", 0)]) and good[0][1] == 0 and good[-1][1] == 1
    xb, yb = collate_fn_ignore_none(items)
    # the reference's collate refills the hole with the first survivor (util/utils.py:361-369)
    assert xb.dtype == torch.uint8 and tuple(xb.shape) == (7, 112, 112, 3) and yb.tolist() == [0, 0, 0, 1, 1, 1, 0]


@pytest.mark.gpu
def test_train_driver_with_gpu_input_pipeline(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylegan-for-facerec_amd")
    env = dict(os.environ, PYTHONPATH=root)
    argv = ["train.py", "--config", "configs/config_synthetic_smoke.py", "--synthetic", "12x10", "--max-steps", "3"]
    cfg_patch = ("import configs.config_synthetic_smoke as c; c.configurations[1].update(BATCH_SIZE=20, "
                 "GPU_INPUT_PIPELINE=True, MODEL_ROOT=r'%s', LOG_ROOT=r'%s')" % (tmp_path / "model", tmp_path / "log"))
    code = "import sys, runpy; sys.argv=%r; %s; runpy.run_path('train.py', run_name='__main__')" % (argv, cfg_patch)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Training Loss" in out.stdout and "nan" not in out.stdout.lower()


@pytest.mark.gpu
def test_eval_transforms_on_device_equal_the_host_pil_path():
    """perform_val's centre crop (Resize 128 -> CenterCrop 112) and flip on device tensors == the host PIL round trip of
    the reference (util/utils.py:204-236), bit for bit; also at the uint8 quantisation edges (-1, 1, k/255 steps)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from frhip import synth
    from util.utils import ccrop_batch, hflip_batch
    x = synth.uniform(77, "tta.x", (5, 3, 112, 112), -1.0, 1.0)
    x[0] = 1.0
    x[1] = -1.0
    x[2, :, :, :56] = (torch.arange(56) / 127.5 - 1.0)  # exact k/255 grid values
    for fn in (ccrop_batch, hflip_batch):
        host = fn(x)
        dev = fn(x.cuda())
        assert dev.is_cuda and dev.shape == host.shape
        assert torch.equal(dev.cpu(), host), (fn.__name__, float((dev.cpu() - host).abs().max()))
