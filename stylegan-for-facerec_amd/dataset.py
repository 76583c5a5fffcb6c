"""``from dataset import FacesDataset`` (reference train.py:13, dataset.py:17-91) plus a synthetic stand-in.

``FacesDataset`` reads ``<root>/<identity>/*.jpg``; label = index of the identity (ethnicity prefix removed) in the
sorted identity set; a sample that fails to load returns ``None`` (dropped by ``collate_fn_ignore_none``).  torchvision
is not required: the train transform (resize 128 -> random crop 112 -> h-flip -> [-1,1] CHW float) is implemented with PIL+numpy on the host
worker, as in the reference.  With ``StageTransform`` the worker only decodes and hands over the uint8 HWC image; resize,
crop, flip and normalisation then run on the GPU for the whole batch (frhip/input_pipeline.py, SURVEY.md 8f rank 3;
``GPU_INPUT_PIPELINE=True`` in a train config).
"""
import glob
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset


class TrainTransform(object):
    """Resize(128*S/112) -> RandomCrop(S) -> RandomHorizontalFlip -> ToTensor -> Normalize(mean, std)."""

    def __init__(self, size=112, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        self.size, self.big = size, int(128 * size / 112)
        self.mean = np.asarray(mean, np.float32).reshape(3, 1, 1)
        self.std = np.asarray(std, np.float32).reshape(3, 1, 1)

    def __call__(self, img):
        from PIL import Image
        # grayscale / CMYK / RGBA files: same three channels as StageTransform (the GPU input path) sees, so both input
        # pipelines train on the same sample set (the reference's host transform drops such files in its Normalize)
        img = img.convert("RGB").resize((self.big, self.big), Image.BILINEAR)
        x0, y0 = random.randint(0, self.big - self.size), random.randint(0, self.big - self.size)
        arr = np.asarray(img.crop((x0, y0, x0 + self.size, y0 + self.size)), np.float32) / 255.0
        if random.random() < 0.5:
            arr = arr[:, ::-1]
        return torch.from_numpy((arr.transpose(2, 0, 1) - self.mean) / self.std)


class StageTransform(object):
    """Decode only: PIL image -> uint8 [H, W, 3] tensor for frhip.input_pipeline.GpuTrainTransform.  All images of a
    batch must share one size (the aligned training crops do: dataset.py:62 ``dims = (112, 112, 3)``)."""

    def __call__(self, img):
        return torch.from_numpy(np.array(img.convert("RGB"), dtype=np.uint8))


def identity_of(dirname):
    """Identity code of a sample directory: everything after the last '^' ("Caucasian^m49.r8743" -> "m49.r8743",
    reference dataset.py:47-48); names without '^' are identities as they stand."""
    return dirname[dirname.rfind("^") + 1:] if "^" in dirname else dirname


class FacesDataset(Dataset):
    """``<root>/<identity_code>/<filename.jpg>`` (reference dataset.py:17-91).

    As in the reference: the sample list is ``sorted(glob(root/*/*.jpg))`` (only .jpg; directories without one
    contribute nothing), an identity is its directory name with an ethnicity prefix up to the last '^' removed,
    ``classes = id_list = sorted(set(identities))`` (so "African^x" and "Asian^x" are ONE class and the label order is
    the order of the bare ids, not of the prefixed directory names), ``label = id2label[identity]``, ``transform=None``
    hands the PIL image through, and a sample that fails to load or transform is ``None`` (dropped by
    ``collate_fn_ignore_none``).  The constructor keeps the reference's signature; ``jpeg_loader`` /
    ``loss_weights_file`` / ``return_onehot`` are accepted and unused there too.  ``extensions`` is an opt-in
    extra (e.g. lossless .png fixtures); the default is the reference's '*.jpg'.  One deliberate difference: the
    reference strips the prefix in ``__getitem__`` only for the four RFW ethnicities and raises ``KeyError`` for any
    other 'x^id' directory (dataset.py:72-73 vs :47-48); here both places use the same rule."""

    class2race = {"African": 0, "Asian": 1, "Caucasian": 2, "Indian": 3}
    race2class = ["African", "Asian", "Caucasian", "Indian"]

    def __init__(self, root, transform=None, jpeg_loader=None, loss_weights_file=None, return_onehot=False,
                 id2race_file=None, extensions=(".jpg",)):
        super().__init__()
        self.root, self.transform = root, transform
        names = []
        for ext in extensions:
            names += glob.glob(os.path.join(root, "*", "*" + ext))
        self.filenames = sorted(set(names))
        print("Checking loaded data.")
        print("# filenames:", len(self.filenames))
        print("filenames[:5]", self.filenames[:5])
        self.id_list = sorted(set(identity_of(fn.split(os.sep)[-2]) for fn in self.filenames))
        print("self.id_list[:5]:", self.id_list[:5])
        self.id2race = None
        if id2race_file is not None:
            with open(id2race_file) as f:
                self.id2race = dict(line.split(" ")[:2] for line in f.read().splitlines())
        self.classes = self.id_list
        self.id2label = {identity: label for label, identity in enumerate(self.id_list)}
        self.n_identities = len(self.id_list)
        print("# identities:", self.n_identities)
        self.orig_n_samples = len(self.filenames)
        self.dims = (112, 112, 3)

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        from PIL import Image
        fn = self.filenames[idx]
        label = self.id2label[identity_of(fn.split(os.sep)[-2])]
        try:
            img = Image.open(fn)
            img.load()
        except Exception:  # noqa: BLE001 -- broken file, as in the reference (dataset.py:77-81)
            print("[Image Jpeg loading error]")
            return None
        try:
            if self.transform:
                img = self.transform(img)
        except Exception as e:  # noqa: BLE001
            print("[Error during transforming image] %s: %s" % (fn, e))
            return None
        return (img, label)


class SyntheticFaces(Dataset):
    """``identities`` x ``per_identity`` seeded 112x112 tensors in [-1, 1]; exposes ``.classes`` like FacesDataset."""

    def __init__(self, identities=100, per_identity=12, size=112, seed=900, staged=False):
        self.classes = ["id_%05d" % i for i in range(identities)]
        self.per, self.size, self.seed, self.staged = per_identity, size, seed, staged

    def __len__(self):
        return len(self.classes) * self.per

    def __getitem__(self, idx):
        from frhip import synth
        if self.staged:  # uint8 HWC, as StageTransform delivers decoded files
            # an identity is a coarse 8x8 colour pattern, a sample adds noise to it: unlike white noise this survives the
            # resize / random crop / flip of the input pipeline, so a network can actually learn the identities
            ident = idx // self.per
            cell = (self.size + 7) // 8
            base = synth.uniform(self.seed, "identity%d" % ident, (8, 8, 3), 32.0, 224.0)
            base = base.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:self.size, :self.size]
            img = base + synth.uniform(self.seed, "face%d" % idx, (self.size, self.size, 3), -24.0, 24.0)
            return img.clamp_(0, 255).to(torch.uint8), ident
        return synth.uniform(self.seed, "face%d" % idx, (3, self.size, self.size)), idx // self.per
