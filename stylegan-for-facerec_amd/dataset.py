"""``from dataset import FacesDataset`` (reference train.py:13, dataset.py:17-91) plus a synthetic stand-in.

``FacesDataset`` reads ``<root>/<identity>/*.jpg|png``; label = index of the identity in sorted order; a sample that
fails to load returns ``None`` (dropped by ``collate_fn_ignore_none``).  torchvision is not required: the default
train transform (resize 128 -> random crop 112 -> h-flip -> [-1,1] CHW float) is implemented with PIL+numpy on the host
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
        img = img.resize((self.big, self.big), Image.BILINEAR)
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


class FacesDataset(Dataset):
    def __init__(self, root, transform=None, extensions=(".jpg", ".jpeg", ".png")):
        self.root, self.transform = root, transform or TrainTransform()
        self.classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.samples = []
        for label, ident in enumerate(self.classes):
            for f in sorted(glob.glob(os.path.join(root, ident, "*"))):
                if f.lower().endswith(extensions):
                    self.samples.append((f, label))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        from PIL import Image
        path, label = self.samples[idx]
        try:
            return self.transform(Image.open(path).convert("RGB")), label
        except Exception as e:  # noqa: BLE001 -- broken files are skipped, as in the reference
            print("[FacesDataset] failed on", path, e)
            return None


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
