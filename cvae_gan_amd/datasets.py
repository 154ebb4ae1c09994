"""Training data for the CVAE trainer.

BodyMapDataset follows the reference's datasets.py:21-46: the image list is
<data_root>/<dataset>/input_<dataset>.txt, images live in <data_root>/<dataset>/GT_output,
every image is resized to pix_dim x pix_dim, kept in the BGR channel order cv2.imread
delivers, scaled to [-1, 1] (ToTensor + Normalize(.5, .5), datasets.py:14-18), and the class is
int(name.split('_')[1]) - 2, returned as a one-hot vector of cls_num entries.
Images are decoded with PIL (cv2 is not a dependency of this package).

SyntheticBodyMaps stands in when the data directory is absent (there is no dataset in this
repository): class-dependent smooth blobs plus noise, same tensor shapes and value range."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def label_of(name):
    """datasets.py:40: the second '_'-separated field of the file name, minus 2."""
    return int(name.split("_")[1]) - 2


class BodyMapDataset(Dataset):
    def __init__(self, data_root, dataset, dim=512, max_size=-1, device=None, transform=None, cls_num=12):
        from PIL import Image  # noqa: F401  (fail here, not in a worker)
        self.dim, self.cls_num = dim, cls_num
        self.im_root = os.path.join(data_root, dataset)
        names = np.loadtxt(os.path.join(self.im_root, "input_%s.txt" % dataset), dtype=str, ndmin=1)
        self.names = names[:max_size]                    # the reference's slice: -1 drops the last entry
        self.im_names = [os.path.join(self.im_root, "GT_output", n) for n in self.names]

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        from PIL import Image
        im = Image.open(self.im_names[i]).convert("RGB").resize((self.dim, self.dim), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32)[:, :, ::-1]                  # RGB -> BGR, as cv2.imread
        img = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))) / 255.0
        img = (img - 0.5) / 0.5
        y = torch.zeros(self.cls_num)
        y[label_of(str(self.names[i]))] = 1
        return img, y


class SyntheticBodyMaps(Dataset):
    """Deterministic stand-in: image i has class i % cls_num; a class moves a Gaussian blob."""

    def __init__(self, n, dim, cls_num, seed=0):
        self.n, self.dim, self.cls_num, self.seed = n, dim, cls_num, seed
        g = torch.linspace(-1, 1, dim)
        self.yy, self.xx = torch.meshgrid(g, g, indexing="ij")

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        c = i % self.cls_num
        gen = torch.Generator().manual_seed(self.seed * 1000003 + i)
        ang = 2 * np.pi * c / self.cls_num
        cx, cy = 0.5 * np.cos(ang), 0.5 * np.sin(ang)
        blob = torch.exp(-((self.xx - cx) ** 2 + (self.yy - cy) ** 2) / 0.08)
        img = torch.stack([blob, blob.flip(0), blob.flip(1)]) * 1.6 - 0.9
        img = (img + 0.05 * torch.randn(3, self.dim, self.dim, generator=gen)).clamp(-1, 1)
        y = torch.zeros(self.cls_num)
        y[c] = 1
        return img, y


def make_dataset(data_dir, dataset, pix_dim, cls_num, max_size=-1, synthetic_size=2048):
    """The reference's dataset when its files exist, the synthetic stand-in otherwise."""
    listing = os.path.join(data_dir, dataset, "input_%s.txt" % dataset)
    if os.path.exists(listing):
        return BodyMapDataset(data_dir, dataset, dim=pix_dim, max_size=max_size, cls_num=cls_num), False
    n = synthetic_size if max_size is None or max_size < 0 else max_size
    return SyntheticBodyMaps(n, pix_dim, cls_num), True
