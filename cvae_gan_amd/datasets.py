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
    def __init__(self, data_root, dataset, dim=512, max_size=-1, device=None, transform=None, cls_num=12, raw=False):
        from PIL import Image  # noqa: F401  (fail here, not in a worker)
        self.dim, self.cls_num, self.raw = dim, cls_num, raw
        self.im_root = os.path.join(data_root, dataset)
        names = np.loadtxt(os.path.join(self.im_root, "input_%s.txt" % dataset), dtype=str, ndmin=1)
        self.names = names[:max_size]                    # the reference's slice: -1 drops the last entry
        self.im_names = [os.path.join(self.im_root, "GT_output", n) for n in self.names]

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        from PIL import Image
        im = Image.open(self.im_names[i]).convert("RGB").resize((self.dim, self.dim), Image.BILINEAR)
        if self.raw:                                                      # uint8 HWC BGR + class index: normalised on the device
            a = np.ascontiguousarray(np.asarray(im, dtype=np.uint8)[:, :, ::-1])
            return torch.from_numpy(a), label_of(str(self.names[i]))
        a = np.asarray(im, dtype=np.float32)[:, :, ::-1]                  # RGB -> BGR, as cv2.imread
        img = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))) / 255.0
        img = (img - 0.5) / 0.5
        y = torch.zeros(self.cls_num)
        y[label_of(str(self.names[i]))] = 1
        return img, y


class SyntheticBodyMaps(Dataset):
    """Deterministic stand-in: image i has class i % cls_num; a class moves a Gaussian blob."""

    def __init__(self, n, dim, cls_num, seed=0, raw=False):
        self.n, self.dim, self.cls_num, self.seed, self.raw = n, dim, cls_num, seed, raw
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
        if self.raw:
            u8 = ((img * 0.5 + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8)
            return u8.permute(1, 2, 0).contiguous(), c
        y = torch.zeros(self.cls_num)
        y[c] = 1
        return img, y


def mnist_to_input(gray, pix_dim=32):
    """MNIST-shaped batches for the reference's networks: gray [B,1,S,S] in [0, 1] -> [B,3,P,P] in [-1, 1].
    The networks take 3-channel images (models/networks.py:80) whose side is a multiple of 16
    (models/networks.py:179, :136), so 28 x 28 digits are zero-padded (black) to 32 x 32 around the centre,
    normalised as datasets.py:14-18 does ((v - .5) / .5) and replicated to three channels."""
    B, _, S, _ = gray.shape
    assert S <= pix_dim and (pix_dim - S) % 2 == 0
    lo = (pix_dim - S) // 2
    out = torch.zeros(B, 1, pix_dim, pix_dim, dtype=torch.float32)
    out[:, :, lo:lo + S, lo:lo + S] = gray.float()
    return ((out - 0.5) / 0.5).expand(B, 3, pix_dim, pix_dim).contiguous()


def make_dataset(data_dir, dataset, pix_dim, cls_num, max_size=-1, synthetic_size=2048, raw=False):
    """The reference's dataset when its files exist, the synthetic stand-in otherwise.
    raw=True: items are (uint8 HWC image, class index) for DevicePrefetcher."""
    listing = os.path.join(data_dir, dataset, "input_%s.txt" % dataset)
    if os.path.exists(listing):
        return BodyMapDataset(data_dir, dataset, dim=pix_dim, max_size=max_size, cls_num=cls_num, raw=raw), False
    n = synthetic_size if max_size is None or max_size < 0 else max_size
    return SyntheticBodyMaps(n, pix_dim, cls_num, raw=raw), True


class DevicePrefetcher:
    """Replaces the reference's DataLoaderX / BackgroundGenerator (models/cvae.py:18-21): batches
    of (uint8 HWC images, class indices) from a DataLoader go through pinned host buffers to the
    GPU on a copy stream, one batch ahead of the consumer, and are normalised there
    (cvg_image_u8_to_f32).  Yields (x f32 NCHW in [-1, 1], labels int32), both on the device and
    valid until the next-but-one batch is requested (two buffer sets)."""

    def __init__(self, loader, batch_size, pix_dim, device, y_dim=None):
        from . import ops
        self.loader, self.ops, self.B, self.P = loader, ops, batch_size, pix_dim
        self.y_dim = y_dim                                # class count, to reject labels the one-hot cannot hold
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets = []
        for _ in range(2):
            self.sets.append(dict(
                h_img=torch.empty(batch_size, pix_dim, pix_dim, 3, dtype=torch.uint8).pin_memory(),
                h_lab=torch.empty(batch_size, dtype=torch.int32).pin_memory(),
                d_img=torch.empty(batch_size, pix_dim, pix_dim, 3, dtype=torch.uint8, device=self.device),
                d_lab=torch.empty(batch_size, dtype=torch.int32, device=self.device),
                x=torch.empty(batch_size, 3, pix_dim, pix_dim, dtype=torch.float32, device=self.device),
                ready=torch.cuda.Event(), consumed=torch.cuda.Event()))
        for s in self.sets:
            s["consumed"].record()

    def __len__(self):
        return len(self.loader)

    def _stage(self, s, imgs, labels):
        s["consumed"].synchronize()                       # the consumer's kernels on this set have run
        s["h_img"].copy_(imgs)
        s["h_lab"].copy_(torch.as_tensor(labels).to(torch.int32))
        if self.y_dim is not None:
            lo, hi = int(s["h_lab"].min()), int(s["h_lab"].max())
            if lo < 0 or hi >= self.y_dim:                # the reference fails indexing its one-hot table (models/cvae.py:129)
                raise IndexError("label %d is out of bounds for y_dim %d" % (hi if hi >= self.y_dim else lo, self.y_dim))
        with torch.cuda.stream(self.copy_stream):
            s["d_img"].copy_(s["h_img"], non_blocking=True)
            s["d_lab"].copy_(s["h_lab"], non_blocking=True)
            self.ops.image_u8_to_f32(s["d_img"], s["x"], self.B, self.P)
            s["ready"].record()

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        try:
            imgs, labels = next(it)
        except StopIteration:
            return
        self._stage(self.sets[0], imgs, labels)
        while True:
            cur = self.sets[k % 2]
            nxt = None
            try:
                imgs, labels = next(it)
                nxt = self.sets[(k + 1) % 2]
                self._stage(nxt, imgs, labels)
            except StopIteration:
                pass
            torch.cuda.current_stream().wait_event(cur["ready"])
            yield cur["x"], cur["d_lab"]
            cur["consumed"].record()                      # on the consumer's stream, after its launches
            if nxt is None:
                return
            k += 1
