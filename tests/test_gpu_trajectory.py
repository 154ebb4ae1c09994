"""Training dynamics: a run of AdamW steps of the HIP engine against the CPU oracle (oracle/ref_cpu.py:
train_step = models/cvae.py:132-157) from the same seeded initialisation, on the same images, eps draws
and dropout masks.  One-step parity is covered elsewhere; this checks that the differences do not
compound: the loss curves of the f32 mode stay on the oracle's, the bf16 mode follows within a few
percent, and all three learn (the reconstruction loss leaves its initial value)."""
import math
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobs(P, y, B, g):
    """A blob whose position depends on the class, plus noise (the long-run tool's data, on the CPU)."""
    lin = torch.linspace(-1, 1, P)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    labels = torch.randint(0, y, (B,), generator=g)
    ang = labels.float() * (2 * math.pi / y)
    cx, cy = 0.5 * torch.cos(ang), 0.5 * torch.sin(ang)
    blob = torch.exp(-((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / 0.08)
    img = torch.stack([blob, blob.flip(1), blob.flip(2)], 1) * 1.6 - 0.9
    img = (img + 0.05 * torch.randn(img.shape, generator=g)).clamp(-1, 1).contiguous()
    return img, labels


def test_training_trajectory_tracks_oracle():
    from cvae_gan_amd.engine import Engine
    from oracle import ref_cpu as O
    P, y, z, B, ngf, steps, lr = 32, 10, 16, 32, 16, 150, 1e-3
    g = torch.Generator().manual_seed(0)
    e, d = O.init_params(P, y, z, seed=0, ngf=ngf)
    state = OrderedDict(e)
    state.update(d)
    st = O.TrainState(e, d)
    engines = {"f32": Engine(P, y, z, B, dtype=torch.float32, ngf=ngf, state=state),
               "bf16": Engine(P, y, z, B, dtype=torch.bfloat16, ngf=ngf, state=state)}
    pool = [_blobs(P, y, B, g) for _ in range(4)]
    curves = {"ref": [], "f32": [], "bf16": []}
    for it in range(steps):
        x, labels = pool[it % len(pool)]
        eps = torch.randn(B, z, generator=g)
        mask = (torch.rand(B, 4 * (z + y), generator=g) > 0.5).float()
        r = O.train_step(st, x, labels, eps, mask, y, float(B), lr)
        curves["ref"].append((float(r["KL"]), float(r["LL"])))
        for name, eng in engines.items():
            eng.set_mask(mask)
            eng.train_step(x.cuda(), labels.int().cuda(), eps.cuda(), float(B), lr)
            l = eng.read_losses()
            curves[name].append((l["KL"], l["LL"]))
    dev = {name: [abs(a[1] - b[1]) / b[1] for a, b in zip(curves[name], curves["ref"])] for name in ("f32", "bf16")}
    for it in (0, 10, 25, 50, 100, steps - 1):
        print("step %3d  LL ref %.5f  f32 %.5f  bf16 %.5f   KL ref %.5f  f32 %.5f  bf16 %.5f" % (
            it, curves["ref"][it][1], curves["f32"][it][1], curves["bf16"][it][1],
            curves["ref"][it][0], curves["f32"][it][0], curves["bf16"][it][0]))
    for name in ("f32", "bf16"):
        print(name, "steps off by more than 2e-3:", " ".join("%d:%.4f" % (i, v) for i, v in enumerate(dev[name]) if v > 2e-3))

    def level(name):                     # where the curve has settled: median of the last 30 steps
        return sorted(c[1] for c in curves[name][-30:])[15]

    def within(name, tol):
        return sum(v <= tol for v in dev[name]) / float(steps)

    # all three learn: the loss leaves its initial value by more than a factor of ten and settles at one level
    for name in curves:
        assert level(name) < 0.1 * curves[name][0][1], (name, curves[name][0], level(name))
    # The loss falls 2.86 -> 0.17 within ten steps and then sits near 0.1666 with isolated one-step bumps of
    # 1-4 % (AdamW at lr 1e-3: Adam's normalisation turns rounding-level gradient differences of weakly
    # determined parameters into +-lr steps, so two correct f32 implementations drift apart by bumps that
    # arrive a step earlier or later, while the level they settle at is the same).  Measured with two forms
    # of the BatchNorm statistics that agree to 1e-6 (sums of raw values / sums about the first row): the f32
    # engine stays within 2e-3 of the oracle for 60 steps / for 3 steps, 14 / 36 of 150 steps are off by more
    # than 2e-3, 9 / 17 by more than 2e-2, the largest 0.11 / 0.37 at a step where one curve has a bump; bf16:
    # 39 / 40 steps above 2e-3, 24 above 2e-2, the largest 0.50 / 0.22.  The settled level (median of the
    # last 30 steps) agreed to 5 digits in every run.
    assert max(dev["f32"][:3]) <= 2e-3, dev["f32"][:3]
    assert within("f32", 2e-2) >= 0.8 and max(dev["f32"]) <= 0.6, (within("f32", 2e-2), max(dev["f32"]))
    assert abs(level("f32") - level("ref")) <= 2e-3 * level("ref"), (level("f32"), level("ref"))
    assert within("bf16", 2e-2) >= 0.7 and max(dev["bf16"]) <= 1.0, (within("bf16", 2e-2), max(dev["bf16"]))
    assert abs(level("bf16") - level("ref")) <= 1e-2 * level("ref"), (level("bf16"), level("ref"))
