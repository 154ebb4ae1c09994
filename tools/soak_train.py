#!/usr/bin/env python3
"""Long-run stability check of the bf16 train step: a few thousand AdamW steps at the benchmark shape on
class-dependent synthetic images held on the device; prints the losses as they go and fails on a
non-finite value.    python tools/soak_train.py [steps=3000] [lr=3e-5] [bf16|f32] [seed=0]

At the reference's learning rate (3e-5, train.sh) the run first settles on the mean image (LL 0.0215 on
this data: the decoder ignores z and the label) and leaves that plateau after a number of steps that depends
on the initial draw: seed 3 leaves it between steps 1000 and 2000 and reaches LL 0.0016 by step 6000; seeds
0, 1 and 2 are still on it after 20000 / 6000 / 6000 steps, in f32 mode as in bf16.  The CPU restatement
of the reference's step shows the same plateau (tests/test_gpu_trajectory.py)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd.engine import Engine  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-5
    dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.bfloat16
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    P, y, z, B = 64, 100, 32, 256
    eng = Engine(P, y, z, B, dtype=dtype, seed=seed)
    g = torch.Generator(device="cuda").manual_seed(0)
    lin = torch.linspace(-1, 1, P, device="cuda")
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    pool = 16
    data = []
    for _ in range(pool):           # a blob whose position depends on the class, plus noise
        labels = torch.randint(0, y, (B,), device="cuda", generator=g, dtype=torch.int32)
        ang = labels.float() * (2 * math.pi / y)
        cx, cy = 0.5 * torch.cos(ang), 0.5 * torch.sin(ang)
        blob = torch.exp(-((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / 0.08)
        img = torch.stack([blob, blob.flip(1), blob.flip(2)], 1) * 1.6 - 0.9
        img = (img + 0.05 * torch.randn(img.shape, device="cuda", generator=g)).clamp(-1, 1).contiguous()
        data.append((img, labels))
    eps = torch.empty(B, z, device="cuda")
    first = None
    for it in range(steps):
        x, labels = data[it % pool]
        eps.normal_(generator=g)
        eng.random_mask()
        eng.train_step(x, labels, eps, float(B), lr)
        if it in (0, 100, 250, 500) or it % 1000 == 0 or it == steps - 1:
            l = eng.read_losses()
            first = first or l
            print("step %5d  VAE %.6f  KL %.6f  LL %.6f" % (it, l["VAE"], l["KL"], l["LL"]), flush=True)
            if not all(math.isfinite(v) for v in l.values()):
                sys.exit("non-finite loss")
    assert torch.isfinite(eng.p_arena).all(), "non-finite parameter"
    assert l["LL"] < 0.5 * first["LL"], "the reconstruction loss did not fall"
    print("ok: LL %.5f -> %.5f over %d steps" % (first["LL"], l["LL"], steps))


if __name__ == "__main__":
    main()
