#!/usr/bin/env python3
"""Rate of the trainer's own loop at the benchmark shape, host data included: uint8 HWC batches in
host memory -> pinned buffer -> copy stream -> device normalisation (DevicePrefetcher) -> eager
train step.  This is the PCIe-inclusive figure next to bench.py's resident-input one.
    python tools/trainer_rate.py [steps=100]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd.engine import Engine  # noqa: E402
from cvae_gan_amd.datasets import DevicePrefetcher  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    P, y, z, B = 64, 100, 32, 256
    g = torch.Generator().manual_seed(0)
    pool = [(torch.randint(0, 256, (B, P, P, 3), generator=g, dtype=torch.uint8),
             torch.randint(0, y, (B,), generator=g)) for _ in range(8)]
    eng = Engine(P, y, z, B, dtype=torch.bfloat16, seed=0)
    eps = torch.empty(B, z, device="cuda")

    def run(n):
        loader = [pool[i % len(pool)] for i in range(n)]
        for x, cls in DevicePrefetcher(loader, B, P, "cuda", y_dim=y):
            eps.normal_()
            eng.random_mask()
            eng.train_step(x, cls, eps, float(B), 3e-5)
        torch.cuda.synchronize()

    run(10)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    print("trainer loop, host uint8 batches, eager launches: %.3f ms/step, %.0f images/s (%d steps); losses %s"
          % (dt / steps * 1e3, B * steps / dt, steps, eng.read_losses()))


if __name__ == "__main__":
    main()
