#!/usr/bin/env python3
"""Rate of the trainer's own loop (cvae_gan_amd.cvae.CVAE) at the benchmark shape, host data included:
uint8 HWC batches in host memory -> pinned buffer -> copy stream -> device normalisation
(DevicePrefetcher) -> train step, replayed from a HIP graph as the trainer does on one GPU, and with
eager launches (args.graph = False) for comparison.  This is the PCIe-inclusive figure next to bench.py's
resident-input one.    python tools/trainer_rate.py [steps=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd.cvae import CVAE  # noqa: E402
from cvae_gan_amd.datasets import DevicePrefetcher  # noqa: E402


class Args:
    pass


def trainer(graph, steps):
    a = Args()
    a.epoch, a.batch_size, a.save_dir, a.result_dir = 1, 256, "/tmp/trainer_rate_ck", "/tmp/trainer_rate_res"
    a.dataset, a.data_dir, a.data_size = "Synth", "/tmp/trainer_rate_none", 256 * 16
    a.device, a.y_dim, a.z_dim, a.pix_dim, a.lrG = "cuda", 100, 32, 64, 3e-5
    a.worker, a.resume, a.testmode, a.pkl, a.dtype, a.graph = 0, False, False, "", "bf16", graph
    m = CVAE(a)
    eps = torch.empty(m.local_batch, m.z_dim, device=m.device)
    g = torch.Generator().manual_seed(0)
    B, P, y = a.batch_size, a.pix_dim, a.y_dim
    # batches as a DataLoader with enough workers would deliver them: uint8 HWC images + class indices in host memory
    pool = [(torch.randint(0, 256, (B, P, P, 3), generator=g, dtype=torch.uint8),
             torch.randint(0, y, (B,), generator=g)) for _ in range(8)]

    def run(n):
        loader = [pool[i % len(pool)] for i in range(n)]
        for x, cls in DevicePrefetcher(loader, B, P, "cuda", y_dim=y):
            eps.normal_()
            m.train_step(x, cls, eps)
        torch.cuda.synchronize()

    run(10)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    print("trainer loop, host uint8 batches, %s: %.3f ms/step, %.0f images/s (%d steps); losses %s"
          % ("HIP-graph replay" if graph else "eager launches", dt / steps * 1e3, a.batch_size * steps / dt, steps,
             m.read_losses()), flush=True)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    trainer(True, steps)
    trainer(False, steps)


if __name__ == "__main__":
    main()
