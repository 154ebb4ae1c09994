#!/usr/bin/env python3
"""Eval-mode throughput of the inference paths (models/cvae.py:197-220 and :167-178): encoder +
sampler + decoder forward, and decoder only, replayed from a HIP graph.
    python tools/infer_bench.py [P=128] [y=100] [B=256] [reps=20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd.engine import Engine  # noqa: E402


def main():
    a = sys.argv[1:]
    P = int(a[0]) if len(a) > 0 else 128
    y = int(a[1]) if len(a) > 1 else 100
    B = int(a[2]) if len(a) > 2 else 256
    reps = int(a[3]) if len(a) > 3 else 20
    z = 32
    eng = Engine(P, y, z, B, dtype=torch.bfloat16, seed=0)
    eng.training = False
    x = torch.rand(B, 3, P, P, device="cuda") * 2 - 1
    labels = torch.randint(0, y, (B,), device="cuda", dtype=torch.int32)
    eps = torch.randn(B, z, device="cuda")
    zs = torch.randn(B, z, device="cuda")
    ys = torch.zeros(B, y, device="cuda")
    ys[torch.arange(B), labels.long()] = 1

    def timed(fn, name):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-28s P %d y %d B %d: %8.3f ms  %9.1f images/s" % (name, P, y, B, dt * 1e3, B / dt))

    timed(lambda: eng.reconstruct(x, labels, eps), "encoder+decoder (eval)")
    timed(lambda: eng.decode(zs, ys), "decoder only (eval)")


if __name__ == "__main__":
    main()
