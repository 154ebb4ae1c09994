#!/usr/bin/env python3
"""Throughput of the adversarial step (cvae_gan_amd/gan.py) on synthetic data:
    python tools/gan_bench.py [config=c3] [steps=10] [dtype=bf16] [--eager]
One step = E/G forward (reconstruction + prior sample), D and C over the joint 3B batch (forward, own
backward, frozen backward to the images), E/G backward, three AdamW updates.  Launched eagerly."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd.gan import CVAEGAN  # noqa: E402

CONFIGS = {"c2": (32, 10, 32, 256), "c3": (64, 100, 32, 256)}


def main():
    a = sys.argv[1:]
    P, y, z, B = CONFIGS[a[0] if a and a[0] in CONFIGS else "c3"]
    steps = int(a[1]) if len(a) > 1 else 10
    dtype = torch.float32 if (len(a) > 2 and a[2] == "f32") else torch.bfloat16
    net = CVAEGAN(P, y, z, B, dtype=dtype)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, 3, P, P, device="cuda", generator=g) * 2 - 1
    labels = torch.randint(0, y, (B,), device="cuda", generator=g, dtype=torch.int32)
    eps = torch.randn(B, z, device="cuda", generator=g)
    z_p = torch.randn(B, z, device="cuda", generator=g)
    y_p = torch.nn.functional.one_hot(torch.randint(0, y, (B,), device="cuda", generator=g), y).float()
    engines = (net.vae, net.gen2, net.D.trunk, net.C.trunk)

    def one():
        net.vae.random_mask()
        net.gen2.random_mask()
        net.train_step(x, labels, eps, z_p, y_p, float(B), 3e-5)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    run = one
    if "--eager" not in a:             # the whole step (about 900 launches) as one HIP graph
        for e in engines:
            e.pack_weights(force=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for e in engines:
                e._packed_version = -1
            one()
        run = graph.replay
        run()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("CVAE-GAN step %s B=%d %s (%s): %.2f ms/step, %.0f real images/s; losses %s"
          % (a[0] if a else "c3", B, "f32" if dtype == torch.float32 else "bf16",
             "eager" if run is one else "hip-graph", dt * 1e3, B / dt, net.read_losses()))


if __name__ == "__main__":
    main()
