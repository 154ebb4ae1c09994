#!/usr/bin/env python3
"""Calibration: what the vendor GEMM (torch.mm -> hipBLASLt) reaches in bf16 at the shapes of the
up-convolutions' low-resolution GEMMs at c3, batch 256: V = x W (forward at low resolution), dX = Z W (input
gradient), dW = Z^T x (weight gradient).  Timing only; nothing in the product path uses it."""
import torch

LEVELS = [(256 * 32 * 32, 128, 64), (256 * 16 * 16, 256, 128), (256 * 8 * 8, 512, 256), (256 * 4 * 4, 512, 512)]


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    torch.manual_seed(0)
    for M, cin, cout in LEVELS:
        x = torch.randn(M, cin, device="cuda").to(torch.bfloat16)
        z = torch.randn(M, 9 * cout, device="cuda").to(torch.bfloat16)
        w = torch.randn(cin, 9 * cout, device="cuda").to(torch.bfloat16)          # [cin][9 cout]
        wt = w.t().contiguous()
        v = torch.empty(M, 9 * cout, device="cuda", dtype=torch.bfloat16)
        dx = torch.empty(M, cin, device="cuda", dtype=torch.bfloat16)
        dw = torch.empty(9 * cout, cin, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * cin * 9 * cout
        for name, fn in (("V  = x W   ", lambda: torch.mm(x, w, out=v)),
                         ("dX = Z W^T ", lambda: torch.mm(z, wt, out=dx)),
                         ("dW = Z^T x ", lambda: torch.mm(z.t(), x, out=dw))):
            t = timed(fn)
            print("%s M %7d cin %4d 9cout %5d  %7.1f us  %7.1f TFLOP/s" % (name, M, cin, 9 * cout, t * 1e6, fl / t / 1e12), flush=True)


if __name__ == "__main__":
    main()
