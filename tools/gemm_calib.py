#!/usr/bin/env python3
"""Calibration: what the vendor GEMM (torch.mm -> hipBLASLt) reaches in bf16 at the GEMM shapes of
this network's 3x3 convolutions (M = B*OH*OW, N = Cout, K = 9*Cin at batch 256, 64x64 images), with
the im2col matrix already materialised, next to a large square product.  Timing only; nothing in
the product path uses it."""
import torch

SHAPES = [("square 8192", 8192, 8192, 8192),
          ("64ch  @64x64", 256 * 64 * 64, 64, 576),
          ("128ch @64x64 (decoder 3)", 256 * 64 * 64, 64, 1152),
          ("128ch @32x32", 256 * 32 * 32, 128, 1152),
          ("256->128 @32x32 (decoder 2)", 256 * 32 * 32, 128, 2304),
          ("256ch @16x16", 256 * 16 * 16, 256, 2304),
          ("512->256 @16x16 (decoder 1)", 256 * 16 * 16, 256, 4608),
          ("512ch @8x8", 256 * 8 * 8, 512, 4608)]


def main():
    torch.manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            torch.mm(a, b, out=c)
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.mm(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        print("%-30s M %8d N %4d K %5d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
        del a, b, c


if __name__ == "__main__":
    main()
