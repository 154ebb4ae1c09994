#!/usr/bin/env python3
"""Time one convolution layer (forward / weight gradient) of the step in isolation.
    python tools/layer_bench.py B H Cin Cout [k=3] [pad=1] [mode=fwd|wgrad|both] [reps=20] [dtype=bf16]
pad = 1: zero-padded 3x3 (encoder), pad = 0: valid conv on an (H+k-1) input (decoder up-conv),
pad = -1: reflection padding (decoder tail)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd import ops  # noqa: E402
from cvae_gan_amd.engine import _desc, _wmap  # noqa: E402
from cvae_gan_amd._lib import PAD_REFLECT, PAD_ZERO  # noqa: E402


def main():
    a = sys.argv[1:]
    B, H, cin, cout = [int(v) for v in a[:4]]
    k = int(a[4]) if len(a) > 4 else 3
    pad = int(a[5]) if len(a) > 5 else 1
    mode = a[6] if len(a) > 6 else "both"
    reps = int(a[7]) if len(a) > 7 else 20
    dtype = torch.float32 if (len(a) > 8 and a[8] == "f32") else torch.bfloat16
    T = k * k
    if pad == 0:
        Hin = H + k - 1
        d = _desc(dtype, B, Hin, Hin, cin, H, H, max(cout, 4), k, T)
    elif pad < 0:
        Hin = H
        d = _desc(dtype, B, H, H, cin, H, H, max(cout, 4), k, T, off=(-(k // 2), -(k // 2)), pad_mode=PAD_REFLECT)
    else:
        Hin = H
        d = _desc(dtype, B, H, H, cin, H, H, max(cout, 4), k, T, off=(-(k // 2), -(k // 2)))
    wm = _wmap(1, cin * T, 0, cin, 1, T, cout, cin, T)
    N = d.N
    x = torch.randn(B * Hin * Hin, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * T) ** 0.5
    y = torch.zeros(B * H * H, N, device="cuda", dtype=dtype)
    dy = torch.randn(B * H * H, max(N, 8), device="cuda").to(dtype)
    gw = torch.zeros_like(w)
    packed = torch.zeros(d.w_rows * d.w_ld, dtype=dtype, device="cuda")
    ops.weight_pack(d, wm, w, packed)
    ws = torch.zeros(max(ops.conv_fwd_workspace(d), ops.conv_wgrad_workspace(d), 1024) // 4 + 16, device="cuda")
    flops = 2.0 * B * H * H * cout * cin * T

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e-3

    if mode in ("fwd", "both"):
        t = timeit(lambda: ops.conv_fwd(d, x, packed, y, ws=ws))
        print("fwd   B%d H%d %d->%d k%d pad%d: %8.1f us  %7.1f TF" % (B, H, cin, cout, k, pad, t * 1e6, flops / t / 1e12))
        from cvae_gan_amd._lib import lib
        if hasattr(lib, "cvg_debug_timing_dump"):      # only in instrumented experiment builds
            lib.cvg_debug_timing_dump()
    if mode in ("wgrad", "both"):
        t = timeit(lambda: ops.conv_wgrad(d, x, dy, dy.shape[1], wm, gw, ws))
        print("wgrad B%d H%d %d->%d k%d pad%d: %8.1f us  %7.1f TF" % (B, H, cin, cout, k, pad, t * 1e6, flops / t / 1e12))


if __name__ == "__main__":
    main()
