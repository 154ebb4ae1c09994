#!/usr/bin/env python3
"""Time of the tail's forward launch (7 row taps x 64 channels into 28 (kernel column, colour) outputs per padded
column) and of the stem's image-gradient launch of the same shape:  python tools/tail_timing.py
CVG_NO_RES_ROWS=1 selects the plain four-wave kernel for the comparison."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd import ops  # noqa: E402
from cvae_gan_amd.engine import _desc, _wmap, PAD_REFLECT  # noqa: E402


def run(name, d, wm, x_elems, wshape, y_elems, reps=20):
    T = torch.bfloat16
    x = torch.randn(x_elems, device="cuda").to(T)
    w = torch.randn(*wshape, device="cuda") * 0.05
    y = torch.zeros(y_elems, device="cuda", dtype=T)
    packed = torch.zeros(d.w_rows * d.w_ld, dtype=T, device="cuda")
    ops.weight_pack(d, wm, w, packed)
    ws = torch.zeros(max(ops.conv_fwd_workspace(d), 1024) // 4 + 16, device="cuda")
    for _ in range(3):
        ops.conv_fwd(d, x, packed, y, ws=ws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv_fwd(d, x, packed, y, ws=ws)
    e.record()
    torch.cuda.synchronize()
    print("%s: %s  %.1f us per launch" % (name, ops.conv_fwd_kernel(d), s.elapsed_time(e) / reps * 1e3), flush=True)


def main():
    T = torch.bfloat16
    B, P, ngf = 256, 64, 64
    d = _desc(T, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT)
    wm = _wmap(4, 1, ngf * 49, ngf, 7, 49, 28, ngf, 7, nr_valid=3)
    run("tail forward", d, wm, B * P * P * ngf, (3, ngf, 7, 7), B * P * (P + 6) * 32)
    d = _desc(T, B, P, P, ngf, P + 6, P, 32, 1, 7, off=(-6, 0))
    wm = _wmap(4, 1, 49, ngf, 7, 103 * 49, 28, ngf, 7, tap_mode=1, nr_valid=3)
    run("stem image gradient", d, wm, B * P * P * ngf, (ngf, 103, 7, 7), B * (P + 6) * P * 32)


if __name__ == "__main__":
    main()
