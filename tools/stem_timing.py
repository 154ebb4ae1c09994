#!/usr/bin/env python3
"""Phase stamps of the stem-shaped launch (7 row taps over 8-pixel x 4-channel windows) and of a 64-channel
3x3 layer, from the instrumented build:  make timing && CVG_LIB=build_exp/lib_timing.so python tools/stem_timing.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd import ops  # noqa: E402
from cvae_gan_amd._lib import lib  # noqa: E402
from cvae_gan_amd.engine import _desc, _wmap  # noqa: E402


def run(name, d, wm, x_elems, wshape, y_elems, reps=5):
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
    if hasattr(lib, "cvg_debug_timing_dump"):
        lib.cvg_debug_timing_dump()
        sys.stdout.flush()


def main():
    T = torch.bfloat16
    B, P, ngf, y = 256, 64, 64, 100
    Hp, Wp = P + 6, P + 8
    table = [(kh * 7 + kw) if kw < 7 else -1 for kh in range(7) for kw in range(8)]
    d = _desc(T, B, Hp, Wp, 32, P, P, ngf, 1, 7, x_pix=4)
    wm = _wmap(1, (3 + y) * 49, 0, 4, 1, 49, ngf, 3, 56, tap_mode=2, table=table)
    run("stem", d, wm, B * Hp * Wp * 4 + 64, (ngf, 3 + y, 7, 7), B * P * P * ngf)
    d = _desc(T, B, P, P, ngf, P, P, ngf, 3, 9, off=(-1, -1))
    wm = _wmap(1, ngf * 9, 0, ngf, 1, 9, ngf, ngf, 9)
    run("64ch 3x3", d, wm, B * P * P * ngf, (ngf, ngf, 3, 3), B * P * P * ngf)


if __name__ == "__main__":
    main()
