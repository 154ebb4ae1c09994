#!/usr/bin/env python3
"""Streaming rates of the device memory at sizes beyond the 256 MB Infinity Cache: fill (write only), copy (read + write)
and a read-only reduction of a 2 GiB bf16 tensor, with torch's own kernels.  For judging the HBM-bound layers of the
128 x 128 configurations:  python tools/hbm_stream.py"""
import torch

n = 1 << 30
a = torch.empty(n, dtype=torch.bfloat16, device="cuda")
b = torch.empty(n, dtype=torch.bfloat16, device="cuda")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


gb = n * 2 / 1e9
t = timed(lambda: a.fill_(1.0))
print("fill  %.2f GB written            %.3f ms  %.2f TB/s" % (gb, t * 1e3, gb / t / 1e3))
t = timed(lambda: b.copy_(a))
print("copy  %.2f GB read + %.2f written %.3f ms  %.2f TB/s" % (gb, gb, t * 1e3, 2 * gb / t / 1e3))
t = timed(lambda: a.view(torch.int16).max())
print("max   %.2f GB read               %.3f ms  %.2f TB/s" % (gb, t * 1e3, gb / t / 1e3))
