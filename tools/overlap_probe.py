#!/usr/bin/env python3
"""How well does an MFMA-bound convolution overlap with HBM-bound BatchNorm passes of OTHER data when the two
run on different streams?  Decides whether splitting the forward pass into two half-batch chains can hide its
memory passes.    python tools/overlap_probe.py

Prints microseconds per repetition for: the convolution alone, the BatchNorm passes alone, both on one stream,
both on two streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvae_gan_amd import ops  # noqa: E402
from cvae_gan_amd.engine import Engine, EP  # noqa: E402


def main():
    eng = Engine(64, 100, 32, 256, dtype=torch.bfloat16, seed=0)
    eng.pack_weights()
    torch.cuda.synchronize()
    for conv_blk in (1, 3):
        blk = eng.enc["blocks"][conv_blk]          # 128 -> 128 at 32x32 (M = 262144) / 512 -> 512 at 8x8
        big = eng.enc["blocks"][0]                 # BatchNorm over 1M x 64
        M, C = big["M"], big["cin"]
        p = big["prefix"] + "res.res.1"
        gamma, beta = eng.params[p + ".weight"], eng.params[p + ".bias"]
        rm, rv, nbt = eng.buffers[p + ".running_mean"], eng.buffers[p + ".running_var"], eng.buffers[p + ".num_batches_tracked"]

        def conv():
            ops.conv_fwd(blk["conv1"].desc, blk["b1"], blk["conv1"].w, blk["c1"], ws=eng.ws)

        def bn():
            ops.bn_stats(big["c1"], M, C, big["st1"], rm, rv, nbt, eng.red_ws)
            ops.bn_act_fwd(big["c1"], None, big["b1"], M, C, big["st1"], gamma, beta)

        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        reps = 200

        def timed(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(reps)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1000.0 / reps

        def only_conv(n=1):
            for _ in range(n):
                conv()

        def only_bn(n=1):
            for _ in range(n):
                bn()

        def serial(n=1):
            for _ in range(n):
                conv()
                bn()

        def two_streams(n=1):
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            s1.wait_event(ev)
            s2.wait_event(ev)
            with torch.cuda.stream(s1):
                for _ in range(n):
                    conv()
            with torch.cuda.stream(s2):
                for _ in range(n):
                    bn()
            for s in (s1, s2):
                e = torch.cuda.Event()
                e.record(s)
                cur.wait_event(e)

        a, b, c, d = timed(only_conv), timed(only_bn), timed(serial), timed(two_streams)
        print("block %d conv %.1f us   bn stats+apply (1M x 64) %.1f us   one stream %.1f us   two streams %.1f us   "
              "(hidden: %.0f %% of the shorter)" % (conv_blk, a, b, c, d, 100.0 * (a + b - d) / min(a, b)))


if __name__ == "__main__":
    main()
