#!/usr/bin/env python3
"""Throughput of the CVAE train step on MI355X (BASELINE.json metric: train images/sec at
64x64, batch 256 per GPU), with the roofline of the dominant kernel and a CPU baseline.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = forward + KL/MSE losses + backward + gradient all-reduce (N > 1) + AdamW over one
synthetic batch that is already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (pix_dim, y_dim, z_dim, per-GPU batch, down/up stages)   -- z_dim 32 from the reference's train.sh:6
    "c2": (32, 10, 32, 256, 4),
    "c3": (64, 100, 32, 256, 4),
    "c4": (128, 100, 32, 64, 4),
    "c4d": (128, 100, 32, 64, 5),          # "deeper stacks": ndown = nup = 5 (networks.py:165, :133)
    "cli": (512, 1000, 256, 50, 4),        # the reference's command-line defaults (starter.py:19-27)
    "c5": (128, 100, 32, 1024, 4),         # inference only: eval-mode encoder + sampler + decoder forward (cvae.py:197-220)
}
# forward multiply-adds per image of the reference's formulation (networks.py:164-196, :91-161) with the label planes
# folded into a class bias, for the "reference_formulation" entries of the JSON line: (forward GMAC, stem GMAC)
REFERENCE_GMAC = {"c2": (0.652, 0.0096), "c3": (2.611, 0.0385), "c4": (10.447, 0.1541), "c4d": (10.921, 0.1541),
                  "c5": (10.447, 0.1541)}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def make_batches(eng, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.rand(eng.B, 3, eng.P, eng.P, device="cuda", generator=g) * 2 - 1
        labels = torch.randint(0, eng.y, (eng.B,), device="cuda", generator=g, dtype=torch.int32)
        eps = torch.randn(eng.B, eng.z, device="cuda", generator=g)
        out.append((x, labels, eps))
    return out


def gemm_flops(g):
    """Algorithmic FLOPs of one launch of a packed GEMM (forward, dgrad or wgrad): 2*M*N*K over
    the layer's real channels and taps (the zero padding the kernels add is not counted)."""
    d, m = g.desc, g.wmap
    M = d.B * d.OH * d.OW
    n = m.n_valid
    k = m.kc_valid * (sum(1 for i in range(m.ntaps) if m.tap_table[i] >= 0) if m.tap_mode == 2 else m.ntaps)
    return 2.0 * M * n * k


class KernelTimer:
    """HIP-event brackets around every implicit-GEMM launch, on the stream they are launched on."""

    def __init__(self, eng):
        self.eng, self.records, self.hbm, self.empty = eng, [], [], []
        self._overhead = None
        self._fwd, self._wgrad, self._bn, self._bn_bwd = eng._fwd, eng._wgrad, eng._bn, eng._bn_bwd
        self._fwd_folded = eng._fwd_folded

    def __enter__(self):
        eng = self.eng

        def fwd(g, x, y, bias=None, **kw):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._fwd(g, x, y, bias=bias, **kw)
            b.record()
            self.records.append((self.kernel_name(g, "fwd"), gemm_flops(g), a, b, g))
            self.empty_pair()

        def fwd_folded(g, x, y, scale, shift, **kw):       # inference: BatchNorm folded into the epilogue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._fwd_folded(g, x, y, scale, shift, **kw)
            b.record()
            self.records.append((self.kernel_name(g, "fwd"), gemm_flops(g), a, b, g))
            self.empty_pair()

        def wgrad(g, x, dy, dy_ld):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._wgrad(g, x, dy, dy_ld)
            b.record()
            self.records.append((self.kernel_name(g, "wgrad"), gemm_flops(g), a, b, g))

        es = 2 if eng.dtype == torch.bfloat16 else 4

        # HBM-bound passes, algorithmic bytes: BatchNorm forward = statistics (1 read) + apply (1 read,
        # + the residual, 1 write); backward = two passes over dy, x (and y where the gate is read from it) + 1 write
        def bn(x, res, y, M, C, stats, prefix, partial_rows=0):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._bn(x, res, y, M, C, stats, prefix, partial_rows=partial_rows)
            b.record()
            # with partial_rows the producing convolution has already summed x: no statistics read
            passes = (2 if partial_rows else 3) + (res is not None)
            self.hbm.append(("batchnorm+lrelu forward (statistics, apply)", M * C * es * passes, a, b))

        def bn_bwd(dy, y, x, dx, M, C, stats, prefix, **kw):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._bn_bwd(dy, y, x, dx, M, C, stats, prefix, **kw)
            b.record()
            self.hbm.append(("batchnorm+lrelu backward (sums, dx)", M * C * es * (2 * (2 + (y is not None)) + 1), a, b))

        eng._fwd, eng._wgrad, eng._bn, eng._bn_bwd, eng._fwd_folded = fwd, wgrad, bn, bn_bwd, fwd_folded
        return self

    def empty_pair(self):
        """Two events with nothing between them, in the same stream of queued work: what a bracket adds to the
        kernel it times (the second event's packet is processed that much after the kernel ends)."""
        c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c.record()
        d.record()
        self.empty.append((c, d))

    def overhead_s(self):
        """Median empty-pair time, subtracted from every bracket (rocprofv3's kernel durations have no such term)."""
        if self._overhead is None:
            torch.cuda.synchronize()
            ts = sorted(c.elapsed_time(d) for c, d in self.empty)
            self._overhead = ts[len(ts) // 2] * 1e-3 if ts else 0.0
        return self._overhead

    def __exit__(self, *exc):
        self.eng._fwd, self.eng._wgrad, self.eng._bn, self.eng._bn_bwd = self._fwd, self._wgrad, self._bn, self._bn_bwd
        self.eng._fwd_folded = self._fwd_folded

    def timed(self, label, nbytes, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.hbm.append((label, nbytes, a, b))

    def hbm_summary(self, reps):
        torch.cuda.synchronize()
        agg = {}
        for label, nbytes, a, b in self.hbm:
            s = agg.setdefault(label, [0.0, 0.0])
            s[0] += nbytes
            s[1] += a.elapsed_time(b) * 1e-3
        return {k: {"GB/s": round(v[0] / v[1] / 1e9, 1), "frac_of_8TB/s": round(v[0] / v[1] / 8e12, 3),
                    "ms_per_step": round(v[1] / reps * 1e3, 3)} for k, v in agg.items()}

    @staticmethod
    def kernel_name(g, kind):
        import ctypes
        from cvae_gan_amd._lib import lib
        if kind == "fwd":
            return lib.cvg_conv_fwd_kernel(ctypes.byref(g.desc)).decode()
        return lib.cvg_conv_wgrad_kernel(ctypes.byref(g.desc)).decode() + "(+reduce)"

    def per_layer(self, reps):
        """One line per GEMM launch site: layer, kind, GEMM shape, time, algorithmic TFLOP/s."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, a, b, g in self.records:
            d = g.desc
            key = (("E." + g.key[8:]) if g.key.startswith("encoder.") else ("D." + g.key[8:]) if g.key.startswith("decoder.") else g.key, name,
                   d.B * d.OH * d.OW, d.N, d.ntaps * d.Cin, d.stride, d.oy_off * 2 + d.ox_off, id(g))
            s = agg.setdefault(key, [fl, 0.0])
            s[1] += max(a.elapsed_time(b) * 1e-3 - self.overhead_s(), 1e-7) / reps
        lines = []
        for (key, name, M, N, K, st, cls, _), (fl, t) in agg.items():
            lines.append("%-28s %-30s M %8d N %5d K %5d s%d c%d  %8.1f us  %7.1f TF" % (
                key, name, M, N, K, st, cls, t * 1e6, fl / t / 1e12))
        return "\n".join(lines)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, fl, a, b, _ in self.records:
            t = max(a.elapsed_time(b) * 1e-3 - self.overhead_s(), 1e-7)
            s = agg.setdefault(name, [0.0, 0.0, 0])
            s[0] += fl
            s[1] += t
            s[2] += 1
        return agg


def mfma_rate_measured(achieved_tflops, dtype):
    """A pure MFMA loop on this GPU, now (cvg_mfma_rate_probe: independent v_mfma_f32_32x32x16_bf16 from registers on
    every SIMD, no memory traffic): the rate the matrix pipes deliver at the clock the chip sustains under matrix load,
    beside the data-sheet peak that `roofline.frac` is quoted against."""
    if dtype != "bf16":
        return None
    from cvae_gan_amd import ops
    out = torch.empty(1024 * 256, dtype=torch.float32, device="cuda")
    ops.mfma_rate_probe(out, 1024, 200)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        flops = ops.mfma_rate_probe(out, 1024, 4000)
        b.record()
        torch.cuda.synchronize()
        best = max(best, flops / (a.elapsed_time(b) * 1e-3) / 1e12)
    return {"tflops": round(best, 1), "frac_of_it": round(achieved_tflops / best, 4),
            "what": "pure v_mfma_f32_32x32x16_bf16 loop, 4 waves per CU x 4 workgroups, measured in this run"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(P, y, z, B, n_stage, seconds_budget, pad_from=None):
    """Train steps per second of the CPU oracle (a port of the reference's step, pinned to the reference by
    the golden fixtures) on this host: fp32, same synthetic distribution.  pad_from: MNIST-shaped inputs
    (gray pad_from x pad_from, zero-padded to P and replicated to 3 channels: the reference's networks
    take 3-channel images whose side is a multiple of 16)."""
    from oracle import ref_cpu as O
    e, d = O.init_params(P, y, z, seed=0, n_stage=n_stage)
    st = O.TrainState(e, d)
    g = torch.Generator().manual_seed(0)
    F = 4 * (z + y)

    def one():
        if pad_from:
            from cvae_gan_amd.datasets import mnist_to_input
            x = mnist_to_input(torch.rand(B, 1, pad_from, pad_from, generator=g), P)
        else:
            x = torch.rand(B, 3, P, P, generator=g) * 2 - 1
        labels = torch.randint(0, y, (B,), generator=g)
        eps = torch.randn(B, z, generator=g)
        mask = (torch.rand(B, F, generator=g) < 0.5).float()
        O.train_step(st, x, labels, eps, mask, y, B, 3e-5)

    one()                                   # warm-up
    t0, n = time.time(), 0
    while n < 3 or (time.time() - t0 < seconds_budget and n < 10):
        one()
        n += 1
    return B * n / (time.time() - t0), n


def cpu_baseline(P, y, z, n_stage=4, seconds_budget=14.0):
    """The benchmark's shape at batch 32, plus BASELINE configuration 1 (MNIST-shaped 28 x 28 gray images,
    10 classes, batch 64, run as 32 x 32 x 3)."""
    B = 32
    rate, n = _time_oracle(P, y, z, B, n_stage, seconds_budget)
    out = {"value": round(rate, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
           "kind": "port", "sample": "%d steps of the fp32 CPU oracle at %dx%d, %d classes, batch %d" % (n, P, P, y, B)}
    r1, n1 = _time_oracle(32, 10, z, 64, 4, 6.0, pad_from=28)
    out["config1"] = {"value": round(r1, 2), "unit": "images/sec",
                      "sample": "%d steps of the fp32 CPU oracle on MNIST-shaped inputs (28x28 gray zero-padded to "
                                "32x32x3), 10 classes, batch 64" % n1}
    return out


def load_traffic(config, family):
    """Memory-side bytes per launch of a kernel family from the newest counter file under profiles/ for this
    configuration (profiles/<tag>_hbm_traffic_<config>.json from tools/profile_round.sh), or None."""
    import glob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    # newest = last by name (round3_b_... after round3_a_... after the unprefixed first file): modification times
    # do not survive a checkout or a copy of the tree
    paths = sorted(glob.glob(os.path.join(here, "*hbm_traffic_%s.json" % config)),
                   key=lambda q: (os.path.basename(q).startswith("round"), os.path.basename(q)))
    for path in reversed(paths):
        with open(path) as f:
            got = json.load(f).get(family.replace("(+reduce)", ""), {}).get("bytes")
        if got is not None:
            return got
    return None


def infer_main(args, eng, batches, json_fd, P, y, z, B, n_stage, world):
    """--config c5: eval-mode reconstruction (encoder + sampler + decoder forward, BatchNorm folded into the
    convolution epilogues, no dropout) of `B` images per step, replayed from a HIP graph."""
    assert world == 1, "the inference configuration runs on one GPU"
    eng.training = False
    x0, l0, e0 = batches[0]
    gx, gl, ge = x0.clone(), l0.clone(), e0.clone()
    for _ in range(max(args.warmup, 1)):
        eng.reconstruct(gx, gl, ge)
    torch.cuda.synchronize()
    use_graph = not args.no_graph
    if use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eng.reconstruct(gx, gl, ge)

    def step(i):
        x, labels, eps = batches[i % len(batches)]
        gx.copy_(x)
        gl.copy_(labels)
        ge.copy_(eps)
        if use_graph:
            graph.replay()
        else:
            eng.reconstruct(gx, gl, ge)

    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # roofline of the dominant kernel: event brackets around every GEMM launch of two eager passes
    wa = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
    wb = torch.empty_like(wa)
    with KernelTimer(eng) as kt:
        for i in range(2):
            for _ in range(24):
                torch.mm(wa, wa, out=wb)
            eng.reconstruct(gx, gl, ge)
        agg = kt.summary()
        if args.per_layer:
            print(kt.per_layer(2), file=sys.stderr)
    name, (fl, t, n) = max(agg.items(), key=lambda kv: kv[1][1])
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    executed = sum(v[0] for v in agg.values()) / 2
    roofline = {"bound": "mfma", "kernel": name, "achieved": round(fl / t / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(fl / t / 1e12 / peak, 4), "traffic": load_traffic(args.config, name),
                "launches_per_step": n // 2, "avg_launch_ms": round(t / n * 1e3, 4),
                "event_pair_overhead_us": round(kt.overhead_s() * 1e6, 2),
                "mfma_rate_measured": mfma_rate_measured(fl / t / 1e12, args.dtype),
                "all_gemm_ms_per_step": round(sum(v[1] for v in agg.values()) / 2 * 1e3, 3),
                "executed_tflop_per_step": round(executed / 1e12, 4),
                "step_tflops_executed": round(executed / (dt / args.steps) / 1e12, 1),
                "per_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / 2 * 1e3, 3)}
                               for k, v in sorted(agg.items())}}
    out = {
        "metric": "inference images/sec", "value": round(B * args.steps / dt, 1), "unit": "images/sec",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "CVAE inference (eval-mode encoder + sampler + decoder forward), %dx%dx3, %d classes, z_dim %d, "
                               "%d down/up stages, batch %d, random-init weights" % (P, P, y, z, n_stage, B),
                   "name": args.config, "global_batch": B, "parallelism": "dp1",
                   "launch": "hip-graph" if use_graph else "eager"},
        "roofline": roofline,
    }
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the config's batch is the GLOBAL batch, split over the ranks "
                         "(the reference's nn.DataParallel semantics); default is weak scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--lr", type=float, default=3e-5)
    ap.add_argument("--per-layer", action="store_true", help="print a per-layer GEMM table to stderr")
    args = ap.parse_args()

    trace = os.environ.get("CVG_BENCH_TRACE") is not None
    # Libraries write to file descriptor 1 behind Python's back (RCCL prints a version banner there
    # when its first communicator comes up).  stdout must carry the one JSON line only: everything
    # else goes to stderr, and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def note(msg):
        if trace:
            print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1:
        backend = os.environ.get("CVG_BENCH_BACKEND") or os.environ.get("CVG_DDP_BACKEND", "nccl")   # "gloo": rehearsal of N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    if world == 1 and os.environ.get("CVG_BENCH_FORCE_DDP") == "nccl":
        # rehearsal of the N > 1 schedule on one GPU with real RCCL calls (a one-rank group)
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500),
                                rank=0, world_size=1, device_id=torch.device("cuda", local))

    from cvae_gan_amd.engine import Engine
    from cvae_gan_amd import ddp

    P, y, z, B, n_stage = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    if args.strong:
        assert B % world == 0, "global batch %d does not split over %d ranks" % (B, world)
        B //= world
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    note("process group up")
    eng = Engine(P, y, z, B, dtype=dtype, seed=0, n_stage=n_stage)
    ddp.broadcast_state(eng)
    note("state broadcast")
    batches = make_batches(eng, 4, 1234 + rank)
    batch_size_flag = float(B * world)       # the reference divides both losses by --batch_size (global)
    loss_scale = 1.0 / world

    # N > 1: the gradient arena is exchanged in three ranges (it is laid out in the order backward
    # finishes gradients), each as one SUM all-reduce on a side stream, by the same GradSync the trainer
    # uses: the decoder's (33 MB at c3) travels while the encoder's head and deep blocks run backward,
    # theirs (45 MB: nearly all of the encoder's parameters) while the shallow blocks and the stem run,
    # which are the longest part of the encoder's backward pass; only the last range (1.5 MB) is exposed.
    force_ddp = os.environ.get("CVG_BENCH_FORCE_DDP") is not None      # rehearse the N > 1 control flow on one rank
    multi = world > 1 or force_ddp
    sync = ddp.GradSync.for_engine(eng) if multi else None
    if multi:
        (dec_lo, dec_hi), (up_lo, up_hi) = eng.arena_range("dec"), eng.enc_upper_range()
        assert dec_lo == 0 and up_lo == dec_hi, "arena order: decoder, encoder head + deep blocks, the rest"

    if args.config == "c5":
        return infer_main(args, eng, batches, json_fd, P, y, z, B, n_stage, world)

    def step(i):
        x, labels, eps = batches[i % len(batches)]
        eng.random_mask()
        if multi:
            sync.begin_step()
            eng.grad_ready = sync.on_ready                   # exchanges start as backward reports its stages
            eng.forward_backward(x, labels, eps, batch_size_flag, loss_scale)
            eng.grad_ready = None
            sync.finish()
        else:
            eng.forward_backward(x, labels, eps, batch_size_flag, loss_scale)
        note("step %d enqueued" % i)
        eng.adamw_step(args.lr)

    # A HIP graph of one whole step removes the per-launch host cost (about 170 launches per step).  With
    # N > 1 the step is four graphs with the gradient exchanges between them (cvae_gan_amd.ddp.GraphedStep, the
    # schedule the multi-rank trainer runs as well).
    use_graph = not args.no_graph
    for i in range(max(args.warmup, 1)):
        step(i)
    torch.cuda.synchronize()
    note("eager warm-up done")
    if use_graph:
        gstep = ddp.GraphedStep(eng, batch_size_flag, args.lr, loss_scale, sync=sync if multi else None)
        gstep.capture(*batches[0])

        def step(i):                                    # noqa: F811
            gstep.run(*batches[i % len(batches)])

        step(0)
        torch.cuda.synchronize()
        note("graphs captured and replayed once")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = eng.read_losses()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: event-bracketed launches of two more (eager) steps
    eng.grad_ready = None
    # (everything on one stream so that a bracket times its launch alone, and behind ~30 ms of
    # queued matrix products so that the host has enqueued the whole step before its first kernel
    # starts and the clocks are those of a busy GPU: the events then measure kernels, not launch
    # gaps or a clock ramp)
    overlap, eng.overlap = eng.overlap, False
    eng.set_co_launches(1)               # one stream: every weight-gradient launch has the whole chip
    wa = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
    wb = torch.empty_like(wa)
    with KernelTimer(eng) as kt:
        for i in range(2):
            x, labels, eps = batches[i % len(batches)]
            for _ in range(24):
                torch.mm(wa, wa, out=wb)
            eng.forward_backward(x, labels, eps, batch_size_flag, loss_scale)
            kt.timed("adamw (28 B per parameter)", 28 * eng.arena_numel, lambda: eng.adamw_step(args.lr))
        agg = kt.summary()
        hbm = kt.hbm_summary(2)
        if args.per_layer:
            print(kt.per_layer(2), file=sys.stderr)
    eng.overlap = overlap
    eng.set_co_launches(eng._n_side)
    name, (fl, t, n) = max(agg.items(), key=lambda kv: kv[1][1])
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    achieved = fl / t / 1e12
    total_gemm_t = sum(v[1] for v in agg.values()) / 2
    # HBM-side bytes per launch of that kernel family: PMC counters (FETCH_SIZE doubled as gfx950
    # needs, plus WRITE_SIZE) from two separate rocprofv3 passes over this same command, reduced by
    # tools/pmc_traffic.py into profiles/; null when the file is absent or lacks the family
    traffic = load_traffic(args.config, name)
    roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "launches_per_step": n // 2, "avg_launch_ms": round(t / n * 1e3, 4),
                # median of empty event pairs recorded between the launches, already subtracted from every bracket
                "event_pair_overhead_us": round(kt.overhead_s() * 1e6, 2),
                "mfma_rate_measured": mfma_rate_measured(achieved, args.dtype),
                "all_gemm_ms_per_step": round(total_gemm_t * 1e3, 3),
                # FLOPs the step's GEMM launches execute (the gradients of the up-convolutions are taken at low
                # resolution: a quarter of the multiplications of the reference's formulation for those layers),
                # and what the reference's own formulation of the step would take (forward GMAC g, stem s:
                # 2 * (3 g - s) per image, label planes folded)
                "executed_tflop_per_step": round(sum(v[0] for v in agg.values()) / 2 / 1e12, 4),
                "step_tflops_executed": round(sum(v[0] for v in agg.values()) / 2 / (dt / args.steps) / 1e12, 1),
                "reference_formulation_tflop_per_step": (round(2 * (3 * REFERENCE_GMAC[args.config][0] - REFERENCE_GMAC[args.config][1]) * B / 1e3, 4)
                                                         if args.config in REFERENCE_GMAC else None),
                "per_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / 2 * 1e3, 3)}
                               for k, v in sorted(agg.items())},
                # the step's HBM-bound passes: algorithmic bytes / event-bracketed time (small finalize
                # kernels included), against the 8 TB/s HBM3E peak
                "hbm_kernels": hbm}

    out = {
        "metric": "train images/sec", "value": round(B * world * args.steps / dt, 1), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "CVAE train step (encoder+sampler+decoder fwd/bwd, KL+MSE, AdamW), %dx%dx3, %d classes, "
                               "z_dim %d, %d down/up stages, per-GPU batch %d, random-init weights" % (P, P, y, z, n_stage, B),
                   "name": args.config, "global_batch": B * world, "parallelism": "dp%d" % world,
                   "launch": "hip-graph" if use_graph else "eager"},
        "losses_last_step": {k: round(v, 6) for k, v in losses.items()},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(P, y, z, n_stage)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
