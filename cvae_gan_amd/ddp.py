"""Data-parallel training: one process per GPU, bucketed all-reduce of the flat gradient
arena over RCCL (torch.distributed backend "nccl" on ROCm), issued on a side stream while
backward is still producing the remaining gradients.

Replaces the reference's nn.DataParallel wrapping (models/cvae.py:57-58,
models/networks.py:307): every rank holds a full replica, rank r works on rows
[r*B/N, (r+1)*B/N) of each global batch, BatchNorm statistics stay per replica (DataParallel
normalises each chunk separately too), each rank scales its loss by 1/world so that the SUM
all-reduce yields the gradient of the global-batch mean loss, and rank 0's BatchNorm running
statistics are the ones checkpointed (DataParallel keeps replica 0's).

The arena is laid out in the order backward finishes gradients (decoder tail ... encoder
stem), so every stage of backward completes one contiguous range.  The trainer
(cvae_gan_amd/cvae.py), bench.py and the tests all exchange through GradSync.
"""
import os

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def stages(n_stage=4):
    """Backward stages in completion order (the names Engine.grad_ready reports)."""
    return ["dec", "enc.head"] + ["enc.blk%d" % i for i in range(n_stage - 1, -1, -1)] + ["enc.stem"]


STAGES = stages(4)


def default_groups(n_stage=4):
    """Three exchanges per step: the decoder's range (travels while the encoder's head and deep
    blocks run backward), the head + deep blocks (nearly all encoder parameters; travels while the
    shallow blocks and the stem run, the longest part of the encoder's backward pass), and the
    rest (the only exposed exchange, ~1.5 MB at 64x64).  The split between deep and shallow
    blocks is Engine.enc_split = n_stage // 2."""
    st = stages(n_stage)
    split = n_stage // 2
    upper = ["enc.head"] + ["enc.blk%d" % i for i in range(n_stage - 1, split - 1, -1)]
    lower = ["enc.blk%d" % i for i in range(split - 1, -1, -1)] + ["enc.stem"]
    assert ["dec"] + upper + lower == st
    return [["dec"], upper, lower]


def stage_of(key, n_stage=4):
    if key.startswith("decoder."):
        return "dec"
    if ".fc." in key:
        return "enc.head"
    idx = int(key.split(".conv.")[1].split(".")[0])
    if idx <= 2:
        return "enc.stem"
    return "enc.blk%d" % (idx - 4)


def stage_ranges(arena_order, arena_offsets, arena_numel, n_stage=4):
    """{stage: (lo, hi)} element ranges of the arena; stages are contiguous by construction."""
    out = {}
    for i, k in enumerate(arena_order):
        st = stage_of(k, n_stage)
        lo = arena_offsets[k]
        hi = arena_offsets[arena_order[i + 1]] if i + 1 < len(arena_order) else arena_numel
        if st in out:
            assert out[st][1] == lo, "arena order must follow backward order"
            out[st] = (out[st][0], hi)
        else:
            out[st] = (lo, hi)
    return out


class GradSync:
    """All-reduce (SUM) of `arena` in contiguous ranges, driven by backward's progress.

    groups: lists of stage names; the range of a group is exchanged when its last stage reports.
    groups=None: stages are merged until a range holds at least bucket_elems elements.
    Stages report either through on_ready(stage) (hook it to Engine.grad_ready for eager launches)
    or through group_ready(i) (a driver that replays one captured graph per group).
    With the "nccl" backend the reduction runs on the device on a communication stream; any other
    backend (gloo: rehearsals with several ranks on one GPU, CPU arenas in tests) reduces a host copy."""

    def __init__(self, arena, ranges, stage_names, bucket_elems=4 << 20, group=None, groups=None):
        self.arena, self.ranges, self.group = arena, ranges, group
        self.stages = [s for s in stage_names if s in ranges]
        self.bucket_elems = bucket_elems
        self.groups = groups
        self._flush_after = None
        if groups is not None:
            assert [s for grp in groups for s in grp] == self.stages, "groups must list every stage in backward order"
            self._flush_after = {grp[-1] for grp in groups}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = arena.is_cuda
        self.device_reduce = self.cuda and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.comm_stream = torch.cuda.Stream(device=arena.device) if self.cuda else None
        self._pending = None
        self._works = []
        self._events = []
        self.launched = []          # (lo, hi) of every exchange of the current step, for tests

    @classmethod
    def for_engine(cls, eng, group=None, groups="default", bucket_elems=4 << 20):
        """The exchange plan of an Engine's gradient arena."""
        ranges = stage_ranges(eng.arena_order, eng.arena_offsets, eng.arena_numel, eng.n_stage)
        if groups == "default":
            groups = default_groups(eng.n_stage)
        return cls(eng.g_arena, ranges, stages(eng.n_stage), bucket_elems=bucket_elems, group=group, groups=groups)

    def begin_step(self):
        assert self._pending is None and not self._works, "the previous step was not finished"
        self.launched = []
        self._events = []

    def on_ready(self, stage, events=()):
        """events: HIP events recorded on other streams (the engine's weight-gradient side streams) that
        the stage's gradients also wait for; the exchange waits for them on the communication stream, so
        the compute stream that reports the stage never has to."""
        self._events.extend(events)
        lo, hi = self.ranges[stage]
        if self._pending is None:
            self._pending = (lo, hi)
        else:
            assert self._pending[1] == lo, "stages must arrive in arena order"
            self._pending = (self._pending[0], hi)
        if self._flush_after is not None:
            flush = stage in self._flush_after
        else:
            flush = stage == self.stages[-1] or self._pending[1] - self._pending[0] >= self.bucket_elems
        if flush:
            self._flush()

    def group_ready(self, i):
        """Every stage of groups[i] has been enqueued on the current stream."""
        for stage in self.groups[i]:
            self.on_ready(stage)

    def _flush(self):
        lo, hi = self._pending
        self._pending = None
        self.launched.append((lo, hi))
        if self.world == 1 and not self.device_reduce:
            return
        chunk = self.arena[lo:hi]
        if not self.cuda:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record()                                   # gradients of this range are enqueued up to here
        events, self._events = self._events, []
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            for e in events:                          # ... and on the side streams up to these
                self.comm_stream.wait_event(e)
            if self.device_reduce:
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
            else:
                self.comm_stream.synchronize()
                host = chunk.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                chunk.copy_(host)

    def finish(self):
        """Make the compute stream wait for every exchange (call before the optimiser)."""
        assert self._pending is None, "a stage never reported its gradients"
        for w in self._works:
            w.wait()
        self._works = []
        if self.cuda:
            torch.cuda.current_stream(self.arena.device).wait_stream(self.comm_stream)


class GraphedStep:
    """One train step (dropout draw, forward, both losses, backward, gradient exchange, AdamW, weight repack)
    replayed from HIP graphs on static input buffers: what bench.py times and what the trainer runs after its
    first eager steps, so both exercise one schedule.

    One replica: ONE graph of the whole step.  With a GradSync (N > 1 ranks): FOUR graphs with the gradient
    exchanges between them, each exchange on the communication stream while the next graph runs:
    [forward + decoder backward] -> all-reduce of the decoder's range -> [encoder head + deep blocks backward] ->
    all-reduce of their range -> [shallow blocks + stem backward] -> all-reduce of the rest -> [AdamW].
    Captured in thread-local mode: other threads of the process (a DataLoader's pin-memory thread, the process
    group's watchdog) may call into the HIP runtime while this thread captures."""

    def __init__(self, eng, batch_size, lr, loss_scale=1.0, sync=None):
        self.eng, self.batch_size, self.lr, self.loss_scale, self.sync = eng, float(batch_size), lr, loss_scale, sync
        self.graphs, self.inputs, self.out = None, None, None

    def capture(self, x, labels, eps):
        """x, labels, eps: tensors of the step's shapes and dtypes (their values are not used)."""
        eng, sync = self.eng, self.sync
        self.inputs = (torch.empty_like(x), torch.empty_like(labels), torch.empty_like(eps))
        gx, gl, ge = self.inputs
        eng.grad_ready = None
        eng.pack_weights(force=True)
        torch.cuda.synchronize()
        saved = eng.step_count
        mode = {"capture_error_mode": "thread_local"}
        if sync is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                eng.random_mask()
                eng._packed_version = -1             # the captured step repacks the weights it updated last time
                self.out = eng.forward_backward(gx, gl, ge, self.batch_size, self.loss_scale)
                eng.adamw_step(self.lr)
            self.graphs = [g]
        else:
            assert sync.groups is not None and len(sync.groups) == 3, "the default exchange plan"
            if "CVG_FORK_SITES" not in os.environ:
                # every graph of this schedule ends in a join that waits for the side launches: capturing them first
                # at every fork measured faster here (6.70 against 6.79 ms per c3 step), the opposite of the one-graph step
                eng._fork_sites = set()
            assert sync.groups[1][-1] == "enc.blk%d" % eng.enc_split, "graph boundaries follow the exchange groups"
            gs = [torch.cuda.CUDAGraph() for _ in range(4)]
            mode["pool"] = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(gs[0], **mode):
                eng.random_mask()
                eng._packed_version = -1
                self.out = eng.forward_backward_decoder(gx, gl, ge, self.batch_size, self.loss_scale)
            with torch.cuda.graph(gs[1], **mode):
                eng.enc_backward(eng.enc["dh"], part="upper")
            with torch.cuda.graph(gs[2], **mode):
                eng.enc_backward(eng.enc["dh"], part="lower")
            with torch.cuda.graph(gs[3], **mode):
                eng.adamw_step(self.lr)
            self.graphs = gs
        eng.step_count = saved
        return self

    def run(self, x, labels, eps):
        """Copies the inputs in and enqueues the step; returns the reconstruction (a static buffer)."""
        for dst, src in zip(self.inputs, (x, labels, eps)):
            dst.copy_(src)
        self.replay()
        return self.out

    def replay(self):
        eng, sync = self.eng, self.sync
        if sync is None:
            self.graphs[0].replay()
        else:
            sync.begin_step()
            self.graphs[0].replay()
            sync.group_ready(0)                          # the decoder's range
            self.graphs[1].replay()
            sync.group_ready(1)                          # encoder head + deep blocks
            self.graphs[2].replay()
            sync.group_ready(2)                          # shallow blocks + stem
            sync.finish()
            self.graphs[3].replay()
        eng.step_count += 1
        eng.params_changed()                             # host-side version: engines that share the weights repack


def broadcast_state(engine, src=0, group=None):
    """Start every replica from rank `src`'s parameters and BatchNorm buffers (the reference's
    DataParallel replicates module 0's on every forward)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    device_ok = dist.get_backend(group) == "nccl"

    def bcast(t):
        if device_ok or not t.is_cuda:
            dist.broadcast(t, src=src, group=group)
        else:
            host = t.cpu()
            dist.broadcast(host, src=src, group=group)
            t.copy_(host)

    bcast(engine.p_arena)
    for t in engine.buffers.values():
        bcast(t)
    engine.params_changed()


def allreduce_mean(t, group=None):
    """Mean over the ranks of a small device tensor (the two loss scalars, for logging)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        out = t.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    else:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        out = host.to(t.device)
    return out / dist.get_world_size(group)


def allreduce_sum_(t, group=None):
    """In-place SUM over the ranks of a device tensor on the current stream (a critic's gradient arena)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    return t


class ArenaExchange:
    """SUM all-reduce of whole gradient arenas on a communication stream: start(arena) when everything that writes
    the arena is enqueued on the current stream, finish() before the optimiser.  The adversarial step has three
    arenas that become final one after the other (discriminator, classifier, CVAE); each travels while the rest
    of the step runs.  "nccl": the reduction runs on the device; any other backend reduces a host copy (rehearsals)."""

    def __init__(self, device, group=None):
        self.group = group
        self.stream = torch.cuda.Stream(device=device)

    def start(self, arena):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            allreduce_sum_(arena, self.group)

    def finish(self):
        torch.cuda.current_stream().wait_stream(self.stream)


def env_world():
    """(rank, local_rank, world_size) from the launcher's environment (torchrun); (0, 0, 1) without one."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(device, backend=None):
    """Join the job's process group; call after torch.cuda.set_device(device) and before any
    collective.  backend: "nccl" (RCCL, the product path) unless CVG_DDP_BACKEND names another
    (gloo: several ranks sharing one GPU in rehearsals and tests)."""
    if dist.is_initialized():
        return
    backend = backend or os.environ.get("CVG_DDP_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=device)
    else:
        dist.init_process_group(backend=backend)


class ShardedBatchSampler(Sampler):
    """Rank r's rows [r*B/N, (r+1)*B/N) of every global batch of B indices (drop_last, one seeded
    permutation per epoch shared by all ranks): the rows nn.DataParallel's scatter hands replica r
    (models/cvae.py:57-58) of the batches DataLoader(shuffle=True, drop_last=True) forms (:47-54)."""

    def __init__(self, n_items, global_batch, rank, world, seed=0, shuffle=True):
        assert global_batch % world == 0, "--batch_size %d does not split over %d ranks" % (global_batch, world)
        self.n, self.B, self.rank, self.world, self.seed, self.shuffle = n_items, global_batch, rank, world, seed, shuffle
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.n // self.B

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        per = self.B // self.world
        for b in range(self.n // self.B):
            lo = b * self.B + self.rank * per
            yield perm[lo:lo + per]
        self.epoch += 1
