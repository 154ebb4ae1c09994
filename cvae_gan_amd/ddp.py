"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce of the
flat gradient arena over RCCL (torch.distributed backend "nccl" on ROCm), issued on
a side stream while backward is still producing the remaining gradients.

Replaces the reference's nn.DataParallel wrapping (models/cvae.py:57-58,
models/networks.py:307): every rank holds a full replica, BatchNorm statistics stay
per replica (DataParallel normalises each chunk separately too), each rank scales its
loss by 1/world so that the SUM all-reduce yields the gradient of the global-batch
mean loss, and rank 0's BatchNorm running statistics are the ones checkpointed.

The arena is laid out in the order backward finishes gradients (decoder tail ...
encoder stem), so every stage of backward completes one contiguous range.
"""
import torch
import torch.distributed as dist

# backward stages in completion order and the parameter-key prefixes they finish
STAGES = ["dec", "enc.head", "enc.blk3", "enc.blk2", "enc.blk1", "enc.blk0", "enc.stem"]


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
    """Bucketed all-reduce of `arena` driven by the engine's grad-ready notifications."""

    def __init__(self, arena, ranges, stages, bucket_elems=4 << 20, group=None):
        self.arena, self.ranges, self.group = arena, ranges, group
        self.stages = [s for s in stages if s in ranges]
        self.bucket_elems = bucket_elems
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = arena.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self._pending = None
        self._works = []
        self.launched = []          # (lo, hi) of every bucket of the current step, for tests

    def begin_step(self):
        self._pending, self._works, self.launched = None, [], []

    def on_ready(self, stage):
        lo, hi = self.ranges[stage]
        if self._pending is None:
            self._pending = (lo, hi)
        else:
            assert self._pending[1] == lo, "stages must arrive in arena order"
            self._pending = (self._pending[0], hi)
        last = stage == self.stages[-1]
        if last or self._pending[1] - self._pending[0] >= self.bucket_elems:
            self._flush()

    def _flush(self):
        lo, hi = self._pending
        self._pending = None
        self.launched.append((lo, hi))
        if self.world == 1:
            return
        chunk = self.arena[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()                                   # gradients of this bucket are enqueued up to here
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every bucket (call before the optimiser)."""
        assert self._pending is None, "a stage never reported its gradients"
        for w in self._works:
            w.wait()
        self._works = []


def broadcast_state(engine, src=0, group=None):
    """Start every replica from rank `src`'s parameters and BatchNorm buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.broadcast(engine.p_arena, src=src, group=group)
    for t in engine.buffers.values():
        dist.broadcast(t, src=src, group=group)
    engine.params_changed()
