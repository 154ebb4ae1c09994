"""Data-parallel gradient exchange on the CPU with gloo, world size 2.

Each rank computes the gradients of ITS half of the batch with the oracle (loss divided by the
global batch_size flag and scaled by 1/world, as the HIP engine does), writes them into a flat
arena laid out like the engine's, and runs cvae_gan_amd.ddp.GradSync with the engine's
grad-ready stage notifications.  The all-reduced arena must equal the gradient of the
reference's nn.DataParallel step on the full batch (per-chunk BatchNorm, losses on the
gathered batch), which oracle.dp_forward_backward restates."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ref_cpu as O
    from helpers import Case
    from cvae_gan_amd.engine import arena_layout
    from cvae_gan_amd import ddp

    c = Case("ragged_p32_ngf8")                 # B = 6 -> 3 images per rank
    lay = arena_layout(c.P, c.y, c.z, 4, c.ngf)
    arena = torch.zeros(lay["numel"])
    e, d = c.init_state()
    x, labels, eps, mask = c.inputs(0)
    n = c.B // world
    sl = slice(rank * n, (rank + 1) * n)
    # local step: mean over the LOCAL chunk, divided by the global batch_size flag, times 1/world
    st = O.TrainState(e, d)
    params = {}
    for sd in (st.enc, st.dec):
        for k, t in sd.items():
            if not O.is_buffer(k):
                sd[k] = t.detach().clone().requires_grad_(True)
                params[k] = sd[k]
    dec, mu, sigma, _ = O.cvae_forward(st.enc, st.dec, x[sl], labels[sl], eps[sl], mask[sl], c.y, True, None)
    vae, kl, ll = O.losses(dec, x[sl], mu, sigma, c.bs_flag)
    grads = dict(zip(params.keys(), torch.autograd.grad(vae / world, list(params.values()))))

    sync = ddp.GradSync(arena, ddp.stage_ranges(lay["order"], lay["offsets"], lay["numel"]), ddp.STAGES,
                        bucket_elems=20000)
    sync.begin_step()
    # backward order of the engine: decoder, encoder head, blocks 3..0, stem
    for stage in ddp.STAGES:
        for k in lay["order"]:
            if ddp.stage_of(k) == stage:
                off = lay["offsets"][k]
                arena[off:off + grads[k].numel()] = grads[k].reshape(-1)
        sync.on_ready(stage)
    sync.finish()
    if rank == 0:
        torch.save({"arena": arena, "buckets": sync.launched}, os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_dataparallel_step():
    from oracle import ref_cpu as O
    from helpers import Case
    from cvae_gan_amd.engine import arena_layout
    world = 2
    with tempfile.TemporaryDirectory() as td:
        port = 29500 + os.getpid() % 2000
        mp.spawn(_worker, args=(world, port, td), nprocs=world, join=True)
        got = torch.load(os.path.join(td, "r0.pt"))
    c = Case("ragged_p32_ngf8")
    lay = arena_layout(c.P, c.y, c.z, 4, c.ngf)
    e, d = c.init_state()
    x, labels, eps, mask = c.inputs(0)
    _, ref = O.dp_forward_backward(e, d, x, labels, eps, mask, c.y, c.bs_flag, world)
    zero_grad = {"encoder.conv.1.bias", "encoder.fc.0.bias"}   # exactly 0 up to rounding noise
    for k in lay["order"]:
        if k in zero_grad:
            continue
        off = lay["offsets"][k]
        g = got["arena"][off:off + ref[k].numel()].view(ref[k].shape)
        den = float(ref[k].abs().max()) or 1.0
        assert float((g - ref[k]).abs().max()) / den < 2e-4, k
    # the exchange really was bucketed, in arena order, covering the arena exactly once
    b = got["buckets"]
    assert len(b) >= 2 and b[0][0] == 0 and b[-1][1] == lay["numel"]
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))


def test_stage_ranges_are_contiguous_and_complete():
    from cvae_gan_amd.engine import arena_layout
    from cvae_gan_amd import ddp
    lay = arena_layout(64, 100, 32)
    r = ddp.stage_ranges(lay["order"], lay["offsets"], lay["numel"])
    assert [s for s in ddp.STAGES if s in r] == ddp.STAGES
    pos = 0
    for s in ddp.STAGES:
        assert r[s][0] == pos
        pos = r[s][1]
    assert pos == lay["numel"] == sum((int(torch.Size(lay["shapes"][k]).numel()) + 3) // 4 * 4 for k in lay["order"])
    # parameter count of the reference at 64x64 / 100 classes / z 32 (19,880,467: 11,558,720 encoder + 8,321,747 decoder)
    assert sum(int(torch.Size(lay["shapes"][k]).numel()) for k in lay["order"]) == 19880467


def test_exchange_groups_and_batch_shards():
    from cvae_gan_amd import ddp
    for n_stage in (4, 5):
        g = ddp.default_groups(n_stage)
        assert [s for grp in g for s in grp] == ddp.stages(n_stage) and len(g) == 3
    assert ddp.default_groups(4)[1] == ["enc.head", "enc.blk3", "enc.blk2"]
    # rank r takes rows [r*B/N, (r+1)*B/N) of every global batch; the ranks' shards tile the batches of
    # one shared permutation, and the permutation changes with the epoch
    world, B, n = 4, 8, 37
    samplers = [ddp.ShardedBatchSampler(n, B, r, world, seed=3) for r in range(world)]
    epochs = []
    for _ in range(2):
        per_rank = [list(s) for s in samplers]
        assert all(len(p) == n // B for p in per_rank)
        batches = [sum((per_rank[r][b] for r in range(world)), []) for b in range(n // B)]
        flat = sum(batches, [])
        assert len(set(flat)) == len(flat) == (n // B) * B and all(len(p[0]) == B // world for p in per_rank)
        epochs.append(flat)
    assert epochs[0] != epochs[1]
    g = torch.Generator().manual_seed(3 * 1000003)
    assert epochs[0] == torch.randperm(n, generator=g).tolist()[:(n // B) * B]
    with pytest.raises(AssertionError):
        ddp.ShardedBatchSampler(n, 10, 0, 4)
