"""One rank of tests/test_gpu_ddp.py (started as a plain subprocess with RANK / WORLD_SIZE / MASTER_* set).

Both ranks share GPU 0 and exchange over gloo (host-staged): the control flow, stream ordering and
arena ranges are those of the RCCL path, only the transport differs."""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    from cvae_gan_amd import ddp
    from cvae_gan_amd.engine import Engine
    from helpers import Case
    rank, _, world = ddp.env_world()
    torch.cuda.set_device(0)
    ddp.init_process_group(torch.device("cuda", 0), backend="gloo")
    c = Case("ragged_p32_ngf8")                 # B = 6 -> 3 images per rank
    e, d = c.init_state()
    state = OrderedDict(e)
    state.update(d)
    n = c.B // world
    sl = slice(rank * n, (rank + 1) * n)
    eng = Engine(c.P, c.y, c.z, n, dtype=torch.float32, ngf=c.ngf, n_stage=c.n_stage, state=state if rank == 0 else None,
                 seed=100 + rank)               # rank 1 starts from other weights: broadcast_state must fix that
    ddp.broadcast_state(eng)
    sync = ddp.GradSync.for_engine(eng)
    eng.grad_ready = sync.on_ready
    result = {}
    for step in range(2):
        x, labels, eps, mask = c.inputs(step)
        eng.set_mask(mask[sl])
        sync.begin_step()
        eng.forward_backward(x[sl].cuda(), labels[sl].int().cuda(), eps[sl].cuda(), float(c.bs_flag), loss_scale=1.0 / world)
        sync.finish()
        torch.cuda.synchronize()
        if step == 0:
            result["grads"] = eng.g_arena.cpu().clone()
            result["launched"] = list(sync.launched)
            result["losses"] = ddp.allreduce_mean(eng.losses[:2]).cpu()
        eng.adamw_step(c.lr)
    torch.cuda.synchronize()
    result["params"] = eng.p_arena.cpu().clone()
    result["running_mean0"] = eng.buffers["encoder.conv.2.running_mean"].cpu().clone()
    torch.save(result, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
