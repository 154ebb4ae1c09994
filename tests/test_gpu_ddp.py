"""Data-parallel training with the real engine: two processes (both on GPU 0, gloo transport) run
Engine.forward_backward on their half of a fixture batch with cvae_gan_amd.ddp.GradSync hooked to
Engine.grad_ready, the path the trainer and bench.py use.  The exchanged gradient must be the one of
the reference's nn.DataParallel step on the whole batch (models/cvae.py:57-59: per-chunk BatchNorm,
sampler and losses on the gathered batch), which oracle.dp_forward_backward restates; f32, 1e-3 of
each tensor's maximum.  After the optimiser both replicas hold bit-identical parameters."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import Case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(cmd_tail, world, extra_env=None, timeout=170):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable] + cmd_tail, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    return outs


def test_two_ranks_equal_dataparallel_step(tmp_path):
    from oracle import ref_cpu as O
    from cvae_gan_amd.engine import arena_layout
    world = 2
    _run_ranks([os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), str(tmp_path)], world)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    c = Case("ragged_p32_ngf8")
    lay = arena_layout(c.P, c.y, c.z, c.n_stage, c.ngf)
    e, d = c.init_state()
    x, labels, eps, mask = c.inputs(0)
    ref_losses, ref = O.dp_forward_backward(e, d, x, labels, eps, mask, c.y, c.bs_flag, world)
    zero_grad = {O.EP + "conv.1.bias", O.EP + "fc.0.bias"}
    for k in lay["order"]:
        if k in zero_grad:
            continue
        off = lay["offsets"][k]
        g = r0["grads"][off:off + ref[k].numel()].view(ref[k].shape)
        den = float(ref[k].abs().max()) or 1.0
        assert float((g - ref[k]).abs().max()) / den < 1e-3, k
    assert torch.equal(r0["grads"], r1["grads"])
    kl, ll = r0["losses"].tolist()
    assert abs(kl - float(ref_losses["KL"])) <= 1e-4 * abs(float(ref_losses["KL"]))
    assert abs(ll - float(ref_losses["LL"])) <= 1e-4 * abs(float(ref_losses["LL"]))
    # three exchanges, in arena order, covering the arena exactly once
    b = r0["launched"]
    assert len(b) == 3 and b[0][0] == 0 and b[-1][1] == lay["numel"]
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    # replicas stay in step; BatchNorm running statistics are per replica (each saw its own rows)
    assert torch.equal(r0["params"], r1["params"])
    assert not torch.equal(r0["running_mean0"], r1["running_mean0"])


def test_trainer_two_ranks_share_one_gpu(tmp_path):
    """The command line under a launcher: two ranks, --gpus 0 shared (CVG_DDP_SHARE_GPU), gloo transport.
    Rank 0 alone prints and writes; the global batch of 16 is split 8 + 8."""
    out = _run_ranks(["-m", "cvae_gan_amd.starter", "--dataset", "Synth", "--data_dir", str(tmp_path / "none"),
                      "--data_size", "64", "--pix_dim", "32", "--y_dim", "4", "--z_dim", "8", "--batch_size", "16",
                      "--epoch", "2", "--max_iters", "6", "--lrG", "1e-3", "--worker", "0", "--gpus", "0",
                      "--save_dir", str(tmp_path / "ck"), "--result_dir", str(tmp_path / "res")], 2,
                     extra_env={"CVG_DDP_BACKEND": "gloo", "CVG_DDP_SHARE_GPU": "1"})
    # (steps 3..6 are replayed from the four graphs of ddp.GraphedStep with the exchanges between them)
    assert "Training finish" in out[0] and "Training finish" not in out[1]
    assert "capture of the train step failed" not in out[0]
    d = os.path.join(str(tmp_path / "ck"), "VAE_data_Synth_pix_32_batch_16_embed_8")
    enc = torch.load(os.path.join(d, "Synth_encoder.pkl"), weights_only=True)
    assert len(enc) == 68 and all(torch.isfinite(v.float()).all() for v in enc.values())


def test_gan_trainer_two_ranks(tmp_path):
    """--gan_type CVAEGAN under two ranks: the discriminator's, the classifier's and the CVAE's gradient arenas
    are summed over the ranks, each as soon as it is final; two eager steps, then three from the four graphs of
    gan.GraphedGanStep with the exchanges between them; rank 0 writes all four checkpoints."""
    out = _run_ranks(["-m", "cvae_gan_amd.starter", "--dataset", "Synth", "--data_dir", str(tmp_path / "none"),
                      "--data_size", "128", "--pix_dim", "32", "--y_dim", "4", "--z_dim", "8", "--batch_size", "16",
                      "--epoch", "1", "--max_iters", "5", "--lrG", "1e-3", "--worker", "0", "--gpus", "0",
                      "--gan_type", "CVAEGAN", "--save_dir", str(tmp_path / "ck"), "--result_dir", str(tmp_path / "res")], 2,
                     extra_env={"CVG_DDP_BACKEND": "gloo", "CVG_DDP_SHARE_GPU": "1"})
    assert "Training finish" in out[0]
    assert "capture of the adversarial step failed" not in out[0]
    d = os.path.join(str(tmp_path / "ck"), "VAE_data_Synth_pix_32_batch_16_embed_8")
    for name in ("encoder", "decoder", "discriminator", "classifier"):
        sd = torch.load(os.path.join(d, "Synth_%s.pkl" % name), weights_only=True)
        assert all(torch.isfinite(v.float()).all() for v in sd.values()), name


def test_bench_two_ranks_one_json_line():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), both ranks on
    this GPU with the gloo transport: the four-graph schedule with the exchanges between the graphs runs, rank 0
    prints exactly one JSON line for the two-rank job, and the losses are finite."""
    import json
    env = dict(os.environ, CVG_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c2",
           "--batch", "32", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64
    assert d["config"]["launch"] == "hip-graph" and d["scaling"] == "weak"
    assert all(v == v and abs(v) < 1e6 for v in d["losses_last_step"].values())
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
