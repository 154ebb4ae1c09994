"""Trainer drop-in and inference path on the GPU: eval-mode forward against the fixture recorded
from the reference (models/cvae.py:212 after three optimiser steps), a short training run on the
synthetic stand-in, checkpoint round trip with the reference's file names and keys."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import Case
from oracle.fixtures import compare

pytestmark = pytest.mark.gpu


def _after3_state(c):
    st = OrderedDict()
    for k in c.npz.files:
        if k.startswith("after3/"):
            st[k[len("after3/"):]] = torch.from_numpy(c.npz[k])
    return st


@pytest.mark.parametrize("name", ["tiny_p16_ngf8", "ragged_p32_ngf8"])
def test_eval_forward_matches_reference(name):
    from cvae_gan_amd.engine import Engine
    c = Case(name)
    st = _after3_state(c)
    assert len(st) > 0
    eng = Engine(c.P, c.y, c.z, c.B, dtype=torch.float32, ngf=c.ngf, n_stage=c.n_stage, state=st)
    eng.training = False
    x, labels, _, _ = c.inputs(c.n_steps - 1)
    eps = torch.from_numpy(c.npz["eval/eps"])
    out = eng.reconstruct(x.cuda(), labels.int().cuda(), eps.cuda())
    ok, msg = compare(c.npz, "eval/full", out.cpu().numpy(), 1e-3, 3e-4)
    assert ok, msg
    # decoder only, eval mode (cvae.py:171)
    dec = eng.decode(torch.from_numpy(c.npz["eval/z"]).cuda(), torch.from_numpy(c.npz["eval/y"]).cuda())
    ok, msg = compare(c.npz, "eval/dec", dec.cpu().numpy(), 1e-3, 3e-4)
    assert ok, msg


def test_single_image_batch():
    """One sample: train mode raises what nn.BatchNorm1d raises (models/networks.py:188); eval mode
    (running statistics) gives the same row as the fixture's batch does."""
    from cvae_gan_amd.engine import Engine
    c = Case("tiny_p16_ngf8")
    st = _after3_state(c)
    x, labels, _, _ = c.inputs(c.n_steps - 1)
    eps = torch.from_numpy(c.npz["eval/eps"])
    eng = Engine(c.P, c.y, c.z, 1, dtype=torch.float32, ngf=c.ngf, state=st)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        eng.forward_backward(x[:1].cuda(), labels[:1].int().cuda(), eps[:1].cuda(), 1.0)
    eng.training = False
    out = eng.reconstruct(x[:1].cuda(), labels[:1].int().cuda(), eps[:1].cuda()).cpu().numpy()
    ref = c.npz["eval/full"][:1]
    assert np.abs(out - ref).max() <= 1e-3 * np.abs(ref).max(), np.abs(out - ref).max()


def test_label_out_of_range():
    """The data path rejects a class index the one-hot cannot hold (IndexError, as the reference's
    table lookup at models/cvae.py:129).  The engine, which only sees device labels, keeps indices
    inside its tables while the step runs and raises at the next read_losses()."""
    from cvae_gan_amd.datasets import DevicePrefetcher
    from cvae_gan_amd.engine import Engine
    imgs = torch.zeros(4, 16, 16, 3, dtype=torch.uint8)
    loader = [(imgs, torch.tensor([0, 1, 7, 2]))]
    with pytest.raises(IndexError, match="out of bounds"):
        list(DevicePrefetcher(loader, 4, 16, "cuda", y_dim=4))
    assert len(list(DevicePrefetcher(loader, 4, 16, "cuda", y_dim=8))) == 1
    eng = Engine(16, 4, 8, 4, dtype=torch.float32, ngf=8, seed=0)
    x = torch.rand(4, 3, 16, 16, device="cuda") * 2 - 1
    eps = torch.randn(4, 8, device="cuda")
    eng.random_mask(seed=1)
    eng.forward_backward(x, torch.tensor([0, 1, 7, -3], dtype=torch.int32, device="cuda"), eps, 4.0)
    with pytest.raises(ValueError, match="class label outside"):
        eng.read_losses()
    a = eng.losses[:2].tolist()
    eng.random_mask(seed=1)
    eng.forward_backward(x, torch.tensor([0, 1, 3, 0], dtype=torch.int32, device="cuda"), eps, 4.0)
    b = eng.read_losses()                                 # the flag was cleared by the raise
    assert [b["KL"], b["LL"]] == a                        # the bad step ran as 7 -> 3, -3 -> 0: nothing left its buffers


class Args:
    pass


def _args(tmp_path, **kw):
    a = Args()
    a.epoch, a.batch_size, a.save_dir, a.result_dir = 1, 16, str(tmp_path / "ckpt"), str(tmp_path / "res")
    a.dataset, a.data_dir, a.data_size = "Synth", str(tmp_path / "nodata"), 64
    a.device, a.y_dim, a.z_dim, a.pix_dim, a.lrG = "cuda", 4, 8, 32, 1e-3
    a.worker, a.resume, a.testmode, a.pkl, a.dtype = 0, False, False, "", "bf16"
    a.log_every = 1
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_trainer_runs_saves_and_reloads(tmp_path):
    from cvae_gan_amd.cvae import CVAE
    torch.manual_seed(0)
    np.random.seed(0)
    m = CVAE(_args(tmp_path, epoch=4))
    assert m.synthetic
    m.train()
    ll = m.train_hist["LL_loss"]
    assert len(ll) == 16 and all(np.isfinite(ll))
    assert np.mean(ll[-4:]) < np.mean(ll[:4])              # it learns the blobs
    d = os.path.join(m.save_dir, m.model_dir)
    enc = torch.load(os.path.join(d, "Synth_encoder.pkl"), weights_only=True)
    dec = torch.load(os.path.join(d, "Synth_decoder.pkl"), weights_only=True)
    assert len(enc) == 68 and len(dec) == 14               # the reference's checkpoint entries
    assert tuple(enc["encoder.conv.1.weight"].shape) == (64, 3 + 4, 7, 7)
    assert tuple(dec["decoder.conv.5.weight"].shape) == (3, 64, 7, 7)
    assert os.path.exists(os.path.join(d, "Synth_history.pkl"))
    pngs = [f for f in os.listdir(os.path.join(m.result_dir, m.model_dir)) if f.endswith(".png")]
    assert len(pngs) == 4
    # a second trainer resumes from the files and holds the same parameters; test() writes its grid
    m2 = CVAE(_args(tmp_path, resume=True))
    for k, v in m.engine.state_dict("encoder").items():
        assert torch.equal(v, m2.engine.state_dict("encoder")[k]), k
    path = m2.test()
    assert os.path.exists(path)
    grid = m2.sample_grid()
    assert grid.shape == (10 * 4, 32, 32, 3) and np.isfinite(grid).all() and np.abs(grid).max() <= 1.0


def test_trainer_graph_replay_equals_eager_launches(tmp_path):
    """The single-rank trainer captures its step into a HIP graph at the third call: the same kernels in the
    same order on the same draws (device-side dropout counter, AdamW step count and bias corrections advance
    under replay), so after eight steps every parameter, BatchNorm buffer and logged loss is bit-identical to a
    trainer that launches eagerly."""
    from cvae_gan_amd.cvae import CVAE
    runs = []
    for graph in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        m = CVAE(_args(tmp_path / ("g%d" % graph), epoch=2, graph=graph))
        m.train()
        assert (m._graph is not None) == graph
        sd = dict(m.engine.state_dict("encoder"))
        sd.update(m.engine.state_dict("decoder"))
        runs.append((sd, list(m.train_hist["LL_loss"]), list(m.train_hist["KL_loss"])))
    (a, lla, kla), (b, llb, klb) = runs
    assert len(lla) == 8 and lla == llb and kla == klb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_trainer_graph_capture_with_loader_workers(tmp_path):
    """train.sh passes --worker 48: the DataLoader's worker processes and its pin-memory thread (which allocates
    pinned host memory through the HIP runtime) are alive while the third step is captured.  The capture runs in
    thread-local mode, so it succeeds, and the run equals eager launches bit for bit."""
    from cvae_gan_amd.cvae import CVAE
    runs = []
    for graph in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        m = CVAE(_args(tmp_path / ("w%d" % graph), epoch=2, graph=graph, worker=2))
        m.train()
        assert (m._graph is not None) == graph
        sd = dict(m.engine.state_dict("encoder"))
        sd.update(m.engine.state_dict("decoder"))
        runs.append((sd, list(m.train_hist["LL_loss"])))
    (a, lla), (b, llb) = runs
    assert len(lla) == 8 and lla == llb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_trainer_falls_back_to_eager_when_capture_fails(tmp_path, monkeypatch):
    """A failed capture must not end the run: the trainer reports it and goes on launching eagerly."""
    from cvae_gan_amd import ddp
    from cvae_gan_amd.cvae import CVAE

    def broken(self, *a, **k):
        raise RuntimeError("capture refused (test)")

    monkeypatch.setattr(ddp.GraphedStep, "capture", broken)
    torch.manual_seed(0)
    np.random.seed(0)
    m = CVAE(_args(tmp_path, epoch=2))
    m.train()
    assert m._graph is None and m._use_graph is False
    assert len(m.train_hist["LL_loss"]) == 8 and all(np.isfinite(m.train_hist["LL_loss"]))


def test_device_prefetcher_matches_host_pipeline(tmp_path):
    """uint8 batches normalised on the GPU (cvg_image_u8_to_f32 behind pinned double buffers) equal
    ToTensor + Normalize(.5, .5) of the same bytes (reference datasets.py:14-18), batch after batch."""
    from torch.utils.data import DataLoader
    from cvae_gan_amd.datasets import SyntheticBodyMaps, DevicePrefetcher
    ds = SyntheticBodyMaps(40, 16, 5, raw=True)
    loader = DataLoader(ds, batch_size=8, shuffle=False, drop_last=True)
    got = [(x.clone(), l.clone()) for x, l in DevicePrefetcher(loader, 8, 16, "cuda")]
    assert len(got) == 5
    for k, (x, l) in enumerate(got):
        u8 = torch.stack([ds[8 * k + i][0] for i in range(8)])                 # [8,16,16,3]
        ref = ((u8.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)
        assert torch.allclose(x.cpu(), ref, atol=1e-6)
        assert l.cpu().tolist() == [(8 * k + i) % 5 for i in range(8)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_eval_conv_epilogue_batchnorm_equals_separate_pass(dtype, tol, monkeypatch):
    """Inference folds running-statistics BatchNorm + LeakyReLU (+ the skip tensor) into the
    convolution's epilogue (cvg_conv_bn_act_fwd).  Same network, same inputs, with the separate
    normalisation pass forced instead: f32 results agree to rounding, bf16 to one rounding of the
    intermediate the fused form no longer stores."""
    from cvae_gan_amd import ops
    from cvae_gan_amd.engine import Engine
    P, y, z, B = 32, 10, 32, 16
    eng = Engine(P, y, z, B, dtype=dtype, seed=9)
    g = torch.Generator(device="cuda").manual_seed(1)
    for k, buf in eng.buffers.items():          # non-trivial running statistics
        if k.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, device="cuda", generator=g) * 0.1)
        elif k.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, device="cuda", generator=g) + 0.5)
    x = torch.rand(B, 3, P, P, device="cuda", generator=g) * 2 - 1
    labels = torch.randint(0, y, (B,), device="cuda", generator=g, dtype=torch.int32)
    eps = torch.randn(B, z, device="cuda", generator=g)
    eng.training = False
    assert ops.conv_fwd_kernel(eng.enc["blocks"][0]["conv1"].desc).startswith("patch_fwd")
    fused = eng.reconstruct(x, labels, eps).clone()
    h_fused = eng.enc["h"].float().clone()
    monkeypatch.setattr(ops, "conv_fwd_kernel", lambda desc: "separate")
    plain = eng.reconstruct(x, labels, eps).clone()
    h_plain = eng.enc["h"].float().clone()
    torch.cuda.synchronize()
    assert float((h_fused - h_plain).abs().max()) <= tol * max(float(h_plain.abs().max()), 1.0)
    # the decoder holds no BatchNorm; at a random initialisation its tanh is saturated, so the images
    # are compared in f32 only (in bf16 a 1e-2 change of h flips saturated pixels)
    if dtype == torch.float32:
        assert float((fused - plain).abs().max()) <= 1e-3


def test_starter_command_line(tmp_path):
    """`python -m cvae_gan_amd.starter` with the reference's flags in a fresh process (the package is
    imported before torch there: the library must still bind to torch's HIP runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "cvae_gan_amd.starter", "--dataset", "Synth", "--data_dir", str(tmp_path / "none"),
           "--data_size", "64", "--pix_dim", "32", "--y_dim", "4", "--z_dim", "8", "--batch_size", "16", "--epoch", "1",
           "--max_iters", "3", "--lrG", "1e-3", "--worker", "0", "--gpus", "0",
           "--save_dir", str(tmp_path / "ck"), "--result_dir", str(tmp_path / "res")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Training finish" in r.stdout
    d = os.path.join(str(tmp_path / "ck"), "VAE_data_Synth_pix_32_batch_16_embed_8")
    assert os.path.exists(os.path.join(d, "Synth_encoder.pkl")) and os.path.exists(os.path.join(d, "Synth_decoder.pkl"))


def test_bench_prints_one_json_line():
    """bench.py's contract: stdout is exactly one JSON line with the metric, the roofline of the
    dominant kernel and (single GPU) the CPU baseline fields; everything else goes to stderr."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "bench.py", "--config", "c2", "--batch", "32", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert all(v == v and abs(v) < 1e6 for v in d["losses_last_step"].values())      # finite


def test_schedule_switches_do_not_change_results(tmp_path):
    """The step's schedule (which child of a fork is captured first, the repack beside the stem, the resident
    row-tap kernel) only moves launches between streams and kernels of equal arithmetic: after three graph-replayed
    steps at the reference's width every parameter and BatchNorm buffer is bit-identical whatever the switches say."""
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from cvae_gan_amd import ddp
from cvae_gan_amd.engine import Engine
eng = Engine(32, 10, 32, 16, dtype=torch.bfloat16, seed=3)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(16, 3, 32, 32, device="cuda", generator=g) * 2 - 1
labels = torch.randint(0, 10, (16,), device="cuda", generator=g, dtype=torch.int32)
eps = torch.randn(16, 32, device="cuda", generator=g)
step = ddp.GraphedStep(eng, 16.0, 1e-3).capture(x, labels, eps)
for _ in range(3):
    step.run(x, labels, eps)
torch.cuda.synchronize()
sd = dict(eng.state_dict("encoder"))
sd.update(eng.state_dict("decoder"))
sd["losses"] = eng.losses.cpu().clone()
torch.save(sd, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("default", {}), ("old_order", {"CVG_FORK_SITES": "none"}), ("all_sites", {"CVG_FORK_SITES": "blk,down,low,tail,head,fc"}),
                      ("whole_repack", {"CVG_PACK_WHOLE": "1"}), ("no_lowk", {"CVG_NO_LOWK": "1"}),
                      ("no_walking", {"CVG_NO_RES_ROWS": "1", "CVG_NO_WALK64": "1", "CVG_NO_FWD_WALK": "1"})):
        # (CVG_NO_OVERLAP=1 is not in the list: on one stream every weight-gradient launch is planned for the whole chip,
        # i.e. with another split count, which is a different, equally valid, summation order)
        path = str(tmp_path / (name + ".pt"))
        p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, (name, p.stderr[-2000:])
        runs[name] = torch.load(path, weights_only=True)
    ref = runs.pop("default")
    assert all(torch.isfinite(v.float()).all() for v in ref.values())
    for name, sd in runs.items():
        for k in ref:
            assert torch.equal(ref[k], sd[k]), (name, k)

