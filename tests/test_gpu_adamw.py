"""torch.optim.AdamW(lr, betas=(0.5, 0.999), eps=1e-8, weight_decay=0.01) of models/cvae.py:67, :157 on the
device: the kernels alone against the oracle's update rule, and the engine's whole step (parameters and
BatchNorm buffers after 1 and 3 steps) against the values recorded from the reference's own optimiser,
launched eagerly and replayed from a captured HIP graph (the device-side step counter must advance the
bias corrections on every replay)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import Case
from oracle import ref_cpu as O
from oracle.fixtures import compare, compare_update

pytestmark = pytest.mark.gpu

ZERO_GRAD = {O.EP + "conv.1.bias", O.EP + "fc.0.bias"}
AFTER_ZERO_GRAD = {O.EP + "conv.2.running_mean", O.EP + "fc.1.running_mean"}


@pytest.mark.parametrize("dev_counter", [False, True])
def test_adamw_kernels_match_update_rule(dev_counter):
    from cvae_gan_amd import ops
    n = 100003                                   # not a multiple of 4: the tail path runs too
    g = torch.Generator().manual_seed(11)
    p = torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.cuda(), m.cuda(), v.cuda()
    step_dev = torch.zeros(1, dtype=torch.long, device="cuda")
    bc = torch.zeros(2, device="cuda")
    lr, wd = 3e-3, 1e-2
    gmax = torch.zeros(n, dtype=torch.float64)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 2, (n,), generator=g).float())
        p0 = p
        gmax = torch.maximum(gmax, grad.abs().double())
        p, m, v = O.adamw_update(p.double(), grad.double(), m.double(), v.double(), step, lr)
        if dev_counter:
            ops.adamw_dev(pd, grad.cuda(), md, vd, n, lr, 0.5, 0.999, 1e-8, wd, step_dev, bc)
        else:
            ops.adamw(pd, grad.cuda(), md, vd, n, lr, 0.5, 0.999, 1e-8, wd, step)
        torch.cuda.synchronize()
        # 1e-6 relative to the result, plus a few f32 roundings of the largest term that ever entered it
        # (a moment may be much smaller than the gradients it averages, and keeps their rounding)
        scale = {"p": p0.abs().double() + lr, "m": gmax, "v": gmax ** 2}
        for name, got, ref in (("p", pd, p), ("m", md, m), ("v", vd, v)):
            err = (got.cpu().double() - ref).abs()
            ok = err <= 1e-6 * ref.abs() + 3e-7 * scale[name]
            assert bool(ok.all()), (step, name, float((err / ref.abs().clamp_min(1e-30)).max()))
    if dev_counter:
        assert int(step_dev) == 3


def _check_against_fixture(c, eng, tag, step, init):
    bad = []
    for which in ("encoder", "decoder"):
        for k, v in eng.state_dict(which).items():
            if k.endswith("num_batches_tracked"):
                if int(v) != step:
                    bad.append("%s = %d, expected %d" % (k, int(v), step))
                continue
            if k in ZERO_GRAD:
                continue
            if O.is_buffer(k):
                drift = 2.2 * c.lr * step if k in AFTER_ZERO_GRAD else 0.0
                ok, msg = compare(c.npz, tag + k, v.numpy(), 2e-3, 2e-5 + drift)
            else:
                # (after the first step the reference's own first gradient tells which elements' directions are
                # rounding noise: see compare_update)
                ok, msg = compare_update(c.npz, tag + k, v.numpy(), init[k].numpy(), c.lr, step,
                                         frac_ok=0.97 if step == 1 else 0.90,
                                         grad_key="s0/grad/" + k if step == 1 else None)
            if not ok:
                bad.append(msg)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["tiny_p16_ngf8", "ragged_p32_ngf8"])
@pytest.mark.parametrize("graph", [False, True])
def test_engine_updates_match_reference_after_1_and_3_steps(name, graph):
    from cvae_gan_amd.engine import Engine
    c = Case(name)
    e, d = c.init_state()
    init = OrderedDict(e)
    init.update(d)
    eng = Engine(c.P, c.y, c.z, c.B, dtype=torch.float32, ngf=c.ngf, n_stage=c.n_stage, state=init)
    gx = torch.zeros(c.B, 3, c.P, c.P, device="cuda")
    gl = torch.zeros(c.B, dtype=torch.int32, device="cuda")
    ge = torch.zeros(c.B, c.z, device="cuda")
    replay = None
    if graph:
        # one eager step first (streams, kernel attributes and workspaces come into being outside the
        # capture), then everything it changed is put back: weights, buffers, moments, step counter
        x, labels, eps, mask = c.inputs(0)
        eng.set_mask(mask)
        eng.train_step(x.cuda(), labels.int().cuda(), eps.cuda(), float(c.bs_flag), c.lr)
        torch.cuda.synchronize()
        eng.load_state(init)
        eng.m_arena.zero_()
        eng.v_arena.zero_()
        eng.step_dev.zero_()
        eng.pack_weights(force=True)
        torch.cuda.synchronize()
        replay = torch.cuda.CUDAGraph()
        with torch.cuda.graph(replay):
            eng._packed_version = -1
            eng.train_step(gx, gl, ge, float(c.bs_flag), c.lr)
    for step in range(c.n_steps):
        x, labels, eps, mask = c.inputs(step)
        eng.set_mask(mask)
        gx.copy_(x)
        gl.copy_(labels.int())
        ge.copy_(eps)
        if graph:
            replay.replay()          # every launch of every step, AdamW's included, comes from the graph
        else:
            eng.train_step(gx, gl, ge, float(c.bs_flag), c.lr)
        torch.cuda.synchronize()
        got = eng.read_losses()
        for k in ("VAE", "KL", "LL"):
            refv = float(c.npz["s%d/%s" % (step, k)])
            tol = 2e-5 if step == 0 else 2e-3
            assert abs(got[k] - refv) <= tol * abs(refv), (step, k, got[k], refv)
        if step in (0, c.n_steps - 1):
            _check_against_fixture(c, eng, "after%d/" % (step + 1), step + 1, init)
    assert int(eng.step_dev) == c.n_steps
