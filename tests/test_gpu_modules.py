"""The drop-in module surface (encoder / decoder / CVAE_T / latent_loss) driven exactly like
the reference's training loop (models/cvae.py:125-157): torch losses on the module outputs,
loss.backward(), torch.optim.AdamW -- against the fixture recorded from the reference."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import Case
from oracle.fixtures import compare

pytestmark = pytest.mark.gpu


class Args:
    pass


def make(case, dtype="f32"):
    from cvae_gan_amd import networks as N
    a = Args()
    a.z_dim, a.y_dim, a.pix_dim, a.device, a.dtype, a.ngf = case.z, case.y, case.P, torch.device("cuda"), dtype, case.ngf
    enc, dec = N.encoder(a), N.decoder(a)
    e, d = case.init_state()
    enc.load_state_dict(e)
    dec.load_state_dict(d)
    model = N.CVAE_T(a, enc, dec)
    return N, enc, dec, model


def test_state_dict_keys_match_reference_checkpoints():
    c = Case("tiny_p16_ngf8")
    N, enc, dec, model = make(c)
    e, d = c.init_state()
    assert list(enc.state_dict().keys()) == list(e.keys())
    assert list(dec.state_dict().keys()) == list(d.keys())
    for k, v in enc.state_dict().items():
        assert tuple(v.shape) == tuple(e[k].shape), k


@pytest.mark.parametrize("name", ["tiny_p16_ngf8", "ragged_p32_ngf8"])
def test_reference_style_training_loop(name):
    c = Case(name)
    N, enc, dec, model = make(c)
    opt = torch.optim.AdamW(model.parameters(), lr=c.lr, betas=(0.5, 0.999), eps=1e-8)   # cvae.py:67
    mse = torch.nn.MSELoss()
    model.train()
    x, labels, eps, mask = c.inputs(0)
    planes = torch.zeros(c.B, c.y, c.P, c.P)
    planes[torch.arange(c.B), labels] = 1
    yvec = torch.zeros(c.B, c.y)
    yvec[torch.arange(c.B), labels] = 1
    x, planes, yvec = x.cuda(), planes.cuda(), yvec.cuda()

    # the reference draws eps with numpy inside _sample_latent and the dropout mask with torch's
    # RNG; feed the fixture's values through the same two doors
    orig_normal = np.random.normal
    np.random.normal = lambda *a, **k: eps.double().numpy()
    eng = enc._pair.engine(c.B)
    eng.random_mask = lambda seed=None: eng.set_mask(mask)
    try:
        out = model(x, planes, yvec)
    finally:
        np.random.normal = orig_normal
    opt.zero_grad()
    kl = N.latent_loss(model.z_mean, model.z_sigma) / c.bs_flag          # cvae.py:137
    ll = mse(out, x) / c.bs_flag * 50.0                                   # cvae.py:139
    vae = ll + kl
    vae.backward()
    for k, v in (("VAE", vae), ("KL", kl), ("LL", ll)):
        ref = float(c.npz["s0/" + k])
        assert abs(float(v) - ref) <= 2e-5 * abs(ref), (k, float(v), ref)
    ok, msg = compare(c.npz, "s0/dec", out.detach().cpu().numpy(), 1e-3, 3e-4)
    assert ok, msg
    zero_grad = {"encoder.conv.1.bias", "encoder.fc.0.bias"}
    for mod in (enc, dec):
        for k, p in mod.named_parameters():
            if k in zero_grad:
                continue
            ok, msg = compare(c.npz, "s0/grad/" + k, p.grad.detach().cpu().numpy(), 2e-3, 1e-6, 1e-3)
            assert ok, msg
    opt.step()
    # eval-mode decoder-only sampling (cvae.py:167-171) after the step, different batch size
    model.eval()
    enc.eval()
    dec.eval()
    with torch.no_grad():
        zs = torch.from_numpy(c.npz["eval/z"]).cuda()
        ys = torch.from_numpy(c.npz["eval/y"]).cuda()
        samples = dec(zs[:3], ys[:3])
    assert samples.shape == (3, 3, c.P, c.P) and bool(torch.isfinite(samples).all())
    assert float(samples.abs().max()) <= 1.0


def test_module_surface_runs_through_torch_custom_ops(monkeypatch):
    """encoder / decoder forward and backward go through torch.ops.cvaegan.* (cvae_gan_amd/torch_ops.py),
    and a tensor-only operator gives what the direct binding gives."""
    from cvae_gan_amd import torch_ops, ops
    c = Case("tiny_p16_ngf8")
    N, enc, dec, model = make(c)
    calls = []
    real = torch_ops._engine
    monkeypatch.setattr(torch_ops, "_engine", lambda h: (calls.append(h), real(h))[1])
    x, labels, eps, mask = c.inputs(0)
    yvec = torch.zeros(c.B, c.y)
    yvec[torch.arange(c.B), labels] = 1
    model.train()
    out = model(x.cuda(), labels.cuda(), yvec.cuda())
    out.square().mean().backward()
    torch.cuda.synchronize()
    assert len(calls) == 4 and len(set(calls)) == 1          # enc fwd, dec fwd, dec bwd, enc bwd on one engine
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.cvaegan, name)
    g = torch.Generator(device="cuda").manual_seed(3)
    p0 = torch.randn(1001, device="cuda", generator=g)
    grad = torch.randn(1001, device="cuda", generator=g)
    a, b = [p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)], [p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)]
    torch.ops.cvaegan.adamw(a[0], grad, a[1], a[2], 1e-3, 0.5, 0.999, 1e-8, 0.01, 1)
    ops.adamw(b[0], grad, b[1], b[2], 1001, 1e-3, 0.5, 0.999, 1e-8, 0.01, 1)
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and not torch.equal(a[0], p0)
    with pytest.raises(NotImplementedError):                  # no CPU implementation: no silent other path
        torch.ops.cvaegan.adamw(p0.cpu(), grad.cpu(), p0.cpu(), p0.cpu(), 1e-3, 0.5, 0.999, 1e-8, 0.01, 1)


def test_module_surface_with_z_dim_off_the_vector_width():
    """z_dim = 10: the head's rows hold 20 values and are padded to 24 inside the engine; the module surface
    still hands torch [B, 2z] tensors and takes [B, 2z] gradients.  z_mean / z_sigma (which depend on no random
    draw) against the oracle in f32, gradients finite and of the reference's shapes."""
    from cvae_gan_amd import networks as N
    from oracle import ref_cpu as O
    P, y, z, B, ngf = 32, 6, 10, 5, 8
    a = Args()
    a.z_dim, a.y_dim, a.pix_dim, a.device, a.dtype, a.ngf = z, y, P, torch.device("cuda"), "f32", ngf
    enc, dec = N.encoder(a), N.decoder(a)
    e, d = O.init_params(P, y, z, seed=3, ngf=ngf)
    enc.load_state_dict(e)
    dec.load_state_dict(d)
    model = N.CVAE_T(a, enc, dec)
    model.train()
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, P, P, generator=g) * 2 - 1
    labels = torch.randint(0, y, (B,), generator=g)
    yvec = O.one_hot(labels, y)
    out = model(x.cuda(), labels.cuda(), yvec.cuda())
    h_ref = O.encoder_forward(e, x, O.label_planes(labels, y, P), True, {})
    _, mu, sigma = O.sample_latent(h_ref, torch.zeros(B, z), z)
    assert tuple(model.z_mean.shape) == (B, z) and tuple(model.z_sigma.shape) == (B, z)
    assert float((model.z_mean.detach().cpu() - mu).abs().max()) <= 1e-4 * float(mu.abs().max())
    assert float((model.z_sigma.detach().cpu() - sigma).abs().max()) <= 1e-4 * float(sigma.abs().max())
    ll = torch.nn.functional.mse_loss(out, x.cuda()) / B * 50
    kl = N.latent_loss(model.z_mean, model.z_sigma) / B
    (kl + ll).backward()
    torch.cuda.synchronize()
    for k, shape in (("encoder.fc.3.weight", (2 * z, 4 * z)), ("encoder.fc.3.bias", (2 * z,))):
        gw = enc.state_dict(keep_vars=True)[k].grad
        assert gw is not None and tuple(gw.shape) == shape and bool(torch.isfinite(gw).all()) and float(gw.abs().max()) > 0, k
