#!/usr/bin/env python3
"""Generate the *.npz fixtures in this directory from the reference's own network classes.

Runs in the build container only (it needs /root/reference); the GPU box and the
test-suite read the committed fixtures, never the reference.  The reference's
models/networks.py imports packages that are not installed here and that the
network classes never use (torchvision, cv2, torchsummary, skimage, imageio);
they are replaced by empty stand-in modules in sys.modules before the import.  No
reference file is modified and no bytecode is written next to it.

What is recorded per case: inputs (x, labels, eps, dropout keep-mask), the initial
state_dicts under the checkpoint key names, forward intermediates, the three
losses, every parameter gradient, and parameters / BatchNorm buffers after 1 and
3 AdamW steps, plus an eval-mode decoder output.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import ref_cpu as O          # noqa: E402
from oracle.fixtures import put          # noqa: E402

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    ident = lambda *a, **k: None
    tv = mod("torchvision")
    tv.datasets = mod("torchvision.datasets")
    tv.transforms = mod("torchvision.transforms", Compose=lambda x: x, ToTensor=ident,
                        Normalize=ident)
    mod("cv2", imread=ident, imwrite=ident, connectedComponents=ident, resize=ident)
    mod("torchsummary", summary=ident)
    sk = mod("skimage")
    sk.transform = mod("skimage.transform", resize=ident)
    sk.io = mod("skimage.io")
    mod("imageio")
    mod("visdom", Visdom=object)
    mod("prefetch_generator", BackgroundGenerator=object)


def import_reference():
    _stub_modules()
    sys.path.insert(0, REF)
    import models.networks as N  # noqa: E402
    return N


class _EncAdapter(torch.nn.Module):
    """encoder.forward (models/networks.py:85-89) without the .to('cuda') ctor."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x, planes):
        return self.net(torch.cat([x, planes], dim=1))


class _DecAdapter(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, z, y):
        return self.net(torch.cat([z, y], dim=1))


class _Args:
    pass


def wrappers_on_cpu(N):
    """The reference's `encoder` / `decoder` wrapper classes call init_net(), which ends in
    .to('cuda') (networks.py:309) and cannot run here.  Replace that one function, in the imported
    module object only, by the same body without the device move; the wrappers then construct on
    the CPU and consume the random generator exactly as on a GPU box."""
    def init_net_cpu(net, init_type="normal"):
        net = torch.nn.DataParallel(net)
        N.init_weights(net, init_type)
        return net
    N.init_net = init_net_cpu


def build(N, P, y, z, seed, ngf, oracle_init, n_stage=4):
    """The reference's inner nets with the wrappers' ctor arguments (networks.py:82, :60; ngf is
    64 there, smaller here for the tiny cases), constructed and initialised in the wrappers' order
    (cvae.py:57-58: encoder constructed + init_weights, then decoder constructed + init_weights).
    oracle_init=True: the fixture does not store the initial weights; oracle.init_params(seed)
    must then reproduce the reference's draw bit for bit, which is asserted here."""
    torch.manual_seed(seed)
    norm = N.get_norm_layer("batch")
    nl = N.get_non_linearity("lrelu")
    enc = N.ResNetEncoder(P, 3 + y, 2 * z, ngf, n_stage, norm, nl)
    N.init_weights(torch.nn.DataParallel(enc), "kaiming")
    dec = N.ConvUpSampleDecoder(P, z + y, ngf, n_stage, norm, nl)
    N.init_weights(torch.nn.DataParallel(dec), "kaiming")
    if oracle_init:
        e, d = O.init_params(P, y, z, seed=seed, ngf=ngf, n_stage=n_stage)
        for k, v in enc.state_dict().items():
            assert torch.equal(v, e[O.EP + k]), "oracle init differs from the reference at " + k
        for k, v in dec.state_dict().items():
            assert torch.equal(v, d[O.DP + k]), "oracle init differs from the reference at " + k
    return enc, dec


def init_case(N, name, P, y, z, seed):
    """Seed parity of the initialisation and the checkpoint key names, from the reference's own
    wrapper classes: torch.manual_seed(seed); encoder(args); decoder(args) (cvae.py:57-58), then
    the state dicts cvae.py:234-235 saves (En.module.state_dict() with En = DataParallel(wrapper),
    i.e. the wrapper's own state dict).  Stored per tensor: shape, a strided sample, sum, L2."""
    wrappers_on_cpu(N)
    args = _Args()
    args.z_dim, args.y_dim, args.pix_dim, args.device = z, y, P, torch.device("cpu")
    torch.manual_seed(seed)
    En = torch.nn.DataParallel(N.encoder(args))
    De = torch.nn.DataParallel(N.decoder(args))
    out = {"meta": np.array([P, y, z, seed], dtype=np.int64)}
    for half, sd in (("encoder", En.module.state_dict()), ("decoder", De.module.state_dict())):
        out[half + "_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out["shape/" + k] = np.array(tuple(v.shape), dtype=np.int64)
            put(out, "init/" + k, v.detach().numpy(), small=False)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6), len(out["encoder_keys"]), "+",
          len(out["decoder_keys"]), "keys; first:", out["encoder_keys"][0], out["decoder_keys"][0])


def sd_with_prefix(net, prefix):
    return {prefix + k: v.detach().clone().numpy() for k, v in net.state_dict().items()}


def _run_case(N, name, P, y, z, B, batch_size_flag, lr, seed, ngf, small, n_steps=3,
             layer_taps=True, n_stage=4):
    enc, dec = build(N, P, y, z, seed, ngf, oracle_init=not small, n_stage=n_stage)
    args = _Args()
    args.z_dim, args.device = z, torch.device("cpu")
    model = N.CVAE_T(args, _EncAdapter(enc), _DecAdapter(dec))
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.5, 0.999), eps=1e-8)  # cvae.py:67
    mse = torch.nn.MSELoss()
    model.train()

    meta = [P, y, z, B, batch_size_flag, n_steps, ngf, seed, int(small)]
    if n_stage != 4:
        meta.append(n_stage)                # older fixtures have nine fields and mean four stages
    out = {"meta": np.array(meta, dtype=np.int64),
           "lr": np.array(lr, dtype=np.float64)}
    if small:
        out.update({"init/" + k: v for k, v in sd_with_prefix(enc, O.EP).items()})
        out.update({"init/" + k: v for k, v in sd_with_prefix(dec, O.DP).items()})

    g = torch.Generator().manual_seed(seed + 1000)
    drop_mod = dec.fc[2]
    captured = {}

    def hook(mod, inp, outp):
        i, o = inp[0].detach(), outp.detach()
        keep = torch.where(i != 0, (o != 0).float(), torch.ones_like(o))
        captured["mask"] = keep

    hd = drop_mod.register_forward_hook(hook)

    acts = {}
    hooks = []
    if layer_taps:
        def tap(key):
            def f(mod, inp, outp):
                acts[key] = outp.detach().clone().numpy()
            return f
        hooks.append(enc.conv[1].register_forward_hook(tap("stem")))
        hooks.append(enc.conv[4].res.res[0].register_forward_hook(tap("b0_c1")))
        hooks.append(enc.conv[4].res.res[3].register_forward_hook(tap("b0_c2")))
        hooks.append(enc.conv[4].register_forward_hook(tap("b0_out")))
        hooks.append(enc.conv[7].register_forward_hook(tap("b3_out")))
        hooks.append(dec.conv[0].register_forward_hook(tap("up0")))
        hooks.append(dec.conv[3].register_forward_hook(tap("up3")))

    for step in range(n_steps):
        x = torch.rand(B, 3, P, P, generator=g) * 2 - 1
        labels = torch.randint(0, y, (B,), generator=g)
        planes = torch.zeros(B, y, P, P)
        planes[torch.arange(B), labels] = 1                       # cvae.py:92-94, :129
        yvec = torch.zeros(B, y)
        yvec[torch.arange(B), labels] = 1
        np.random.seed(seed + 77 + step)
        eps = np.random.normal(0, 1, size=(B, z))
        np.random.seed(seed + 77 + step)                          # networks.py:32 redraws the same
        dec_out = model(x, planes, yvec)
        opt.zero_grad()
        kl = N.latent_loss(model.z_mean, model.z_sigma) / batch_size_flag   # cvae.py:137
        ll = mse(dec_out, x) / batch_size_flag * 50.0                       # cvae.py:139
        vae = ll + kl
        vae.backward()
        s = "s%d/" % step
        out[s + "x"] = x.numpy()
        out[s + "labels"] = labels.numpy()
        out[s + "eps"] = eps.astype(np.float32)
        out[s + "mask"] = captured["mask"].numpy().astype(np.uint8)
        out[s + "VAE"] = vae.detach().numpy()
        out[s + "KL"] = kl.detach().numpy()
        out[s + "LL"] = ll.detach().numpy()
        if step == 0:
            out["s0/dec"] = dec_out.detach().numpy()
            out["s0/z_mean"] = model.z_mean.detach().numpy()
            out["s0/z_sigma"] = model.z_sigma.detach().numpy()
            for k, v in acts.items():
                put(out, "s0/act/" + k, v, small)
            for k, p in enc.named_parameters():
                put(out, "s0/grad/" + O.EP + k, p.grad.detach().clone().numpy(), small)
            for k, p in dec.named_parameters():
                put(out, "s0/grad/" + O.DP + k, p.grad.detach().clone().numpy(), small)
        opt.step()
        if step in (0, n_steps - 1):
            tag = "after%d/" % (step + 1)
            for k, v in sd_with_prefix(enc, O.EP).items():
                put(out, tag + k, v, small)
            for k, v in sd_with_prefix(dec, O.DP).items():
                put(out, tag + k, v, small)
    for h in hooks:
        h.remove()
    hd.remove()

    # eval-mode decoder-only sample (cvae.py:167-171) and full eval forward (cvae.py:212)
    model.eval()
    with torch.no_grad():
        zs = torch.randn(B, z, generator=g)
        ys = torch.zeros(B, y)
        ys[torch.arange(B), torch.arange(B) % y] = 1
        out["eval/z"] = zs.numpy()
        out["eval/y"] = ys.numpy()
        out["eval/dec"] = dec(torch.cat([zs, ys], 1)).numpy()
        np.random.seed(seed + 999)
        e2 = np.random.normal(0, 1, size=(B, z))
        np.random.seed(seed + 999)
        out["eval/eps"] = e2.astype(np.float32)
        out["eval/full"] = model(x, planes, yvec).numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f MB" % (os.path.getsize(path) / 1e6),
          "VAE/KL/LL step0 =", float(out["s0/VAE"]), float(out["s0/KL"]), float(out["s0/LL"]))


def main():
    N = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]

    def run_case(N, name, **kw):            # noqa: F811  (optional filter: names on the command line)
        if not only or name in only:
            _run_case(N, name, **kw)
    # tiny: P=16 is the smallest size the nets accept (pix_dim must be a multiple of
    # 16: networks.py:179, :136); ngf=8 keeps every tensor small enough to store whole.
    run_case(N, "tiny_p16_ngf8", P=16, y=4, z=8, B=4, batch_size_flag=4, lr=1e-3, seed=0,
             ngf=8, small=True)
    # odd batch (not a multiple of any tile), batch_size flag != B, non-square-friendly y
    run_case(N, "ragged_p32_ngf8", P=32, y=5, z=16, B=6, batch_size_flag=64, lr=3e-4, seed=1,
             ngf=8, small=True, layer_taps=False)
    # the reference's real width (ngf=64), CIFAR-shaped; weights come from
    # oracle.init_params(seed), big tensors are stored as samples + moments.
    run_case(N, "c2_p32_ngf64", P=32, y=10, z=32, B=8, batch_size_flag=256, lr=1e-3, seed=2,
             ngf=64, small=False)
    # five down / up stages instead of the wrappers' four (ndown, nup of networks.py:165, :133): the
    # "deeper stacks" variant of the 128 x 128 configuration, here at 64 x 64 -> 2 x 2 with ngf=8
    run_case(N, "deep5_p64_ngf8", P=64, y=5, z=8, B=4, batch_size_flag=4, lr=1e-3, seed=3,
             ngf=8, small=True, layer_taps=False, n_stage=5)
    # initial state and checkpoint keys of the real wrapper classes (ngf 64, four stages) per seed
    if not only or "init_p16_seed7" in only:
        init_case(N, "init_p16_seed7", P=16, y=4, z=8, seed=7)
    if not only or "init_c3_seed0" in only:
        init_case(N, "init_c3_seed0", P=64, y=100, z=32, seed=0)


if __name__ == "__main__":
    main()
