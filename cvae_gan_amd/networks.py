"""Drop-in module surface of the reference's models/networks.py on the HIP engine.

    encoder(args), decoder(args), CVAE_T(args, encoder, decoder), latent_loss(...)

keep the reference's constructor arguments, call signatures, attributes (z_mean, z_sigma)
and checkpoint keys ("encoder.conv.1.weight", ...), so the training loop of
models/cvae.py:125-157 runs on them unchanged: forward, torch losses on the outputs,
loss.backward(), torch.optim.AdamW.step().  Parameters are ordinary nn.Parameters whose
storage is the engine's canonical f32 arena.

Differences a caller can observe:
  * label1 may be the reference's one-hot label PLANES [B, y, P, P] (models/cvae.py:129), a
    one-hot matrix [B, y] or an integer vector [B].  Planes are folded into a per-class bias
    of the stem convolution (exact for the constant planes the trainer builds); pass
    check_planes=True to verify that the planes really are constant one-hot planes.
  * activations are bf16 by default (args.dtype = 'f32' selects exact-f32 MFMA arithmetic);
  * more than one GPU: run one process per GPU (cvae_gan_amd.ddp), not nn.DataParallel.
"""
import numpy as np
import torch
import torch.nn as nn

from . import torch_ops
from .engine import Engine, param_shapes, init_state, is_buffer, EP, DP


def _dtype_of(args):
    d = getattr(args, "dtype", "bf16")
    if isinstance(d, torch.dtype):
        return d
    return torch.float32 if str(d) in ("f32", "fp32", "float32") else torch.bfloat16


class _Holder(nn.Module):
    """Parameter container that reproduces one level of the reference's module tree."""


def _plant(root, key, tensor):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Holder())
        mod = getattr(mod, p)
    if is_buffer(key):
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _leaf(root, key):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        mod = getattr(mod, p)
    return mod, parts[-1]


class _Half(nn.Module):
    """Common part of encoder / decoder: owns the parameters of one checkpoint file."""
    _which = None

    def __init__(self, args, dataset="1101"):
        super().__init__()
        self.z_dim_arg, self.y_dim, self.pix_dim = args.z_dim, args.y_dim, args.pix_dim
        self.ngf = getattr(args, "ngf", 64)
        self.dtype = _dtype_of(args)
        self.device = torch.device(getattr(args, "device", "cuda"))
        self._keys = list(param_shapes(self.pix_dim, self.y_dim, self.z_dim_arg, 4, self.ngf)[0 if self._which == "enc" else 1])
        e, d = init_state(self.pix_dim, self.y_dim, self.z_dim_arg, seed=getattr(args, "seed", 0), ngf=self.ngf)
        st = e if self._which == "enc" else d
        for k in self._keys:
            _plant(self, k, st[k].to(self.device))
        self._pair = None            # set by CVAE_T / _Pair

    def _pair_or_solo(self):
        if self._pair is None:
            _Pair(self if self._which == "enc" else None, self if self._which == "dec" else None)
        return self._pair


class _Pair:
    """The engines (one per batch size) shared by an encoder and a decoder module."""

    def __init__(self, enc, dec):
        some = enc if enc is not None else dec
        self.enc, self.dec = enc, dec
        self.cfg = (some.pix_dim, some.y_dim, some.z_dim_arg, some.ngf, some.dtype, some.device)
        self.engines = {}
        for m in (enc, dec):
            if m is not None:
                m._pair = self

    def engine(self, B):
        eng = self.engines.get(B)
        if eng is None:
            P, y, z, ngf, dtype, dev = self.cfg
            first = next(iter(self.engines.values()), None)
            state = None
            if first is None:
                state = {}
                for m in (self.enc, self.dec):
                    if m is not None:
                        state.update({k: v.detach() for k, v in m.state_dict().items()})
            eng = Engine(P, y, z, B, dtype=dtype, ngf=ngf, device=dev, share=first)
            if first is None:
                eng.load_state(state)
                # from now on the modules' parameters and buffers ARE the engine's arena
                for m in (self.enc, self.dec):
                    if m is None:
                        continue
                    for k in m._keys:
                        mod, name = _leaf(m, k)
                        if is_buffer(k):
                            mod._buffers[name] = eng.buffers[k]
                        else:
                            mod._parameters[name].data = eng.params[k]
            self.engines[B] = eng
        return eng


class _EncFn(torch.autograd.Function):
    """ResNetEncoder.forward and its backward through torch.ops.cvaegan (cvae_gan_amd/torch_ops.py)."""

    @staticmethod
    def forward(ctx, eng, keys, x, labels, *params):
        ctx.eng, ctx.keys, ctx.handle = eng, keys, torch_ops.register_engine(eng)
        return torch.ops.cvaegan.encoder_forward(ctx.handle, x, labels)

    @staticmethod
    def backward(ctx, dh):
        torch.ops.cvaegan.encoder_backward(ctx.handle, dh)
        return (None, None, None, None) + tuple(ctx.eng.grads[k] for k in ctx.keys)


class _DecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, keys, z, y_vec, *params):
        ctx.eng, ctx.keys, ctx.handle = eng, keys, torch_ops.register_engine(eng)
        return torch.ops.cvaegan.decoder_forward(ctx.handle, z, y_vec, bool(eng.training))

    @staticmethod
    def backward(ctx, d_dec):
        dz = torch.ops.cvaegan.decoder_backward(ctx.handle, d_dec)
        return (None, None, dz, None) + tuple(ctx.eng.grads[k] for k in ctx.keys)


class encoder(_Half):
    """models/networks.py:70-89.  forward(input [B,3,P,P], label) -> [B, 2*z_dim]."""
    _which = "enc"

    def __init__(self, args, dataset="1101", check_planes=False):
        super().__init__(args, dataset)
        self.z_dim = args.z_dim * 2
        self.in_dim = 3 + args.y_dim
        self.check_planes = check_planes

    def labels_of(self, label, x):
        if label.dim() == 4:
            lab = label[:, :, 0, 0].argmax(dim=1)
            if self.check_planes:
                ref = torch.zeros_like(label)
                ref[torch.arange(label.shape[0]), lab] = 1
                if not torch.equal(ref, label):
                    raise ValueError("label planes are not constant one-hot planes; the folded stem does not apply")
            return lab.int()
        if label.dim() == 2:
            return label.argmax(dim=1).int()
        return label.int()

    def forward(self, input, label):
        pair = self._pair_or_solo()
        eng = pair.engine(input.shape[0])
        eng.training = self.training
        keys = [k for k in self._keys if not is_buffer(k)]
        params = [_leaf(self, k)[0]._parameters[_leaf(self, k)[1]] for k in keys]
        return _EncFn.apply(eng, keys, input, self.labels_of(label, input).to(input.device), *params)


class decoder(_Half):
    """models/networks.py:49-68.  forward(input [B,z], label [B,y]) -> [B,3,P,P] in (-1, 1)."""
    _which = "dec"

    def __init__(self, args, dataset="1101"):
        super().__init__(args, dataset)
        self.z_dim = args.z_dim
        self.in_dim = args.z_dim + args.y_dim

    def forward(self, input, label):
        pair = self._pair_or_solo()
        eng = pair.engine(input.shape[0])
        eng.training = self.training
        keys = [k for k in self._keys if not is_buffer(k)]
        params = [_leaf(self, k)[0]._parameters[_leaf(self, k)[1]] for k in keys]
        return _DecFn.apply(eng, keys, input, label, *params)


def _unwrap(m):
    return m.module if isinstance(m, nn.DataParallel) else m


class CVAE_T(nn.Module):
    """models/networks.py:16-42.  After forward(): .z_mean [B,z], .z_sigma [B,z] (sigma)."""

    def __init__(self, args, encoder, decoder):
        super().__init__()
        self.z_dim = args.z_dim
        self.encoder = encoder
        self.decoder = decoder
        self.device = args.device
        _Pair(_unwrap(encoder), _unwrap(decoder))

    def _sample_latent(self, h_enc):
        """z ~ N(mu, sigma^2) with eps drawn on the host by numpy, as the reference does
        (models/networks.py:32), so that a seeded numpy reproduces the reference's draw."""
        mu = h_enc[:, :self.z_dim]
        sigma = torch.exp(h_enc[:, self.z_dim:])
        std_z = torch.from_numpy(np.random.normal(0, 1, size=tuple(sigma.size()))).float().to(h_enc.device)
        self.z_mean, self.z_sigma = mu, sigma
        return mu + sigma * std_z

    def forward(self, state, label1, label2):
        h_enc = self.encoder(state, label1)
        z = self._sample_latent(h_enc)
        return self.decoder(z, label2)


def latent_loss(z_mean, z_stddev):
    """models/networks.py:44-47."""
    mean_sq = z_mean * z_mean
    stddev_sq = z_stddev * z_stddev
    return torch.mean(mean_sq + stddev_sq - torch.log(stddev_sq) - 1)
