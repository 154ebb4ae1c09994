"""CPU oracle for the CVAE train-step hot path (TEST INFRASTRUCTURE ONLY).

This file restates, in functional PyTorch on the CPU (fp32, or fp64 when asked),
the arithmetic of the reference's encoder, sampler, decoder, losses and optimiser.
It is the checker the HIP path is compared against; it is never the thing that is
shipped or measured as the product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  tests/golden/gen_goldens.py imports the reference's own network
classes (models/networks.py) on the CPU, runs them on seeded inputs and writes the
fixtures under tests/golden/; tests/test_oracle_golden.py checks every function
here against those fixtures.

All tensors use the reference's layouts: images NCHW, conv weights OIHW, linear
weights [out, in].  Parameters travel in two dicts that use the reference's
checkpoint keys (models/cvae.py:234-235):
    encoder:  "encoder.conv.1.weight", ... "encoder.fc.3.bias"
    decoder:  "decoder.conv.0.2.weight", ... "decoder.fc.3.bias"
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2      # models/networks.py:343
BN_EPS = 1e-5          # torch.nn.BatchNorm2d default used at models/networks.py:329
BN_MOMENTUM = 0.1
NGF = 64               # models/networks.py:60, :82
N_STAGE = 4            # ndown = nup = 4, models/networks.py:60, :82
LL_WEIGHT = 50.0       # models/cvae.py:139
ADAM_BETAS = (0.5, 0.999)   # models/cvae.py:67
ADAM_EPS = 1e-8
ADAM_WD = 1e-2         # torch.optim.AdamW default, not overridden at models/cvae.py:67

EP = "encoder."      # En.module.state_dict(): the wrapper's child attribute is `encoder` (networks.py:82)
DP = "decoder."      # De.module.state_dict(): child attribute `decoder` (networks.py:60)


# --------------------------------------------------------------------------- shapes
def stage_channels(n_stage=N_STAGE, ngf=NGF):
    """(in, out) channel pairs of the encoder's down blocks.
    models/networks.py:177-183: prev starts at 1, cur = min(8, prev*2)."""
    out, prev = [], 1
    for _ in range(n_stage):
        cur = min(8, prev * 2)
        out.append((ngf * prev, ngf * cur))
        prev = cur
    return out


def up_channels(n_stage=N_STAGE, ngf=NGF):
    """(in, out) channel pairs of the decoder's up stages.
    models/networks.py:140-144: prev starts at 8, cur = min(prev, 2**i), i = nup-1..0."""
    out, prev = [], 8
    for i in range(n_stage - 1, -1, -1):
        cur = min(prev, 2 ** i)
        out.append((ngf * prev, ngf * cur))
        prev = cur
    return out


def param_shapes(pix_dim, y_dim, z_dim, n_stage=N_STAGE, ngf=NGF):
    """Ordered {key: shape} for both checkpoints, parameters and buffers, in
    state_dict order (models/networks.py:164-191 encoder, :129-157 decoder)."""
    enc = OrderedDict()
    enc[EP + "conv.1.weight"] = (ngf, 3 + y_dim, 7, 7)
    enc[EP + "conv.1.bias"] = (ngf,)
    _bn_keys(enc, EP + "conv.2", ngf)
    for i, (cin, cout) in enumerate(stage_channels(n_stage, ngf)):
        p = EP + "conv.%d." % (4 + i)
        enc[p + "res.res.0.weight"] = (cin, cin, 3, 3)
        _bn_keys(enc, p + "res.res.1", cin)
        enc[p + "res.res.3.weight"] = (cin, cin, 3, 3)
        _bn_keys(enc, p + "res.res.4", cin)
        enc[p + "conv.weight"] = (cout, cin, 3, 3)
    s = pix_dim // (2 ** n_stage)
    c_last = stage_channels(n_stage, ngf)[-1][1]
    nz = 2 * z_dim                      # models/networks.py:76
    fc_dim = 2 * nz                     # models/networks.py:169
    enc[EP + "fc.0.weight"] = (fc_dim, s * s * c_last)
    enc[EP + "fc.0.bias"] = (fc_dim,)
    _bn_keys(enc, EP + "fc.1", fc_dim)
    enc[EP + "fc.3.weight"] = (nz, fc_dim)
    enc[EP + "fc.3.bias"] = (nz,)

    dec = OrderedDict()
    for i, (cin, cout) in enumerate(up_channels(n_stage, ngf)):
        dec[DP + "conv.%d.2.weight" % i] = (cout, cin, 3, 3)
        dec[DP + "conv.%d.2.bias" % i] = (cout,)
    dec[DP + "conv.%d.weight" % (n_stage + 1)] = (3, ngf, 7, 7)
    dec[DP + "conv.%d.bias" % (n_stage + 1)] = (3,)
    in_dim = z_dim + y_dim              # models/networks.py:58
    fcd = 4 * in_dim                    # models/networks.py:137
    dec[DP + "fc.0.weight"] = (fcd, in_dim)
    dec[DP + "fc.0.bias"] = (fcd,)
    dec[DP + "fc.3.weight"] = (s * s * ngf * 8, fcd)
    dec[DP + "fc.3.bias"] = (s * s * ngf * 8,)
    return enc, dec


def _bn_keys(d, prefix, c):
    d[prefix + ".weight"] = (c,)
    d[prefix + ".bias"] = (c,)
    d[prefix + ".running_mean"] = (c,)
    d[prefix + ".running_var"] = (c,)
    d[prefix + ".num_batches_tracked"] = ()


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def init_params(pix_dim, y_dim, z_dim, seed=0, dtype=torch.float32, n_stage=N_STAGE,
                ngf=NGF):
    """The state torch.manual_seed(seed); encoder(args); decoder(args) produces
    (models/cvae.py:57-58).  Each wrapper constructs its network (the nn.Conv2d and
    nn.Linear constructors draw kaiming_uniform_(a=sqrt(5)) weights and uniform biases,
    in definition order: models/networks.py:171-191, :139-157) and then calls
    init_weights(net, 'kaiming') (models/networks.py:283-304): conv / linear weights
    kaiming_normal_(a=0, fan_in), biases 0; BatchNorm2d gamma ~ N(1, 0.02), beta 0;
    BatchNorm1d left at its constructor values (gamma 1, beta 0) because the class-name
    test at models/networks.py:299 only matches 'BatchNorm2d'.  Module definition order,
    net.apply() order and state_dict order coincide for both networks."""
    torch_init = torch.nn.init
    g = torch.Generator().manual_seed(seed)
    enc_s, dec_s = param_shapes(pix_dim, y_dim, z_dim, n_stage, ngf)
    out = []
    for shapes in (enc_s, dec_s):
        for k, shp in shapes.items():
            if len(shp) >= 2:                       # constructor draws, overwritten below
                w = torch.empty(shp)
                torch_init.kaiming_uniform_(w, a=math.sqrt(5), generator=g)
                if k.replace(".weight", ".bias") in shapes:
                    fan_in = w[0].numel()
                    torch.empty(shp[0]).uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in), generator=g)
        sd = OrderedDict()
        for k, shp in shapes.items():
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.zeros((), dtype=torch.long)
            elif k.endswith("running_mean"):
                sd[k] = torch.zeros(shp, dtype=dtype)
            elif k.endswith("running_var"):
                sd[k] = torch.ones(shp, dtype=dtype)
            elif len(shp) >= 2:
                fan_in = 1
                for s_ in shp[1:]:
                    fan_in *= s_
                std = math.sqrt(2.0) / math.sqrt(fan_in)
                sd[k] = torch.empty(shp).normal_(0, std, generator=g).to(dtype)
            elif k.endswith(".bias"):
                sd[k] = torch.zeros(shp, dtype=dtype)
            else:  # BatchNorm weight
                is_bn1d = ".fc.1." in k
                sd[k] = torch.ones(shp, dtype=dtype) if is_bn1d else \
                    torch.empty(shp).normal_(1.0, 0.02, generator=g).to(dtype)
        out.append(sd)
    return out[0], out[1]


# --------------------------------------------------------------------------- pieces
def _bn(sd, prefix, x, train, new_stats):
    """BatchNorm (2d or 1d) with the reference's defaults.  In training mode the
    batch statistics normalise (biased variance) and the running statistics move
    with momentum 0.1 using the unbiased variance; the updated buffers are
    returned through new_stats, the input dict is left untouched."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not train:
        return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    rm2, rv2 = rm.detach().clone(), rv.detach().clone()
    y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
    if new_stats is not None:
        new_stats[prefix + ".running_mean"] = rm2
        new_stats[prefix + ".running_var"] = rv2
        new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    return y


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def label_planes(labels, y_dim, pix_dim, dtype=torch.float32):
    """One-hot label PLANES [B, y, P, P] as the trainer builds them
    (models/cvae.py:92-94 and :129)."""
    b = labels.shape[0]
    planes = torch.zeros(b, y_dim, pix_dim, pix_dim, dtype=dtype)
    planes[torch.arange(b), labels] = 1
    return planes


def one_hot(labels, y_dim, dtype=torch.float32):
    return F.one_hot(labels, y_dim).to(dtype)


def stem_conv(sd, xin):
    """ReflectionPad2d(3) -> Conv2d(3+y, 64, 7) (models/networks.py:172-173)."""
    return F.conv2d(F.pad(xin, (3, 3, 3, 3), mode="reflect"),
                    sd[EP + "conv.1.weight"], sd[EP + "conv.1.bias"])


def res_block(sd, prefix, x, train, new_stats, taps=None):
    """BasicResBlock (models/networks.py:233-243): conv3x3-BN-LReLU-conv3x3-BN, + x."""
    c1 = F.conv2d(x, sd[prefix + "res.res.0.weight"], None, 1, 1)
    b1 = _lrelu(_bn(sd, prefix + "res.res.1", c1, train, new_stats))
    c2 = F.conv2d(b1, sd[prefix + "res.res.3.weight"], None, 1, 1)
    n2 = _bn(sd, prefix + "res.res.4", c2, train, new_stats)
    if taps is not None:
        taps.update(c1=c1, b1=b1, c2=c2)
    return n2 + x


def down_block(sd, prefix, x, train, new_stats, taps=None):
    """ConvResBlock direction='down' (models/networks.py:220-221):
    conv3x3 stride 2 (bias-free) applied to LReLU(res(x))."""
    r = _lrelu(res_block(sd, prefix, x, train, new_stats, taps))
    out = F.conv2d(r, sd[prefix + "conv.weight"], None, 2, 1)
    if taps is not None:
        taps.update(r=r, out=out)
    return out


def count_stages(sd):
    """Number of down blocks (encoder dict) or up stages (decoder dict) a state dict holds:
    ndown / nup of models/networks.py:165, :133 (4 in the wrappers, :82 and :60)."""
    return sum(1 for k in sd if (k.startswith(EP) and k.endswith(".conv.weight"))
               or (k.startswith(DP) and k.endswith(".2.weight")))


def encoder_forward(sd, x, planes, train=True, new_stats=None, taps=None, n_stage=None):
    """encoder.forward (models/networks.py:85-89) followed by ResNetEncoder.forward
    (models/networks.py:193-196).  x [B,3,P,P], planes [B,y,P,P] -> [B, 2z]."""
    n_stage = n_stage or count_stages(sd)
    h = stem_conv(sd, torch.cat([x, planes], dim=1))
    if taps is not None:
        taps["stem"] = h
    h = _lrelu(_bn(sd, EP + "conv.2", h, train, new_stats))
    if taps is not None:
        taps["a0"] = h
    for i in range(n_stage):
        t = {} if taps is not None else None
        h = down_block(sd, EP + "conv.%d." % (4 + i), h, train, new_stats, t)
        if taps is not None:
            taps["block%d" % i] = t
    h = h.reshape(h.shape[0], -1)
    f0 = F.linear(h, sd[EP + "fc.0.weight"], sd[EP + "fc.0.bias"])
    f1 = _lrelu(_bn(sd, EP + "fc.1", f0, train, new_stats))
    if taps is not None:
        taps.update(f0=f0, f1=f1)
    return F.linear(f1, sd[EP + "fc.3.weight"], sd[EP + "fc.3.bias"])


def sample_latent(h_enc, eps, z_dim):
    """CVAE_T._sample_latent (models/networks.py:24-37) with eps supplied by the
    caller.  Returns (z, z_mean, z_sigma)."""
    mu = h_enc[:, :z_dim]
    sigma = torch.exp(h_enc[:, z_dim:])
    return mu + sigma * eps, mu, sigma


def latent_loss(z_mean, z_stddev):
    """models/networks.py:44-47 (no 1/2 factor)."""
    m2 = z_mean * z_mean
    s2 = z_stddev * z_stddev
    return torch.mean(m2 + s2 - torch.log(s2) - 1)


def upconv(sd, prefix, x):
    """deconv3x3 (models/networks.py:274-281): bilinear x2 (align_corners=False),
    ReflectionPad2d(1), Conv2d 3x3 with bias."""
    u = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    u = F.pad(u, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(u, sd[prefix + "weight"], sd[prefix + "bias"])


def decoder_forward(sd, z, y_vec, drop_mask=None, train=True, taps=None, n_stage=None):
    """decoder.forward (models/networks.py:63-68) and ConvUpSampleDecoder.forward
    (models/networks.py:159-162).  drop_mask is the 0/1 keep mask of nn.Dropout(0.5)
    (models/networks.py:155); kept activations are scaled by 2.  In eval mode the
    dropout is the identity."""
    n_stage = n_stage or count_stages(sd)
    h = torch.cat([z, y_vec], dim=1)
    g0 = F.linear(h, sd[DP + "fc.0.weight"], sd[DP + "fc.0.bias"])
    h = _lrelu(g0)
    if train:
        h = h * drop_mask * 2.0
    g1 = h
    h = F.linear(h, sd[DP + "fc.3.weight"], sd[DP + "fc.3.bias"])
    if taps is not None:
        taps.update(g0=g0, g1=g1)
    c0 = sd[DP + "conv.0.2.weight"].shape[1]
    s = int(round(math.sqrt(h.shape[1] // c0)))
    h = h.view(h.shape[0], c0, s, s)
    if taps is not None:
        taps["x0"] = h
    for i in range(n_stage):
        h = upconv(sd, DP + "conv.%d.2." % i, h)
        if taps is not None:
            taps["up%d" % i] = h
    k = DP + "conv.%d." % (n_stage + 1)
    h = F.conv2d(F.pad(h, (3, 3, 3, 3), mode="reflect"), sd[k + "weight"], sd[k + "bias"])
    if taps is not None:
        taps["pre"] = h
    return torch.tanh(h)


def losses(dec, x, z_mean, z_sigma, batch_size):
    """models/cvae.py:137-140.  batch_size is the --batch_size FLAG, not B."""
    kl = latent_loss(z_mean, z_sigma) / batch_size
    ll = F.mse_loss(dec, x) / batch_size * LL_WEIGHT
    return ll + kl, kl, ll


def cvae_forward(enc_sd, dec_sd, x, labels, eps, drop_mask, y_dim, train=True,
                 new_stats=None):
    """CVAE_T.forward (models/networks.py:39-42) with explicit randomness."""
    z_dim = eps.shape[1]
    planes = label_planes(labels, y_dim, x.shape[-1], x.dtype)
    h_enc = encoder_forward(enc_sd, x, planes, train, new_stats)
    z, mu, sigma = sample_latent(h_enc, eps, z_dim)
    dec = decoder_forward(dec_sd, z, one_hot(labels, y_dim, x.dtype), drop_mask, train)
    return dec, mu, sigma, h_enc


def adamw_update(p, g, m, v, step, lr, betas=ADAM_BETAS, eps=ADAM_EPS, wd=ADAM_WD):
    """One torch.optim.AdamW step on a single tensor (decoupled weight decay,
    bias-corrected), as configured at models/cvae.py:67.  step counts from 1."""
    b1, b2 = betas
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


class TrainState:
    """Parameters + AdamW moments for repeated train_step() calls."""

    def __init__(self, enc_sd, dec_sd):
        self.enc = OrderedDict((k, v.clone()) for k, v in enc_sd.items())
        self.dec = OrderedDict((k, v.clone()) for k, v in dec_sd.items())
        self.m, self.v, self.step = {}, {}, 0
        for sd in (self.enc, self.dec):
            for k, t in sd.items():
                if not is_buffer(k):
                    self.m[k] = torch.zeros_like(t)
                    self.v[k] = torch.zeros_like(t)


def train_step(st, x, labels, eps, drop_mask, y_dim, batch_size, lr):
    """The body of CVAE.train()'s inner loop, models/cvae.py:132-157: forward,
    KL + 50*MSE loss, backward, AdamW.  Returns dict(VAE, KL, LL, dec, grads)."""
    params = {}
    for sd in (st.enc, st.dec):
        for k, t in sd.items():
            if not is_buffer(k):
                sd[k] = t.detach().clone().requires_grad_(True)
                params[k] = sd[k]
    new_stats = {}
    dec, mu, sigma, h_enc = cvae_forward(st.enc, st.dec, x, labels, eps, drop_mask,
                                         y_dim, True, new_stats)
    vae, kl, ll = losses(dec, x, mu, sigma, batch_size)
    grads = torch.autograd.grad(vae, list(params.values()))
    grads = dict(zip(params.keys(), grads))
    st.step += 1
    with torch.no_grad():
        for sd in (st.enc, st.dec):
            for k in list(sd.keys()):
                if k in params:
                    p, m, v = adamw_update(sd[k].detach(), grads[k], st.m[k], st.v[k],
                                           st.step, lr)
                    sd[k], st.m[k], st.v[k] = p, m, v
                elif k in new_stats:
                    sd[k] = new_stats[k]
    return dict(VAE=vae.detach(), KL=kl.detach(), LL=ll.detach(), dec=dec.detach(),
                z_mean=mu.detach(), z_sigma=sigma.detach(), h_enc=h_enc.detach(),
                grads=grads)


def dp_forward_backward(enc_sd, dec_sd, x, labels, eps, drop_mask, y_dim, batch_size, n_replica):
    """Gradient of one step under the reference's nn.DataParallel wrapping
    (models/cvae.py:57-59): the batch is split into n_replica equal chunks, encoder and
    decoder run per chunk (so BatchNorm normalises each chunk by its own statistics), the
    sampler and both losses see the gathered full batch.  Returns (losses, grads)."""
    params = {}
    for sd in (enc_sd, dec_sd):
        for k, t in sd.items():
            if not is_buffer(k):
                sd[k] = t.detach().clone().requires_grad_(True)
                params[k] = sd[k]
    z_dim = eps.shape[1]
    xs, ls = x.chunk(n_replica), labels.chunk(n_replica)
    h = torch.cat([encoder_forward(enc_sd, xc, label_planes(lc, y_dim, x.shape[-1], x.dtype), True, None)
                   for xc, lc in zip(xs, ls)])
    z, mu, sigma = sample_latent(h, eps, z_dim)
    dec = torch.cat([decoder_forward(dec_sd, zc, one_hot(lc, y_dim, x.dtype), mc, True)
                     for zc, lc, mc in zip(z.chunk(n_replica), ls, drop_mask.chunk(n_replica))])
    vae, kl, ll = losses(dec, x, mu, sigma, batch_size)
    grads = torch.autograd.grad(vae, list(params.values()))
    return dict(VAE=vae.detach(), KL=kl.detach(), LL=ll.detach()), dict(zip(params.keys(), grads))
