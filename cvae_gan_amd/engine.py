"""The CVAE train-step executor for MI355X.

Engine owns the canonical parameters (f32, reference layouts and checkpoint keys,
models/cvae.py:234-235:
"encoder.conv.1.weight", ...), their gradient / AdamW-moment arenas, the packed weight
shadows and every activation buffer, and drives the HIP kernels stage by stage:

    enc_forward -> sample -> dec_forward -> [loss] -> dec_backward ->
    sample_backward -> enc_backward -> adamw

Every stage only enqueues kernels on the current stream (no host sync), so a whole
step can be captured into a HIP graph and replayed.

Reference map (all in /root/reference):
    enc_forward / enc_backward    models/networks.py:85-89, :164-196, :198-249
    sample / sample_backward      models/networks.py:24-47, models/cvae.py:137
    dec_forward / dec_backward    models/networks.py:63-68, :129-162, :263-281
    losses                        models/cvae.py:137-140
    adamw                         models/cvae.py:67, :157
"""
import math
from collections import OrderedDict

import contextlib
import os

import torch

from . import ops
from ._lib import ConvDesc, WMap, PAD_ZERO, PAD_REFLECT, FORM_S2_TRANSPOSED, CONV_ALONE, WS_HEADER

# Checkpoint key prefixes.  The reference saves self.En.module.state_dict() (models/cvae.py:234-235):
# En.module is the `encoder` wrapper, whose one child attribute is `self.encoder` (the inner
# ResNetEncoder; the nn.DataParallel that init_net builds around it is dropped, models/networks.py:82-83),
# so the keys read "encoder.conv.1.weight", "decoder.conv.0.2.weight", ...
EP = "encoder."
DP = "decoder."
ADAM_BETAS = (0.5, 0.999)        # models/cvae.py:67
ADAM_EPS = 1e-8
ADAM_WD = 1e-2                   # torch.optim.AdamW default


def _rup(v, m):
    return (v + m - 1) // m * m


def stage_channels(n_stage, ngf):
    out, prev = [], 1
    for _ in range(n_stage):
        cur = min(8, prev * 2)
        out.append((ngf * prev, ngf * cur))
        prev = cur
    return out


def up_channels(n_stage, ngf):
    out, prev = [], 8
    for i in range(n_stage - 1, -1, -1):
        cur = min(prev, 2 ** i)
        out.append((ngf * prev, ngf * cur))
        prev = cur
    return out


def param_shapes(P, y_dim, z_dim, n_stage=4, ngf=64):
    """Ordered {key: shape} of both checkpoints in state_dict order."""
    def bn(d, prefix, c):
        d[prefix + ".weight"] = (c,)
        d[prefix + ".bias"] = (c,)
        d[prefix + ".running_mean"] = (c,)
        d[prefix + ".running_var"] = (c,)
        d[prefix + ".num_batches_tracked"] = ()
    enc = OrderedDict()
    enc[EP + "conv.1.weight"] = (ngf, 3 + y_dim, 7, 7)
    enc[EP + "conv.1.bias"] = (ngf,)
    bn(enc, EP + "conv.2", ngf)
    for i, (cin, cout) in enumerate(stage_channels(n_stage, ngf)):
        p = EP + "conv.%d." % (4 + i)
        enc[p + "res.res.0.weight"] = (cin, cin, 3, 3)
        bn(enc, p + "res.res.1", cin)
        enc[p + "res.res.3.weight"] = (cin, cin, 3, 3)
        bn(enc, p + "res.res.4", cin)
        enc[p + "conv.weight"] = (cout, cin, 3, 3)
    s = P // (2 ** n_stage)
    c_last = stage_channels(n_stage, ngf)[-1][1]
    nz, fc_dim = 2 * z_dim, 4 * z_dim
    enc[EP + "fc.0.weight"] = (fc_dim, s * s * c_last)
    enc[EP + "fc.0.bias"] = (fc_dim,)
    bn(enc, EP + "fc.1", fc_dim)
    enc[EP + "fc.3.weight"] = (nz, fc_dim)
    enc[EP + "fc.3.bias"] = (nz,)
    dec = OrderedDict()
    for i, (cin, cout) in enumerate(up_channels(n_stage, ngf)):
        dec[DP + "conv.%d.2.weight" % i] = (cout, cin, 3, 3)
        dec[DP + "conv.%d.2.bias" % i] = (cout,)
    dec[DP + "conv.%d.weight" % (n_stage + 1)] = (3, ngf, 7, 7)
    dec[DP + "conv.%d.bias" % (n_stage + 1)] = (3,)
    in_dim = z_dim + y_dim
    fcd = 4 * in_dim
    dec[DP + "fc.0.weight"] = (fcd, in_dim)
    dec[DP + "fc.0.bias"] = (fcd,)
    dec[DP + "fc.3.weight"] = (s * s * ngf * 8, fcd)
    dec[DP + "fc.3.bias"] = (s * s * ngf * 8,)
    return enc, dec


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def init_state(P, y_dim, z_dim, seed=0, n_stage=4, ngf=64):
    """The parameters torch.manual_seed(seed); encoder(args); decoder(args) leaves behind
    (models/cvae.py:57-58), bit for bit.  Host-side, once; returns CPU tensors.

    Each wrapper first constructs its network, whose nn.Conv2d / nn.Linear constructors draw
    their default initialisation (weight kaiming_uniform_(a=sqrt 5), then the bias uniform in
    +-1/sqrt(fan_in)) in definition order, and then runs init_weights(net, 'kaiming')
    (models/networks.py:283-304), which walks the modules in the same order and draws conv and
    linear weights kaiming_normal_(a=0, fan_in), sets their biases to 0, draws BatchNorm2d gamma
    ~ N(1, 0.02) and leaves BatchNorm1d alone (gamma 1: the class-name test at :299 only matches
    'BatchNorm2d').  The first round of draws is overwritten but moves the generator, so it is
    replayed here on scratch tensors."""
    from torch.nn import init
    g = torch.Generator().manual_seed(seed)
    enc_s, dec_s = param_shapes(P, y_dim, z_dim, n_stage, ngf)
    out = []
    for shapes in (enc_s, dec_s):
        for k, shp in shapes.items():               # constructors (models/networks.py:164-191, :129-157)
            if len(shp) < 2:
                continue
            init.kaiming_uniform_(torch.empty(shp), a=math.sqrt(5), generator=g)
            if k[:-len("weight")] + "bias" in shapes:
                bound = 1.0 / math.sqrt(int(torch.Size(shp[1:]).numel()))
                init.uniform_(torch.empty(shp[0]), -bound, bound, generator=g)
        sd = OrderedDict()
        for k, shp in shapes.items():               # init_weights' net.apply(init_func)
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.zeros((), dtype=torch.long)
            elif k.endswith("running_mean"):
                sd[k] = torch.zeros(shp)
            elif k.endswith("running_var"):
                sd[k] = torch.ones(shp)
            elif len(shp) >= 2:
                sd[k] = init.kaiming_normal_(torch.empty(shp), a=0, mode="fan_in", generator=g)
            elif k.endswith(".bias"):
                sd[k] = torch.zeros(shp)
            elif ".fc.1." in k:
                sd[k] = torch.ones(shp)
            else:
                sd[k] = init.normal_(torch.empty(shp), 1.0, 0.02, generator=g)
        out.append(sd)
    return out[0], out[1]


def arena_layout(P, y_dim, z_dim, n_stage=4, ngf=64, parts="both", extra=None):
    """Flat f32 arena shared by parameters, gradients and AdamW moments.  Tensors are placed in
    the order in which backward finishes their gradients (decoder tail first, encoder stem
    last), so that contiguous ranges can be all-reduced while backward is still running.
    parts="encoder" leaves the decoder out (a critic is the trunk alone); extra: {name: shape} of
    further trainable tensors (heads), placed first."""
    enc_s, dec_s = param_shapes(P, y_dim, z_dim, n_stage, ngf)
    if parts == "encoder":
        dec_s = OrderedDict()
    shapes = OrderedDict()
    shapes.update(extra or {})
    shapes.update(enc_s)
    shapes.update(dec_s)
    trainable = [k for k in shapes if not is_buffer(k)]
    dec_tr = [k for k in trainable if k.startswith(DP)]
    enc_tr = [k for k in trainable if k.startswith(EP)]

    def dec_rank(k):
        if ".fc.0." in k:
            return 100
        if ".fc.3." in k:
            return 99
        return n_stage + 1 - int(k[len(DP) + 5:].split(".")[0])      # conv.N...

    def enc_rank(k):
        if ".fc." in k:
            return 3 - int(k.split(".fc.")[1].split(".")[0])
        return 100 - int(k[len(EP) + 5:].split(".")[0])

    order = [k for k in (extra or {})] + sorted(dec_tr, key=dec_rank) + sorted(enc_tr, key=enc_rank)
    offs, n = {}, 0
    for k in order:
        offs[k] = n
        n += _rup(int(torch.Size(shapes[k]).numel()), 4)
    return dict(enc_keys=list(enc_s), dec_keys=list(dec_s), shapes=shapes, order=order, offsets=offs, numel=n)


# ------------------------------------------------------------------------------------
# descriptors
# ------------------------------------------------------------------------------------
def _desc(dtype, B, H, W, Cin, OH, OW, N, KW, ntaps, stride=1, tstep=(1, 1), off=(0, 0),
          pad_mode=PAD_ZERO, x_pix=None, x_row=None, x_img=None, y_pix=None, y_row=None, y_img=None,
          o_mul=(1, 1), o_off=(0, 0), y_dims=None):
    d = ConvDesc()
    d.dtype = ops.dt(dtype)
    d.B, d.H, d.W, d.Cin = B, H, W, Cin
    d.x_pix_pitch = x_pix if x_pix is not None else Cin
    d.x_row_pitch = x_row if x_row is not None else W * d.x_pix_pitch
    d.x_img_pitch = x_img if x_img is not None else H * d.x_row_pitch
    d.OH, d.OW, d.N, d.KW, d.ntaps = OH, OW, N, KW, ntaps
    d.stride = stride
    d.tstep_y, d.tstep_x = tstep
    d.off_y, d.off_x = off
    d.pad_mode = pad_mode
    yh, yw = y_dims if y_dims is not None else (OH, OW)
    d.y_pix_pitch = y_pix if y_pix is not None else N
    d.y_row_pitch = y_row if y_row is not None else yw * d.y_pix_pitch
    d.y_img_pitch = y_img if y_img is not None else yh * d.y_row_pitch
    d.oy_mul, d.ox_mul = o_mul
    d.oy_off, d.ox_off = o_off
    bk = 64 if dtype == torch.bfloat16 else 32
    d.w_ld = _rup(ntaps * Cin, bk)
    d.w_rows = _rup(N, 128)
    return d


def _wmap(nd, na, nb, kd, ka, kb, n_valid, kc_valid, ntaps, tap_mode=0, table=None, nr_valid=0):
    m = WMap()
    m.nr_valid = nr_valid
    m.nd, m.na, m.nb, m.kd, m.ka, m.kb = nd, na, nb, kd, ka, kb
    m.n_valid, m.kc_valid, m.ntaps, m.tap_mode = n_valid, kc_valid, ntaps, tap_mode
    if table is not None:
        assert len(table) <= 64
        for i, t in enumerate(table):
            m.tap_table[i] = t
    return m


class _Gemm:
    """One packed-weight layout + the descriptor that consumes it."""

    def __init__(self, eng, desc, wmap, canonical_key):
        self.desc, self.wmap, self.key = desc, wmap, canonical_key
        self.w = torch.zeros(desc.w_rows * desc.w_ld, dtype=eng.dtype, device=eng.device)
        eng._gemms.append(self)
        eng._ws_bytes = max(eng._ws_bytes, ops.conv_fwd_workspace(desc))

    def pack(self, eng):
        ops.weight_pack(self.desc, self.wmap, eng.params[self.key], self.w)


class _WgradOnly:
    """A weight-gradient geometry without a packed-weight layout of its own."""

    def __init__(self, eng, desc, wmap, canonical_key):
        self.desc, self.wmap, self.key = desc, wmap, canonical_key
        eng._wgrad_descs.append(desc)


class Engine:
    def __init__(self, pix_dim, y_dim, z_dim, batch, dtype=torch.bfloat16, ngf=64, n_stage=4,
                 device="cuda", seed=0, state=None, share=None, parts="both", extra_params=None,
                 own_grads=False, stem_input_grad=False):
        """share: another Engine of the same network; this one then aliases its parameter,
        gradient and optimiser arenas and BatchNorm buffers (only the batch size differs);
        own_grads gives it a gradient arena of its own all the same (two passes over shared
        weights whose gradients are added afterwards).  parts="encoder": the trunk alone
        (y_dim may then be 0: no label planes); extra_params: {name: shape} of further trainable
        tensors in the same arenas; stem_input_grad: also plan the gradient w.r.t. the image."""
        assert pix_dim % (2 ** n_stage) == 0, "pix_dim must be a multiple of %d" % (2 ** n_stage)
        assert z_dim >= 2 and z_dim % 2 == 0, "z_dim must be even (the head's 4 * z_dim features are handled 8 at a time)"
        assert ngf % 8 == 0 and (ngf & (ngf - 1)) == 0, "ngf must be a power of two >= 8"
        self.P, self.y, self.z, self.B = pix_dim, y_dim, z_dim, batch
        self.dtype, self.ngf, self.n_stage = dtype, ngf, n_stage
        self.device = torch.device(device)
        self.s = pix_dim // (2 ** n_stage)
        self._gemms, self._ws_bytes = [], 0
        self.overlap = os.environ.get("CVG_NO_OVERLAP") is None     # A/B switch for benchmarking
        self.fused_s2_dgrad = os.environ.get("CVG_S2_DGRAD_CLASSES") is None    # A/B: four gather convolutions instead
        # the forward pass runs on one stream with nothing beside it (the weight-gradient side streams only exist
        # in backward): its convolutions may keep their workgroups resident (see CVG_CONV_ALONE)
        self._fwd_flags = CONV_ALONE if os.environ.get("CVG_NO_FWD_WALK") is None else 0
        self._fork_sites = set(os.environ.get("CVG_FORK_SITES", "blk").split(","))    # A/B: where the chain's launch is enqueued first
        self.lowres_up_bwd = os.environ.get("CVG_UP_BWD_HIRES") is None      # A/B: gradients of the up-convolutions on the up-sampled grid
        # The forward pass the same way (one-tap GEMM with 9 cout columns, then the spatial operator) does a quarter
        # of the multiplications too, but moves as many bytes as the convolution over the up-sampled tensor (v is as
        # large as u), and that convolution runs at 1 PFLOP/s: measured 0.16 ms per step SLOWER with the present
        # one-tap GEMM and operator kernels.  Opt-in.
        self.lowres_up_fwd = os.environ.get("CVG_UP_FWD_LOWRES") is not None
        self._side, self._side_dirty = None, False
        self._wgrad_descs = []          # weight-gradient-only descriptors (workspace sizing)
        self._pack_batch = None
        self.step_count = 0
        self.training = True
        self.grad_ready = None      # callable(stage) -> None, see cvae_gan_amd/ddp.py
        self._frozen = False
        self._share, self._own_grads = share, own_grads
        # position of this engine's dropout stream: the mask is a hash of (element, counter) and the
        # counter advances by one per step, so streams start 2^32 steps apart per engine of a network
        # (two generator passes over shared weights must not drop the same units) and 2^40 per seed
        self._n_shares = 0
        salt = 0
        if share is not None:
            share._n_shares += 1
            salt = share._n_shares
        self._dropout_start = ((int(seed) & 0x3FFFFF) << 40) | (salt << 32)
        self.parts, self._extra = parts, OrderedDict(extra_params or {})
        self.stem_input_grad = stem_input_grad
        self.dec = None
        self._build_params(seed, state)
        self._build_encoder()
        if parts == "both":
            self._build_decoder()
        else:
            self.losses = self._f(4)
            self.step_dev = torch.zeros(1, dtype=torch.long, device=self.device)
            self.seed_dev = torch.full((1,), self._dropout_start, dtype=torch.long, device=self.device)
            self.bc_dev = self._f(2)
        self._build_workspaces()
        if share is not None:        # one optimiser state for all engines of the network
            self.step_dev, self.bc_dev = share.step_dev, share.bc_dev
        self._packed_version = -1
        self._pack_fork, self._pack_done, self._pack_stream = None, None, None
        self._split_pack = os.environ.get("CVG_PACK_WHOLE") is None     # A/B: the whole repack in front of the step

    # canonical parameters may be written by any engine sharing the arenas or by an outside
    # optimiser; one counter tells every engine when its packed shadows are stale
    @property
    def _param_version(self):
        return self._vers[0]

    @_param_version.setter
    def _param_version(self, v):
        self._vers[0] = v

    # ------------------------------------------------------------------ parameters
    def _build_params(self, seed, state):
        lay = arena_layout(self.P, self.y, self.z, self.n_stage, self.ngf, self.parts, self._extra)
        self.enc_keys, self.dec_keys, self.shapes = lay["enc_keys"], lay["dec_keys"], lay["shapes"]
        self.arena_order, self.arena_offsets, self.arena_numel = lay["order"], lay["offsets"], lay["numel"]
        shapes, order, offs, n = self.shapes, self.arena_order, self.arena_offsets, self.arena_numel
        dev = self.device
        sh = self._share
        if sh is not None:
            assert (sh.P, sh.y, sh.z, sh.ngf, sh.n_stage) == (self.P, self.y, self.z, self.ngf, self.n_stage)
            self.p_arena, self.g_arena, self.m_arena, self.v_arena = sh.p_arena, sh.g_arena, sh.m_arena, sh.v_arena
            self.params, self.grads, self.buffers, self._vers = sh.params, sh.grads, sh.buffers, sh._vers
            if self._own_grads:
                self.g_arena = torch.zeros(n, dtype=torch.float32, device=dev)
                self.grads = OrderedDict((k, self.g_arena[offs[k]:offs[k] + v.numel()].view(v.shape))
                                         for k, v in sh.grads.items())
            return
        self._vers = [0]
        self.p_arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g_arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m_arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v_arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params, self.grads, self.buffers = OrderedDict(), OrderedDict(), OrderedDict()
        for k in shapes:
            if is_buffer(k):
                continue
            cnt = int(torch.Size(shapes[k]).numel())
            self.params[k] = self.p_arena[offs[k]:offs[k] + cnt].view(shapes[k])
            self.grads[k] = self.g_arena[offs[k]:offs[k] + cnt].view(shapes[k])
        for k in shapes:
            if is_buffer(k):
                self.buffers[k] = torch.zeros(shapes[k], device=dev,
                                              dtype=torch.long if k.endswith("tracked") else torch.float32)
        if state is None:
            e, d = init_state(self.P, self.y, self.z, seed, self.n_stage, self.ngf)
            state = OrderedDict(e)
            if self.parts == "both":
                state.update(d)
        self.load_state(state)

    def load_state(self, state):
        """Copy reference-layout tensors (either checkpoint's keys) into the engine."""
        with torch.no_grad():
            for k, v in state.items():
                if k not in self.params and k not in self.buffers and ".module." in k:
                    k = k.replace(".module.", ".", 1)       # an inner net saved with its nn.DataParallel level
                if k in self.params:
                    assert tuple(v.shape) == tuple(self.shapes[k]), (k, tuple(v.shape), self.shapes[k])
                    self.params[k].copy_(v.to(torch.float32))
                elif k in self.buffers:
                    self.buffers[k].copy_(v)
                else:
                    raise KeyError("unexpected key %r" % k)
        self._param_version += 1

    def state_dict(self, which):
        """'encoder' / 'decoder' checkpoint dict with the reference's keys, on the CPU."""
        keys = {"encoder": self.enc_keys, "decoder": self.dec_keys,
                "critic": list(self._extra) + self.enc_keys}[which]      # critic: trunk + head
        out = OrderedDict()
        for k in keys:
            t = self.buffers[k] if is_buffer(k) else self.params[k]
            out[k] = t.detach().cpu().clone()
        return out

    def params_changed(self):
        """Call after writing to the canonical parameters from outside the engine."""
        self._param_version += 1

    # ------------------------------------------------------------------ layer plans
    def _t(self, *shape):
        return torch.zeros(*shape, dtype=self.dtype, device=self.device)

    def _f(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def _conv3x3_plans(self, key, B, H, cin, cout, stride):
        """zero-padded 3x3 conv (models/networks.py:246-249): forward, dgrad, wgrad."""
        T = self.dtype
        OH = H // stride
        fwd = _Gemm(self, _desc(T, B, H, H, cin, OH, OH, cout, 3, 9, stride=stride, off=(-1, -1)),
                    _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9), key)
        fwd.desc.flags = self._fwd_flags
        if stride == 1:
            dg = [_Gemm(self, _desc(T, B, H, H, cout, H, H, cin, 3, 9, off=(-1, -1)),
                        _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9, tap_mode=1), key)]
        else:
            # transposed stride-2 conv.  Preferred: one launch over the dY grid whose nine taps each feed the
            # accumulators of their own output-parity class (rows 2a+py receive kh = 1 (from dY row a) when
            # py = 0, kh = 0 (row a+1) and kh = 2 (row a) when py = 1; columns likewise)
            dt = _desc(T, B, OH, OH, cout, OH, OH, cin, 3, 9, y_dims=(H, H))
            dt.form = FORM_S2_TRANSPOSED
            if self.fused_s2_dgrad and ops.conv_fwd_kernel(dt) == "patch_dgrad_s2":
                return fwd, [_Gemm(self, dt, _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9), key)]
            # otherwise (narrow test networks): the four classes as separate gather convolutions in one launch
            dg = []
            for py in (0, 1):
                for px in (0, 1):
                    khs = [1] if py == 0 else [0, 2]
                    kws = [1] if px == 0 else [0, 2]
                    table = [kh * 3 + kw for kh in khs for kw in kws]
                    d = _desc(T, B, OH, OH, cout, OH, OH, cin, len(kws), len(table),
                              tstep=(-1, -1), off=(1 if py else 0, 1 if px else 0),
                              o_mul=(2, 2), o_off=(py, px), y_dims=(H, H))
                    dg.append(_Gemm(self, d, _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, len(table),
                                                  tap_mode=2, table=table), key))
        return fwd, dg

    def _build_encoder(self):
        B, P, ngf, T = self.B, self.P, self.ngf, self.dtype
        e = self.enc = {}
        # stem: reflection-padded NHWC4 image, 7 row-taps of 8 pixels x 4 channels (kw = 7 and
        # channel 3 carry zero weights); label planes folded into a per-class bias
        Hp, Wp = P + 6, P + 8
        table = [(kh * 7 + kw) if kw < 7 else -1 for kh in range(7) for kw in range(8)]
        e["stem"] = _Gemm(self, _desc(T, B, Hp, Wp, 32, P, P, ngf, 1, 7, x_pix=4),
                          _wmap(1, (3 + self.y) * 49, 0, 4, 1, 49, ngf, 3, 56, tap_mode=2, table=table),
                          EP + "conv.1.weight")
        e["xpad"] = self._t(B, Hp, Wp, 4)
        e["cbias"] = self._f(self.y, ngf) if self.y > 0 else None
        if self.stem_input_grad:
            # gradient w.r.t. the reflection-padded image = full correlation of d_s0 with the
            # flipped kernel (zero padding 6), channels 3.. dropped; folded back by cvg_reflect_fold
            e["stem_dg"], e["stem_dg_v"] = None, None
            if ngf % 32 == 0 and os.environ.get("CVG_STEM_DGRAD_49") is None:
                # as 7 row taps with (kernel column, input channel) as 32 "output channels", finished by a 7-column
                # sum (cvg_window_col_sum): a seventh of the multiplications of the 49-tap form below, whose GEMM has
                # 3 real output rows out of 32
                e["stem_dg_v"] = _Gemm(self, _desc(T, B, P, P, ngf, P + 6, P, 32, 1, 7, off=(-6, 0)),
                                       _wmap(4, 1, 49, ngf, 7, (3 + self.y) * 49, 28, ngf, 7, tap_mode=1, nr_valid=3),
                                       EP + "conv.1.weight")
                e["v_dg"] = self._t(B * (P + 6) * P, 32)
            else:
                e["stem_dg"] = _Gemm(self, _desc(T, B, P, P, ngf, P + 6, P + 6, 8, 7, 49, off=(-6, -6)),
                                     _wmap(1, 49, 0, ngf, 1, (3 + self.y) * 49, 3, ngf, 49, tap_mode=1),
                                     EP + "conv.1.weight")
            e["d_xp"] = self._t(B * (P + 6) * (P + 6), 8)
            e["d_x8"] = self._t(B * P * P, 8)
            e["d_img"] = self._f(B, 3, P, P)
        e["s0"] = self._t(B * P * P, ngf)
        e["a0"] = self._t(B * P * P, ngf)
        e["st0"] = self._f(2, ngf)
        e["d_a0"] = self._t(B * P * P, ngf)
        e["d_s0"] = self._t(B * P * P, ngf)
        e["blocks"] = []
        H = P
        for i, (cin, cout) in enumerate(stage_channels(self.n_stage, ngf)):
            p = EP + "conv.%d." % (4 + i)
            M = B * H * H
            blk = {"prefix": p, "H": H, "cin": cin, "cout": cout, "M": M}
            blk["conv1"], blk["dg1"] = self._conv3x3_plans(p + "res.res.0.weight", B, H, cin, cin, 1)
            blk["conv2"], blk["dg2"] = self._conv3x3_plans(p + "res.res.3.weight", B, H, cin, cin, 1)
            blk["down"], blk["dgd"] = self._conv3x3_plans(p + "conv.weight", B, H, cin, cout, 2)
            blk["dgd_group"] = (ops.ConvGroup([g.desc for g in blk["dgd"]], [g.w for g in blk["dgd"]])
                                if len(blk["dgd"]) > 1 else None)
            for name in ("c1", "b1", "c2", "r", "d_r", "d_c2", "d_b1", "d_c1"):
                blk[name] = self._t(M, cin)
            blk["st1"], blk["st2"] = self._f(2, cin), self._f(2, cin)
            Mo = B * (H // 2) * (H // 2)
            blk["out"] = self._t(Mo, cout)
            blk["d_out"] = self._t(Mo, cout)
            e["blocks"].append(blk)
            H //= 2
        s, C = self.s, stage_channels(self.n_stage, ngf)[-1][1]
        fin, fcd, nz = s * s * C, 4 * self.z, 2 * self.z
        nz_ld = _rup(nz, 8)              # row pitch of h / dh: padded when 2 * z_dim is not a multiple of 8
        e["fin"], e["fcd"], e["nz"], e["nz_ld"], e["C_last"] = fin, fcd, nz, nz_ld, C
        # Linear layers as 1x1 "convolutions" over a 1x1 image; the flatten order of the
        # reference is (c, h, w), ours (h, w, c): the weight maps carry the permutation.
        e["fc0"] = _Gemm(self, _desc(T, B, 1, 1, fin, 1, 1, fcd, 1, 1),
                         _wmap(1, fin, 0, C, 1, s * s, fcd, C, s * s), EP + "fc.0.weight")
        e["fc0_dg"] = _Gemm(self, _desc(T, B, 1, 1, fcd, 1, 1, fin, 1, 1),
                            _wmap(C, 1, s * s, fcd, 0, fin, fin, fcd, 1), EP + "fc.0.weight")
        e["fc3"] = _Gemm(self, _desc(T, B, 1, 1, fcd, 1, 1, nz_ld, 1, 1),
                         _wmap(1, fcd, 0, fcd, 0, 1, nz, fcd, 1), EP + "fc.3.weight")
        e["fc3_dg"] = _Gemm(self, _desc(T, B, 1, 1, nz_ld, 1, 1, fcd, 1, 1),
                            _wmap(1, 1, 0, nz_ld, 0, fcd, fcd, nz, 1), EP + "fc.3.weight")
        e["f0"], e["f1"] = self._t(B, fcd), self._t(B, fcd)
        e["d_f0"], e["d_f1"] = self._t(B, fcd), self._t(B, fcd)
        e["stf"] = self._f(2, fcd)
        e["h"] = self._t(B, nz_ld)         # columns from nz on: the padded bias (zero), never read
        e["dh"] = self._t(B, nz_ld)        # columns from nz on stay zero
        # padded copies of the head bias / its gradient when the rows are (the kernels read N = nz_ld entries)
        e["bias_h"] = self._f(nz_ld) if nz_ld != nz else None
        e["bgrad_h"] = self._f(nz_ld) if nz_ld != nz else None
        e["labels"] = torch.zeros(B, dtype=torch.int32, device=self.device)

    def _build_decoder(self):
        B, P, ngf, T, s = self.B, self.P, self.ngf, self.dtype, self.s
        d = self.dec = {}
        C0 = ngf * 8
        zin_ld = _rup(self.z + self.y, 8)
        F = 4 * (self.z + self.y)
        F8 = _rup(F, 8)
        nout = s * s * C0
        d.update(zin_ld=zin_ld, F=F, F8=F8, nout=nout, C0=C0)
        d["zin"] = self._t(B, zin_ld)
        d["dzin"] = self._t(B, zin_ld)
        d["fc0"] = _Gemm(self, _desc(T, B, 1, 1, zin_ld, 1, 1, F8, 1, 1),
                         _wmap(1, self.z + self.y, 0, zin_ld, 0, 1, F, self.z + self.y, 1), DP + "fc.0.weight")
        d["fc0_dg"] = _Gemm(self, _desc(T, B, 1, 1, F8, 1, 1, zin_ld, 1, 1),
                            _wmap(1, 1, 0, F8, 0, self.z + self.y, self.z + self.y, F, 1), DP + "fc.0.weight")
        d["fc3"] = _Gemm(self, _desc(T, B, 1, 1, F8, 1, 1, nout, 1, 1),
                         _wmap(C0, F, s * s * F, F8, 0, 1, nout, F, 1), DP + "fc.3.weight")
        d["fc3_dg"] = _Gemm(self, _desc(T, B, 1, 1, nout, 1, 1, F8, 1, 1),
                            _wmap(1, 1, 0, C0, F, s * s * F, F, C0, s * s), DP + "fc.3.weight")
        d["bias0"] = self._f(F8)
        d["bias3"] = self._f(nout)
        d["g0"], d["g1"] = self._t(B, F8), self._t(B, F8)
        d["d_g0"], d["d_g1"] = self._t(B, F8), self._t(B, F8)
        d["mask"] = torch.ones(B, F8, dtype=torch.uint8, device=self.device)
        d["x0"] = self._t(B * s * s, C0)
        d["d_x0"] = self._t(B * s * s, C0)
        d["ups"] = []
        h = s
        for i, (cin, cout) in enumerate(up_channels(self.n_stage, ngf)):
            key = DP + "conv.%d.2." % i
            U = 2 * h + 2
            up = {"key": key, "h": h, "cin": cin, "cout": cout}
            # valid 3x3 conv on the up-sampled, reflection-padded tensor U
            if not (self.lowres_up_fwd and self.lowres_up_bwd):
                up["fwd"] = _Gemm(self, _desc(T, B, U, U, cin, 2 * h, 2 * h, cout, 3, 9),
                                  _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9), key + "weight")
                up["fwd"].desc.flags = self._fwd_flags
            # dU = full correlation of dY with the flipped kernel (zero padding 2)
            if not self.lowres_up_bwd:
                up["dg"] = _Gemm(self, _desc(T, B, 2 * h, 2 * h, cout, U, U, cin, 3, 9, off=(-2, -2)),
                                 _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9, tap_mode=1), key + "weight")
            if not (self.lowres_up_fwd and self.lowres_up_bwd):
                up["u"] = self._t(B * U * U, cin)
            up["y"] = self._t(B * 4 * h * h, cout)
            up["d_y"] = self._t(B * 4 * h * h, cout)
            if self.lowres_up_fwd:
                # Forward at LOW resolution (see cvg_upsample_taps_fwd): v = x . W' with 9 cout columns, then the
                # spatial operator; the up-sampled tensor is never built
                up["v"] = self._t(B * h * h, 9 * cout)
                up["v_gemm"] = _Gemm(self, _desc(T, B, h, h, cin, h, h, 9 * cout, 1, 1),
                                     _wmap(cout, 1, cin * 9, cin, 0, 9, 9 * cout, cin, 1), key + "weight")
            if self.lowres_up_bwd:
                # Backward at LOW resolution (see cvg_upsample_taps_bwd): z holds, per input pixel, the nine
                # tap-shifted transposed up-samplings of d_y; the input gradient is then a one-tap GEMM
                # d_x = z . W' with K = 9 cout, the weight gradient dW' = z^T x, both over B h h rows instead of
                # B (2h + 2)^2 / B (2h)^2 rows with the same K: a quarter of the multiplications, and neither
                # d_u nor the transposed up-sampling pass.
                up["z"] = self._t(B * h * h, 9 * cout)
                up["dg_low"] = _Gemm(self, _desc(T, B, h, h, 9 * cout, h, h, cin, 1, 1),
                                     _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9), key + "weight")
                up["wg_low"] = _WgradOnly(self, _desc(T, B, h, h, cin, h, h, 9 * cout, 1, 1),
                                          _wmap(cout, 1, cin * 9, cin, 0, 9, 9 * cout, cin, 1), key + "weight")
                d["ups"].append(up)
                h *= 2
                continue
            up["d_u"] = self._t(B * U * U, cin)
            d["ups"].append(up)
            h *= 2
        key = DP + "conv.%d." % (self.n_stage + 1)
        d["tail_key"] = key
        d["tail"] = _Gemm(self, _desc(T, B, P, P, ngf, P, P, 4, 7, 49, off=(-3, -3), pad_mode=PAD_REFLECT),
                          _wmap(1, ngf * 49, 0, ngf, 1, 49, 3, ngf, 49), key + "weight")
        # input gradient w.r.t. the reflection-padded tensor = full correlation of dpre with the
        # flipped kernel; dpre sits zero-padded by 6 in a 4-channel image so that, like the
        # stem, it is 7 row taps of 8-pixel x 4-channel windows
        table = [((6 - kh) * 7 + (6 - kw)) if kw < 7 else -1 for kh in range(7) for kw in range(8)]
        d["tail_dg"] = _Gemm(self, _desc(T, B, P + 12, P + 14, 32, P + 6, P + 6, ngf, 1, 7, x_pix=4),
                             _wmap(1, 49, 0, 4, 1, ngf * 49, ngf, 3, 56, tap_mode=2, table=table), key + "weight")
        d["dpre_win"] = self._t(B, P + 12, P + 14, 4)
        # weight gradient of the tail through the same windows: with x' = x + kw,
        #   dW[n][c][kh][kw] = sum_{y,x'} win[y][x'][(6 - kw, n)] * Xpad[y + kh][x'][c]
        # i.e. 7 row taps over the reflection-padded input and 32 "output channels" (window pixel,
        # n) instead of 49 taps with 3 real output rows out of 32: a seventh of the MFMA work.
        # dY = rows 6.. of the window image, read through the descriptor's output pitches.
        Wp = P + 14
        d["tail_wg_desc"] = _desc(T, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT,
                                  y_pix=4, y_row=Wp * 4, y_img=(P + 12) * Wp * 4)
        d["tail_wg_map"] = _wmap(4, -1, ngf * 49, ngf, 1, 49, 28, ngf, 7, tap_mode=2,
                                 table=[kh * 7 + 6 for kh in range(7)], nr_valid=3)
        d["tail_wg_dy"] = d["dpre_win"][:, 6:]
        # forward the same way round: per padded column x', the 7 kernel rows for every kernel column
        # kw as 32 output channels (kw, n); cvg_tail_sum_tanh adds the 7 columns of each pixel
        d["tail_v"] = None
        if ngf % 32 == 0:
            d["tail_v"] = _Gemm(self, _desc(T, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT),
                                _wmap(4, 1, ngf * 49, ngf, 7, 49, 28, ngf, 7, nr_valid=3), key + "weight")
            d["v"] = self._t(B * P * (P + 6), 32)
        # (narrow networks, ngf < 32, keep the 49-tap form on the gather kernel)
        d["tail_wg_windows"] = ops.conv_wgrad_kernel(d["tail_wg_desc"]).startswith("patch_wgrad")
        if d["tail_wg_windows"]:
            self._wgrad_descs.append(d["tail_wg_desc"])
        d["tail_bias"] = self._f(4)
        d["pre"] = self._t(B * P * P, 4)
        d["dpre"] = self._t(B * P * P, 8)
        d["dxp"] = self._t(B * (P + 6) * (P + 6), ngf)
        d["dec"] = self._f(B, 3, P, P)
        d["d_dec"] = self._f(B, 3, P, P)
        d["bgrad8"] = self._f(8)
        d["bgradF8"] = self._f(F8)
        self.eps = self._f(B, self.z)
        self.z_mean, self.z_sigma = self._f(B, self.z), self._f(B, self.z)
        self.losses = self._f(4)          # [kl, ll, bad-label flag, -]
        self.step_dev = torch.zeros(1, dtype=torch.long, device=self.device)     # AdamW step count
        self.seed_dev = torch.full((1,), self._dropout_start, dtype=torch.long, device=self.device)   # dropout stream position
        self.bc_dev = self._f(2)

    def _build_workspaces(self):
        # backward's weight-gradient launches alternate between two side streams (CVG_SIDE_STREAMS): two of
        # them run side by side, each planned for half of the CUs, which halves their split count and with
        # it the partial-slab traffic (measured -0.17 ms per c3 step against one side stream)
        self._n_side = max(1, int(os.environ.get("CVG_SIDE_STREAMS", "2"))) if self.overlap else 1
        wg = 0
        for share in sorted({1, self._n_side}):          # workspaces fit either plan (see set_co_launches)
            self.set_co_launches(share)
            for g in self._gemms:
                wg = max(wg, ops.conv_wgrad_workspace(g.desc))
            for desc in self._wgrad_descs:
                wg = max(wg, ops.conv_wgrad_workspace(desc))
        self.ws = torch.zeros(max(wg, self._ws_bytes, 1 << 20) // 4 + 16, dtype=torch.float32, device=self.device)
        max_c = max(self.ngf * 8, 4 * self.z)
        if self.dec is not None:
            max_c = max(max_c, self.dec["nout"], self.dec["F8"])
        # reduction workspaces start with WS_HEADER arrival counters (zero here, left at zero by every kernel that
        # counts in them); kernels that only need scratch rows get the part behind them (red_rows)
        self.red_ws = torch.zeros(WS_HEADER + 2 * 1024 * max(max_c, 1024), dtype=torch.float32, device=self.device)
        # the weight-gradient branches of backward run on side streams with workspaces of their own
        self._ws_sides = [torch.zeros(max(wg, 1 << 20) // 4 + 16, dtype=torch.float32, device=self.device)
                          for _ in range(self._n_side)]
        self._red_ws_sides = [torch.zeros_like(self.red_ws) for _ in range(self._n_side)]
        self._side_i = 0
        self._bn_scratch = (self._f(max_c), self._f(max_c))

    def set_co_launches(self, n):
        """Plan every weight-gradient launch for 1/n of the CUs (n = the number of side streams while they
        overlap; 1 for a driver that runs the step on one stream, e.g. to time kernels one by one)."""
        for desc in [g.desc for g in self._gemms] + list(self._wgrad_descs):
            desc.co_launches = n

    # ------------------------------------------------------------------ helpers
    def pack_weights(self, force=False, early_first=False):
        """Rebuild the packed shadows if the parameters changed.  early_first (enc_forward): only the layouts of the
        stem and the first block now; the rest (nearly all of the bytes: an HBM-bound pass) is left to _pack_rest(),
        which the caller invokes once the stem's launches are enqueued, on another stream beside them, and
        _pack_wait() before the first launch that reads one of those layouts."""
        if not force and self._packed_version == self._param_version:
            return
        if self._pack_batch is None:
            early = lambda g: g.key.startswith(EP + "conv.1.") or g.key.startswith(EP + "conv.4.")
            item = lambda g: (g.desc, g.wmap, self.params[g.key], g.w)
            first = [item(g) for g in self._gemms if early(g)]
            rest = [item(g) for g in self._gemms if not early(g)]
            self._pack_batch = (ops.PackBatch(first, self.device) if first else None,
                                ops.PackBatch(rest, self.device) if rest else None)
        if self._pack_batch[0] is not None:
            self._pack_batch[0].run()
        if self.y > 0:
            ops.label_bias(self.params[EP + "conv.1.weight"], self.enc["cbias"], self.ngf, self.y)
        self._packed_version = self._param_version
        if early_first and self.overlap and self._split_pack:
            self._pack_fork = torch.cuda.Event()
            self._pack_fork.record()
            return
        self._pack_rest_launches()

    def _pack_rest_launches(self):
        if self._pack_batch[1] is not None:
            self._pack_batch[1].run()
        if getattr(self, "enc", None) and self.enc.get("bias_h") is not None:
            ops.permute_f32(self.params[EP + "fc.3.bias"], self.enc["bias_h"], self.enc["nz"])
        d = self.dec
        if d is not None:
            ops.permute_f32(self.params[DP + "fc.0.bias"], d["bias0"], d["F"])
            ops.permute_f32(self.params[DP + "fc.3.bias"], d["bias3"], d["nout"], d["C0"], 1, self.s * self.s)
            ops.permute_f32(self.params[d["tail_key"] + "bias"], d["tail_bias"], 3)

    def _pack_rest(self):
        if self._pack_fork is None:
            return
        if self._pack_stream is None:
            self._pack_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._pack_stream):
            self._pack_stream.wait_event(self._pack_fork)
            self._pack_rest_launches()
            self._pack_done = torch.cuda.Event()
            self._pack_done.record()
        self._pack_fork = None

    def _pack_wait(self):
        self._pack_rest()
        if self._pack_done is not None:
            torch.cuda.current_stream().wait_event(self._pack_done)
            self._pack_done = None

    def _bn(self, x, res, y, M, C, stats, prefix, partial_rows=0):
        """partial_rows > 0: self.red_ws already holds that many rows of column sums / sums of squares of x
        (left there by the convolution that produced x)."""
        if self.training and partial_rows:
            ops.bn_stats_finalize(self.red_rows, partial_rows, M, C, stats, self.buffers[prefix + ".running_mean"],
                                  self.buffers[prefix + ".running_var"], self.buffers[prefix + ".num_batches_tracked"])
        elif self.training:
            ops.bn_stats(x, M, C, stats, self.buffers[prefix + ".running_mean"],
                         self.buffers[prefix + ".running_var"], self.buffers[prefix + ".num_batches_tracked"],
                         self.red_ws)
        else:
            ops.bn_eval_stats(self.buffers[prefix + ".running_mean"], self.buffers[prefix + ".running_var"], C, stats)
        ops.bn_act_fwd(x, res, y, M, C, stats, self.params[prefix + ".weight"], self.params[prefix + ".bias"])

    def _conv_bn(self, g, x, conv_out, res, y, M, C, stats, prefix, bias=None, **kw):
        """conv -> BatchNorm(+res) -> LeakyReLU.  Training: the convolution, then batch statistics and
        the apply pass.  Inference: running statistics fold into one multiply-add per channel, applied
        in the convolution's epilogue (no normalisation pass, conv_out is not written)."""
        if not self.training and ops.conv_fwd_kernel(g.desc).startswith("patch_fwd"):
            sc, sh = self._bn_scratch[0][:C], self._bn_scratch[1][:C]
            ops.bn_eval_scale_shift(self.buffers[prefix + ".running_mean"], self.buffers[prefix + ".running_var"],
                                    self.params[prefix + ".weight"], self.params[prefix + ".bias"], C, sc, sh)
            self._fwd_folded(g, x, y, sc, sh, bias=bias, res=res, **kw)
            return
        nblk = self._stat_tiles(g)
        if self.training and nblk:
            # opt-in (see _stat_tiles): the convolution's epilogue leaves per-tile column sums of what it
            # stores, so the statistics need no pass over conv_out
            assert 2 * nblk * C <= self.red_rows.numel()
            self._fwd(g, x, conv_out, bias=bias, stat_partials=self.red_rows, **kw)
            self._bn(conv_out, res, y, M, C, stats, prefix, partial_rows=nblk)
            return
        self._fwd(g, x, conv_out, bias=bias, **kw)
        self._bn(conv_out, res, y, M, C, stats, prefix)

    def _stat_tiles(self, g):
        """Partial rows of the statistics-emitting convolution for this layer, 0 = use the separate pass.
        Off unless CVG_BN_EPILOGUE_STATS is set: measured at 64x64 / batch 256 in bf16 the step takes 7.73 ms
        with it against 7.65 ms without.  The sums cost the 64-channel convolutions 12-15 us each (a barrier,
        an LDS round and a dependent store per 256-pixel tile of a power-limited kernel) and the finalize over
        4096 tile rows is slower than over 512 block rows, which together outweigh the 20 us statistics pass."""
        if os.environ.get("CVG_BN_EPILOGUE_STATS") is None:
            return 0
        n = getattr(g, "_stat_tiles", None)
        if n is None:
            n = g._stat_tiles = ops.conv_stats_partials(g.desc)
        return n

    def _bn_bwd(self, dy, y, x, dx, M, C, stats, prefix, gated=False):
        """y=None for a BatchNorm without a residual input: the gate is recomputed from x;
        gated=True: dy is already the gradient of the BatchNorm's output."""
        if self._frozen:        # parameters are constants in this pass: dgamma / dbeta only serve dx
            dg, db = self._bn_scratch[0][:C], self._bn_scratch[1][:C]
        else:
            dg, db = self.grads[prefix + ".weight"], self.grads[prefix + ".bias"]
        ops.bn_act_bwd(dy, y, x, dx, M, C, stats, self.params[prefix + ".weight"], dg, db, self.red_ws,
                       beta=None if gated else self.params[prefix + ".bias"])

    def _ready(self, stage):
        """Report that every gradient of `stage` has been enqueued.  Part of them may still be running on the
        weight-gradient side streams: the listener gets one event per side stream to wait for, on whatever
        stream it exchanges on, and this (the input-gradient) stream goes on without joining them."""
        if self.grad_ready is not None:
            events = []
            if self._side_dirty:
                for side in self._side:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    events.append(ev)
            self.grad_ready(stage, events)

    def _fwd(self, g, x, y, bias=None, stat_partials=None, **kw):
        if stat_partials is not None:
            ops.conv_fwd_stats(g.desc, x, g.w, y, stat_partials, bias=bias, **kw)
        else:
            ops.conv_fwd(g.desc, x, g.w, y, bias=bias, ws=self.ws, **kw)

    def _fwd_folded(self, g, x, y, scale, shift, bias=None, res=None, **kw):
        """Inference: the convolution with BatchNorm's running statistics (one multiply-add per channel), the residual
        and LeakyReLU in its epilogue."""
        ops.conv_bn_act_fwd(g.desc, x, g.w, y, scale, shift, bias=bias, res=res, **kw)

    def _wgrad(self, g, x, dy, dy_ld):
        if self._frozen:
            return
        ops.conv_wgrad(g.desc, x, dy, dy_ld, g.wmap, self.grads[g.key], self.ws_side)

    def _fork(self, chain, params, site="blk"):
        """chain(): the next launch of the input-gradient chain; params(): parameter-gradient launches (weight-gradient
        GEMMs, bias column sums) that depend on what is enqueued so far and run on a side stream.  The chain's launch
        is ENQUEUED first and the side stream then waits for an event recorded before it: in a captured graph the
        node that is captured first stays on its parent's queue, and the other one starts behind a cross-queue signal
        (about 10 us with nothing in flight).  With the side launches captured first that delay sits on the chain.
        Measured per group of sites (`site`), because the gain is not uniform: at the two stride-1 weight gradients of
        each encoder block ("blk") the chain-first order takes 0.14 ms off a c3 step; at the stride-2 layers ("down")
        it changes nothing, at the decoder's up-convolutions ("low") it costs 0.05 ms, at the tail 0.2 ms and at the
        Linear layers ("head", "fc") 0.5 ms: there the side launches are what the join at the end of the decoder's
        pass waits for.  CVG_FORK_SITES lists the groups that use the chain-first order (default "blk")."""
        if not self.overlap or not (site in self._fork_sites or site.rstrip("0123456789") in self._fork_sites):
            with self._branch():
                params()
            chain()
            return
        ev = torch.cuda.Event()
        ev.record()
        chain()
        with self._branch(after=ev):
            params()

    @contextlib.contextmanager
    def _branch(self, after=None):
        """Launches inside run on the side stream, after everything enqueued so far on the current
        stream (or up to the event `after`, recorded on it earlier).  Backward uses it for parameter gradients:
        nothing downstream in backward reads them, so they overlap the input-gradient chain; _join() makes the
        current stream wait for them."""
        if not self.overlap:
            yield
            return
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self._n_side)]
        self._side_i = (self._side_i + 1) % self._n_side
        side = self._side[self._side_i]
        ev = after
        if ev is None:
            ev = torch.cuda.Event()
            ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            yield
        self._side_dirty = True

    @property
    def ws_side(self):
        return self._ws_sides[self._side_i]

    @property
    def red_ws_side(self):
        return self._red_ws_sides[self._side_i]

    @property
    def red_rows(self):
        return self.red_ws[WS_HEADER:]

    def _join(self):
        if self._side_dirty:
            for side in self._side:
                ev = torch.cuda.Event()
                ev.record(side)
                torch.cuda.current_stream().wait_event(ev)
            self._side_dirty = False

    # ------------------------------------------------------------------ encoder
    def enc_forward(self, x, labels):
        """x: f32 NCHW [B,3,P,P] on the device; labels: int32 [B].  Returns h [B, 2z]."""
        e, B, P, ngf = self.enc, self.B, self.P, self.ngf
        if self.training and B == 1:
            # what nn.BatchNorm1d (models/networks.py:188) raises in train mode for one sample
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             "torch.Size([1, %d])" % e["fcd"])
        self.pack_weights(early_first=True)
        if self.y > 0:
            # class indices address the per-class bias table, its gradient rows and the one-hot
            # columns of the decoder input: kept inside [0, y) so that a bad index cannot leave
            # those buffers; losses[2] records that one was outside and read_losses() raises
            assert labels.numel() == B
            if labels.dtype != torch.int32 or not labels.is_contiguous():
                labels = labels.to(torch.int32).contiguous()
            ops.labels_prepare(labels, e["labels"], B, self.y, self.losses[2:3])
        ops.pack_image(x, e["xpad"], B, P)
        self._conv_bn(e["stem"], e["xpad"], e["s0"], None, e["a0"], B * P * P, ngf, e["st0"], EP + "conv.2",
                      bias=self.params[EP + "conv.1.bias"], cbias=e["cbias"],
                      labels=e["labels"] if self.y > 0 else None)
        self._pack_rest()                    # the other layouts are repacked beside the stem ...
        xin = e["a0"]
        for i, blk in enumerate(e["blocks"]):
            if i == 1:
                self._pack_wait()            # ... and are first needed here
            p, M, C = blk["prefix"], blk["M"], blk["cin"]
            self._conv_bn(blk["conv1"], xin, blk["c1"], None, blk["b1"], M, C, blk["st1"], p + "res.res.1")
            self._conv_bn(blk["conv2"], blk["b1"], blk["c2"], xin, blk["r"], M, C, blk["st2"], p + "res.res.4")
            self._fwd(blk["down"], blk["r"], blk["out"])
            blk["xin"] = xin
            xin = blk["out"]
        self._pack_wait()
        self._fwd(e["fc0"], xin, e["f0"], bias=self.params[EP + "fc.0.bias"])
        self._bn(e["f0"], None, e["f1"], B, e["fcd"], e["stf"], EP + "fc.1")
        self._fwd(e["fc3"], e["f1"], e["h"], bias=e["bias_h"] if e["bias_h"] is not None else self.params[EP + "fc.3.bias"])
        return e["h"]

    def enc_backward(self, dh, param_grads=True, input_grad=False, part=None):
        """param_grads=False: the parameters are constants in this pass (a critic judging the
        generator): no weight / bias gradients are produced and self.grads is left alone.
        input_grad=True (needs stem_input_grad at construction): also returns dL/dx as f32 NCHW.
        part: None = the whole pass; "upper" = the head and the deep blocks (index >= enc_split),
        which hold nearly all of the encoder's parameters; "lower" = the remaining blocks and the
        stem.  A data-parallel driver exchanges the upper gradients while "lower" runs."""
        self._frozen = not param_grads
        try:
            return self._enc_backward(dh, input_grad, part)
        finally:
            self._frozen = False

    @property
    def enc_split(self):
        return len(self.enc["blocks"]) // 2

    def arena_range(self, prefix):
        """[lo, hi) of the arena covered by the parameters whose stage (cvae_gan_amd.ddp.stage_of) is `prefix`."""
        from .ddp import stage_ranges
        return stage_ranges(self.arena_order, self.arena_offsets, self.arena_numel, self.n_stage)[prefix]

    def enc_upper_range(self):
        """[lo, hi) of the gradient arena written by enc_backward(part="upper")."""
        lo = min(self.arena_offsets[k] for k in self.arena_order if k.startswith(EP + "fc."))
        first_lower = EP + "conv.%d." % (4 + self.enc_split - 1)
        hi = min(self.arena_offsets[k] for k in self.arena_order if k.startswith(first_lower))
        return lo, hi

    def _enc_backward(self, dh, input_grad, part=None):
        e, B, P, ngf = self.enc, self.B, self.P, self.ngf
        frozen = self._frozen
        last = e["blocks"][-1]
        if part != "lower":
            self._enc_backward_head(dh)
        hi = len(e["blocks"]) - 1 if part != "lower" else self.enc_split - 1
        lo = 0 if part != "upper" else self.enc_split
        for i in range(hi, lo - 1, -1):
            self._enc_backward_block(i)
        if part == "upper":
            self._join()
            return None
        return self._enc_backward_stem(input_grad)

    def _enc_backward_head(self, dh):
        e, B = self.enc, self.B
        frozen = self._frozen
        last = e["blocks"][-1]
        def fc3_params():
            self._wgrad(e["fc3"], e["f1"], dh, e["nz_ld"])
            if not frozen and e["bgrad_h"] is None:
                ops.colsum(dh, B, e["nz"], e["nz"], self.grads[EP + "fc.3.bias"], self.red_ws_side)
            elif not frozen:
                ops.colsum(dh, B, e["nz_ld"], e["nz_ld"], e["bgrad_h"], self.red_ws_side)
                ops.permute_f32(e["bgrad_h"], self.grads[EP + "fc.3.bias"], e["nz"])

        def fc0_params():
            self._wgrad(e["fc0"], last["out"], e["d_f0"], e["fcd"])
            if not frozen:
                ops.colsum(e["d_f0"], B, e["fcd"], e["fcd"], self.grads[EP + "fc.0.bias"], self.red_ws_side)

        self._fork(lambda: self._fwd(e["fc3_dg"], dh, e["d_f1"]), fc3_params, "head")
        self._bn_bwd(e["d_f1"], None, e["f0"], e["d_f0"], B, e["fcd"], e["stf"], EP + "fc.1")
        self._fork(lambda: self._fwd(e["fc0_dg"], e["d_f0"], last["d_out"]), fc0_params, "head")
        self._ready("enc.head")

    def _enc_backward_block(self, i):
        e = self.enc
        blk = e["blocks"][i]
        p, M, C = blk["prefix"], blk["M"], blk["cin"]
        # d_r: gradient of the block's pre-activation sum (the stride-2 input gradient times
        # LeakyReLU'(r), applied as that kernel stores): r is read once, not by all three consumers
        def down_dgrad():
            if blk["dgd_group"] is None:
                self._fwd(blk["dgd"][0], blk["d_out"], blk["d_r"], gate=blk["r"])
            else:
                blk["dgd_group"].run(blk["d_out"], blk["d_r"], gate=blk["r"])

        self._fork(down_dgrad, lambda: self._wgrad(blk["down"], blk["r"], blk["d_out"], blk["cout"]), "down%d" % i)
        self._bn_bwd(blk["d_r"], None, blk["c2"], blk["d_c2"], M, C, blk["st2"], p + "res.res.4", gated=True)
        self._fork(lambda: self._fwd(blk["dg2"][0], blk["d_c2"], blk["d_b1"]),
                   lambda: self._wgrad(blk["conv2"], blk["b1"], blk["d_c2"], C))
        self._bn_bwd(blk["d_b1"], None, blk["c1"], blk["d_c1"], M, C, blk["st1"], p + "res.res.1")
        d_xin = e["d_a0"] if i == 0 else e["blocks"][i - 1]["d_out"]
        # d_xin = conv1^T(d_c1) + d_r   (the residual branch)
        self._fork(lambda: self._fwd(blk["dg1"][0], blk["d_c1"], d_xin, res=blk["d_r"]),
                   lambda: self._wgrad(blk["conv1"], blk["xin"], blk["d_c1"], C))
        self._ready("enc.blk%d" % i)

    def _enc_backward_stem(self, input_grad):
        e, B, P, ngf = self.enc, self.B, self.P, self.ngf
        frozen = self._frozen
        self._bn_bwd(e["d_a0"], None, e["s0"], e["d_s0"], B * P * P, ngf, e["st0"], EP + "conv.2")
        with self._branch():
            self._wgrad(e["stem"], e["xpad"], e["d_s0"], ngf)
            if not frozen:
                ops.label_bias_bwd(e["d_s0"], e["labels"], B, P * P, ngf, self.y, self.grads[EP + "conv.1.weight"],
                                   self.grads[EP + "conv.1.bias"], self.red_ws_side[WS_HEADER:])
        d_img = None
        if input_grad:
            if e["stem_dg_v"] is not None:
                self._fwd(e["stem_dg_v"], e["d_s0"], e["v_dg"])
                ops.window_col_sum(e["v_dg"], e["d_xp"], B, P + 6, P)
            else:
                self._fwd(e["stem_dg"], e["d_s0"], e["d_xp"])
            ops.reflect_fold(e["d_xp"], e["d_x8"], B, P, P, 8, 3)
            ops.nhwc_to_nchw(e["d_x8"], e["d_img"], B, P * P, 3, 8)
            d_img = e["d_img"]
        self._ready("enc.stem")
        self._join()
        return d_img

    # ------------------------------------------------------------------ sampler
    def sample(self, h, eps, batch_size):
        d = self.dec
        ops.reparam_kl_fwd(h, eps, self.enc["labels"], self.B, self.z, self.y, d["zin_ld"], d["zin"],
                           self.z_mean, self.z_sigma, self.losses[0:1], batch_size, h_ld=self.enc["nz_ld"])
        return d["zin"]

    def sample_backward(self, dzin, eps, g_mean, g_sigma, kl_scale, batch_size):
        ops.reparam_kl_bwd(dzin, self.dec["zin_ld"], eps, self.z_mean, self.z_sigma, g_mean, g_sigma,
                           kl_scale, batch_size, self.B, self.z, self.enc["dh"], h_ld=self.enc["nz_ld"])
        return self.enc["dh"]

    def set_decoder_input(self, z, y_vec):
        """zin = [z | y_vec | 0] from f32 tensors (decoder.forward's torch.cat, networks.py:64)."""
        d = self.dec
        buf = torch.zeros(self.B, d["zin_ld"], dtype=torch.float32, device=self.device)
        buf[:, :self.z] = z
        buf[:, self.z:self.z + self.y] = y_vec
        ops.cast(buf, d["zin"], buf.numel())
        return d["zin"]

    # ------------------------------------------------------------------ decoder
    def dec_forward(self, zin, mask=None):
        """zin [B, zin_ld]; mask: uint8 keep mask [B, F8] (training) or None (eval).
        Returns dec f32 NCHW [B,3,P,P]."""
        d, B, P, s = self.dec, self.B, self.P, self.s
        self.pack_weights()
        self._pack_wait()
        self._fwd(d["fc0"], zin, d["g0"], bias=d["bias0"])
        ops.lrelu_dropout_fwd(d["g0"], mask if self.training else None, d["g1"], B * d["F8"])
        self._fwd(d["fc3"], d["g1"], d["x0"], bias=d["bias3"])
        x = d["x0"]
        for up in d["ups"]:
            if self.lowres_up_fwd:
                self._fwd(up["v_gemm"], x, up["v"])
                ops.upsample_taps_fwd(up["v"], self.params[up["key"] + "bias"], up["y"], B, up["h"], up["h"], up["cout"])
                if self.training and not self.lowres_up_bwd:        # (A/B) the weight gradient on the up-sampled grid reads u
                    ops.upsample_pad_fwd(x, up["u"], B, up["h"], up["h"], up["cin"])
            else:
                ops.upsample_pad_fwd(x, up["u"], B, up["h"], up["h"], up["cin"])
                self._fwd(up["fwd"], up["u"], up["y"], bias=self.params[up["key"] + "bias"])
            x = up["y"]
        if d["tail_v"] is not None:
            self._fwd(d["tail_v"], x, d["v"])
            ops.tail_sum_tanh(d["v"], self.params[d["tail_key"] + "bias"], d["pre"], d["dec"], B, P)
        else:
            self._fwd(d["tail"], x, d["pre"], bias=d["tail_bias"])
            ops.tanh_fwd(d["pre"], d["dec"], B, P)
        return d["dec"]

    def dec_backward(self, d_dec, mask=None):
        d, B, P, s, ngf = self.dec, self.B, self.P, self.s, self.ngf
        ops.tanh_bwd(d_dec, d["dec"], d["dpre"], B, P, d["dpre_win"])
        last = d["ups"][-1]
        def tail_params():
            if d["tail_wg_windows"]:
                ops.conv_wgrad_strided(d["tail_wg_desc"], last["y"], d["tail_wg_dy"], d["tail_wg_map"],
                                       self.grads[d["tail_key"] + "weight"], self.ws_side)
            else:
                self._wgrad(d["tail"], last["y"], d["dpre"], 8)
            ops.colsum(d["dpre"], B * P * P, 8, 8, d["bgrad8"], self.red_ws_side)
            ops.permute_f32(d["bgrad8"], self.grads[d["tail_key"] + "bias"], 3)

        self._fork(lambda: self._fwd(d["tail_dg"], d["dpre_win"], d["dxp"]), tail_params, "tail")
        ops.reflect_fold(d["dxp"], last["d_y"], B, P, P, ngf, 3)
        for i in range(len(d["ups"]) - 1, -1, -1):
            up = d["ups"][i]
            M = B * 4 * up["h"] * up["h"]
            d_x = d["d_x0"] if i == 0 else d["ups"][i - 1]["d_y"]
            if self.lowres_up_bwd:
                x_in = d["x0"] if i == 0 else d["ups"][i - 1]["y"]
                ops.upsample_taps_bwd(up["d_y"], up["z"], B, up["h"], up["h"], up["cout"])

                def low_params(up=up, x_in=x_in):
                    self._wgrad(up["wg_low"], x_in, up["z"], 9 * up["cout"])
                    # the bias gradient sum_p d_y[p] from z as well: the up-sampling weights of every position sum
                    # to 1, so sum_q z_t[q] = sum_p d_y[p] for every tap t; the centre tap's columns of z are a
                    # quarter of the rows of d_y
                    if not self._frozen:
                        N = up["cout"]
                        ops.colsum(up["z"].view(-1)[4 * N:], B * up["h"] * up["h"], N, 9 * N,
                                   self.grads[up["key"] + "bias"], self.red_ws_side)

                self._fork(lambda up=up, d_x=d_x: self._fwd(up["dg_low"], up["z"], d_x), low_params, "low%d" % i)
                continue
            def hires_params(up=up, M=M):
                self._wgrad(up["fwd"], up["u"], up["d_y"], up["cout"])
                ops.colsum(up["d_y"], M, up["cout"], up["cout"], self.grads[up["key"] + "bias"], self.red_ws_side)

            self._fork(lambda up=up: self._fwd(up["dg"], up["d_y"], up["d_u"]), hires_params, "low")
            ops.upsample_pad_bwd(up["d_u"], d_x, B, up["h"], up["h"], up["cin"])

        def fc3_params():
            self._wgrad(d["fc3"], d["g1"], d["d_x0"], d["nout"])
            ops.colsum(d["d_x0"], B, d["nout"], d["nout"], self.grads[DP + "fc.3.bias"], self.red_ws_side,
                       d["C0"], 1, s * s)

        def fc0_params():
            self._wgrad(d["fc0"], d["zin"], d["d_g0"], d["F8"])
            ops.colsum(d["d_g0"], B, d["F8"], d["F8"], d["bgradF8"], self.red_ws_side)
            ops.permute_f32(d["bgradF8"], self.grads[DP + "fc.0.bias"], d["F"])

        self._fork(lambda: self._fwd(d["fc3_dg"], d["d_x0"], d["d_g1"]), fc3_params, "fc")
        ops.lrelu_dropout_bwd(d["d_g1"], d["g1"], mask if self.training else None, d["d_g0"], B * d["F8"])
        self._fork(lambda: self._fwd(d["fc0_dg"], d["d_g0"], d["dzin"]), fc0_params, "fc")
        self._ready("dec")
        # (this join stays even though enc_backward() joins again at its end: without it the captured step is
        # 0.4 ms SLOWER, 6.83 against 6.42 ms; the runtime places a graph's branches on its queues by their shape)
        self._join()
        return d["dzin"]

    # ------------------------------------------------------------------ fused step
    def set_mask(self, keep):
        """keep: [B, F] 0/1 tensor (any dtype) -> padded uint8 device mask."""
        d = self.dec
        d["mask"].zero_()
        d["mask"][:, :d["F"]] = keep.to(self.device).to(torch.uint8)

    def random_mask(self, seed=None):
        """Fresh Bernoulli(0.5) keep mask.  seed=None draws from the device-side counter
        (advances on every call, also under HIP-graph replay)."""
        d = self.dec
        if seed is None:
            ops.dropout_mask_dev(d["mask"], d["mask"].numel(), self.seed_dev)
        else:
            ops.dropout_mask(d["mask"], d["mask"].numel(), seed)

    def forward_backward(self, x, labels, eps, batch_size, loss_scale=1.0):
        """Forward + loss + backward of models/cvae.py:132-152 with every reduction on
        the device.  x f32 NCHW, labels int32, eps f32 [B,z]; the dropout mask is the
        one last set with set_mask()/random_mask().  Gradients land in self.grads;
        self.losses = [KL, LL]."""
        dec = self.forward_backward_decoder(x, labels, eps, batch_size, loss_scale)
        self.enc_backward(self.enc["dh"])
        return dec

    def forward_backward_decoder(self, x, labels, eps, batch_size, loss_scale=1.0):
        """First part of forward_backward(): forward, losses and the backward pass down to the
        encoder output (every decoder gradient is final, enc["dh"] holds dL/dh).  A data-parallel
        driver can start the decoder's gradient exchange here and run enc_backward(enc["dh"])
        meanwhile."""
        d = self.dec
        h = self.enc_forward(x, labels)
        zin = self.sample(h, eps, batch_size)
        dec = self.dec_forward(zin, d["mask"])
        ops.mse_fwd_bwd(dec, x, d["d_dec"], self.losses[1:2], dec.numel(), batch_size, loss_scale, self.red_rows)
        dzin = self.dec_backward(d["d_dec"], d["mask"])
        self.sample_backward(dzin, eps, None, None, loss_scale, batch_size)
        return dec

    # ------------------------------------------------------------------ inference
    def reconstruct(self, x, labels, eps):
        """Forward only: decoder(sample(encoder(x, labels)), labels) (models/cvae.py:212).  With
        self.training False, BatchNorm uses its running statistics and Dropout is the identity."""
        h = self.enc_forward(x, labels)
        zin = self.sample(h, eps, 1.0)
        return self.dec_forward(zin, self.dec["mask"])

    def decode(self, z, y_vec):
        """Decoder only, from f32 latents [B,z] and one-hot labels [B,y] (models/cvae.py:171)."""
        zin = self.set_decoder_input(z, y_vec)
        return self.dec_forward(zin, self.dec["mask"])

    def adamw_step(self, lr, betas=ADAM_BETAS, eps=ADAM_EPS, weight_decay=ADAM_WD):
        self.step_count += 1
        ops.adamw_dev(self.p_arena, self.g_arena, self.m_arena, self.v_arena, self.arena_numel, lr, betas[0],
                      betas[1], eps, weight_decay, self.step_dev, self.bc_dev)
        self._param_version += 1

    def train_step(self, x, labels, eps, batch_size, lr):
        dec = self.forward_backward(x, labels, eps, batch_size)
        self.adamw_step(lr)
        return dec

    def read_losses(self):
        kl, ll, bad = self.losses[:3].tolist()
        if bad != 0.0:
            self.losses[2:3].zero_()
            raise ValueError("a class label outside [0, %d) reached the encoder since the last read_losses()" % self.y)
        return {"VAE": kl + ll, "KL": kl, "LL": ll}
