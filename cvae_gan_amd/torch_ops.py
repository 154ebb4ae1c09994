"""The kernels as PyTorch custom operators: torch.ops.cvaegan.*.

The module surface (cvae_gan_amd/networks.py) calls these from its autograd Functions; each
operator enqueues the kernels of one or more C entry points of include/cvaegan.h on the current
stream.  Two groups:

* module-level operators on an engine handle (an int from register_engine): encoder_forward /
  encoder_backward / decoder_forward / decoder_backward replace the reference's
  ResNetEncoder.forward, ConvUpSampleDecoder.forward (models/networks.py:193-196, :159-162) and
  their autograd backward passes;
* tensor-only operators that map one to one onto C entry points without a descriptor struct:
  bn_stats, bn_act_fwd, bn_act_bwd, reparam_kl_fwd, mse_fwd_bwd, adamw.

They are implemented for the CUDA dispatch key only (ROCm devices dispatch there); there is no
CPU implementation, so a CPU tensor fails in the dispatcher instead of taking another path."""
import weakref

import torch
from torch.library import Library

from . import ops

_ENGINES = weakref.WeakValueDictionary()
_next_handle = [1]


def register_engine(engine):
    """-> int handle under which torch.ops.cvaegan.* find this engine (held weakly)."""
    h = getattr(engine, "_op_handle", None)
    if h is None:
        h = engine._op_handle = _next_handle[0]
        _next_handle[0] += 1
    _ENGINES[h] = engine
    return h


def _engine(handle):
    eng = _ENGINES.get(int(handle))
    if eng is None:
        raise RuntimeError("torch.ops.cvaegan: unknown engine handle %d" % handle)
    return eng


_DEF = Library("cvaegan", "DEF")
_DEF.define("encoder_forward(int engine, Tensor x, Tensor labels) -> Tensor")
_DEF.define("encoder_backward(int engine, Tensor dh) -> ()")
_DEF.define("decoder_forward(int engine, Tensor z, Tensor y_vec, bool fresh_mask) -> Tensor")
_DEF.define("decoder_backward(int engine, Tensor d_dec) -> Tensor")
_DEF.define("bn_stats(Tensor x, Tensor(a!) stats, Tensor(b!)? running_mean, Tensor(c!)? running_var, "
            "Tensor(d!)? num_batches_tracked, Tensor(e!) ws) -> ()")
_DEF.define("bn_act_fwd(Tensor x, Tensor? res, Tensor(a!) y, Tensor stats, Tensor gamma, Tensor beta) -> ()")
_DEF.define("bn_act_bwd(Tensor dy, Tensor? y, Tensor x, Tensor(a!) dx, Tensor stats, Tensor gamma, Tensor? beta, "
            "Tensor(b!) dgamma, Tensor(c!) dbeta, Tensor(d!) ws) -> ()")
_DEF.define("reparam_kl_fwd(Tensor h, Tensor eps, Tensor labels, int y_dim, Tensor(a!) zin, Tensor(b!) z_mean, "
            "Tensor(c!) z_sigma, Tensor(d!) kl, float batch_size) -> ()")
_DEF.define("mse_fwd_bwd(Tensor dec, Tensor x, Tensor(a!) d_dec, Tensor(b!) ll, float batch_size, float scale, "
            "Tensor(c!) ws) -> ()")
_DEF.define("adamw(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, "
            "float weight_decay, int step) -> ()")

_IMPL = Library("cvaegan", "IMPL", "CUDA")


def _encoder_forward(engine, x, labels):
    eng = _engine(engine)
    eng.params_changed()                 # an outside optimiser may have stepped
    return eng.enc_forward(x.contiguous().float(), labels)[:, :eng.enc["nz"]].float()


def _encoder_backward(engine, dh):
    eng = _engine(engine)
    if eng.enc["nz_ld"] == eng.enc["nz"]:
        ops.cast(dh.contiguous().float(), eng.enc["dh"], dh.numel())
    else:                                # padded rows: the columns from 2 * z_dim on stay zero
        eng.enc["dh"][:, :eng.enc["nz"]].copy_(dh)
    eng.enc_backward(eng.enc["dh"])


def _decoder_forward(engine, z, y_vec, fresh_mask):
    eng = _engine(engine)
    eng.params_changed()
    zin = eng.set_decoder_input(z.float(), y_vec.float())
    if fresh_mask:
        eng.random_mask()
    return eng.dec_forward(zin, eng.dec["mask"]).clone()


def _decoder_backward(engine, d_dec):
    eng = _engine(engine)
    dzin = eng.dec_backward(d_dec.contiguous().float(), eng.dec["mask"])
    return dzin[:, :eng.z].float()


def _bn_stats(x, stats, running_mean, running_var, num_batches_tracked, ws):
    ops.bn_stats(x, x.shape[0], x.shape[1], stats, running_mean, running_var, num_batches_tracked, ws)


def _bn_act_fwd(x, res, y, stats, gamma, beta):
    ops.bn_act_fwd(x, res, y, x.shape[0], x.shape[1], stats, gamma, beta)


def _bn_act_bwd(dy, y, x, dx, stats, gamma, beta, dgamma, dbeta, ws):
    ops.bn_act_bwd(dy, y, x, dx, x.shape[0], x.shape[1], stats, gamma, dgamma, dbeta, ws, beta=beta)


def _reparam_kl_fwd(h, eps, labels, y_dim, zin, z_mean, z_sigma, kl, batch_size):
    ops.reparam_kl_fwd(h, eps, labels, eps.shape[0], eps.shape[1], y_dim, zin.shape[1], zin, z_mean, z_sigma, kl, batch_size)


def _mse_fwd_bwd(dec, x, d_dec, ll, batch_size, scale, ws):
    ops.mse_fwd_bwd(dec, x, d_dec, ll, dec.numel(), batch_size, scale, ws)


def _adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    ops.adamw(p, g, m, v, p.numel(), lr, beta1, beta2, eps, weight_decay, step)


for _name, _fn in (("encoder_forward", _encoder_forward), ("encoder_backward", _encoder_backward),
                   ("decoder_forward", _decoder_forward), ("decoder_backward", _decoder_backward),
                   ("bn_stats", _bn_stats), ("bn_act_fwd", _bn_act_fwd), ("bn_act_bwd", _bn_act_bwd),
                   ("reparam_kl_fwd", _reparam_kl_fwd), ("mse_fwd_bwd", _mse_fwd_bwd), ("adamw", _adamw)):
    _IMPL.impl(_name, _fn)

OPS = ["encoder_forward", "encoder_backward", "decoder_forward", "decoder_backward", "bn_stats", "bn_act_fwd",
       "bn_act_bwd", "reparam_kl_fwd", "mse_fwd_bwd", "adamw"]
