"""Full train-step parity of the HIP engine (through the C ABI) against the CPU oracle
on the golden cases.

Tolerances (error relative to the largest magnitude of each reference tensor):
  f32 mode : activations 1e-4, gradients 1e-3, losses 1e-5.  The arithmetic is exact
             f32 fma; what remains is summation order amplified by the tiny-batch
             BatchNorms (the oracle itself differs from the reference by ~1e-4 there).
             A LeakyReLU input within rounding of zero can get the other sign on the other
             side; that one gate then scales its gradient element by 5 or 1/5, and the
             difference spreads down the backward pass.  parity.gate_flips counts such gates
             from the forward activations: with none, the max-norm bounds above hold; a
             case that has some (one element in 131072 in the 32 x 32 / ngf 64 fixture) is held
             to cosine >= 0.9999 per gradient tensor instead, forward bounds unchanged.
  bf16 mode: the fixtures are batches of 4-8 images, where a BatchNorm statistic is a sum
             of a few hundred terms and BatchNorm1d normalises 4-8 samples: bf16 rounding
             is amplified far more than at a real batch.  Stated per test below; the
             benchmark's own shape (64 x 64, ngf 64, batch 32) is held to losses 3e-2,
             forward cosine 0.998 and gradient cosine 0.98 in tests/test_gpu_fullsize.py.
"""
import numpy as np
import pytest
import torch

from helpers import CASES, Case
from parity import run_parity, format_rows

pytestmark = pytest.mark.gpu

ZERO_GRAD = {"encoder.conv.1.bias", "encoder.fc.0.bias"}   # true gradient is exactly 0


@pytest.mark.parametrize("name", CASES)
def test_step_parity_f32(name):
    rows, eng, ref = run_parity(name, torch.float32)
    flips = ref["flips"]
    assert flips <= 4, flips
    bad = []
    for kind, k, r, m, c in rows:
        if k in ZERO_GRAD:
            continue
        tol = {"act": 1e-4, "dact": 1e-3, "grad": 1e-3, "loss": 1e-5, "stat": 1e-4}[kind]
        if k == "dec":
            tol = 1e-3      # tanh of pre-activations of magnitude ~60: their rounding error times the slope
        ok = r <= tol
        if flips and kind in ("dact", "grad"):
            ok = c >= 0.9999
        if not ok:
            bad.append((kind, k, r, m, c))
    assert not bad, "flipped gates: %d\n" % flips + format_rows(bad)
    # losses also against the fixture recorded from the reference classes
    c = Case(name)
    got = eng.read_losses()
    for k in ("VAE", "KL", "LL"):
        refv = float(c.npz["s0/" + k])
        assert abs(got[k] - refv) <= 2e-5 * abs(refv), (k, got[k], refv)


def _bf16_check(rows, eng, ref, act_cos, loss_tol, grads):
    """Forward rows and losses always; `grads` = None (gradient rows are printed, not judged) or a dict
    agg_enc, agg_dec (cosine of all of a side's parameter gradients, each tensor scaled to unit reference
    norm, taken as one vector), big_enc, big_dec (worst tensor of at least 4096 elements), any (every tensor)."""
    from parity import cosine
    bad = []
    for kind, k, r, m, c in rows:
        if k in ZERO_GRAD:
            continue
        if kind in ("loss", "stat"):
            ok = r <= loss_tol
        elif kind == "act":
            ok = c >= act_cos
        elif kind == "dact" and k == "dec":
            ok = c >= 0.98           # the loss gradient itself: one elementwise step behind the forward pass
        else:
            continue
        if not ok:
            bad.append((kind, k, r, m, c))
    summary = {}
    for side in ("encoder.", "decoder."):
        ks = [k for k in ref["pgrads"] if k.startswith(side) and k not in ZERO_GRAD]
        got = {k: eng.grads[k].detach().cpu() for k in ks}
        a = torch.cat([(got[k].flatten().double() / max(float(ref["pgrads"][k].norm()), 1e-30)) for k in ks])
        b = torch.cat([(ref["pgrads"][k].flatten().double() / max(float(ref["pgrads"][k].norm()), 1e-30)) for k in ks])
        per = {k: cosine(got[k], ref["pgrads"][k]) for k in ks}
        big = [v for k, v in per.items() if ref["pgrads"][k].numel() >= 4096]
        summary[side] = (cosine(a, b), min(big) if big else 1.0, min(per.values()), min(per, key=per.get))
    print("bf16 gradients (aggregate, worst large tensor, worst tensor):",
          {s: (round(v[0], 4), round(v[1], 4), round(v[2], 4), v[3]) for s, v in summary.items()})
    if grads is not None:
        for side, tag in (("encoder.", "enc"), ("decoder.", "dec")):
            agg, big, worst, wk = summary[side]
            if agg < grads["agg_" + tag]:
                bad.append(("grad", side + "* aggregate", 0.0, 0.0, agg))
            if big < grads["big_" + tag]:
                bad.append(("grad", side + "* worst large tensor", 0.0, 0.0, big))
            if worst < grads["any"]:
                bad.append(("grad", wk, 0.0, 0.0, worst))
    return bad


# bf16 bars per case, for the statistics form that ships (sums about the first row, cvg_bn_stats).  Where a bar is
# looser than 0.998 / 3e-2 / 0.98 the reason is named; tools/bf16_chain.py prints the whole backward chain.
RAW_BARS = {
    # raw initialisation: the decoder's pre-tanh activations reach +-60, so the loss gradient (1 - tanh^2) lives on
    # the few unsaturated pixels, where a bf16-sized error of the pre-activation (+-0.5) changes it by a factor of e.
    # tiny_p16_ngf8 (16 x 16, batch 4) has so few of them that the direction of everything behind `pre` is decided
    # by rounding: three numerically equivalent variants of this code base measured aggregate gradient cosines of
    # +0.98, -0.33 and -0.96 (this tree: -0.31), while the forward pass and the losses hold.  Its gradients are
    # therefore not judged here; the conditioned test below judges them.
    "tiny_p16_ngf8": dict(act_cos=0.98, loss_tol=3e-2, grads=None),
    "ragged_p32_ngf8": dict(act_cos=0.998, loss_tol=5e-2,        # KL: log sigma reaches +-3 and exp() turns its bf16 rounding into 2 % of sigma^2
                            grads=dict(agg_enc=0.97, big_enc=0.97, agg_dec=0.97, big_dec=0.97, any=0.93)),   # measured 0.993 / 0.993, worst tensor 0.972
    "c2_p32_ngf64": dict(act_cos=0.99, loss_tol=3e-2,            # the saturated tail: cos(pre-tanh) 0.997, cos(loss gradient) 0.90 with bf16 storage of the weights alone at 0.97
                         grads=dict(agg_enc=0.92, big_enc=0.90, agg_dec=0.92, big_dec=0.90, any=0.88)),      # measured 0.960 / 0.954, worst tensor 0.924
    "deep5_p64_ngf8": dict(act_cos=0.998, loss_tol=3e-2,
                           grads=dict(agg_enc=0.95, big_enc=0.95, agg_dec=0.97, big_dec=0.97, any=0.85)),    # measured 0.975 / 0.994, worst tensor 0.893 (stem BatchNorm bias)
}
CONDITIONED_BARS = {
    # encoder bounds: the batch-4 cases with a 1 x 1 / 2 x 2 final map lose the direction at ONE place, the backward pass
    # of the head's BatchNorm1d (4 samples per feature: g - mean g - xhat mean(g xhat) keeps 2 of 4 dimensions and
    # cancels to the last bits of xhat): cos(d f1) 0.9996 -> cos(d f0) 0.90 on deep5_p64_ngf8, unchanged with f0, f1
    # and their gradients kept in f32 (tried), so it is the 1 % of bf16 error the 15 layers in front accumulate in
    # xhat, not this layer's arithmetic; everything behind it carries that cosine.  c2_p32_ngf64 (batch 8) drops
    # there to 0.973, which bf16 storage of the weights and the image alone already does to the f32 oracle (0.975).
    "tiny_p16_ngf8": dict(agg_enc=0.98, big_enc=0.98, agg_dec=0.998, big_dec=0.995, any=0.95),     # measured 0.995 / 0.9997, worst 0.983
    "ragged_p32_ngf8": dict(agg_enc=0.98, big_enc=0.98, agg_dec=0.998, big_dec=0.995, any=0.95),   # measured 0.993 / 0.9999, worst 0.975
    "c2_p32_ngf64": dict(agg_enc=0.95, big_enc=0.94, agg_dec=0.998, big_dec=0.995, any=0.90),      # measured 0.966 / 0.9996, worst 0.939
    "deep5_p64_ngf8": dict(agg_enc=0.85, big_enc=0.85, agg_dec=0.998, big_dec=0.995, any=0.5),     # measured 0.894 / 0.9996, worst 0.636 (stem BatchNorm bias, 8 values)
}


@pytest.mark.parametrize("name", CASES)
def test_step_parity_bf16(name):
    """bf16 activations / weight shadows against the f32 oracle at the reference's own initialisation: the forward
    pass, both losses, the loss gradient and (except on the 16 x 16 batch-4 case) the parameter gradients, with the
    bars of RAW_BARS."""
    rows, eng, ref = run_parity(name, torch.bfloat16)
    bad = _bf16_check(rows, eng, ref, **RAW_BARS[name])
    assert not bad, "\n" + format_rows(bad)


@pytest.mark.parametrize("name", CASES)
def test_step_parity_bf16_conditioned(name):
    """Same comparison with every decoder weight scaled by 0.55 (both sides), which brings the pre-tanh
    activations to O(1): forward cosine >= 0.999, losses 5e-2 (the KL term of the batch-6 case), gradients with the
    bars of CONDITIONED_BARS; at the benchmark width with batch 32 and 256 every tensor is held to >= 0.97
    (tests/test_gpu_fullsize.py)."""
    def shrink(state):
        e, d = state
        for k in d:
            if k.endswith("weight"):
                d[k] = d[k] * 0.55
        return e, d
    rows, eng, ref = run_parity(name, torch.bfloat16, state_fn=shrink)
    bad = _bf16_check(rows, eng, ref, act_cos=0.999, loss_tol=5e-2, grads=CONDITIONED_BARS[name])
    assert not bad, "\n" + format_rows(bad)


def test_three_steps_f32_match_golden():
    """Three AdamW steps from the fixture's initial weights: per-step losses against the
    values recorded from the reference classes."""
    from collections import OrderedDict
    from cvae_gan_amd.engine import Engine
    c = Case("ragged_p32_ngf8")
    e, d = c.init_state()
    st = OrderedDict(e)
    st.update(d)
    eng = Engine(c.P, c.y, c.z, c.B, dtype=torch.float32, ngf=c.ngf, n_stage=c.n_stage, state=st)
    for step in range(c.n_steps):
        x, labels, eps, mask = c.inputs(step)
        eng.set_mask(mask)
        eng.train_step(x.cuda(), labels.int().cuda(), eps.cuda(), float(c.bs_flag), c.lr)
        got = eng.read_losses()
        for k in ("VAE", "KL", "LL"):
            refv = float(c.npz["s%d/%s" % (step, k)])
            tol = 2e-5 if step == 0 else 2e-3     # AdamW's first steps amplify rounding (sign of ~0 gradients)
            assert abs(got[k] - refv) <= tol * abs(refv), (step, k, got[k], refv)


@pytest.mark.parametrize("z,y", [(260, 21), (256, 16)])
def test_wide_latent_and_label_dims_f32(z, y):
    """Channel counts beyond 1024 that are no multiple of 512 in the row kernels: the encoder head's
    BatchNorm1d over 4 z features (1040 for z = 260) and the decoder's first Linear over 4 (z + y)
    (1124 -> padded to 1128).  The reference's own command-line defaults (z_dim 256, y_dim 1000,
    starter.py:25-26) are of this kind.  f32 against the oracle at 16 x 16, ngf 8."""
    from parity import synthetic_parity, worst_grads
    r = synthetic_parity(16, y, z, 6, torch.float32, seed=31, init_seed=32, ngf=8)
    (wk, werr), (ck, wcos) = worst_grads(r)
    assert r["loss"]["KL"] <= 2e-5 and r["loss"]["LL"] <= 2e-5, r["loss"]
    assert r["dec_err"] <= 1e-4
    assert wcos >= 0.9999 and (werr <= 2e-3 or r["flips"] > 0), ((wk, werr), (ck, wcos), r["flips"])
