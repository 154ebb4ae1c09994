"""The benchmark configuration itself (c3: 64x64, 100 classes, ngf 64, z 32).

* f32 mode at batch 32 against the CPU oracle on the same seeded inputs (one oracle step at this
  width takes a few seconds): losses within 2e-5 relative, reconstruction within 1e-4 of its range,
  every parameter gradient at cosine >= 0.9999 and, when no LeakyReLU gate differs between the two
  sides (parity.gate_flips; an input within rounding of zero can take either sign), within 5e-3 of
  the tensor's maximum (sums over 131k pixels behind nine BatchNorm backward passes in two different
  orders; biases in front of a BatchNorm, whose true gradient is zero, excluded).
* bf16 mode at batch 32 against the same oracle step: the mode and width the benchmark is quoted in.
  Tolerances: both losses within 3e-2 relative, encoder output and reconstruction at cosine >= 0.998
  with the f32 oracle, every parameter gradient at cosine >= 0.97 (decoder weights scaled by 0.55 on
  both sides as in the f32 test, so that the tanh is not saturated and gradients are O(1)).  The worst
  tensor is a BatchNorm bias of the first res blocks (a 64- or 128-vector of sums over 131k / 33k pixels
  with heavy cancellation); measured 0.983 and 0.979 with two summation orders of the BatchNorm
  statistics that agree to 1e-7 in f32, i.e. the value moves by +-0.003 with any rounding-level change.
* 128 x 128 (the paper-scale configurations): f32 at batch 4 against the oracle with four and with
  five down / up stages at ngf 64, the eval-mode encoder + decoder forward and the decoder-only
  sampling path against the oracle in eval mode, and the bf16 full-batch (64) step for determinism.
* bf16 mode at the full batches the benchmark is quoted on (c3 at 256, c4 at 64 with four and five stages)
  against one oracle step each (a few seconds of CPU time), and through properties that do not depend on
  size: the step is run-to-run bit-identical, and the backward pass is linear in the loss scale (a scale of 2
  doubles every gradient exactly: powers of two commute with rounding).
"""
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu

P, Y, Z = 64, 100, 32


def _inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, P, P, generator=g) * 2 - 1
    labels = torch.randint(0, Y, (B,), generator=g)
    eps = torch.randn(B, Z, generator=g)
    mask = (torch.rand(B, 4 * (Z + Y), generator=g) > 0.5).float()
    return x, labels, eps, mask


def test_c3_width_f32_matches_oracle():
    from parity import synthetic_parity, worst_grads
    r = synthetic_parity(P, Y, Z, 32, torch.float32, seed=3, init_seed=5)
    (wk, werr), (ck, wcos) = worst_grads(r)
    print("c3 f32 B=32: loss err", r["loss"], "dec err %.2e" % r["dec_err"], "flipped gates", r["flips"],
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.6f" % wcos)
    assert r["loss"]["KL"] <= 2e-5 and r["loss"]["LL"] <= 2e-5, r["loss"]
    assert r["dec_err"] <= 1e-4
    # 131k pixels x 64 channels per layer: a few LeakyReLU inputs sit within rounding of zero and take the
    # other sign (parity.gate_flips); with none the max-norm bound holds, otherwise direction and norm do
    assert r["flips"] <= 64, r["flips"]
    assert wcos >= 0.9999, (ck, wcos)
    assert werr <= 5e-3 or r["flips"] > 0, (wk, werr)


def test_c3_width_bf16_matches_oracle():
    from parity import synthetic_parity, worst_grads
    r = synthetic_parity(P, Y, Z, 32, torch.bfloat16, seed=3, init_seed=5)
    (wk, werr), (ck, wcos) = worst_grads(r)
    print("c3 bf16 B=32: loss err", r["loss"], "dec cos %.5f h cos %.5f" % (r["dec_cos"], r["h_cos"]),
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.5f" % wcos)
    assert r["loss"]["KL"] <= 3e-2 and r["loss"]["LL"] <= 3e-2, r["loss"]
    assert r["h_cos"] >= 0.998 and r["dec_cos"] >= 0.998, (r["h_cos"], r["dec_cos"])
    assert wcos >= 0.97, (ck, wcos)


def test_c3_batch256_bf16_matches_oracle():
    """The headline shape itself: c3 at the benchmark's per-GPU batch of 256 in bf16 against one step of the f32
    CPU oracle on the same seeded inputs (models/cvae.py:132-152; a few seconds of CPU time).  Tolerances: both
    losses within 3e-2 relative, encoder output and reconstruction at cosine >= 0.998, every parameter gradient
    at cosine >= 0.97, the decoder's gradients as one vector at >= 0.999 and the encoder's at >= 0.985 (measured:
    losses 1.2e-3 / 2.6e-5, forward cosines 0.9999, worst tensor 0.985 (the stem BatchNorm's bias, a 64-vector of
    sums over a million pixels each), decoder 0.99997, encoder 0.9915)."""
    from parity import synthetic_parity, worst_grads, cosine
    r = synthetic_parity(P, Y, Z, 256, torch.bfloat16, seed=31, init_seed=5)
    (wk, werr), (ck, wcos) = worst_grads(r)
    agg = {}
    for net in ("encoder.", "decoder."):
        ks = [k for k in r["grads"] if k.startswith(net)]
        got = torch.cat([r["eng"].grads[k].detach().cpu().flatten() for k in ks])
        ref = torch.cat([r["ref"]["pgrads"][k].flatten() for k in ks])
        agg[net] = cosine(got, ref)
    print("c3 bf16 B=256: loss err", r["loss"], "dec cos %.5f h cos %.5f" % (r["dec_cos"], r["h_cos"]),
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.5f" % wcos, "aggregate", agg)
    assert r["loss"]["KL"] <= 3e-2 and r["loss"]["LL"] <= 3e-2, r["loss"]
    assert r["h_cos"] >= 0.998 and r["dec_cos"] >= 0.998, (r["h_cos"], r["dec_cos"])
    assert wcos >= 0.97, (ck, wcos)
    assert agg["encoder."] >= 0.985 and agg["decoder."] >= 0.999, agg


@pytest.mark.parametrize("n_stage", [4, 5])
def test_c4_batch64_bf16_matches_oracle(n_stage):
    """BASELINE configuration 4 at its per-GPU batch: 128 x 128, 100 classes, batch 64, bf16, with four and with
    five down / up stages, against one f32 oracle step.  Same tolerances as the c3 test."""
    from parity import synthetic_parity, worst_grads
    r = synthetic_parity(128, Y, Z, 64, torch.bfloat16, seed=33, init_seed=9, n_stage=n_stage)
    (wk, werr), (ck, wcos) = worst_grads(r)
    print("c4 bf16 B=64 stages %d: loss err" % n_stage, r["loss"], "dec cos %.5f h cos %.5f" % (r["dec_cos"], r["h_cos"]),
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.5f" % wcos)
    assert r["loss"]["KL"] <= 3e-2 and r["loss"]["LL"] <= 3e-2, r["loss"]
    assert r["h_cos"] >= 0.998 and r["dec_cos"] >= 0.998, (r["h_cos"], r["dec_cos"])
    assert wcos >= 0.97, (ck, wcos)
    del r
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n_stage", [4, 5])
def test_p128_f32_matches_oracle(n_stage):
    """BASELINE configuration 4 (128 x 128, 100 classes, ngf 64) with the wrappers' four stages and with
    the "deeper stacks" five, at batch 4."""
    from parity import synthetic_parity, worst_grads
    r = synthetic_parity(128, Y, Z, 4, torch.float32, seed=8, init_seed=9, n_stage=n_stage)
    (wk, werr), (ck, wcos) = worst_grads(r)
    print("P128 f32 stages %d: loss err" % n_stage, r["loss"], "dec err %.2e" % r["dec_err"], "flipped gates", r["flips"],
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.6f" % wcos)
    assert r["loss"]["KL"] <= 2e-5 and r["loss"]["LL"] <= 2e-5, r["loss"]
    assert r["dec_err"] <= 1e-4
    assert r["flips"] <= 64, r["flips"]
    assert wcos >= 0.9999, (ck, wcos)
    assert werr <= 5e-3 or r["flips"] > 0, (wk, werr)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_p128_eval_paths_match_oracle(dtype):
    """BASELINE configuration 5's path (models/cvae.py:197-220 and :167-178) at 128 x 128: eval-mode
    encoder + sampler + decoder, and the decoder alone on given latents, against the oracle with
    train=False (running statistics, no dropout).  f32: 1e-4 of the output range; bf16: cosine >= 0.998."""
    from cvae_gan_amd.engine import Engine
    from oracle import ref_cpu as O
    from parity import synthetic_inputs, synthetic_state, cosine
    B, PP = 4, 128
    x, labels, eps, _ = synthetic_inputs(PP, Y, Z, B, 12)
    e, d = synthetic_state(PP, Y, Z, 13)
    # Running statistics as a trained network has them: those of the data.  One train-mode pass of the
    # oracle moves every buffer by momentum 0.1 towards its batch statistic, which is solved for here.
    # (With the constructor's (0, 1), or random values, nothing normalises the untrained encoder: log
    # sigma reaches +-10, z = mu + exp(log sigma) eps reaches 1e4 and the comparison measures the
    # conditioning of exp, not the kernels.)
    g = torch.Generator().manual_seed(14)
    new_stats = {}
    with torch.no_grad():
        O.encoder_forward(e, x, O.label_planes(labels, Y, PP), True, new_stats)
    for k, v in new_stats.items():
        if not k.endswith("tracked"):
            e[k] = (v - (1 - O.BN_MOMENTUM) * e[k]) / O.BN_MOMENTUM
    with torch.no_grad():
        full_ref, _, _, _ = O.cvae_forward(e, d, x, labels, eps, None, Y, train=False)
        zs = torch.randn(B, Z, generator=g)
        dec_ref = O.decoder_forward(d, zs, O.one_hot(labels, Y), None, train=False)
    state = OrderedDict(e)
    state.update(d)
    eng = Engine(PP, Y, Z, B, dtype=dtype, state=state)
    eng.training = False
    full = eng.reconstruct(x.cuda(), labels.int().cuda(), eps.cuda()).cpu().clone()
    dec = eng.decode(zs.cuda(), O.one_hot(labels, Y).cuda()).cpu().clone()
    print("P128 eval", dtype, "full err %.2e cos %.6f  dec err %.2e cos %.6f  max sigma %.2f" % (
        float((full - full_ref).abs().max()), cosine(full, full_ref), float((dec - dec_ref).abs().max()),
        cosine(dec, dec_ref), float(eng.z_sigma.max())))
    if dtype == torch.float32:
        assert float((full - full_ref).abs().max()) <= 1e-4
        assert float((dec - dec_ref).abs().max()) <= 1e-4
    else:
        assert cosine(full, full_ref) >= 0.998 and cosine(dec, dec_ref) >= 0.998, (cosine(full, full_ref), cosine(dec, dec_ref))


def test_c5_batch_1024_eval_forward():
    """BASELINE configuration 5 in one call: eval-mode encoder + decoder forward at 128 x 128, batch 1024,
    bf16.  The decoder's last up-sampled, reflection-padded tensor holds 1024 x 130 x 130 x 128 = 2.2e9
    elements, more than 32-bit element offsets reach.  In eval mode no operation couples the images of a
    batch, so rows of the big batch must agree with the same images run as a batch of 4 (first rows, and
    the last rows, which sit beyond the 2^31st element of that tensor), up to the summation order of the
    Linear layers, whose split depends on the batch: cosine >= 0.9995, and both within cosine 0.998 of the
    f32 oracle.  Weights and running statistics as in test_p128_eval_paths_match_oracle."""
    from cvae_gan_amd.engine import Engine
    from oracle import ref_cpu as O
    from parity import synthetic_inputs, synthetic_state, cosine
    B, PP = 1024, 128
    x, labels, eps, _ = synthetic_inputs(PP, Y, Z, B, 41)
    e, d = synthetic_state(PP, Y, Z, 13)
    rows = torch.cat([torch.arange(4), torch.arange(B - 4, B)])
    new_stats = {}
    with torch.no_grad():
        O.encoder_forward(e, x[rows], O.label_planes(labels[rows], Y, PP), True, new_stats)
        for k, v in new_stats.items():
            if not k.endswith("tracked"):
                e[k] = (v - (1 - O.BN_MOMENTUM) * e[k]) / O.BN_MOMENTUM
        ref, _, _, _ = O.cvae_forward(e, d, x[rows], labels[rows], eps[rows], None, Y, train=False)
    state = OrderedDict(e)
    state.update(d)
    big = Engine(PP, Y, Z, B, dtype=torch.bfloat16, state=state)
    big.training = False
    out = big.reconstruct(x.cuda(), labels.int().cuda(), eps.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    got = out[rows.cuda()].cpu()
    small = Engine(PP, Y, Z, 8, dtype=torch.bfloat16, share=big)
    small.training = False
    same = small.reconstruct(x[rows].cuda(), labels[rows].int().cuda(), eps[rows].cuda()).cpu()
    print("c5 B=1024: cos(big, small) %.6f  cos(big, oracle) %.6f  max|big - small| %.3e" % (
        cosine(got, same), cosine(got, ref), float((got - same).abs().max())))
    assert cosine(got[:4], same[:4]) >= 0.9995 and cosine(got[4:], same[4:]) >= 0.9995
    assert cosine(got[:4], ref[:4]) >= 0.998 and cosine(got[4:], ref[4:]) >= 0.998
    dec = big.decode(eps.cuda(), torch.nn.functional.one_hot(labels, Y).float().cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(dec).all() and float(dec.abs().max()) <= 1.0
    del big, small
    torch.cuda.empty_cache()


def test_c4_full_batch_bf16_is_deterministic():
    """128 x 128 at the configuration's per-GPU batch of 64, bf16, four and five stages: finite,
    non-zero and run-to-run bit-identical gradients and losses."""
    from cvae_gan_amd.engine import Engine
    from parity import synthetic_inputs
    for n_stage in (4, 5):
        B = 64
        x, labels, eps, mask = synthetic_inputs(128, Y, Z, B, 21)
        x, labels, eps = x.cuda(), labels.int().cuda(), eps.cuda()
        eng = Engine(128, Y, Z, B, dtype=torch.bfloat16, seed=2, n_stage=n_stage)
        eng.set_mask(mask)
        runs = []
        for _ in range(2):
            eng.forward_backward(x, labels, eps, float(B))
            torch.cuda.synchronize()
            runs.append((eng.g_arena.clone(), eng.losses[:2].clone()))
        assert torch.isfinite(runs[0][0]).all() and float(runs[0][0].abs().max()) > 0
        assert torch.isfinite(runs[0][1]).all()
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        del eng
        torch.cuda.empty_cache()


def test_c3_full_batch_bf16_is_deterministic_and_linear():
    from cvae_gan_amd.engine import Engine
    B = 256
    x, labels, eps, mask = _inputs(B, 4)
    x, labels, eps = x.cuda(), labels.int().cuda(), eps.cuda()
    eng = Engine(P, Y, Z, B, dtype=torch.bfloat16, seed=2)
    eng.set_mask(mask)
    runs = []
    for scale in (1.0, 1.0, 2.0):
        eng.forward_backward(x, labels, eps, float(B), loss_scale=scale)
        torch.cuda.synchronize()
        runs.append((eng.g_arena.clone(), eng.losses[:2].clone()))
    assert torch.isfinite(runs[0][0]).all() and float(runs[0][0].abs().max()) > 0
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[2][1], runs[0][1])                 # the reported losses are unscaled
    assert torch.equal(runs[2][0], 2.0 * runs[0][0])


def test_split_encoder_backward_equals_whole():
    """enc_backward(part="upper") + enc_backward(part="lower") (the data-parallel schedule) writes the
    same gradients, bit for bit, as the single pass, and the upper part touches exactly the arena
    range enc_upper_range() names."""
    from cvae_gan_amd.engine import Engine
    B = 16
    x, labels, eps, mask = _inputs(B, 6)
    x, labels, eps = x.cuda(), labels.int().cuda(), eps.cuda()
    eng = Engine(P, Y, Z, B, dtype=torch.bfloat16, seed=3)
    eng.set_mask(mask)
    eng.forward_backward(x, labels, eps, float(B))
    torch.cuda.synchronize()
    whole = eng.g_arena.clone()
    eng.forward_backward_decoder(x, labels, eps, float(B))
    lo, hi = eng.enc_upper_range()
    eng.g_arena[lo:].fill_(float("nan"))
    eng.enc_backward(eng.enc["dh"], part="upper")
    torch.cuda.synchronize()
    assert torch.equal(eng.g_arena[lo:hi], whole[lo:hi])
    assert torch.isnan(eng.g_arena[hi:]).all()          # (arena padding included: nothing below was written)
    eng.enc_backward(eng.enc["dh"], part="lower")
    torch.cuda.synchronize()
    pad = torch.isnan(eng.g_arena)                      # alignment gaps between tensors stay untouched
    assert torch.equal(eng.g_arena[~pad], whole[~pad])
    assert int(pad.sum()) < 64


@pytest.mark.parametrize("pix", [256, 512])
def test_large_images_f32_match_oracle(pix):
    """256 x 256 and 512 x 512 (the reference's --pix_dim default) at the reference's width, batch 3: a tile of
    such an image holds one image, so the 32-bit offsets inside a tile fit and every stride-1 layer stays on the
    patch kernels.  Losses 2e-5, reconstruction 1e-4; with hundreds of thousands of pixels per channel a few
    dozen LeakyReLU gates sit within rounding of zero (parity.gate_flips), so gradients are held by direction."""
    from parity import synthetic_parity, worst_grads
    from cvae_gan_amd import ops
    r = synthetic_parity(pix, 16, 32, 3, torch.float32, seed=4, init_seed=6)
    (wk, werr), (ck, wcos) = worst_grads(r)
    print("P %d f32: loss err" % pix, r["loss"], "dec err %.2e" % r["dec_err"], "flipped gates", r["flips"],
          "worst grad err", wk, "%.3e" % werr, "worst grad cos", ck, "%.6f" % wcos)
    kinds = {ops.conv_fwd_kernel(g.desc) for g in r["eng"]._gemms}
    assert any(k.startswith("patch_fwd") and "64ch" in k for k in kinds), kinds      # the full-resolution layers too
    assert r["loss"]["KL"] <= 2e-5 and r["loss"]["LL"] <= 2e-5, r["loss"]
    assert r["dec_err"] <= 1e-4
    assert r["flips"] <= 256, r["flips"]
    assert wcos >= 0.9999, (ck, wcos)
    assert werr <= 5e-3 or r["flips"] > 0, (wk, werr)
