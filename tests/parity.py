"""Layer-by-layer comparison of the HIP engine against the CPU oracle.

run_parity() replays one golden case through both and returns, for every forward
activation, every activation gradient and every parameter gradient, the error of
the HIP path relative to the largest reference magnitude of that tensor.  The
pytest GPU tests put thresholds on these numbers; tools/gpu_diag.py prints them."""
from collections import OrderedDict

import torch

from oracle import ref_cpu as O
from helpers import Case


def nchw(t, B, H, W, C, ld=None):
    """engine rows [B*H*W][ld] -> f32 NCHW on the CPU"""
    ld = ld or C
    return t.detach().float().view(B, H, W, ld)[..., :C].permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(got, ref):
    ref = ref.detach().double()
    got = got.detach().double()
    if got.shape != ref.shape:
        return float("inf"), float(ref.abs().max())
    den = float(ref.abs().max())
    return float((got - ref).abs().max()) / (den if den > 0 else 1.0), den


def cosine(got, ref):
    a, b = got.detach().double().flatten(), ref.detach().double().flatten()
    d = float(a.norm() * b.norm())
    return float((a * b).sum()) / d if d > 0 else 1.0


def oracle_pass(case, step=0, state=None):
    """Forward + backward of the oracle on a golden case with every intermediate retained."""
    e, d = state if state is not None else case.init_state()
    x, labels, eps, mask = case.inputs(step)
    return oracle_pass_raw(e, d, x, labels, eps, mask, case.y, case.z, case.P, case.bs_flag, case.n_stage)


def oracle_pass_raw(e, d, x, labels, eps, mask, y_dim, z_dim, P, bs_flag, n_stage, retain=True):
    """retain=False keeps the forward intermediates but not their gradients (large cases)."""
    params = {}
    for sd in (e, d):
        for k, t in sd.items():
            if not O.is_buffer(k):
                sd[k] = t.detach().clone().requires_grad_(True)
                params[k] = sd[k]
    et, dt = {}, {}
    new_stats = {}
    planes = O.label_planes(labels, y_dim, P)
    h = O.encoder_forward(e, x, planes, True, new_stats, et)
    z, mu, sigma = O.sample_latent(h, eps, z_dim)
    dec = O.decoder_forward(d, z, O.one_hot(labels, y_dim), mask, True, dt)
    vae, kl, ll = O.losses(dec, x, mu, sigma, bs_flag)
    acts = OrderedDict()
    acts["s0"] = et["stem"]
    acts["a0"] = et["a0"]
    for i in range(n_stage):
        b = et["block%d" % i]
        for n in ("c1", "b1", "c2", "r", "out"):
            acts["blk%d.%s" % (i, n)] = b[n]
    acts["f0"], acts["f1"], acts["h"] = et["f0"], et["f1"], h
    acts["g0"], acts["g1"], acts["x0"] = dt["g0"], dt["g1"], dt["x0"]
    for i in range(n_stage):
        acts["up%d" % i] = dt["up%d" % i]
    acts["pre"], acts["dec"] = dt["pre"], dec
    agrads = OrderedDict()
    if retain:
        for t in acts.values():
            t.retain_grad()
        vae.backward()
        agrads = OrderedDict((k, t.grad) for k, t in acts.items() if t.grad is not None)
        for k in list(agrads):
            if k.endswith(".r"):       # the engine keeps the gradient of the pre-activation sum: r.grad * LeakyReLU'(r)
                agrads[k] = agrads[k] * torch.where(acts[k] > 0, 1.0, O.LRELU_SLOPE).to(agrads[k].dtype)
        pgrads = OrderedDict((k, p.grad) for k, p in params.items())
    else:
        pgrads = OrderedDict(zip(params.keys(), torch.autograd.grad(vae, list(params.values()))))
        acts = OrderedDict((k, t.detach()) for k, t in acts.items())
    return dict(acts=acts, agrads=agrads, pgrads=pgrads, losses=dict(VAE=vae, KL=kl, LL=ll),
                z_mean=mu, z_sigma=sigma, new_stats=new_stats, dec=dec.detach(), h_enc=h.detach())


def is_lrelu_output(key):
    return key in ("a0", "f1", "g1") or key.endswith((".b1", ".r"))


def gate_flips(ref_acts, got_acts):
    """Number of LeakyReLU outputs whose sign differs between the oracle and the engine.  Only inputs
    within rounding of zero can flip, but a flipped gate multiplies that element's gradient by 5 (or
    1/5), far more than any rounding: tensors with 10^5..10^7 elements hold such an input now and then."""
    n = 0
    for k, t in ref_acts.items():
        if is_lrelu_output(k) and k in got_acts:
            n += int(((t.detach() * got_acts[k]) < 0).sum())
    return n


def engine_pass(case, dtype, step=0, state=None):
    from cvae_gan_amd.engine import Engine
    e, d = state if state is not None else case.init_state()
    st = OrderedDict(e)
    st.update(d)
    eng = Engine(case.P, case.y, case.z, case.B, dtype=dtype, ngf=case.ngf, n_stage=case.n_stage, state=st)
    x, labels, eps, mask = case.inputs(step)
    eng.set_mask(mask)
    xd = x.cuda()
    eng.forward_backward(xd, labels.int().cuda(), eps.cuda(), float(case.bs_flag))
    torch.cuda.synchronize()
    return eng


def engine_tensors(eng):
    B, P, ngf, s = eng.B, eng.P, eng.ngf, eng.s
    e, d = eng.enc, eng.dec
    acts, agrads = OrderedDict(), OrderedDict()
    acts["s0"] = nchw(e["s0"], B, P, P, ngf)
    acts["a0"] = nchw(e["a0"], B, P, P, ngf)
    agrads["s0"] = nchw(e["d_s0"], B, P, P, ngf)
    agrads["a0"] = nchw(e["d_a0"], B, P, P, ngf)
    for i, blk in enumerate(e["blocks"]):
        H, C, Co = blk["H"], blk["cin"], blk["cout"]
        for n in ("c1", "b1", "c2", "r"):
            acts["blk%d.%s" % (i, n)] = nchw(blk[n], B, H, H, C)
            agrads["blk%d.%s" % (i, n)] = nchw(blk["d_" + n], B, H, H, C)
        acts["blk%d.out" % i] = nchw(blk["out"], B, H // 2, H // 2, Co)
        agrads["blk%d.out" % i] = nchw(blk["d_out"], B, H // 2, H // 2, Co)
    nz = e["nz"]
    acts["f0"], acts["f1"], acts["h"] = e["f0"].float().cpu(), e["f1"].float().cpu(), e["h"].float().cpu()[:, :nz]
    agrads["f0"], agrads["f1"], agrads["h"] = e["d_f0"].float().cpu(), e["d_f1"].float().cpu(), e["dh"].float().cpu()[:, :nz]
    F = d["F"]
    acts["g0"], acts["g1"] = d["g0"].float().cpu()[:, :F], d["g1"].float().cpu()[:, :F]
    agrads["g0"], agrads["g1"] = d["d_g0"].float().cpu()[:, :F], d["d_g1"].float().cpu()[:, :F]
    acts["x0"] = nchw(d["x0"], B, s, s, d["C0"])
    agrads["x0"] = nchw(d["d_x0"], B, s, s, d["C0"])
    for i, up in enumerate(d["ups"]):
        hh = 2 * up["h"]
        acts["up%d" % i] = nchw(up["y"], B, hh, hh, up["cout"])
        agrads["up%d" % i] = nchw(up["d_y"], B, hh, hh, up["cout"])
    acts["pre"] = nchw(d["pre"], B, P, P, 3, 4)
    agrads["pre"] = nchw(d["dpre"], B, P, P, 3, 8)
    acts["dec"] = d["dec"].cpu()
    agrads["dec"] = d["d_dec"].cpu()
    pgrads = OrderedDict((k, g.detach().cpu().clone()) for k, g in eng.grads.items())
    return acts, agrads, pgrads


def run_parity(name, dtype, step=0, state_fn=None):
    """-> list of (kind, key, rel_err, ref_max, cosine).  state_fn, if given, edits the initial
    (encoder, decoder) state dicts before both sides see them."""
    case = Case(name)
    st_ref = state_fn(case.init_state()) if state_fn else None
    st_eng = state_fn(case.init_state()) if state_fn else None
    ref = oracle_pass(case, step, st_ref)
    eng = engine_pass(case, dtype, step, st_eng)
    acts, agrads, pgrads = engine_tensors(eng)
    ref["flips"] = gate_flips(ref["acts"], acts)
    rows = []
    for k, t in ref["acts"].items():
        r, m = rel_err(acts[k], t)
        rows.append(("act", k, r, m, cosine(acts[k], t)))
    for k in reversed(list(ref["agrads"].keys())):
        r, m = rel_err(agrads[k], ref["agrads"][k])
        rows.append(("dact", k, r, m, cosine(agrads[k], ref["agrads"][k])))
    for k in eng.arena_order:
        r, m = rel_err(pgrads[k], ref["pgrads"][k])
        rows.append(("grad", k, r, m, cosine(pgrads[k], ref["pgrads"][k])))
    got = eng.read_losses()
    for k in ("VAE", "KL", "LL"):
        rv = float(ref["losses"][k].detach())
        rows.append(("loss", k, abs(got[k] - rv) / abs(rv), rv, 1.0))
    r, m = rel_err(eng.z_mean.cpu(), ref["z_mean"])
    rows.append(("act", "z_mean", r, m, 1.0))
    r, m = rel_err(eng.z_sigma.cpu(), ref["z_sigma"])
    rows.append(("act", "z_sigma", r, m, 1.0))
    for k, v in ref["new_stats"].items():
        if k.endswith("tracked"):
            rows.append(("stat", k, float(abs(int(eng.buffers[k]) - int(v))), float(v), 1.0))
        else:
            r, m = rel_err(eng.buffers[k].cpu(), v)
            rows.append(("stat", k, r, m, 1.0))
    return rows, eng, ref


def format_rows(rows):
    out = []
    for kind, k, r, m, c in rows:
        out.append("%-5s %-46s rel_err %.3e  ref_max %.3e  cos %.6f" % (kind, k, r, m, c))
    return "\n".join(out)


# ---------------------------------------------------------------------------------------------------
# seeded synthetic cases at sizes no fixture covers (the oracle is run on the spot)
# ---------------------------------------------------------------------------------------------------
ZERO_GRAD_KEYS = (O.EP + "conv.1.bias", O.EP + "fc.0.bias")     # a bias in front of a BatchNorm: true gradient 0


def synthetic_inputs(P, Y, Z, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, P, P, generator=g) * 2 - 1
    labels = torch.randint(0, Y, (B,), generator=g)
    eps = torch.randn(B, Z, generator=g)
    mask = (torch.rand(B, 4 * (Z + Y), generator=g) > 0.5).float()
    return x, labels, eps, mask


def synthetic_state(P, Y, Z, init_seed, ngf=64, n_stage=4, dec_scale=0.55):
    """The reference's seeded initialisation with every decoder weight scaled by dec_scale, which
    brings the generator's pre-tanh activations to O(1): at the raw initialisation the tanh is
    saturated and every gradient is ~1e-9 of its conditioned size."""
    e, d = O.init_params(P, Y, Z, seed=init_seed, ngf=ngf, n_stage=n_stage)
    d = OrderedDict((k, v * dec_scale if k.endswith("weight") else v) for k, v in d.items())
    return e, d


def synthetic_parity(P, Y, Z, B, dtype, seed=3, init_seed=5, ngf=64, n_stage=4, dec_scale=0.55):
    """One forward + backward of the oracle and of the engine on the same seeded inputs.
    -> dict(loss={k: rel err}, dec_err (max abs), dec_cos, h_cos, grads={key: (err / max|ref|, cosine)},
            flips (LeakyReLU gates whose sign differs, see gate_flips), eng, ref)"""
    from cvae_gan_amd.engine import Engine
    x, labels, eps, mask = synthetic_inputs(P, Y, Z, B, seed)
    e, d = synthetic_state(P, Y, Z, init_seed, ngf, n_stage, dec_scale)
    state = OrderedDict(e)
    state.update(d)
    ref = oracle_pass_raw(OrderedDict(e), OrderedDict(d), x, labels, eps, mask, Y, Z, P, float(B), n_stage, retain=False)
    eng = Engine(P, Y, Z, B, dtype=dtype, ngf=ngf, n_stage=n_stage, state=state)
    eng.set_mask(mask)
    dec = eng.forward_backward(x.cuda(), labels.int().cuda(), eps.cuda(), float(B))
    torch.cuda.synchronize()
    got = eng.read_losses()
    acts, _, _ = engine_tensors(eng)
    out = {"loss": {k: abs(got[k] - float(ref["losses"][k])) / max(abs(float(ref["losses"][k])), 1e-3) for k in ("KL", "LL")},
           "dec_err": float((dec.cpu() - ref["dec"]).abs().max()), "dec_cos": cosine(dec.cpu(), ref["dec"]),
           "h_cos": cosine(eng.enc["h"].float().cpu()[:, :eng.enc["nz"]], ref["h_enc"]), "grads": {}, "eng": eng, "ref": ref,
           "flips": gate_flips(ref["acts"], acts)}
    del acts
    for k, gref in ref["pgrads"].items():
        if k in ZERO_GRAD_KEYS:
            continue
        out["grads"][k] = (rel_err(eng.grads[k].cpu(), gref)[0], cosine(eng.grads[k].cpu(), gref))
    return out


def worst_grads(res):
    """(key, err) with the largest relative error and (key, cos) with the smallest cosine."""
    we = max(res["grads"].items(), key=lambda kv: kv[1][0])
    wc = min(res["grads"].items(), key=lambda kv: kv[1][1])
    return (we[0], we[1][0]), (wc[0], wc[1][1])


def bf16_storage_response(name, step=0, state_fn=None):
    """How far bf16 STORAGE alone moves each gradient of the f32 computation: the oracle is run a second
    time with its weights and input image rounded to bf16 and compared with its own exact run.
    -> {(kind, key): cosine}.  A gradient that this rounding already turns (cosine well below 1) is not a
    meaningful direction to hold a bf16 implementation to: the engine rounds every activation as well."""
    case = Case(name)

    def rounded(state):
        e, d = state
        for sd in (e, d):
            for k in sd:
                if not O.is_buffer(k):
                    sd[k] = sd[k].to(torch.bfloat16).to(torch.float32)
        return e, d

    st = state_fn(case.init_state()) if state_fn else case.init_state()
    st_q = rounded(state_fn(case.init_state()) if state_fn else case.init_state())
    x, labels, eps, mask = case.inputs(step)
    args = (labels, eps, mask, case.y, case.z, case.P, case.bs_flag, case.n_stage)
    ref = oracle_pass_raw(st[0], st[1], x, *args)
    refq = oracle_pass_raw(st_q[0], st_q[1], x.to(torch.bfloat16).to(torch.float32), *args)
    out = {}
    for k, t in ref["agrads"].items():
        out[("dact", k)] = cosine(refq["agrads"][k], t)
    for k, t in ref["pgrads"].items():
        out[("grad", k)] = cosine(refq["pgrads"][k], t)
    return out
