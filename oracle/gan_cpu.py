"""CPU oracle for the adversarial extension (TEST INFRASTRUCTURE ONLY; imported by tests/ alone).

Parity status: UNPINNED.  The reference repository names CVAE-GAN (arXiv 1703.10155) in its
README.md:3 but contains no discriminator, classifier or adversarial loss, and holds no fixture
for them.  This file is therefore a restatement of cvae_gan_amd/gan.py's definition of the step in
PyTorch autograd, built from the pinned pieces of oracle/ref_cpu.py (encoder trunk = ResNetEncoder,
models/networks.py:164-196; decoder; sampler; KL and reconstruction losses, models/cvae.py:137-140).
It checks that the HIP path computes the gradients of the losses it says it computes; it cannot
check them against the reference.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ref_cpu as R


def critic_params(pix_dim, n_out, features, seed=0, n_stage=R.N_STAGE, ngf=R.NGF):
    """Trunk parameters (the encoder's keys, no label planes) + 'head.weight' / 'head.bias'."""
    enc, _ = R.init_params(pix_dim, 0, features // 2, seed, n_stage=n_stage, ngf=ngf)
    g = torch.Generator().manual_seed(seed + 1)
    enc["head.weight"] = torch.randn(n_out, features, generator=g) * (2.0 / features) ** 0.5
    enc["head.bias"] = torch.zeros(n_out)
    return enc


def trunk(sd, x3, n_stage):
    planes = x3.new_zeros(x3.shape[0], 0, x3.shape[2], x3.shape[3])
    return R.encoder_forward(sd, x3, planes, True, None, None, n_stage)


def _leaf(sd):
    return OrderedDict((k, v.clone().requires_grad_(True) if v.is_floating_point() and not R.is_buffer(k) else v)
                       for k, v in sd.items())


def _const(sd):
    return OrderedDict((k, v.detach()) for k, v in sd.items())


def class_centres(fC_real, labels, centres, seen, momentum):
    """Moving-average centres of the real images' classifier features per class (a constant target):
    a class seen for the first time takes its batch mean as is.  Returns new (centres, seen)."""
    centres, seen = centres.clone(), seen.clone()
    for c in labels.unique().tolist():
        m = fC_real[labels == c].mean(0).detach()
        centres[c] = (1 - momentum) * centres[c] + momentum * m if bool(seen[c]) else m
        seen[c] = 1
    return centres, seen


def step_gradients(enc_sd, dec_sd, d_sd, c_sd, x, labels, eps, mask_f, z_p, labels_p, mask_p, y_dim, batch_size,
                   lam_gd, lam_gc, n_stage=R.N_STAGE, lam_pd=0.0, lam_pc=0.0, per_class=False, centres=None, seen=None,
                   momentum=0.1):
    """Losses and gradients of one CVAE-GAN step as cvae_gan_amd/gan.py defines it.
    Returns (losses dict, grads dict per network, images dict); with per_class the images dict also
    holds the updated class centres."""
    enc, dec, dd, cc = _leaf(enc_sd), _leaf(dec_sd), _leaf(d_sd), _leaf(c_sd)
    B = x.shape[0]
    planes = R.label_planes(labels, y_dim, x.shape[-1], x.dtype)
    h_enc = R.encoder_forward(enc, x, planes, True, None, None, n_stage)
    z, mu, sigma = R.sample_latent(h_enc, eps, eps.shape[1])
    dec_f = R.decoder_forward(dec, z, R.one_hot(labels, y_dim, x.dtype), mask_f, True, None, n_stage)
    dec_p = R.decoder_forward(dec, z_p, R.one_hot(labels_p, y_dim, x.dtype), mask_p, True, None, n_stage)

    # critics' own losses: the images are constants
    x3c = torch.cat([x, dec_f.detach(), dec_p.detach()], 0)
    hD = trunk(dd, x3c, n_stage)
    logit = F.linear(hD, dd["head.weight"], dd["head.bias"])[:, 0]
    target = torch.cat([torch.ones(B), torch.zeros(2 * B)]).to(x.dtype)
    L_D = F.binary_cross_entropy_with_logits(logit, target)
    hC = trunk(cc, x3c, n_stage)
    L_C = F.cross_entropy(F.linear(hC[:B], cc["head.weight"], cc["head.bias"]), labels)

    # generator / encoder: the critics are constants, the images carry the graph
    x3 = torch.cat([x, dec_f, dec_p], 0)
    fD = trunk(_const(dd), x3, n_stage)
    fC = trunk(_const(cc), x3, n_stage)
    L_GD = 0.5 * ((fD[2 * B:].mean(0) - fD[:B].mean(0).detach()) ** 2).sum()
    extra = {}
    if per_class:
        centres, seen = class_centres(fC[:B], labels, centres, seen, momentum)
        L_GC = fC.new_zeros(())
        for c in labels_p.unique().tolist():
            if bool(seen[c]):
                L_GC = L_GC + 0.5 * ((fC[2 * B:][labels_p == c].mean(0) - centres[c]) ** 2).sum()
        extra.update(centres=centres, seen=seen)
    else:
        L_GC = 0.5 * ((fC[2 * B:].mean(0) - fC[:B].mean(0).detach()) ** 2).sum()
    # pairwise feature reconstruction: the reconstruction's features against the real image's (constants)
    L_PD = 0.5 * ((fD[B:2 * B] - fD[:B].detach()) ** 2).sum(1).mean()
    L_PC = 0.5 * ((fC[B:2 * B] - fC[:B].detach()) ** 2).sum(1).mean()
    vae, kl, ll = R.losses(dec_f, x, mu, sigma, batch_size)
    L_EG = vae + lam_gd * L_GD + lam_gc * L_GC + lam_pd * L_PD + lam_pc * L_PC

    def grads(loss, sd):
        keys = [k for k, v in sd.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [sd[k] for k in keys], retain_graph=True, allow_unused=True)
        return OrderedDict((k, g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(keys, gs))

    out_g = dict(D=grads(L_D, dd), C=grads(L_C, cc), E=grads(L_EG, enc), G=grads(L_EG, dec))
    losses = {k: float(v.detach()) for k, v in dict(KL=kl, LL=ll, D=L_D, C=L_C, GD=L_GD, GC=L_GC, PD=L_PD, PC=L_PC).items()}
    extra.update(dec_f=dec_f.detach(), dec_p=dec_p.detach(), hD=hD.detach(), hC=hC.detach())
    return losses, out_g, extra
