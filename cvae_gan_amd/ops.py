"""Thin torch-tensor wrappers over the C ABI (include/cvaegan.h).

torch supplies device memory and the current HIP stream; every function here
enqueues exactly the kernels of one C entry point on that stream."""
import ctypes as C

import torch

from ._lib import lib, check, ConvDesc, WMap, PackJob, CvgError, F32, BF16, PAD_ZERO, PAD_REFLECT  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    return _DT[d]


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "cvae_gan_amd ops need device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def elt(dtype):
    return 2 if dtype == torch.bfloat16 else 4


def conv_fwd(desc, x, w, y, bias=None, cbias=None, labels=None, res=None, gate=None, ws=None):
    check(lib.cvg_conv_fwd(C.byref(desc), ptr(x), ptr(w), ptr(y), ptr(bias), ptr(cbias), ptr(labels),
                           ptr(res), ptr(gate), ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0,
                           stream()))


def conv_fwd_workspace(desc):
    return int(lib.cvg_conv_fwd_workspace(C.byref(desc)))


def conv_wgrad(desc, x, dy, dy_ld, wmap, grad, ws, accumulate=False):
    check(lib.cvg_conv_wgrad(C.byref(desc), ptr(x), ptr(dy), dy_ld, C.byref(wmap), ptr(grad), int(accumulate),
                             ptr(ws), ws.numel() * ws.element_size(), stream()))


class ConvGroup:
    """Up to four convolutions of one input over one output grid, launched together."""

    def __init__(self, descs, weights):
        assert 1 <= len(descs) <= 4 and len(descs) == len(weights)
        self.n = len(descs)
        self.descs = (ConvDesc * self.n)(*descs)
        self.w = (C.c_void_p * self.n)(*[ptr(w) for w in weights])
        self._keep = weights

    def run(self, x, y, gate=None):
        """gate (laid out as y): y = conv * LeakyReLU'(gate)."""
        check(lib.cvg_conv_fwd_group(self.n, self.descs, ptr(x), self.w, ptr(y), ptr(gate), stream()))


def conv_wgrad_workspace(desc):
    return int(lib.cvg_conv_wgrad_workspace(C.byref(desc)))


def conv_fwd_kernel(desc):
    """Name of the kernel family the library picks for this descriptor."""
    return lib.cvg_conv_fwd_kernel(C.byref(desc)).decode()


def conv_wgrad_kernel(desc):
    return lib.cvg_conv_wgrad_kernel(C.byref(desc)).decode()


def conv_wgrad_strided(desc, x, dy, wmap, grad, ws, accumulate=False):
    """dy addressed through desc.y_*_pitch (see include/cvaegan.h)."""
    check(lib.cvg_conv_wgrad_strided(C.byref(desc), ptr(x), ptr(dy), C.byref(wmap), ptr(grad), int(accumulate),
                                     ptr(ws), ws.numel() * ws.element_size(), stream()))


def weight_pack(desc, wmap, canonical, packed):
    assert packed.numel() >= desc.w_rows * desc.w_ld and dt(packed) == desc.dtype
    check(lib.cvg_weight_pack(C.byref(desc), C.byref(wmap), ptr(canonical), ptr(packed), stream()))


class PackBatch:
    """Every packed-weight layout of a network, repacked by one launch.
    items: (desc, wmap, canonical f32 tensor, packed tensor); the tensors must stay alive and in place."""

    def __init__(self, items, device):
        jobs = (PackJob * len(items))()
        blocks = 0
        for i, (desc, wmap, canonical, packed) in enumerate(items):
            assert packed.numel() >= desc.w_rows * desc.w_ld and dt(packed) == desc.dtype
            nb = lib.cvg_weight_pack_job(C.byref(desc), C.byref(wmap), ptr(canonical), ptr(packed), blocks,
                                         C.byref(jobs[i]))
            if nb < 0:
                raise CvgError(lib.cvg_last_error().decode())
            blocks += nb
        self.n_jobs, self.blocks = len(items), blocks
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        self.table = raw.to(device)
        self._keep = items

    def run(self):
        check(lib.cvg_weight_pack_batch(ptr(self.table), self.n_jobs, self.blocks, stream()))


def label_bias(w_stem, cbias, cout, y_dim):
    check(lib.cvg_label_bias(ptr(w_stem), ptr(cbias), cout, y_dim, stream()))


def label_bias_bwd(dy, labels, B, pixels, cout, y_dim, dw, dbias, ws):
    check(lib.cvg_label_bias_bwd(ptr(dy), dt(dy), ptr(labels), B, pixels, cout, y_dim, ptr(dw), ptr(dbias),
                                 ptr(ws), stream()))


def pack_image(x, xpad, B, P):
    check(lib.cvg_pack_image(ptr(x), ptr(xpad), B, P, dt(xpad), stream()))


def bn_stats_blocks(M, C_):
    return int(lib.cvg_bn_stats_blocks(M, C_))


def bn_stats(x, M, C_, stats, running_mean, running_var, nbt, ws):
    check(lib.cvg_bn_stats(ptr(x), dt(x), M, C_, ptr(stats), ptr(running_mean), ptr(running_var), ptr(nbt),
                           ptr(ws), stream()))


def conv_stats_partials(desc):
    """Partial rows cvg_conv_fwd_stats writes for this layer (0: the layer runs on the gather kernel)."""
    return int(lib.cvg_conv_stats_partials(C.byref(desc)))


def conv_fwd_stats(desc, x, w, y, partials, bias=None, cbias=None, labels=None):
    check(lib.cvg_conv_fwd_stats(C.byref(desc), ptr(x), ptr(w), ptr(y), ptr(bias), ptr(cbias), ptr(labels),
                                 ptr(partials), stream()))


def bn_stats_finalize(partials, nblk, M, C_, stats, running_mean, running_var, nbt):
    check(lib.cvg_bn_stats_finalize(ptr(partials), nblk, M, C_, ptr(stats), ptr(running_mean), ptr(running_var),
                                    ptr(nbt), stream()))


def bn_stats_static(x, M, C_, stats, running_mean, running_var, nbt):
    """bn_stats without a workspace: the partials go to a static array inside the library (C <= 1024)."""
    check(lib.cvg_bn_stats_static(ptr(x), dt(x), M, C_, ptr(stats), ptr(running_mean), ptr(running_var), ptr(nbt), stream()))


def bn_eval_stats(rm, rv, C_, stats):
    check(lib.cvg_bn_eval_stats(ptr(rm), ptr(rv), C_, ptr(stats), stream()))


def bn_act_fwd(x, res, y, M, C_, stats, gamma, beta):
    check(lib.cvg_bn_act_fwd(ptr(x), ptr(res), ptr(y), dt(x), M, C_, ptr(stats), ptr(gamma), ptr(beta), stream()))


def bn_act_bwd(dy, y, x, dx, M, C_, stats, gamma, dgamma, dbeta, ws, beta=None):
    """y=None with beta: the activation's sign is recomputed from x instead of read from y;
    y=None without beta: dy already carries the activation's derivative."""
    check(lib.cvg_bn_act_bwd(ptr(dy), ptr(y), ptr(x), ptr(dx), dt(x), M, C_, ptr(stats), ptr(gamma), ptr(beta),
                             ptr(dgamma), ptr(dbeta), ptr(ws), stream()))


def colsum(x, M, C_, ld, out, ws, map_d=1, map_a=1, map_b=0):
    check(lib.cvg_colsum(ptr(x), dt(x), M, C_, ld, ptr(out), map_d, map_a, map_b, ptr(ws), stream()))


def permute_f32(src, out, n, d=1, a=1, b=0):
    check(lib.cvg_permute_f32(ptr(src), ptr(out), n, d, a, b, stream()))


def reparam_kl_fwd(h, eps, labels, B, z, y_dim, zin_ld, zin, z_mean, z_sigma, kl, batch_size, h_ld=None):
    """h_ld: row pitch of h (default: its second dimension)."""
    h_ld = h.shape[-1] if h_ld is None else h_ld
    check(lib.cvg_reparam_kl_fwd(ptr(h), dt(h), h_ld, ptr(eps), ptr(labels), B, z, y_dim, zin_ld, ptr(zin),
                                 ptr(z_mean), ptr(z_sigma), ptr(kl), float(batch_size), stream()))


def reparam_kl_bwd(dzin, zin_ld, eps, z_mean, z_sigma, g_mean, g_sigma, kl_scale, batch_size, B, z, dh, h_ld=None):
    h_ld = dh.shape[-1] if h_ld is None else h_ld
    check(lib.cvg_reparam_kl_bwd(ptr(dzin), dt(dzin), zin_ld, ptr(eps), ptr(z_mean), ptr(z_sigma), ptr(g_mean),
                                 ptr(g_sigma), float(kl_scale), float(batch_size), B, z, ptr(dh), h_ld, stream()))


def labels_prepare(labels, out, n, n_class, bad):
    check(lib.cvg_labels_prepare(ptr(labels), ptr(out), n, n_class, ptr(bad), stream()))


def lrelu_dropout_fwd(x, mask, y, n):
    check(lib.cvg_lrelu_dropout_fwd(ptr(x), ptr(mask), ptr(y), dt(x), n, stream()))


def lrelu_dropout_bwd(dy, y, mask, dx, n):
    check(lib.cvg_lrelu_dropout_bwd(ptr(dy), ptr(y), ptr(mask), ptr(dx), dt(y), n, stream()))


def dropout_mask(mask, n, seed):
    check(lib.cvg_dropout_mask(ptr(mask), n, seed, stream()))


def upsample_pad_fwd(x, u, B, h, w, C_):
    check(lib.cvg_upsample_pad_fwd(ptr(x), ptr(u), dt(x), B, h, w, C_, stream()))


def upsample_pad_bwd(du, dx, B, h, w, C_):
    check(lib.cvg_upsample_pad_bwd(ptr(du), ptr(dx), dt(du), B, h, w, C_, stream()))


def upsample_taps_fwd(v, bias, y, B, h, w, N):
    """y [B][2h][2w][N] = bias + the up-sampling / padding / tap-shift operator applied to v [B][h][w][9][N]."""
    check(lib.cvg_upsample_taps_fwd(ptr(v), ptr(bias), ptr(y), dt(v), B, h, w, N, stream()))


def upsample_taps_bwd(dy, z, B, h, w, N):
    """z [B][h][w][9][N] = the nine tap-shifted transposed up-samplings of dy [B][2h][2w][N]."""
    check(lib.cvg_upsample_taps_bwd(ptr(dy), ptr(z), dt(dy), B, h, w, N, stream()))


def reflect_fold(dxp, dx, B, H, W, C_, pad):
    check(lib.cvg_reflect_fold(ptr(dxp), ptr(dx), dt(dxp), B, H, W, C_, pad, stream()))


def tanh_fwd(pre, dec, B, P):
    check(lib.cvg_tanh_fwd(ptr(pre), ptr(dec), dt(pre), B, P, stream()))


def image_u8_to_f32(hwc_u8, nchw_f32, B, P):
    assert hwc_u8.dtype == torch.uint8 and nchw_f32.dtype == torch.float32
    check(lib.cvg_image_u8_to_f32(ptr(hwc_u8), ptr(nchw_f32), B, P, stream()))


def mfma_rate_probe(out, workgroups, iters):
    """Measurement aid: a pure MFMA loop (see cvg_mfma_rate_probe); returns the FLOPs it executes."""
    assert out.dtype == torch.float32 and out.numel() >= workgroups * 256
    check(lib.cvg_mfma_rate_probe(ptr(out), workgroups, iters, stream()))
    return workgroups * 4 * iters * 32 * 32768


def window_col_sum(v, out, B, rows, P):
    check(lib.cvg_window_col_sum(ptr(v), ptr(out), dt(v), B, rows, P, stream()))


def tail_sum_tanh(v, bias, pre, dec, B, P):
    check(lib.cvg_tail_sum_tanh(ptr(v), ptr(bias), ptr(pre), ptr(dec), dt(v), B, P, stream()))


def tanh_bwd(d_dec, dec, dpre, B, P, dpre_win=None):
    check(lib.cvg_tanh_bwd(ptr(d_dec), ptr(dec), ptr(dpre), ptr(dpre_win),
                           dt(dpre), B, P, stream()))


def mse_fwd_bwd(dec, x, d_dec, ll, numel, batch_size, scale, ws):
    check(lib.cvg_mse_fwd_bwd(ptr(dec), ptr(x), ptr(d_dec), ptr(ll), numel, float(batch_size), float(scale),
                              ptr(ws), stream()))


def adamw(p, g, m, v, n, lr, beta1, beta2, eps, wd, step):
    check(lib.cvg_adamw(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, wd, step, stream()))


def adamw_dev(p, g, m, v, n, lr, beta1, beta2, eps, wd, step_counter, bc):
    check(lib.cvg_adamw_dev(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, beta1, beta2, eps, wd, ptr(step_counter),
                            ptr(bc), stream()))


def dropout_mask_dev(mask, n, seed_counter):
    check(lib.cvg_dropout_mask_dev(ptr(mask), n, ptr(seed_counter), stream()))


def cast(src, out, n):
    check(lib.cvg_cast(ptr(src), dt(src), ptr(out), dt(out), n, stream()))


def nhwc_to_nchw(src, out, B, HW, C_, ld):
    check(lib.cvg_nhwc_to_nchw(ptr(src), dt(src), ptr(out), B, HW, C_, ld, stream()))


def nchw_to_nhwc(src, out, B, HW, C_, ld):
    check(lib.cvg_nchw_to_nhwc(ptr(src), ptr(out), dt(out), B, HW, C_, ld, stream()))


# ---- adversarial extension (discriminator / classifier heads and losses) -------------------
def gan_d_head(h, ld, w, b, M, F, n_real, grad_scale, dw, db, loss, dh, ws):
    check(lib.cvg_gan_d_head(ptr(h), dt(h), ld, ptr(w), ptr(b), M, F, n_real, grad_scale, ptr(dw), ptr(db), ptr(loss),
                             ptr(dh), ptr(ws), stream()))


def gan_c_head(h, ld, W, b, labels, M, F, y, n_real, grad_scale, dW, db, loss, dh, ws):
    check(lib.cvg_gan_c_head(ptr(h), dt(h), ld, ptr(W), ptr(b), ptr(labels), M, F, y, n_real, grad_scale, ptr(dW),
                             ptr(db), ptr(loss), ptr(dh), ptr(ws), stream()))


def gan_feature_match(h, ld, M, F, r0, r1, f0, f1, grad_scale, loss, dh):
    check(lib.cvg_gan_feature_match(ptr(h), dt(h), ld, M, F, r0, r1, f0, f1, grad_scale, ptr(loss), ptr(dh), stream()))


def gan_class_feature_match(h, ld, F, labels_real, labels_fake, r0, r1, f0, f1, n_class, centres, seen, momentum,
                            grad_scale, loss, dh, ws):
    check(lib.cvg_gan_class_feature_match(ptr(h), dt(h), ld, F, ptr(labels_real), ptr(labels_fake), r0, r1, f0, f1, n_class,
                                          ptr(centres), ptr(seen), momentum, grad_scale, ptr(loss), ptr(dh), ptr(ws),
                                          stream()))


def gan_pair_feature_match(h, ld, F, r0, f0, n, grad_scale, loss, dh, ws):
    check(lib.cvg_gan_pair_feature_match(ptr(h), dt(h), ld, F, r0, f0, n, grad_scale, ptr(loss), ptr(dh), ptr(ws), stream()))


def axpy(y, x, alpha, n=None):
    """y += alpha * x (f32)."""
    assert y.dtype == torch.float32 and x.dtype == torch.float32
    check(lib.cvg_axpy_f32(ptr(y), ptr(x), alpha, y.numel() if n is None else n, stream()))


# ---- inference: eval-mode BatchNorm + LeakyReLU in the convolution's epilogue ----------------
def bn_eval_scale_shift(rm, rv, gamma, beta, C_, scale, shift):
    check(lib.cvg_bn_eval_scale_shift(ptr(rm), ptr(rv), ptr(gamma), ptr(beta), C_, ptr(scale), ptr(shift), stream()))


def conv_bn_act_fwd(desc, x, w, y, scale, shift, bias=None, cbias=None, labels=None, res=None):
    check(lib.cvg_conv_bn_act_fwd(C.byref(desc), ptr(x), ptr(w), ptr(y), ptr(bias), ptr(cbias), ptr(labels),
                                  ptr(scale), ptr(shift), ptr(res), stream()))
