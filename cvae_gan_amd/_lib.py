"""ctypes binding of libcvaegan_hip.so (declarations: include/cvaegan.h).

The library is the only compute path of this package.  If it is missing the import
fails loudly; there is no CPU or PyTorch fallback."""
import ctypes as C
import os

# torch first: it brings the HIP runtime its streams and allocations live in, and the library must
# bind to that same copy.  Loaded the other way round (this library first, e.g. `python -m
# cvae_gan_amd.starter`), the process ends up with two runtimes and every launch on a torch stream
# fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVG_LIB") or os.path.join(_HERE, "lib", "libcvaegan_hip.so")   # CVG_LIB: A/B builds

F32, BF16 = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1
FORM_GATHER, FORM_S2_TRANSPOSED = 0, 1
CONV_ALONE = 1
WS_HEADER = 40960        # floats of arrival counters at the start of a reduction workspace (CVG_WS_HEADER)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "B", "H", "W", "Cin", "x_img_pitch", "x_row_pitch", "x_pix_pitch",
        "OH", "OW", "N", "KW", "ntaps", "stride", "tstep_y", "tstep_x", "off_y", "off_x",
        "pad_mode", "y_img_pitch", "y_row_pitch", "y_pix_pitch", "oy_mul", "oy_off",
        "ox_mul", "ox_off", "w_ld", "w_rows", "co_launches", "form", "flags")]


class WMap(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nd", "na", "nb", "kd", "ka", "kb", "n_valid", "kc_valid", "ntaps", "tap_mode", "nr_valid", "reserved")] + \
        [("tap_table", C.c_int16 * 64)]


class PackJob(C.Structure):
    _fields_ = [("map", WMap), ("canonical", C.c_void_p), ("packed", C.c_void_p), ("n_groups", C.c_int64)] + \
        [(n, C.c_int32) for n in ("block0", "mode", "rows", "ld", "nchunks", "ntaps", "Cin", "dtype")]


class CvgError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "cvae_gan_amd: %s is missing. Build it with `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`) at the repository root; "
            "this package has no fallback compute path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    f64 = C.c_double
    P = C.POINTER
    sig = {
        "cvg_version": (C.c_int, []),
        "cvg_last_error": (C.c_char_p, []),
        "cvg_conv_fwd_workspace": (sz, [P(ConvDesc)]),
        "cvg_conv_fwd": (C.c_int, [P(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "cvg_conv_fwd_group": (C.c_int, [i32, P(ConvDesc), vp, P(vp), vp, vp, vp]),
        "cvg_conv_wgrad_workspace": (sz, [P(ConvDesc)]),
        "cvg_conv_fwd_kernel": (C.c_char_p, [P(ConvDesc)]),
        "cvg_conv_wgrad_kernel": (C.c_char_p, [P(ConvDesc)]),
        "cvg_conv_wgrad": (C.c_int, [P(ConvDesc), vp, vp, i32, P(WMap), vp, i32, vp, sz, vp]),
        "cvg_conv_wgrad_strided": (C.c_int, [P(ConvDesc), vp, vp, P(WMap), vp, i32, vp, sz, vp]),
        "cvg_weight_pack": (C.c_int, [P(ConvDesc), P(WMap), vp, vp, vp]),
        "cvg_weight_pack_job": (C.c_int, [P(ConvDesc), P(WMap), vp, vp, i32, P(PackJob)]),
        "cvg_weight_pack_batch": (C.c_int, [vp, i32, i32, vp]),
        "cvg_label_bias": (C.c_int, [vp, vp, i32, i32, vp]),
        "cvg_label_bias_bwd": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
        "cvg_pack_image": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "cvg_bn_stats_blocks": (i32, [i32, i32]),
        "cvg_bn_stats": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "cvg_conv_stats_partials": (i32, [P(ConvDesc)]),
        "cvg_conv_fwd_stats": (C.c_int, [P(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
        "cvg_bn_stats_finalize": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
        "cvg_bn_stats_static": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
        "cvg_bn_act_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
        "cvg_bn_eval_stats": (C.c_int, [vp, vp, i32, vp, vp]),
        "cvg_bn_eval_scale_shift": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, vp]),
        "cvg_conv_bn_act_fwd": (C.c_int, [P(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "cvg_bn_act_bwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "cvg_colsum": (C.c_int, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp]),
        "cvg_permute_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "cvg_reparam_kl_fwd": (C.c_int, [vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, f32, vp]),
        "cvg_reparam_kl_bwd": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, i32, vp]),
        "cvg_labels_prepare": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "cvg_lrelu_dropout_fwd": (C.c_int, [vp, vp, vp, i32, i64, vp]),
        "cvg_lrelu_dropout_bwd": (C.c_int, [vp, vp, vp, vp, i32, i64, vp]),
        "cvg_dropout_mask": (C.c_int, [vp, i64, C.c_uint64, vp]),
        "cvg_upsample_pad_fwd": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "cvg_upsample_pad_bwd": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "cvg_upsample_taps_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "cvg_upsample_taps_bwd": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "cvg_reflect_fold": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        "cvg_image_u8_to_f32": (C.c_int, [vp, vp, i32, i32, vp]),
        "cvg_tanh_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "cvg_tail_sum_tanh": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "cvg_window_col_sum": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "cvg_mfma_rate_probe": (C.c_int, [vp, i32, i32, vp]),
        "cvg_tanh_bwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "cvg_mse_fwd_bwd": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, vp, vp]),
        "cvg_adamw": (C.c_int, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i32, vp]),
        "cvg_adamw_dev": (C.c_int, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, vp, vp, vp]),
        "cvg_dropout_mask_dev": (C.c_int, [vp, i64, vp, vp]),
        "cvg_cast": (C.c_int, [vp, i32, vp, i32, i64, vp]),
        "cvg_nhwc_to_nchw": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, vp]),
        "cvg_nchw_to_nhwc": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "cvg_gan_d_head": (C.c_int, [vp, i32, i32, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
        "cvg_gan_c_head": (C.c_int, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
        "cvg_gan_feature_match": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
        "cvg_gan_class_feature_match": (C.c_int, [vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, f32, vp, vp,
                                                   vp, vp]),
        "cvg_gan_pair_feature_match": (C.c_int, [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
        "cvg_axpy_f32": (C.c_int, [vp, vp, f32, i64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def check(rc):
    if rc != 0:
        raise CvgError(lib.cvg_last_error().decode())
