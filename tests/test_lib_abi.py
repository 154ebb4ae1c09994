"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU, exports every
symbol include/cvaegan.h declares, and rejects malformed descriptors before launching anything."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cvaegan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cvae_gan_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), "libcvaegan_hip.so does not export %s" % s
    # and the Python binding table covers the header
    assert sorted(_lib.EXPORTS) == syms


def test_version_and_error_text():
    from cvae_gan_amd import _lib
    assert _lib.lib.cvg_version() >= 1
    assert isinstance(_lib.lib.cvg_last_error(), bytes)


def test_bad_descriptor_is_rejected_without_a_gpu():
    from cvae_gan_amd import _lib
    d = _lib.ConvDesc()
    d.dtype, d.B, d.H, d.W, d.Cin, d.OH, d.OW, d.N, d.KW, d.ntaps = 1, 1, 4, 4, 12, 4, 4, 8, 3, 9
    rc = _lib.lib.cvg_conv_fwd(C.byref(d), None, None, None, None, None, None, None, None, None, 0, None)
    assert rc != 0
    assert b"Cin" in _lib.lib.cvg_last_error()


def _desc(**kw):
    from cvae_gan_amd import _lib
    d = _lib.ConvDesc()
    base = dict(dtype=1, B=1, H=8, W=8, Cin=16, OH=8, OW=8, N=16, KW=3, ntaps=9, stride=1, tstep_y=1, tstep_x=1,
                off_y=-1, off_x=-1, pad_mode=0, x_pix_pitch=16, x_row_pitch=8 * 16, x_img_pitch=64 * 16,
                y_pix_pitch=16, y_row_pitch=8 * 16, y_img_pitch=64 * 16, oy_mul=1, ox_mul=1, w_ld=192, w_rows=128)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def _rejected(d, needle):
    from cvae_gan_amd import _lib
    rc = _lib.lib.cvg_conv_fwd(C.byref(d), None, None, None, None, None, None, None, None, None, 0, None)
    assert rc != 0
    assert needle in _lib.lib.cvg_last_error(), _lib.lib.cvg_last_error()


def test_size_and_padding_limits_are_rejected_without_a_gpu():
    """The kernels mirror a coordinate once and need power-of-two channel counts per tap: a descriptor
    beyond a limit must fail in validation, never reach a launch.  (Batches beyond 2^31 elements are
    fine: image bases are 64-bit, tests/test_gpu_fullsize.py runs one.)"""
    # 640 images of 130 x 130 x 256 (a padded 128 x 128 map) = 2.77e9 elements: valid geometry, fails only
    # on the null tensors
    _rejected(_desc(B=640, H=130, W=130, Cin=256, OH=128, OW=128, x_pix_pitch=256, x_row_pitch=130 * 256,
                    x_img_pitch=130 * 130 * 256, w_ld=9 * 256, y_pix_pitch=16, y_row_pitch=128 * 16,
                    y_img_pitch=128 * 128 * 16), b"null tensor")
    # reflection by 3 into a 2-pixel-wide tensor needs two reflections
    _rejected(_desc(H=2, W=2, OH=2, OW=2, KW=7, ntaps=49, off_y=-3, off_x=-3, pad_mode=1, x_row_pitch=2 * 16,
                    x_img_pitch=4 * 16), b"reflection")
    # channels per tap must be a power of two when there is more than one tap
    _rejected(_desc(Cin=24, x_pix_pitch=24, x_row_pitch=8 * 24, x_img_pitch=64 * 24), b"power of two")
    # a null tensor with an otherwise valid descriptor
    _rejected(_desc(), b"null tensor")


def test_struct_layout_matches_header():
    from cvae_gan_amd import _lib
    assert C.sizeof(_lib.ConvDesc) == 31 * 4
    assert C.sizeof(_lib.WMap) == 12 * 4 + 64 * 2
