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


def test_struct_layout_matches_header():
    from cvae_gan_amd import _lib
    assert C.sizeof(_lib.ConvDesc) == 28 * 4
    assert C.sizeof(_lib.WMap) == 12 * 4 + 64 * 2
