"""cvae_gan_amd -- MI355X-native CVAE train step (HIP kernels behind a C ABI).

The compute path is libcvaegan_hip.so (cvae_gan_amd/lib/, built by `make`);
importing this package fails if the library is missing."""
from . import _lib          # noqa: F401  (loads the library or raises)
from ._lib import F32, BF16  # noqa: F401

__all__ = ["F32", "BF16"]
