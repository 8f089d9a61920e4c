"""nerf_amd -- MI355X-native NeRF training/rendering hot path (HIP kernels behind a C ABI).

The numerical work lives in libnerf_amd.so (nerf_amd/csrc, headers in include/).  This package is
the thin host-side mirror used by the tests, bench.py and the data-parallel driver.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library has not been built)
from . import api  # noqa: F401

__all__ = ["api", "_lib"]
