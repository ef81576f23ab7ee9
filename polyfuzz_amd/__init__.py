"""polyfuzz_amd -- MI355X (gfx950) engine behind PolyFuzz's TFIDF / EditDistance matchers.

Host side: Python mirror of the reference's matcher interface
(polyfuzz/models/_base.py, _tfidf.py, _distance.py, _utils.py) over the C ABI of
include/polyfuzz_hip.h (ctypes; no torch in the data path).  Device side:
hand-written HIP kernels in polyfuzz_amd/csrc.  There is no CPU fallback.
"""
from . import _lib                      # noqa: F401
from ._lib import Context, PfzError, PfzNoDevice, PfzUnsupported, device_count  # noqa: F401

__version__ = "0.1.0"
