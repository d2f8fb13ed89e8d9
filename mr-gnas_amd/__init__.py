"""mr_gnas_amd -- MI355X-native relational message-passing operators for MR-GNAS.

The hot path (compose -> gate -> destination-segmented reduce, and the CompGCN
layer) runs in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/mrgnas.h`` (``lib/libmrgnas_hip.so``).  The Python layer mirrors the
reference's operator API (``models/operations_lp.py`` registries and
``nn.Module`` signatures, ``models/compgcn.py``) so the reference's cells and
supernet call into it unchanged.  There is no CPU fallback: using an operator
without the built library, or on non-HIP tensors, raises.
"""
__version__ = "0.1.0"

from . import _lib            # noqa: F401  (does not load the .so until first use)
from .graph import RelGraph   # noqa: F401
