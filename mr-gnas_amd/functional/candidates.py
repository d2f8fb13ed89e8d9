"""The explicit MixedOp candidate objects (Link, Candidate) the operators hand to the fused epilogue.

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib


class Link:
    """Mailbox between ONE producing autograd node (a dense filter, the dense-filter pair, the scalar gate as a row factor) and the
    MixedOp epilogue that is the only reader of its output(s).  The producer's wrapper creates it, hands it to the node's forward
    (which keeps it on its ctx and fills in what the epilogue recomputes candidates from) and returns it inside the Candidate; the
    epilogue's BACKWARD then writes the producer's first backward pass itself and says so here, and the producer's backward reads
    it.  This replaces the attributes rounds 2-3 hung on tensors and on grad_fn objects: nothing is inferred from a
    tensor's identity any more -- a Candidate either reaches mixed_epilogue_prepare with its Link or it is a plain stored tensor.

    slot = which of the node's outputs (0; the pair: 0 = f_dense_comp, 1 = f_comp)."""

    __slots__ = ("folded", "written", "gs_direct", "s", "gate", "row_h", "row_uvc", "row_written")

    def __init__(self, slots=1):
        self.folded = [False] * slots       # claimed by an epilogue: it will write this output's gradient as the producer's `dz`
        self.written = [None] * slots       # address of the gradient buffer the epilogue wrote for the slot (checked on arrival)
        self.gs_direct = None               # gated kinds: the direct term of the gradient w.r.t. s, written by the epilogue
        self.s = self.gate = None           # what a gate-only candidate is recomputed from (the node's own saved tensors)
        self.row_h = self.row_uvc = None    # row factor: h_r = t_r g (1 - g) and the collapsed gate vectors
        self.row_written = None             # row factor: address of the [rows] gradient the epilogue wrote AND folded into gs

    def claim(self, slot):
        if self.folded[slot]:
            return False                    # a second epilogue reads the same output: that one gets the stored form
        self.folded[slot] = True
        return True

    def arrived(self, slot, g, what):
        """In the producer's backward: was gradient `g` of output `slot` written by the epilogue in the folded form?  A folded slot
        whose gradient is NOT the buffer the epilogue wrote means the output had a second consumer and autograd combined the two."""
        if not self.folded[slot] or self.written[slot] is None:
            return False
        if self.written[slot] != g.data_ptr():
            raise _lib.MrgnasError(f"{what}: the folded epilogue gradient was combined with another consumer's gradient; "
                                   "call mixed_epilogue(fold_row_scales=False) when a candidate's output is read elsewhere")
        return True


class Candidate:
    """What an operator's `for_epilogue=True` path hands to mixed_epilogue_prepare instead of a bare [rows, D] tensor.

      kind "stored"     y = the operator's output; `rowscale` (norm, b1, scale_edge, scale_self, gated) lets the epilogue's gradient
                        store perform the producer's first backward pass (dz = g * c_r [* s gate (1 - gate)], direct term)
      kind "gate"       y = f_dense_comp's GATE; the candidate gate * s * c_r is recomputed wherever it is read (never stored)
      kind "rowfactor"  y = f_sparse_comp's gate as ONE factor per row, [rows]; the candidate is s * y[:, None]

    `y` is the tensor that takes part in autograd; `link` the producer's mailbox; `slot` the producer's output index."""

    __slots__ = ("kind", "y", "s", "c", "rowscale", "link", "slot", "b0", "b1")

    def __init__(self, kind, y, link=None, slot=0, s=None, c=None, rowscale=None, b0=0, b1=0):
        self.kind, self.y, self.link, self.slot, self.s, self.c, self.rowscale, self.b0, self.b1 = kind, y, link, slot, s, c, rowscale, b0, b1

    def materialize(self):
        """The candidate as a plain [rows, D] tensor (for consumers other than the fused epilogue)."""
        if self.kind == "stored":
            return self.y
        if self.kind == "rowfactor":
            return self.s * self.y.unsqueeze(1)
        return self.y * self.s * self.c.unsqueeze(1)
