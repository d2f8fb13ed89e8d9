"""a9: fused gather -> compose -> segmented sum (CompGCN aggregation, csrc/fused_gcs.hip) and its backward.

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .._lib import call, f32c, ptr, require_hip, stream_of
from ._base import _cnt, _ws, _ws_bytes


GCS = {"sub": 0, "mul": 1, "copy": 2, "negs": 3, "ccorr": 4, "cconv": 5}


def fused_gcs(mode, X, xi, Y, yi, scal, plan, nseg):
    """out[seg] = sum_{e in seg} combine(mode, X[xi[e]], Y[yi[e]], scal[e]); forward-only primitive
    (mrg_fused_gcs).  `plan` = graph.dst_csr_plan(segment key per element, nseg)."""
    X, Y, scal = f32c(X), f32c(Y), f32c(scal)
    require_hip(X, Y, scal, xi, yi)
    D = X.shape[1]
    out = torch.empty(nseg, D, dtype=torch.float32, device=X.device)
    n_chunks, n_hubs, n_slots = _cnt(plan, "chunks"), _cnt(plan, "hubs"), _cnt(plan, "slots")
    ws = _ws(_ws_bytes("mrg_seg_reduce_workspace_bytes", n_slots, D), X) if n_slots > 0 else None
    E = int(plan["eid"].numel())
    rows_y = Y.shape[0] if Y is not None else 0
    nb = E * (8 + 4 * D) + 4 * (nseg + 1) + 4 * D * (rows_y + nseg)
    call("mrg_fused_gcs", (GCS[mode], ptr(X), ptr(xi), ptr(Y), ptr(yi), ptr(scal), ptr(plan["eid"]), ptr(plan["chunk_node"]),
                           ptr(plan["chunk_start"]), ptr(plan["chunk_end"]), ptr(plan["chunk_slot"]), n_chunks,
                           ptr(plan["hub_node"]), ptr(plan["hub_first"]), ptr(plan["hub_count"]), n_hubs,
                           n_slots, ptr(plan["in_degree"]), ptr(out), ptr(ws), nseg, D, stream_of(X)),
         nbytes=nb, flops=(2 * E * D * D if mode in ("ccorr", "cconv") else 0))
    return out


def span_gcs(mode, X, Y, meta, plan, ext_scal=None):
    """mrg_span_gcs: balanced span form of fused_gcs for the elementwise modes."""
    X, Y, ext_scal = f32c(X), f32c(Y), f32c(ext_scal)
    require_hip(X, Y, meta, ext_scal)
    D, nseg, E = X.shape[1], plan["nseg"], plan["E"]
    out = torch.empty(nseg, D, dtype=torch.float32, device=X.device)     # every row is written: runs, hubs, empty segments
    n_hubs, n_slots = _cnt(plan, "hubs"), _cnt(plan, "slots")
    ws = _ws(_ws_bytes("mrg_seg_reduce_workspace_bytes", n_slots, D), X) if n_slots > 0 else None
    rows_y = Y.shape[0] if Y is not None else 0
    nb = E * (8 + 4 * D) + 4 * (nseg + 1) + 4 * D * (rows_y + nseg)
    call("mrg_span_gcs", (GCS[mode], ptr(X), ptr(Y), ptr(meta), ptr(ext_scal), E, plan["span"], ptr(plan["span_slot"]), ptr(plan.get("span_start")),
                          plan["n_spans"],
                          ptr(plan["hub_seg"]), ptr(plan["hub_first"]), ptr(plan["hub_count"]), n_hubs, n_slots,
                          ptr(plan["seg_len"]), ptr(out), ptr(ws), nseg, D, stream_of(X)), nbytes=nb)
    return out


class ComposePlan:
    """Index structure of one compose-and-aggregate: element e reads node row xi[e] and relation
    row yi[e], is scaled by scal[e] and summed into segment seg[e].  The span plans / packed metadata of the
    forward (by segment) and of the backward (by node row, by relation row), and the chunk plans the O(D^2)
    ccorr kernels use, are each built at first use (HIP plan builders) and settled before they are cached."""

    def __init__(self, xi, yi, seg, scal, n_x, n_y, n_seg):
        i32 = lambda t: t.to(torch.int32).contiguous()
        self.xi, self.yi, self.seg = i32(xi), i32(yi), i32(seg)
        self.scal = None if scal is None else scal.float().contiguous()
        self.n_x, self.n_y, self.n_seg = int(n_x), int(n_y), int(n_seg)
        self._c = {}

    def _lazy(self, key, build):
        v = self._c.get(key)
        if v is None:
            from ..graph import settle
            v = self._c[key] = build()
            settle(self.xi.device)
        return v

    def _span(self, which):
        from ..graph import span_plan
        keys = {"seg": (self.seg, self.n_seg), "x": (self.xi, self.n_x), "y": (self.yi, self.n_y)}[which]
        return self._lazy("sp_" + which, lambda: span_plan(*keys))

    def _chunk(self, which):
        from ..graph import dst_csr_plan
        keys = {"seg": (self.seg, self.n_seg), "x": (self.xi, self.n_x), "y": (self.yi, self.n_y)}[which]
        return self._lazy("by_" + which, lambda: dst_csr_plan(*keys))

    sp_seg = property(lambda self: self._span("seg"))
    sp_x = property(lambda self: self._span("x"))
    sp_y = property(lambda self: self._span("y"))
    by_seg = property(lambda self: self._chunk("seg"))
    by_x = property(lambda self: self._chunk("x"))
    by_y = property(lambda self: self._chunk("y"))

    def _meta(self, key, plan, a, b, scal):
        from ..graph import span_meta
        return self._lazy(key, lambda: span_meta(plan, a, b, scal))

    m_fwd = property(lambda self: self._meta("m_fwd", self.sp_seg, self.xi, self.yi, self.scal))      # out[seg] <- X[xi] (op) Y[yi]*s
    m_bx = property(lambda self: self._meta("m_bx", self.sp_x, self.seg, self.yi, self.scal))          # gX[xi]   <- G[seg] (op) Y[yi]*s
    m_by_g = property(lambda self: self._meta("m_by_g", self.sp_y, self.seg, self.xi, self.scal))      # gY[yi]   <- G[seg] (op) X[xi]*s

    def m_bx_unit(self):
        """metadata of gX[xi] <- G[seg] with unit scale (d/dx of x - y*s)."""
        return self._meta("m_bx_unit", self.sp_x, self.seg, None, None)


class _ComposeAggregate(torch.autograd.Function):
    """A[seg] = sum_e phi(X[xi_e], Y[yi_e] * s_e), phi in {sub, mul, ccorr}
    (reference models/compgcn.py:58-87 without the per-direction linears)."""

    @staticmethod
    def forward(ctx, kind, X, Y, cp):
        X, Y = f32c(X), f32c(Y)
        ctx.kind, ctx.cp = kind, cp
        ctx.save_for_backward(X, Y)
        if kind in ("sub", "mul"):
            return span_gcs(kind, X, Y, cp.m_fwd, cp.sp_seg)
        return fused_gcs(kind, X, cp.xi, Y, cp.yi, cp.scal, cp.by_seg, cp.n_seg)

    @staticmethod
    def backward(ctx, G):
        X, Y = ctx.saved_tensors
        cp, kind = ctx.cp, ctx.kind
        G = f32c(G)
        gX = gY = None
        if kind == "sub":       # x - y s
            if ctx.needs_input_grad[1]:
                gX = span_gcs("copy", G, None, cp.m_bx_unit(), cp.sp_x)
            if ctx.needs_input_grad[2]:
                gY = span_gcs("negs", G, None, cp.m_by_g, cp.sp_y)
        elif kind == "mul":     # x * y s
            if ctx.needs_input_grad[1]:
                gX = span_gcs("mul", G, Y, cp.m_bx, cp.sp_x)
            if ctx.needs_input_grad[2]:
                gY = span_gcs("mul", G, X, cp.m_by_g, cp.sp_y)
        else:                   # ccorr(x, y s)
            if ctx.needs_input_grad[1]:
                gX = fused_gcs("ccorr", G, cp.seg, Y, cp.yi, cp.scal, cp.by_x, cp.n_x)
            if ctx.needs_input_grad[2]:
                gY = fused_gcs("cconv", X, cp.xi, G, cp.seg, cp.scal, cp.by_y, cp.n_y)
        return None, gX, gY, None


def compose_aggregate(kind, X, Y, cp):
    if kind not in ("sub", "mul", "ccorr"):
        raise Exception('Only supports sub, mul, and ccorr')
    return _ComposeAggregate.apply(kind, X, Y, cp)
