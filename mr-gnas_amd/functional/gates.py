"""a2 / a3: collapsed scalar gates f_sparse_comp / f_sparse_last (csrc/gate.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib
from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of
from ._base import _ws, _ws_bytes, gate_ld, same_rows
from .candidates import Candidate, Link


class _Gate(torch.autograd.Function):
    """Parameters come flat, three per row segment (in, out, self): W, b, a -- None for
    an absent segment.  The three segments' parameter collapse / parameter gradient are one launch each
    (mrg_gate_collapse3 / mrg_gate_param_grad3), with the same-rows fold built in."""

    @staticmethod
    def forward(ctx, s, s_in, norm, b0, b1, scale, *params):
        tied = s_in is not None and same_rows(s, s_in)         # [s ; s] : u.s + v.s = (u + v).s -- one operand is streamed
        s, s_in, norm = f32c(s), (None if tied else f32c(s_in)), f32c(norm)
        params = tuple(f32c(p) for p in params)
        require_hip(s, s_in, norm, *params)
        M, D = s.shape
        st = stream_of(s)
        in_dim = 2 * D if (s_in is not None or tied) else D     # inner dimension of the nn.Linear parameters
        Ws, bs, as_ = params[0::3], params[1::3], params[2::3]
        uvc = torch.empty(3, gate_ld(D), dtype=torch.float32, device=s.device)
        call("mrg_gate_collapse3", (ptr_array(Ws), ptr_array(bs), ptr_array(as_), ptr(uvc), D, in_dim, int(tied), st),
             nbytes=4 * D * in_dim * sum(W is not None for W in Ws))
        out = torch.empty_like(s)
        nb = 4 * D * M * (3 if s_in is not None else 2) + (4 * b1 if norm is not None else 0)
        call("mrg_gate_fwd", (ptr(s), ptr(s_in), ptr(norm), ptr(uvc), ptr(out), b0, b1, M, D, scale, st), nbytes=nb)
        ctx.save_for_backward(s, s_in, norm, uvc, *params)
        ctx.cfg = (b0, b1, scale, in_dim, tied)
        return out

    @staticmethod
    def backward(ctx, g):
        s, s_in, norm, uvc, *params = ctx.saved_tensors
        b0, b1, scale, in_dim, tied = ctx.cfg
        g = f32c(g)
        M, D = s.shape
        st = stream_of(s)
        gs = torch.empty_like(s)
        gs_in = torch.empty_like(s) if s_in is not None else None
        d_uvc = torch.empty(3, gate_ld(D), dtype=torch.float32, device=s.device)
        ws = _ws(_ws_bytes("mrg_gate_bwd_workspace_bytes", M, D), s)
        nb = 4 * D * M * (5 if s_in is not None else 3) + (4 * b1 if norm is not None else 0)
        call("mrg_gate_bwd", (ptr(g), ptr(s), ptr(s_in), ptr(norm), ptr(uvc), ptr(gs), ptr(gs_in), ptr(d_uvc), ptr(ws),
                              b0, b1, M, D, scale, st), nbytes=nb)
        Ws, bs, as_ = params[0::3], params[1::3], params[2::3]
        gWs = [None if W is None else torch.empty_like(W) for W in Ws]
        gbs = [None if b is None else torch.empty_like(b) for b in bs]
        gas = [None if a is None else torch.empty_like(a) for a in as_]
        call("mrg_gate_param_grad3", (ptr_array(Ws), ptr_array(bs), ptr_array(as_), ptr(d_uvc), ptr_array(gWs), ptr_array(gbs), ptr_array(gas),
                                      D, in_dim, int(tied), st), nbytes=8 * D * in_dim * sum(W is not None for W in Ws))
        gparams = []
        for gW, gb, ga in zip(gWs, gbs, gas):
            gparams += [gW, gb, ga]
        return (gs, gs_in, None, None, None, None, *gparams)


def gate_comp(s, s_in, norm, b0, b1, W_in, b_in, a_in, W_out, b_out, a_out, W_self, b_self, a_self):
    """f_sparse_comp (reference models/operations_lp.py:317-343)."""
    return _Gate.apply(s, s_in, norm, int(b0), int(b1), 1.0 / 3.0, W_in, b_in, a_in, W_out, b_out, a_out, W_self, b_self, a_self)


def gate_last(s, W, b, a):
    """f_sparse_last (reference models/operations_lp.py:412-416): all rows in segment 2."""
    return _Gate.apply(s, None, None, 0, 0, 1.0, None, None, None, None, None, None, W, b, a)


class _GateRow(torch.autograd.Function):
    """f_sparse_comp as a ROW FACTOR: returns fvec [M] with  f_sparse_comp(s, s_in) == s * fvec[:, None]  bit for bit (the gate is
    one scalar per row: reference models/operations_lp.py:317-343).  Its consumer -- the MixedOp epilogue -- recomputes the
    candidate from s in every pass instead of reading a stored [M, D] tensor, and in backward returns the true gradient
    w.r.t. fvec (the row dots sum_c gy * s).  This node turns it into the parameter and s_in gradients; the part of the gradient
    w.r.t. s that goes through the gate (dz_r * u) is added by the epilogue's gradient store when it says so
    (Link.row_written), and formed here otherwise.  Arguments: the Link (or None), then as _Gate."""

    @staticmethod
    def forward(ctx, link, s, s_in, norm, b0, b1, scale, *params):
        tied = s_in is not None and same_rows(s, s_in)
        s, s_in, norm = f32c(s), (None if tied else f32c(s_in)), f32c(norm)
        params = tuple(f32c(p) for p in params)
        require_hip(s, s_in, norm, *params)
        M, D = s.shape
        st = stream_of(s)
        in_dim = 2 * D if (s_in is not None or tied) else D
        Ws, bs, as_ = params[0::3], params[1::3], params[2::3]
        uvc = torch.empty(3, gate_ld(D), dtype=torch.float32, device=s.device)
        call("mrg_gate_collapse3", (ptr_array(Ws), ptr_array(bs), ptr_array(as_), ptr(uvc), D, in_dim, int(tied), st),
             nbytes=4 * D * in_dim * sum(W is not None for W in Ws))
        fvec = torch.empty(M, dtype=torch.float32, device=s.device)
        hvec = torch.empty(M, dtype=torch.float32, device=s.device)
        nb = 4 * D * M * (2 if s_in is not None else 1) + (4 * b1 if norm is not None else 0) + 8 * M
        call("mrg_gate_row_fwd", (ptr(s), ptr(s_in), ptr(norm), ptr(uvc), ptr(fvec), ptr(hvec), b0, b1, M, D, scale, st), nbytes=nb)
        ctx.save_for_backward(s, s_in, norm, uvc, hvec, *params)
        ctx.cfg = (b0, b1, scale, in_dim, tied)
        ctx.link = link
        if link is not None:
            link.row_h, link.row_uvc = hvec, uvc
        return fvec

    @staticmethod
    def backward(ctx, gq):
        s, s_in, norm, uvc, hvec, *params = ctx.saved_tensors
        b0, b1, scale, in_dim, tied = ctx.cfg
        gq = f32c(gq)
        M, D = s.shape
        st = stream_of(s)
        gs_in = torch.empty_like(s) if s_in is not None else None
        d_uvc = torch.empty(3, gate_ld(D), dtype=torch.float32, device=s.device)
        ws = _ws(_ws_bytes("mrg_gate_bwd_workspace_bytes", M, D), s)
        nb = 4 * D * M * (3 if s_in is not None else 1) + 8 * M
        call("mrg_gate_row_bwd", (ptr(gq), ptr(hvec), ptr(s), ptr(s_in), ptr(uvc), ptr(gs_in), ptr(d_uvc), ptr(ws), b0, b1, M, D, st), nbytes=nb)
        Ws, bs, as_ = params[0::3], params[1::3], params[2::3]
        gWs = [None if W is None else torch.empty_like(W) for W in Ws]
        gbs = [None if b is None else torch.empty_like(b) for b in bs]
        gas = [None if a is None else torch.empty_like(a) for a in as_]
        call("mrg_gate_param_grad3", (ptr_array(Ws), ptr_array(bs), ptr_array(as_), ptr(d_uvc), ptr_array(gWs), ptr_array(gbs), ptr_array(gas),
                                      D, in_dim, int(tied), st), nbytes=8 * D * in_dim * sum(W is not None for W in Ws))
        gparams = []
        for gW, gb, ga in zip(gWs, gbs, gas):
            gparams += [gW, gb, ga]
        gs = None
        folded = ctx.link.row_written if ctx.link is not None else None
        if ctx.link is not None:
            ctx.link.row_written = None
        if folded is not None and folded != gq.data_ptr():
            # the epilogue has ALREADY added its share (dz_r * u_seg) into the direct term of the gated partner; recomputing it here
            # from another gradient tensor would count that share twice (advisor r3) -- the paired dense-filter node raises as well
            raise _lib.MrgnasError("f_sparse_comp (row factor): the MixedOp epilogue folded the gradient w.r.t. s, but a different gradient "
                                   "tensor reached the factor's node -- the factor has a second consumer, which the folded form does not support")
        if folded is None:
            # nobody added dz_r * u_seg to the gradient w.r.t. s (the factor was multiplied out by plain tensor arithmetic)
            dz = gq * hvec
            gs = torch.empty_like(s)
            for seg, (lo, hi) in enumerate(((0, b0), (b0, b1), (b1, M))):
                if hi > lo:
                    torch.mul(dz[lo:hi, None], uvc[seg, :D][None, :], out=gs[lo:hi])
        return (None, gs, gs_in, None, None, None, None, *gparams)


def gate_comp_row_factor(s, s_in, norm, b0, b1, W_in, b_in, a_in, W_out, b_out, a_out, W_self, b_self, a_self):
    """f_sparse_comp as a row factor for mixed_epilogue: Candidate("rowfactor") around the [M] factor; the candidate is
    s * fvec[:, None] (mixed_epilogue_prepare multiplies it out itself when it cannot recompute it in its kernels)."""
    s = f32c(s)
    link = Link()
    fvec = _GateRow.apply(link, s, s_in, norm, int(b0), int(b1), 1.0 / 3.0, W_in, b_in, a_in, W_out, b_out, a_out, W_self, b_self, a_self)
    return Candidate("rowfactor", fvec, link=link, s=s, b0=int(b0), b1=int(b1))
