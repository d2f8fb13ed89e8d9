"""Dense (per-feature) filters f_dense_comp / f_comp / f_dense_last on the MFMA row GEMM (csrc/dense.hip, linear.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import weakref

import torch

from .. import _lib
from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of
from . import switches as SW
from ._base import Fork, _ws, _ws_bytes, same_rows
from .candidates import Candidate, Link


class _DenseFilter(torch.autograd.Function):
    """Three direction segments [0,b0) [b0,b1) [b1,M), each with its own nn.Linear (W, b);
    params flat: W_in, b_in, W_out, b_out, W_self, b_self (None for an absent segment / bias).
    kind 0: sigmoid(W[s;s_in]+b) * s * c   kind 1: (W[s;s_in]) * c,  c = scale * norm on edge rows."""

    @staticmethod
    def forward(ctx, link, kind, s, s_in, norm, b0, b1, scale_edge, scale_self, *params):
        s, s_in, norm = f32c(s), f32c(s_in), f32c(norm)
        params = tuple(f32c(p) for p in params)
        require_hip(s, s_in, norm, *params)
        M, D = s.shape
        st = stream_of(s)
        out = torch.empty_like(s)
        gate = torch.empty_like(s) if kind == 0 else None
        ctx.link = link
        if link is not None:                            # what an epilogue that folds this node's first backward pass reads
            link.s, link.gate = s, (gate.detach() if gate is not None else None)
        K_ = 2 * D if s_in is not None else D
        ws3 = int(_lib.load().mrg_dense_filter3_workspace_bytes(D, K_)) if (SW.GROUPED_SEGMENTS and all(params[2 * i] is not None for i in range(3))) else 0
        if ws3 > 0:                                     # the three direction segments in one weight-split + one grouped GEMM launch
            call("mrg_dense_filter_fwd3", (kind, ptr(s), ptr(s_in), ptr_array([params[0], params[2], params[4]]),
                                           ptr_array([params[1], params[3], params[5]]), ptr(norm), scale_edge, scale_self, ptr(out), ptr(gate),
                                           ptr(_ws(ws3, s)), b0, b1, M, D, st),
                 nbytes=4 * M * (K_ + D * (2 if kind == 0 else 1)), flops=2 * M * K_ * D)
            ctx.cfg = (kind, b0, b1, scale_edge, scale_self)
            ctx.save_for_backward(s, s_in, norm, gate, *params)
            return out
        segs = [(i, lo, hi, scale, edge) for i, (lo, hi, scale, edge) in
                enumerate(((0, b0, scale_edge, True), (b0, b1, scale_edge, True), (b1, M, scale_self, False))) if hi > lo]
        gws = [_ws(_ws_bytes("mrg_gemm_workspace_bytes", K_, D), s) for _ in segs]
        fork = Fork(s.device, len(segs) if M >= SW.FORK_MIN_ROWS else 1)
        for j, (i, lo, hi, scale, edge) in enumerate(segs):
            W, b = params[2 * i], params[2 * i + 1]
            rs = norm[lo:hi] if (edge and norm is not None) else None
            with fork.on(j):
                call("mrg_dense_filter_fwd", (kind, ptr(s[lo:hi]), ptr(s_in[lo:hi]) if s_in is not None else None, ptr(W), ptr(b),
                                              ptr(rs), scale, ptr(out[lo:hi]), ptr(gate[lo:hi]) if gate is not None else None,
                                              ptr(gws[j]), hi - lo, D, stream_of(s)),
                     nbytes=4 * (hi - lo) * (K_ + D * (2 if kind == 0 else 1)), flops=2 * (hi - lo) * K_ * D)
        fork.join()
        ctx.cfg = (kind, b0, b1, scale_edge, scale_self)
        ctx.save_for_backward(s, s_in, norm, gate, *params)
        return out

    @staticmethod
    def backward(ctx, g):
        s, s_in, norm, gate, *params = ctx.saved_tensors
        kind, b0, b1, scale_edge, scale_self = ctx.cfg
        g = f32c(g)
        M, D = s.shape
        st = stream_of(s)
        # a MixedOp epilogue wrote dz (and, kind 0, the direct term of gs) already?  (raises when y had a second consumer whose
        # unscaled gradient autograd summed into the epilogue's pre-scaled one)
        prescaled = ctx.link is not None and ctx.link.arrived(0, g, "dense filter")
        if prescaled and kind == 0:
            gs = ctx.link.gs_direct
            gs.record_stream(torch.cuda.current_stream())   # allocated by the epilogue's backward on ITS stream
        else:
            gs = torch.empty_like(s)
        gs_in = torch.empty_like(s) if s_in is not None else None
        K_ = 2 * D if s_in is not None else D
        lib = _lib.load()
        if SW.GROUPED_SEGMENTS and all(params[2 * i] is not None for i in range(3)):
            wsi, wsw = int(lib.mrg_linear_bwd_input3_workspace_bytes(D, D)), int(lib.mrg_linear_bwd_weight3_workspace_bytes(b0, b1, M, D, K_ - D, D))
            if wsi > 0 and wsw > 0:                     # the three direction segments in one launch per stage: 7 - 9 launches instead of 21 - 27
                Ws = [params[0], params[2], params[4]]
                gWs = [torch.empty_like(W) for W in Ws]
                gbs = [torch.empty_like(params[2 * i + 1]) if params[2 * i + 1] is not None else None for i in range(3)]
                if prescaled:
                    dz = g
                else:
                    dz = torch.empty(M, D, dtype=torch.float32, device=s.device)
                    call("mrg_dense_filter_dz3", (kind, ptr(g), ptr(s), ptr(gate), ptr(norm), scale_edge, scale_self, ptr(dz), ptr(gs), b1, M, D, st),
                         nbytes=4 * M * D * (5 if kind == 0 else 2))
                gwork = dict(nbytes=4 * M * 2 * D + 12 * D * D, flops=2 * M * D * D)
                call("mrg_linear_bwd_input3", (ptr(dz), ptr_array(Ws), ptr(gs), ptr(_ws(wsi, s)), b0, b1, M, D, D, K_, int(kind == 0), st), **gwork)
                if s_in is not None:
                    call("mrg_linear_bwd_input3", (ptr(dz), ptr_array([W[:, D:] for W in Ws]), ptr(gs_in), ptr(_ws(wsi, s)), b0, b1, M, D, D, K_, 0, st),
                         **gwork)
                call("mrg_linear_bwd_weight3", (ptr(dz), ptr(s), ptr(s_in), ptr_array(gWs), ptr_array(gbs), ptr(_ws(wsw, s)), b0, b1, M, D, K_ - D, D, st),
                     nbytes=4 * M * (D + K_), flops=2 * M * K_ * D)
                grads = [t for pair in zip(gWs, gbs) for t in pair]
                return (None, None, gs, gs_in, None, None, None, None, None, *grads)
        grads = []
        segs = ((0, b0, scale_edge, True), (b0, b1, scale_edge, True), (b1, M, scale_self, False))
        work = []
        for i, (lo, hi, scale, edge) in enumerate(segs):               # allocate everything on the main stream first
            W, b = params[2 * i], params[2 * i + 1]
            if W is None:
                grads += [None, None]
                continue
            rows = max(hi - lo, 0)
            gW = torch.empty_like(W)
            gb = torch.empty_like(b) if b is not None else None
            grads += [gW, gb]
            work.append(dict(W=W, gW=gW, gb=gb, rows=rows, sl=slice(lo, hi), scale=scale,
                             rs=norm[lo:hi] if (edge and norm is not None and rows > 0) else None,
                             dz=torch.empty(rows, D, dtype=torch.float32, device=s.device),
                             wt=_ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", D, D), s),
                             wt2=_ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", D, D), s) if s_in is not None else None,
                             ws=_ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", rows, K_, D), s)))
        fork = Fork(s.device, len(work) if M >= SW.FORK_MIN_ROWS else 1)
        # (lab path: one launch per direction segment.  The grouped launch sizes its row blocks for the ranges together: say so, to get
        #  the same partial sums bit for bit -- mrg_wgrad_set_share)
        live_ranges = sum(1 for w in work if w["rows"] > 0)
        lib.mrg_wgrad_set_share(max(1, min(3, live_ranges)))
        for j, w in enumerate(work):
            W, rows, sl = w["W"], w["rows"], w["sl"]
            with fork.on(j):
                st = stream_of(s)
                # 1. dz (+ direct term of gs for the gated kinds); f_comp behind a MixedOp epilogue: the gradient arrives scaled
                if prescaled:
                    w["dz"] = g[sl]
                else:
                    call("mrg_dense_filter_dz", (kind, ptr(g[sl]), ptr(s[sl]), ptr(gate[sl]) if gate is not None else None, ptr(w["rs"]),
                                                 w["scale"], ptr(w["dz"]), ptr(gs[sl]), rows, D, st), nbytes=4 * rows * D * (5 if kind == 0 else 2))
                # 2. gs (+)= dz W[:, :D];  gs_in = dz W[:, D:]
                gwork = dict(nbytes=4 * rows * 2 * D + 4 * D * D, flops=2 * rows * D * D)
                call("mrg_linear_bwd_input", (ptr(w["dz"]), ptr(W), ptr(gs[sl]), ptr(w["wt"]), rows, D, D, K_, int(kind == 0), st), **gwork)
                if s_in is not None:
                    call("mrg_linear_bwd_input", (ptr(w["dz"]), ptr(W[:, D:]), ptr(gs_in[sl]), ptr(w["wt2"]), rows, D, D, K_, 0, st), **gwork)
                # 3. gW = dz^T [s | s_in], gb = column sums of dz
                call("mrg_linear_bwd_weight", (ptr(w["dz"]), ptr(s[sl]), ptr(s_in[sl]) if s_in is not None else None, ptr(w["gW"]),
                                               ptr(w["gb"]), ptr(w["ws"]), rows, D, D if s_in is not None else 0, D, st),
                     nbytes=4 * rows * (D + K_), flops=2 * rows * K_ * D)
        fork.join()
        lib.mrg_wgrad_set_share(1)
        return (None, None, gs, gs_in, None, None, None, None, None, *grads)


class _FoldHalves(torch.autograd.Function):
    """Wt_i = W_i[:, :D] + W_i[:, D:] for up to three nn.Linear(2D, D) weights (None passes through): the weight an
    operator sees when both of its operands are the same rows.  Backward: gW_i = [gWt_i | gWt_i].  One launch each way
    (mrg_fold_halves3 / mrg_unfold_halves3)."""

    @staticmethod
    def forward(ctx, *Ws):
        Ws = tuple(f32c(W) for W in Ws)
        Ws3 = (Ws + (None, None, None))[:3]
        ref = next(W for W in Ws3 if W is not None)
        require_hip(*Ws3)
        D = ref.shape[0]
        buf = torch.empty(3, D, D, dtype=torch.float32, device=ref.device)
        call("mrg_fold_halves3", (ptr_array(Ws3), ptr(buf), D, stream_of(ref)), nbytes=12 * D * D * sum(W is not None for W in Ws3))
        ctx.meta = (D, [W is not None for W in Ws3], len(Ws), ref.device)
        return tuple(buf[i] if Ws3[i] is not None else None for i in range(len(Ws)))

    @staticmethod
    def backward(ctx, *gs):
        D, present, n, dev = ctx.meta
        src = [f32c(gs[i]) if (i < n and present[i] and gs[i] is not None) else None for i in range(3)]
        gWs = [torch.empty(D, 2 * D, dtype=torch.float32, device=dev) if (i < n and present[i]) else None for i in range(3)]
        ref = next(t for t in gWs if t is not None)
        call("mrg_unfold_halves3", (ptr_array(src), ptr_array(gWs), D, stream_of(ref)), nbytes=12 * D * D * sum(present))
        return tuple(gWs[:n])


def dense_filter_comp(kind, s, s_in, norm, b0, b1, W_in, b_in, W_out, b_out, W_self, b_self, self_scale, for_epilogue=False):
    """f_dense_comp (kind 0, self_scale 1/3) / f_comp (kind 1, self_scale 1).  When `s` and `s_in` are the same rows
    the three GEMMs run on folded [D, D] weights (half the flops forward, one input-gradient GEMM and a
    half-width weight-gradient GEMM backward).  for_epilogue: the result goes to mixed_epilogue_prepare and nowhere else -- a
    Candidate whose Link lets that epilogue's gradient store perform this node's first backward pass."""
    if s_in is not None and same_rows(s, s_in):
        W_in, W_out, W_self = _FoldHalves.apply(W_in, W_out, W_self)
        s_in = None
    norm = f32c(norm)        # ONE float32 contiguous [>= b1] vector for the forward, its backward and a folded epilogue gradient
    if norm is not None and norm.numel() < int(b1):
        raise _lib.MrgnasError(f"dense filter: edge norm has {norm.numel()} entries, the edge rows need {int(b1)}")
    fold = for_epilogue and SW.FOLD_ROW_SCALE and f32c(s).is_cuda and torch.is_grad_enabled()
    link = Link() if fold else None
    y = _DenseFilter.apply(link, kind, s, s_in, norm, int(b0), int(b1), 1.0 / 3.0, float(self_scale),
                           W_in, b_in, W_out, b_out, W_self, b_self)
    if fold and y.requires_grad:
        # the backward begins with an elementwise pass over the incoming gradient (f_comp: dz = g * c; f_dense_comp: dz = g c s gate
        # (1 - gate) and the direct term g c gate; c = norm / 3 on edge rows, self_scale on self rows): a MixedOp epilogue that is
        # the only reader of y writes its gradient in that form (Candidate.rowscale + the Link it claims)
        return Candidate("stored", y, link=link, slot=0, rowscale=(norm, int(b1), 1.0 / 3.0, float(self_scale), kind == 0))
    return y


class _DensePair(torch.autograd.Function):
    """f_dense_comp and f_comp of ONE MixedOp as one autograd node (reference models/cell_lp.py:95-113: every first-stage MixedOp
    applies both to the same (h, h_in); models/operations_lp.py:356-390, 266-288).  Forward: the two grouped row GEMMs of
    _DenseFilter (gate epilogue / scale epilogue).  Backward: ONE input-gradient product over the concatenated reduction
    dimension per operand, gs = direct term + [dz_d | dz_c] [W_d[:, :D] ; W_c[:, :D]] (mrg_linear_bwd_input3_pair) -- one
    gradient w.r.t. the shared operand instead of two that the state's fan-in pass would add -- and the two weight gradients.
    params: W_in, b_in, W_out, b_out, W_self, b_self of f_dense_comp, then W_in, W_out, W_self of f_comp (no biases)."""

    @staticmethod
    def forward(ctx, link, s, s_in, norm, b0, b1, gate_only, *params):
        s, s_in, norm = f32c(s), f32c(s_in), f32c(norm)
        params = tuple(f32c(p) for p in params)
        require_hip(s, s_in, norm, *params)
        M, D = s.shape
        st = stream_of(s)
        K_ = 2 * D if s_in is not None else D
        ws3 = int(_lib.load().mrg_dense_filter3_workspace_bytes(D, K_))
        out_c, gate = torch.empty_like(s), torch.empty_like(s)
        # gate_only: f_dense_comp's output is never stored -- the node returns the GATE and its consumer (the MixedOp epilogue)
        # recomputes gate * s * c in every pass that reads the candidate
        out_d = gate if gate_only else torch.empty_like(s)
        dW, dB, cW = [params[0], params[2], params[4]], [params[1], params[3], params[5]], list(params[6:9])
        work = dict(flops=2 * M * K_ * D)
        call("mrg_dense_filter_fwd3", (0, ptr(s), ptr(s_in), ptr_array(dW), ptr_array(dB), ptr(norm), 1.0 / 3.0, 1.0 / 3.0,
                                       None if gate_only else ptr(out_d), ptr(gate), ptr(_ws(ws3, s)), b0, b1, M, D, st),
             nbytes=4 * M * (K_ + (1 if gate_only else 2) * D), **work)
        call("mrg_dense_filter_fwd3", (1, ptr(s), ptr(s_in), ptr_array(cW), ptr_array([None, None, None]), ptr(norm), 1.0 / 3.0, 1.0, ptr(out_c), None,
                                       ptr(_ws(ws3, s)), b0, b1, M, D, st), nbytes=4 * M * (K_ + D), **work)
        ctx.cfg = (b0, b1)
        ctx.save_for_backward(s, s_in, norm, gate, *params)
        ctx.link = link
        if link is not None:
            # detached: with gate_only the gate IS the node's first output -- a plain reference would close the cycle
            # node -> link -> output -> grad_fn = node and keep every tensor of the step alive until the cyclic collector runs
            link.s, link.gate = s, gate.detach()
        return out_d, out_c

    @staticmethod
    def backward(ctx, g_d, g_c):
        s, s_in, norm, gate, *params = ctx.saved_tensors
        b0, b1 = ctx.cfg
        M, D = s.shape
        st = stream_of(s)
        g_d, g_c = f32c(g_d), f32c(g_c)
        pres = [ctx.link is not None and ctx.link.arrived(i, g, "dense filter pair") for i, g in enumerate((g_d, g_c))]
        K_ = 2 * D if s_in is not None else D
        dW, cW = [params[0], params[2], params[4]], list(params[6:9])
        # 1. dz of both candidates (+ the direct term of f_dense_comp's gs), unless the MixedOp epilogue's gradient store did it
        if pres[0]:
            dz_d, gs = g_d, ctx.link.gs_direct
            gs.record_stream(torch.cuda.current_stream())
        else:
            dz_d, gs = torch.empty_like(s), torch.empty_like(s)
            call("mrg_dense_filter_dz3", (0, ptr(g_d), ptr(s), ptr(gate), ptr(norm), 1.0 / 3.0, 1.0 / 3.0, ptr(dz_d), ptr(gs), b1, M, D, st),
                 nbytes=4 * M * D * 5)
        if pres[1]:
            dz_c = g_c
        else:
            dz_c = torch.empty_like(s)
            call("mrg_dense_filter_dz3", (1, ptr(g_c), ptr(s), None, ptr(norm), 1.0 / 3.0, 1.0, ptr(dz_c), None, b1, M, D, st), nbytes=4 * M * D * 2)
        # 2. ONE product per operand: gs += [dz_d | dz_c] [W_d[:, :D] ; W_c[:, :D]],  gs_in = [dz_d | dz_c] [W_d[:, D:] ; W_c[:, D:]]
        lib = _lib.load()
        wsp = int(lib.mrg_linear_bwd_input3_pair_workspace_bytes(D, D))
        gwork = dict(nbytes=4 * M * 3 * D + 24 * D * D, flops=4 * M * D * D)
        call("mrg_linear_bwd_input3_pair", (ptr(dz_d), ptr(dz_c), ptr_array(dW), ptr_array(cW), ptr(gs), ptr(_ws(wsp, s)), b0, b1, M, D, D, K_, 1, st),
             nbytes=4 * M * 4 * D + 24 * D * D, flops=4 * M * D * D)
        gs_in = None
        if s_in is not None:
            gs_in = torch.empty_like(s)
            call("mrg_linear_bwd_input3_pair", (ptr(dz_d), ptr(dz_c), ptr_array([W[:, D:] for W in dW]), ptr_array([W[:, D:] for W in cW]), ptr(gs_in),
                                                ptr(_ws(wsp, s)), b0, b1, M, D, D, K_, 0, st), **gwork)
        # 3. the weight gradients of the two candidates
        wsw = int(lib.mrg_linear_bwd_weight3_workspace_bytes(b0, b1, M, D, K_ - D, D))
        g_dW = [torch.empty_like(W) for W in dW]
        g_dB = [torch.empty_like(params[2 * i + 1]) for i in range(3)]
        g_cW = [torch.empty_like(W) for W in cW]
        wwork = dict(nbytes=4 * M * (D + K_), flops=2 * M * K_ * D)
        call("mrg_linear_bwd_weight3", (ptr(dz_d), ptr(s), ptr(s_in), ptr_array(g_dW), ptr_array(g_dB), ptr(_ws(wsw, s)), b0, b1, M, D, K_ - D, D, st), **wwork)
        call("mrg_linear_bwd_weight3", (ptr(dz_c), ptr(s), ptr(s_in), ptr_array(g_cW), ptr_array([None, None, None]), ptr(_ws(wsw, s)), b0, b1, M, D, K_ - D, D, st),
             **wwork)
        grads_d = [t for pair in zip(g_dW, g_dB) for t in pair]
        return (None, gs, gs_in, None, None, None, None, *grads_d, *g_cW)


def dense_pair_available(D, tied):
    """May f_dense_comp + f_comp run as one node (split core, grouped direction segments, every stage's workspace query answers)?"""
    if not (SW.DENSE_PAIR and SW.GROUPED_SEGMENTS):
        return False
    lib = _lib.load()
    K_ = D if tied else 2 * D
    return (int(lib.mrg_dense_filter3_workspace_bytes(D, K_)) > 0 and int(lib.mrg_linear_bwd_input3_pair_workspace_bytes(D, D)) > 0
            and int(lib.mrg_linear_bwd_weight3_workspace_bytes(1, 2, 3, D, K_ - D, D)) > 0)
_GATED_C = {}


def _gated_rowscale(norm, b1, M, scale_edge, scale_self, device):
    """The gated filter's per-row multiplier for all M rows: scale_edge * norm[r] on the b1 edge rows, scale_self on the self rows
    (float32 products, as the row GEMM's gate epilogue forms them).  Built once per edge-norm vector (a graph's norm_flat() is one
    cached tensor) and kept while that tensor lives."""
    base = None if norm is None else (norm._base if norm._base is not None else norm)      # norm_flat() hands out a fresh view per call
    key = (None if norm is None else (norm.data_ptr(), norm._version), b1, M, scale_edge, scale_self, str(device))
    hit = _GATED_C.get(key)
    if hit is not None and (norm is None or hit[0]() is base):
        return hit[1]
    c = torch.empty(M, dtype=torch.float32, device=device)
    if norm is None:
        c[:b1] = scale_edge
    else:
        torch.mul(norm[:b1], scale_edge, out=c[:b1])
    c[b1:] = scale_self
    if len(_GATED_C) > 64:
        _GATED_C.clear()
    import weakref
    _GATED_C[key] = (weakref.ref(base) if base is not None else None, c)
    return c


def dense_filter_pair(s, s_in, norm, b0, b1, dense_params, comp_weights, gate_only=False, for_epilogue=False):
    """(f_dense_comp(s, s_in), f_comp(s, s_in)) as one autograd node; dense_params = (W_in, b_in, W_out, b_out, W_self, b_self),
    comp_weights = (W_in, W_out, W_self).  Operands that are the same rows use the folded [D, D] weights.
    for_epilogue: both results go to mixed_epilogue_prepare and nowhere else: two Candidates sharing the node's Link.
    gate_only (with for_epilogue): the first is Candidate("gate") around f_dense_comp's GATE -- the epilogue recomputes the
    candidate's value gate * s * c wherever it reads it (the [rows, D] output is never written or re-read)."""
    dW, dB = list(dense_params[0::2]), list(dense_params[1::2])
    cW = list(comp_weights)
    if s_in is not None and same_rows(s, s_in):
        dW = list(_FoldHalves.apply(*dW))
        cW = list(_FoldHalves.apply(*cW))
        s_in = None
    norm = f32c(norm)
    if norm is not None and norm.numel() < int(b1):
        raise _lib.MrgnasError(f"dense filter: edge norm has {norm.numel()} entries, the edge rows need {int(b1)}")
    s = f32c(s)
    gate_only = bool(gate_only and for_epilogue and s.is_cuda)
    fold = for_epilogue and SW.FOLD_ROW_SCALE and s.is_cuda and torch.is_grad_enabled()
    link = Link(2) if fold else None
    y_d, y_c = _DensePair.apply(link, s, s_in, norm, int(b0), int(b1), gate_only, dW[0], dB[0], dW[1], dB[1], dW[2], dB[2], *cW)
    if not for_epilogue:
        return y_d, y_c
    fold = fold and y_d.requires_grad
    rs_d = (norm, int(b1), 1.0 / 3.0, 1.0 / 3.0, True) if fold else None
    rs_c = (norm, int(b1), 1.0 / 3.0, 1.0, False) if fold else None
    if gate_only:
        c_d = Candidate("gate", y_d, link=link, slot=0, s=s, c=_gated_rowscale(norm, int(b1), s.shape[0], 1.0 / 3.0, 1.0 / 3.0, s.device), rowscale=rs_d)
    else:
        c_d = Candidate("stored", y_d, link=link, slot=0, rowscale=rs_d)
    return c_d, Candidate("stored", y_c, link=link, slot=1, rowscale=rs_c)


def dense_filter_single(s, s_in, W, b):
    """f_dense_last (s_in None) / f_dense: sigmoid(W [s ; s_in] + b) * s on all rows."""
    if s_in is not None and same_rows(s, s_in):
        (W,) = _FoldHalves.apply(W)
        s_in = None
    return _DenseFilter.apply(None, 0, s, s_in, None, 0, 0, 1.0, 1.0, None, None, None, None, W, b)
