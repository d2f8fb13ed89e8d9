"""autograd Functions over the C ABI (include/mrgnas.h).

Each Function enqueues HIP kernels of libmrgnas_hip.so on torch's current
stream through ctypes; tensors are only used for memory and stream plumbing.
Inputs are never modified; outputs are fresh tensors.  Every call site states
the algorithmic bytes / flops of the launch so bench.py can
price the kernels against the roofline.
"""
import contextlib
import os
import weakref

import torch

from . import _lib
from ._lib import ptr_array, call, f32c, ptr, require_hip, stream_of

COMPOSE = {"mult": 0, "sub": 1, "add": 2}
REDUCE = {"sum": 0, "mean": 1, "max": 2}
ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2}


def gate_ld(D):
    return 2 * D + 4          # MRG_GATE_LD


def same_rows(a, b):
    """True when the two operands of an operator are the SAME rows (one tensor, or two aliases of one storage --
    e.g. Fan views of the cell's h_in, which the first two MixedOps of a cell receive as both `src_emb` and
    `src_emb_in`, reference models/cell_lp.py:95-104).  Then W [s ; s_in] = (W[:, :D] + W[:, D:]) s: the
    operators fold the weight halves and run at half the inner dimension; the whole input gradient is returned
    through the first operand."""
    if a is None or b is None:
        return False
    if a is b:
        return True
    # aliases fold only when the whole input gradient may leave through the first operand: both need it or neither does
    # (a detached alias next to a tracked one keeps the untied path, which routes each gradient to its own operand);
    # empty tensors all share data_ptr 0 and are never aliases of each other
    return (a.numel() > 0 and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
            and a.dtype == b.dtype and a.requires_grad == b.requires_grad)


def _same_memory(a, b):
    """Do two tensors denote the same [rows, D] block of device memory (e.g. two Fan aliases of one state)?"""
    return a is not None and b is not None and a.numel() > 0 and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()


def _cnt(plan, name):
    """Launch-side count of a plan: the host-known capacity when the plan has one (HIP-built plans pad with -1 and the
    kernels skip the padding: no device-to-host read), else the exact number."""
    cap = "cap_" + name
    if cap in plan and not (hasattr(plan, "try_resolve") and plan.try_resolve()):
        return plan[cap]                         # exact sizes still on their way to the host: launch over the padded capacity
    return plan["n_" + name]


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


_WS_BYTES = {}


def _ws_bytes(name, *args):
    """Workspace size queries are pure functions of the shape: asked once per (entry point, shape)."""
    key = (name, args)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = getattr(_lib.load(), name)(*args)
    return n


# ---------------------------------------------------------------------------
# a1: compose
# ---------------------------------------------------------------------------
class _Compose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, s, hr):
        s, hr = f32c(s), f32c(hr)
        ctx.hr_shape = None
        if hr.shape != s.shape:                 # the reference's `src_emb - hr` broadcasts (e.g. hr [1, D])
            ctx.hr_shape = tuple(hr.shape)
            hr = hr.expand_as(s).contiguous()
        require_hip(s, hr)
        out = torch.empty_like(s)
        rows, D = s.shape
        call("mrg_compose_fwd", (op, ptr(s), ptr(hr), ptr(out), rows, D, stream_of(s)), nbytes=12 * D * rows)
        ctx.op = op
        ctx.save_for_backward(*((s, hr) if op == 0 else ()))
        return out

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        s, hr = ctx.saved_tensors if ctx.op == 0 else (None, None)
        need_s, need_hr = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gs = torch.empty_like(g) if need_s else None
        ghr = torch.empty_like(g) if need_hr else None
        rows, D = g.shape
        nb = 4 * D * rows * (1 + (2 if ctx.op == 0 else 1) * (int(need_s) + int(need_hr)))
        call("mrg_compose_bwd", (ctx.op, ptr(g), ptr(s), ptr(hr), ptr(gs), ptr(ghr), rows, D, stream_of(g)), nbytes=nb)
        if ghr is not None and ctx.hr_shape is not None:
            ghr = ghr.sum_to_size(ctx.hr_shape)
        return None, gs, ghr


def compose(kind, s, hr):
    """s (*|-|+) hr on [rows, D] (reference models/operations_lp.py:71-98).  When both operands are LazyRows -- rows of the
    entity / relation tables that nobody has materialised (the gather G feeding the cell's first stage, reference
    models/model_search_lp.py:135-145) -- gather and compose run as ONE kernel forward and as two balanced segmented sums
    per operand backward: no [M, D] gather output, no [M, D] operand gradients."""
    if isinstance(s, LazyRows) and isinstance(hr, LazyRows):
        return _GatherCompose.apply(COMPOSE[kind], s.table, hr.table, s.gp, hr.gp)
    if isinstance(s, LazyRows):
        s = s.materialize()
    if isinstance(hr, LazyRows):
        hr = hr.materialize()
    return _Compose.apply(COMPOSE[kind], s, hr)


def gather_rows(table, idx32, rel_table=None, rel_idx32=None, kind=None):
    """out[i] = table[idx[i]] (kind None) or table[idx[i]] (op) rel_table[rel_idx[i]].
    Forward-only helper (bit-exact gather, reference models/model_lp.py:131)."""
    table = f32c(table)
    require_hip(table, idx32, rel_table, rel_idx32)
    rows, D = idx32.numel(), table.shape[1]
    out = torch.empty(rows, D, dtype=torch.float32, device=table.device)
    op = -1 if kind is None else COMPOSE[kind]
    nb = rows * (4 * D * (2 if kind is None else 3) + (4 if kind is None else 8))
    call("mrg_gather_compose_fwd", (op, ptr(table), ptr(rel_table), ptr(idx32), ptr(rel_idx32), ptr(out), rows, D,
                                    stream_of(table)), nbytes=nb)
    return out


# ---------------------------------------------------------------------------
# a2 / a3: collapsed scalar gates
# ---------------------------------------------------------------------------
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


ROW_FACTOR = os.environ.get("MRG_ROW_FACTOR", "1") == "1"     # lab switch: 0 = f_sparse_comp's output is stored for the epilogue


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


# ---------------------------------------------------------------------------
# a4 / a5 / a6: destination-segmented reducers
# ---------------------------------------------------------------------------
def _seg_fwd(mode, msg, self_rows, p, N, D, want_arg=True):
    """Launch mrg_seg_reduce_fwd over plan p (graph.dst_csr_plan); returns (out, arg)."""
    out = torch.empty(N, D, dtype=torch.float32, device=msg.device)
    arg = torch.empty(N, D, dtype=torch.int32, device=msg.device) if mode == 2 else None
    n_chunks, n_hubs, n_slots = _cnt(p, "chunks"), _cnt(p, "hubs"), _cnt(p, "slots")
    ws = _ws(_ws_bytes("mrg_seg_reduce_workspace_bytes", n_slots, D), msg) if n_slots > 0 else None
    E = int(p["eid"].numel())
    nb = 4 * D * E + 4 * E + 4 * D * N * (1 + (self_rows is not None) + (mode == 2))
    call("mrg_seg_reduce_fwd", (mode, ptr(msg), ptr(self_rows), ptr(p["eid"]), ptr(p["chunk_node"]), ptr(p["chunk_start"]),
                                ptr(p["chunk_end"]), ptr(p["chunk_slot"]), n_chunks, ptr(p["hub_node"]),
                                ptr(p["hub_first"]), ptr(p["hub_count"]), n_hubs, n_slots, ptr(p["in_degree"]),
                                ptr(out), ptr(arg), ptr(ws), N, D, stream_of(msg)), nbytes=nb)
    return out, arg


def _seg_bwd(mode, g, graph, arg, gmsg, gself, relu_src=None):
    p = graph.plan()
    E, N, D = graph.num_edges(), graph.number_of_nodes(), g.shape[1]
    nb = 4 * D * E * (1 + (relu_src is not None)) + 4 * E + 4 * D * N * (1 + (gself is not None) + (mode == 2))
    call("mrg_seg_reduce_bwd", (mode, ptr(g), ptr(graph.i32("dst")), ptr(p["in_degree"]), ptr(arg), ptr(gmsg), ptr(gself),
                                ptr(relu_src), E, N, D, stream_of(g)), nbytes=nb)


class _SegReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mode, msg, self_rows, graph):
        msg, self_rows = f32c(msg), f32c(self_rows)
        require_hip(msg, self_rows)
        if msg.shape[0] != graph.num_edges():
            raise _lib.MrgnasError(f"message rows {msg.shape[0]} != number of edges {graph.num_edges()}")
        if mode != 2 and hasattr(graph, "agg_plan"):          # sum / mean: balanced span kernel
            sp, meta = graph.agg_plan("mean" if mode == 1 else "sum")
            out, arg = span_gcs("copy", msg, None, meta, sp), None
            if self_rows is not None:
                out += self_rows
        else:
            out, arg = _seg_fwd(mode, msg, self_rows, graph.plan(), graph.number_of_nodes(), msg.shape[1])
        ctx.mode, ctx.graph, ctx.has_self = mode, graph, self_rows is not None
        ctx.save_for_backward(*((arg,) if arg is not None else ()))
        return out

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        arg = ctx.saved_tensors[0] if ctx.mode == 2 else None
        gmsg = torch.empty(ctx.graph.num_edges(), g.shape[1], dtype=torch.float32, device=g.device)
        _seg_bwd(ctx.mode, g, ctx.graph, arg, gmsg, None)
        return None, gmsg, (g if ctx.has_self else None), None


def seg_reduce(kind, msg, self_rows, graph):
    """out[v] = reduce over in-edges of msg (+ self_rows[v]); DGL update_all(copy_e, max|sum|mean)."""
    return _SegReduce.apply(REDUCE[kind], msg, self_rows, graph)


class _AggRows(torch.autograd.Function):
    """a_sum on the reference's [M, D] layout (reference models/operations_lp.py:260-264): rows [0, E) are
    messages, rows [E, M) the residual self rows; out = Dropout(h) + x[E:].  The dropout keep-mask
    (already scaled by 1/(1-p), [N, D]) is passed in, so the whole operator is one autograd node whose
    backward writes the [M, D] gradient once."""

    @staticmethod
    def forward(ctx, mode, x, graph, add_self, keep):
        x = f32c(x)
        require_hip(x, keep)
        E, N, D = graph.num_edges(), graph.number_of_nodes(), x.shape[1]
        if x.shape[0] != E + N:
            raise _lib.MrgnasError(f"expected {E + N} rows (E + N), got {x.shape[0]}")
        if mode != 2 and hasattr(graph, "agg_plan"):
            sp, meta = graph.agg_plan("mean" if mode == 1 else "sum")
            out, arg = span_gcs("copy", x, None, meta, sp), None      # xi < E: only the edge rows of x are gathered
            if keep is not None:
                out *= keep
            if add_self:
                out += x[E:]
        else:
            out, arg = _seg_fwd(mode, x, x[E:] if (add_self and keep is None) else None, graph.plan(), N, D)
            if keep is not None:
                out *= keep
                if add_self:
                    out += x[E:]
        ctx.mode, ctx.graph, ctx.add_self = mode, graph, add_self
        ctx.save_for_backward(*[t for t in (arg, keep) if t is not None])
        ctx.has = (arg is not None, keep is not None)
        # x is an alias handed out by a Fan: a_sum's gradient w.r.t. it is a gather of the [N, D] node gradient, which the fan-in
        # sum can read itself (mrg_sum_rows_gather) instead of receiving an [M, D] copy
        ctx.fan_node = Fan.node_of(x) if (LAZY_ASUM and mode == 0 and x.is_cuda) else None
        return out

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        graph = ctx.graph
        saved = list(ctx.saved_tensors)
        arg = saved.pop(0) if ctx.has[0] else None
        keep = saved.pop(0) if ctx.has[1] else None
        E, N, D = graph.num_edges(), graph.number_of_nodes(), g.shape[1]
        gh = g * keep if keep is not None else g
        node = ctx.fan_node
        if node is not None and ctx.mode == 0 and node.gathered is None:
            # one gathered term per fan; further ones are materialised.  The hand-over bypasses autograd's input buffer (the edge
            # carries None), which is what orders a gradient produced on a candidate's side stream before its consumer: the
            # event below does that instead (advisor r3: MRG_MIXED_STREAMS >= 2)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(g.device))
            node.gathered = (gh, g if ctx.add_self else None, graph, ev)
            return None, None, None, None, None
        gx = torch.empty(E + N, D, dtype=torch.float32, device=g.device)
        if ctx.add_self:
            gx[E:] = g
        else:
            gx[E:].zero_()
        _seg_bwd(ctx.mode, gh, graph, arg, gx, None)
        return None, gx, None, None, None


LAZY_ASUM = os.environ.get("MRG_LAZY_ASUM", "1") == "1"     # lab switch: 0 = a_sum's [M, D] input gradient is materialised for the fan-in sum


def aggregate_rows(kind, x, graph, add_self=True, keep=None):
    return _AggRows.apply(REDUCE[kind], x, graph, add_self, keep)


FUSED_AMAX = os.environ.get("MRG_FUSED_AMAX", "1") == "1"      # lab switch: 0 = linear + segmented max as separate launches
# below this many edges the step is launch-bound and the fused form's extra launches (key memset, unpack pass, mask product)
# cost more than the [E, D] round trip they save: 30 000-edge sampled step 18.4 vs 17.7 ms
FUSED_AMAX_MIN_ROWS = int(os.environ.get("MRG_FUSED_AMAX_MIN_ROWS", "100000"))
FUSED_AMEAN = os.environ.get("MRG_FUSED_AMEAN", "1") == "1"    # lab switch: 0 = linear, then the span reducer over the [E, D] messages


def _fused_agg_ws(N, D):
    """Workspace of the fused a_max (0: not available).  Asked every time: it also answers 0 while mrg_gemm_set_mode(1) keeps
    every GEMM on the exact-f32 core (the fused epilogues exist on the split core only)."""
    return int(_lib.load().mrg_linear_relu_segmax_workspace_bytes(N, D, D))


class _LinReluAgg(torch.autograd.Function):
    """a_max / a_mean as ONE autograd node on the reference's [M, D] layout
    (reference models/operations_lp.py:230-235, 245-250):
        m = ReLU(Linear(x[:E]));  h = reduce_{e -> v} m[e];  out = h + x[E:]
    The backward writes the gradient of x once (rows [0,E) from the input-gradient GEMM, rows [E,M) a copy of
    the incoming gradient) instead of two zero-padded slice gradients that autograd would add, and the ReLU
    mask is applied inside the reducer's backward kernel.  Whenever the split matrix core takes the shape, a_max runs as ONE
    GEMM whose epilogue is the ReLU and the segmented max (mrg_linear_relu_segmax_fwd) and a_mean as a GEMM whose epilogue
    leaves ordered run sums for the heads reducer (mrg_linear_relu_segsum_fwd): m is never written."""

    @staticmethod
    def forward(ctx, mode, x, W, b, graph):
        x, W, b = f32c(x), f32c(W), f32c(b)
        require_hip(x, W, b)
        E, N, D = graph.num_edges(), graph.number_of_nodes(), x.shape[1]
        if x.shape[0] != E + N:
            raise _lib.MrgnasError(f"expected {E + N} rows (E + N), got {x.shape[0]}")
        st = stream_of(x)
        fused_ws = _fused_agg_ws(N, D) if (mode == 2 and FUSED_AMAX and E >= FUSED_AMAX_MIN_ROWS) else 0
        if fused_ws > 0:
            # a_max as ONE GEMM whose epilogue is ReLU + segmented max (the [E, D] messages are never written; the
            # backward's ReLU mask is "the maximum is positive")
            p = graph.plan()
            out = torch.empty(N, D, dtype=torch.float32, device=x.device)
            arg = torch.empty(N, D, dtype=torch.int32, device=x.device)
            mx = torch.empty(N, D, dtype=torch.float32, device=x.device)
            call("mrg_linear_relu_segmax_fwd", (ptr(x), ptr(W), ptr(b), ptr(p["eid"]), ptr(graph.i32("dst")), ptr(x[E:]), ptr(out), ptr(arg),
                                                ptr(mx), ptr(_ws(fused_ws, x)), E, N, D, D, st),
                 nbytes=4 * E * D + 8 * E + 4 * D * D + 4 * N * D * 4, flops=2 * E * D * D)
            ctx.mode, ctx.graph, ctx.fused = mode, graph, True
            ctx.save_for_backward(x, W, arg, mx)
            if MASK_TAP is not None:                       # test instrumentation: which edge won, and whether the maximum is positive
                MASK_TAP(("a_max", W.data_ptr()), [arg, mx > 0])
            return out
        if (mode == 1 and FUSED_AMEAN and E >= FUSED_AMAX_MIN_ROWS and hasattr(graph, "plan")
                and _fused_agg_ws(N, D) > 0):
            # a_mean without the [E, D] messages: the GEMM's epilogue leaves ordered run sums at the head rows of `part` and one
            # ReLU bit per element; the chunk reducer adds a node's head rows
            p = graph.plan()
            part = torch.empty(E, D, dtype=torch.float32, device=x.device)          # only the head rows are written / read
            bits = torch.empty(E, (D + 31) // 32, dtype=torch.int32, device=x.device)
            gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", D, D), x)
            call("mrg_linear_relu_segsum_fwd", (ptr(x), ptr(W), ptr(b), ptr(p["eid"]), ptr(graph.i32("dst")), ptr(part), ptr(bits), ptr(gws),
                                                E, D, D, st), nbytes=4 * E * D + 8 * E + 4 * D * D, flops=2 * E * D * D)
            out = torch.empty(N, D, dtype=torch.float32, device=x.device)
            n_chunks, n_hubs, n_slots = _cnt(p, "chunks"), _cnt(p, "hubs"), _cnt(p, "slots")
            ws = _ws(_ws_bytes("mrg_seg_reduce_workspace_bytes", n_slots, D), x) if n_slots > 0 else None
            call("mrg_seg_reduce_heads_fwd", (1, ptr(part), ptr(x[E:]), ptr(p["rowptr"]), ptr(p["chunk_node"]), ptr(p["chunk_start"]),
                                              ptr(p["chunk_end"]), ptr(p["chunk_slot"]), n_chunks, ptr(p["hub_node"]), ptr(p["hub_first"]),
                                              ptr(p["hub_count"]), n_hubs, n_slots, ptr(p["in_degree"]), ptr(out), ptr(ws), N, D, st),
                 nbytes=8 * N * D + 4 * E)
            ctx.mode, ctx.graph, ctx.fused = mode, graph, "mean"
            ctx.save_for_backward(x, W, bits)
            if MASK_TAP is not None:                       # test instrumentation: the inner ReLU's decisions, one bit per message element
                MASK_TAP(("a_mean", W.data_ptr()), [bits])
            return out
        y = torch.empty(E, D, dtype=torch.float32, device=x.device)
        gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", D, D), x)
        call("mrg_linear_fwd", (ptr(x), ptr(W), ptr(b), ptr(y), ptr(gws), E, D, D, 1, st),
             nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        if mode == 2:
            out, arg = _seg_fwd(2, y, x[E:], graph.plan(), N, D)
        else:
            sp, meta = graph.agg_plan("mean" if mode == 1 else "sum")
            out, arg = span_gcs("copy", y, None, meta, sp), None
            out += x[E:]
        ctx.mode, ctx.graph, ctx.fused = mode, graph, False
        ctx.save_for_backward(x, W, y, *((arg,) if arg is not None else ()))
        return out

    @staticmethod
    def backward(ctx, g):
        graph, mode = ctx.graph, ctx.mode
        g = f32c(g)
        if ctx.fused == "mean":
            x, W, bits = ctx.saved_tensors
        elif ctx.fused:
            x, W, arg, mx = ctx.saved_tensors
        else:
            x, W, y, *rest = ctx.saved_tensors
            arg = rest[0] if rest else None
        E, N, D = graph.num_edges(), graph.number_of_nodes(), x.shape[1]
        st = stream_of(x)
        gx = torch.empty_like(x)
        gy = torch.empty(E, D, dtype=torch.float32, device=x.device)
        if ctx.fused == "mean":
            call("mrg_seg_reduce_bwd_bits", (1, ptr(g), ptr(graph.i32("dst")), ptr(graph.plan()["in_degree"]), ptr(gy), ptr(gx[E:]), ptr(bits),
                                             E, N, D, st), nbytes=4 * D * E + 4 * E + 8 * D * N)
        elif ctx.fused:
            _seg_bwd(mode, g * (mx > 0), graph, arg, gy, None)             # the winning message is ReLU-dead iff the maximum is 0
            gx[E:] = g
        else:
            _seg_bwd(mode, g, graph, arg, gy, gx[E:], relu_src=y)          # gy masked by ReLU; gx[E:] = g
        work = dict(nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        wt = _ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", D, D), x)
        call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(wt), E, D, D, D, 0, st), **work)
        gW = torch.empty_like(W)
        gb = torch.empty(D, dtype=torch.float32, device=x.device)
        ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", E, D, D), x)
        call("mrg_linear_bwd_weight", (ptr(gy), ptr(x), None, ptr(gW), ptr(gb), ptr(ws), E, D, 0, D, st), **work)
        return None, gx, gW, gb, None


def linear_relu_aggregate(kind, x, W, b, graph):
    return _LinReluAgg.apply(REDUCE[kind], x, W, b, graph)


class _LinReluPartial(torch.autograd.Function):
    """The edge part of a_max / a_mean on ONE relation block of a sharded graph (mr-gnas_amd/dist.py): this
    rank's partial  part[v] = max | sum over its LOCAL in-edges of ReLU(W x_e + b)  for all N nodes; the caller
    all-reduces it, scales (mean) and adds the residual self rows.  x is the block's [E_local + n_own, D]
    tensor (rows [E_local, ...) are not read; their gradient is zero here).  One autograd node: the ReLU mask
    is applied inside the reducer's backward kernel, as in _LinReluAgg."""

    @staticmethod
    def forward(ctx, mode, x, W, b, graph):
        x, W, b = f32c(x), f32c(W), f32c(b)
        require_hip(x, W, b)
        E, N, D = graph.num_edges(), graph.number_of_nodes(), x.shape[1]
        st = stream_of(x)
        fused_ws = _fused_agg_ws(N, D) if (mode == 2 and FUSED_AMAX and E >= FUSED_AMAX_MIN_ROWS) else 0
        if fused_ws > 0:                                # one GEMM with the ReLU + segmented-max epilogue, as in _LinReluAgg
            out = torch.empty(N, D, dtype=torch.float32, device=x.device)
            arg = torch.empty(N, D, dtype=torch.int32, device=x.device)
            mx = torch.empty(N, D, dtype=torch.float32, device=x.device)
            call("mrg_linear_relu_segmax_fwd", (ptr(x), ptr(W), ptr(b), ptr(graph.plan()["eid"]), ptr(graph.i32("dst")), None, ptr(out), ptr(arg),
                                                ptr(mx), ptr(_ws(fused_ws, x)), E, N, D, D, st),
                 nbytes=4 * E * D + 8 * E + 4 * D * D + 4 * N * D * 3, flops=2 * E * D * D)
            ctx.mode, ctx.graph, ctx.fused = mode, graph, True
            ctx.save_for_backward(x, W, arg, mx)
            if MASK_TAP is not None:                       # test instrumentation: which edge won, and whether the maximum is positive
                MASK_TAP(("a_max", W.data_ptr()), [arg, mx > 0])
            return out, x[E:].clone()
        if (mode != 2 and FUSED_AMEAN and E >= FUSED_AMAX_MIN_ROWS and _fused_agg_ws(N, D) > 0):
            # the partial SUM of ReLU(linear) without the [E, D] messages (see _LinReluAgg): run sums in the GEMM epilogue + heads reducer
            p = graph.plan()
            part = torch.empty(E, D, dtype=torch.float32, device=x.device)
            bits = torch.empty(E, (D + 31) // 32, dtype=torch.int32, device=x.device)
            gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", D, D), x)
            call("mrg_linear_relu_segsum_fwd", (ptr(x), ptr(W), ptr(b), ptr(p["eid"]), ptr(graph.i32("dst")), ptr(part), ptr(bits), ptr(gws),
                                                E, D, D, st), nbytes=4 * E * D + 8 * E + 4 * D * D, flops=2 * E * D * D)
            out = torch.empty(N, D, dtype=torch.float32, device=x.device)
            n_chunks, n_hubs, n_slots = _cnt(p, "chunks"), _cnt(p, "hubs"), _cnt(p, "slots")
            ws = _ws(_ws_bytes("mrg_seg_reduce_workspace_bytes", n_slots, D), x) if n_slots > 0 else None
            call("mrg_seg_reduce_heads_fwd", (0, ptr(part), None, ptr(p["rowptr"]), ptr(p["chunk_node"]), ptr(p["chunk_start"]),
                                              ptr(p["chunk_end"]), ptr(p["chunk_slot"]), n_chunks, ptr(p["hub_node"]), ptr(p["hub_first"]),
                                              ptr(p["hub_count"]), n_hubs, n_slots, ptr(p["in_degree"]), ptr(out), ptr(ws), N, D, st),
                 nbytes=8 * N * D + 4 * E)
            ctx.mode, ctx.graph, ctx.fused = mode, graph, "sum"
            ctx.save_for_backward(x, W, bits)
            return out, x[E:].clone()
        y = torch.empty(E, D, dtype=torch.float32, device=x.device)
        gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", D, D), x)
        call("mrg_linear_fwd", (ptr(x), ptr(W), ptr(b), ptr(y), ptr(gws), E, D, D, 1, st),
             nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        if mode == 2:
            out, arg = _seg_fwd(2, y, None, graph.plan(), N, D)
        else:
            sp, meta = graph.agg_plan("sum")
            out, arg = span_gcs("copy", y, None, meta, sp), None
        ctx.mode, ctx.graph, ctx.fused = mode, graph, False
        ctx.save_for_backward(x, W, y, *((arg,) if arg is not None else ()))
        return out, x[E:].clone()                       # the residual self rows leave through the same node

    @staticmethod
    def backward(ctx, g, gself):
        if ctx.fused == "sum":
            x, W, bits = ctx.saved_tensors
        elif ctx.fused:
            x, W, arg, mx = ctx.saved_tensors
        else:
            x, W, y, *rest = ctx.saved_tensors
            arg = rest[0] if rest else None
        graph, mode = ctx.graph, ctx.mode
        E, D = graph.num_edges(), x.shape[1]
        g = f32c(g) if g is not None else torch.zeros(graph.number_of_nodes(), D, dtype=torch.float32, device=x.device)
        st = stream_of(x)
        gx = torch.empty_like(x)
        if gself is not None:
            gx[E:] = gself
        else:
            gx[E:].zero_()
        gy = torch.empty(E, D, dtype=torch.float32, device=x.device)
        if ctx.fused == "sum":
            call("mrg_seg_reduce_bwd_bits", (0, ptr(g), ptr(graph.i32("dst")), ptr(graph.plan()["in_degree"]), ptr(gy), None, ptr(bits),
                                             E, graph.number_of_nodes(), D, st), nbytes=4 * D * E + 4 * E + 4 * D * graph.number_of_nodes())
        elif ctx.fused:
            _seg_bwd(mode, g * (mx > 0), graph, arg, gy, None)          # the winning message is ReLU-dead iff the maximum is 0
        else:
            _seg_bwd(mode, g, graph, arg, gy, None, relu_src=y)        # gy masked by ReLU
        work = dict(nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        wt = _ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", D, D), x)
        call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(wt), E, D, D, D, 0, st), **work)
        gW = torch.empty_like(W)
        gb = torch.empty(D, dtype=torch.float32, device=x.device)
        ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", E, D, D), x)
        call("mrg_linear_bwd_weight", (ptr(gy), ptr(x), None, ptr(gW), ptr(gb), ptr(ws), E, D, 0, D, st), **work)
        return None, gx, gW, gb, None


def linear_relu_partial(kind, x, W, b, graph):
    """kind "max" or "sum" (a_mean: the caller divides the all-reduced sum by the global in-degree).
    Returns (partial [N, D], self rows x[E:] [n_own, D]); both gradients return through one [M, D] write."""
    return _LinReluPartial.apply(REDUCE[kind], x, W, b, graph)


class _SumPartial(torch.autograd.Function):
    """a_sum on one relation block: (partial sums over the LOCAL in-edges for all N nodes, self rows x[E:]).
    One node, so the backward writes the [M, D] gradient once instead of two zero-padded slice gradients."""

    @staticmethod
    def forward(ctx, x, graph):
        x = f32c(x)
        require_hip(x)
        E = graph.num_edges()
        sp, meta = graph.agg_plan("sum")
        ctx.graph = graph
        ctx.shape = x.shape
        return span_gcs("copy", x, None, meta, sp), x[E:].clone()

    @staticmethod
    def backward(ctx, g, gself):
        graph = ctx.graph
        E, D = graph.num_edges(), ctx.shape[1]
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=(g if g is not None else gself).device)
        if gself is not None:
            gx[E:] = gself
        else:
            gx[E:].zero_()
        if g is not None:
            _seg_bwd(0, f32c(g), graph, None, gx, None)          # rows [0, E) of gx: g[dst(e)]
        else:
            gx[:E].zero_()
        return gx, None


def sum_partial(x, graph):
    return _SumPartial.apply(x, graph)


# ---------------------------------------------------------------------------
# dense linear on rows (fp32 MFMA)
# ---------------------------------------------------------------------------
WIDE_BWD_INPUT = os.environ.get("MRG_WIDE_BWD_INPUT", "1") == "1"     # lab switch: 0 = the [B, N] scorer's input gradient on the row GEMM


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        x, W, b = f32c(x), f32c(W), f32c(b)
        require_hip(x, W, b)
        rows, K = x.shape
        Nout = W.shape[0]
        if W.shape[1] != K:
            raise _lib.MrgnasError(f"linear: weight {tuple(W.shape)} does not match input width {K}")
        y = torch.empty(rows, Nout, dtype=torch.float32, device=x.device)
        gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", K, Nout), x)
        call("mrg_linear_fwd", (ptr(x), ptr(W), ptr(b), ptr(y), ptr(gws), rows, K, Nout, act, stream_of(x)),
             nbytes=4 * rows * (K + Nout) + 4 * K * Nout, flops=2 * rows * K * Nout)
        ctx.act, ctx.has_b = act, b is not None
        ctx.save_for_backward(x, W, y if act != 0 else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        g = f32c(g)
        rows, K = x.shape
        Nout = W.shape[0]
        st = stream_of(x)
        gx = gW = gb = None
        work = dict(nbytes=4 * rows * (K + Nout) + 4 * K * Nout, flops=2 * rows * K * Nout)
        # a wide, short product ([B, N] scores against the whole entity table: Nout = N >> rows): both gradients reduce over
        # or stream along the N entity rows, so both run on the transposed score gradient g^T [N, B]
        wide = Nout > 1024
        need_w = ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])
        wide_in = wide and WIDE_BWD_INPUT and ctx.needs_input_grad[0] and rows <= 1024 and (rows <= 128 or (rows % 4 == 0 and K % 4 == 0))
        gT = None
        if wide and WIDE_BWD_INPUT and (wide_in or not ctx.needs_input_grad[0]) and (wide_in or need_w):
            # the activation's derivative and the transposition in one pass over g and y (mrg_act_grad_transpose); g itself is not needed
            require_hip(g)
            gT = torch.empty(Nout, rows, dtype=torch.float32, device=x.device)
            call("mrg_act_grad_transpose", (ptr(g), ptr(y), ptr(gT), rows, Nout, ctx.act, st), nbytes=4 * rows * Nout * (3 if ctx.act else 2))
            g = None
        else:
            if ctx.act == 1:
                g = g * (y > 0)           # ReLU mask (elementwise; folded into the fused kernel later)
            elif ctx.act == 2:
                g = g * y * (1 - y)       # sigmoid
            if wide_in or (wide and need_w):
                gT = g.t().contiguous()
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if wide_in:
                # gx = g W = (g^T)^T W is a reduction over N into a [B, K] block: the shape of a weight gradient (rows := N,
                # gY := g^T, X := W).  The row GEMM would give the whole N-long reduction to ceil(B / 128) workgroups
                # (36 ms at B = 256, N = 1 M, K = 256); the split-over-rows kernel spreads it over the chip.
                ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", Nout, K, rows), x)
                call("mrg_linear_bwd_weight", (ptr(gT), ptr(W), None, ptr(gx), None, ptr(ws), Nout, K, 0, rows, st), **work)
            else:
                wt = _ws(_ws_bytes("mrg_linear_bwd_input_workspace_bytes", K, Nout), x)
                call("mrg_linear_bwd_input", (ptr(g), ptr(W), ptr(gx), ptr(wt), rows, K, Nout, K, 0, st), **work)
        if need_w:
            gW = torch.empty_like(W)
            gb = torch.empty(Nout, dtype=torch.float32, device=x.device) if ctx.has_b else None
            if wide:
                # the split-over-rows weight-gradient kernel keeps all of gW's row tiles in registers and does not cover this
                # shape; here gW = g^T x is itself a tall-skinny row GEMM over the transposed operands
                xT = x.t().contiguous()
                gws = _ws(_ws_bytes("mrg_gemm_workspace_bytes", rows, K), x)
                call("mrg_linear_fwd", (ptr(gT), ptr(xT), None, ptr(gW), ptr(gws), Nout, rows, K, 0, st), **work)
                if gb is not None:
                    gb = gT.sum(1)
            else:
                ws = _ws(_ws_bytes("mrg_linear_bwd_weight_workspace_bytes", rows, K, Nout), x)
                call("mrg_linear_bwd_weight", (ptr(g), ptr(x), None, ptr(gW), ptr(gb), ptr(ws), rows, K, 0, Nout, st), **work)
        return gx, gW, gb, None


def linear(x, W, b=None, act=None):
    """act(x W^T + b) with exact-f32 MFMA (nn.Linear semantics)."""
    return _Linear.apply(x, W, b, ACT[act])


NODE_LINEAR = os.environ.get("MRG_NODE_LINEAR", "1") == "1"      # lab switch: 0 = the node-level nn.Linear modules stay on torch (Tensile)


def module_linear(mod, x):
    """An nn.Linear module applied to HIP rows on the library's row GEMM (forward, input and weight gradient) instead of the
    vendor GEMM torch would pick: the entity projection and the cells' concat Linear (reference models/model_search_lp.py:131,
    models/cell_lp.py:186-188) are [N, .] x [., D] products whose Tensile kernels cost 60-120 us each at N = 14 541."""
    if NODE_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
        return linear(x, mod.weight, mod.bias)
    return mod(x)


# ---------------------------------------------------------------------------
# G: row gather with a deterministic (atomic-free) backward
# ---------------------------------------------------------------------------
class GatherPlan:
    """Index of a gather ``out[i] = table[idx[i]]``: the int32 index for the
    forward and a CSR over table rows (graph.dst_csr_plan) for the backward,
    which is a segmented sum of the incoming gradient rows."""

    def __init__(self, idx, num_table_rows):
        self.idx = idx
        self.idx32 = idx.to(torch.int32).contiguous()
        self.rows = int(num_table_rows)
        self._plan = self._sp = self._meta = None      # built on first use: a new step graph per step pays only for what it runs

    @property
    def plan(self):
        if self._plan is None:
            from .graph import dst_csr_plan, settle
            self._plan = dst_csr_plan(self.idx, self.rows)
            settle(self.idx.device)
        return self._plan

    @property
    def sp(self):
        if self._sp is None:
            from .graph import span_plan, settle
            self._sp = span_plan(self.idx, self.rows)
            settle(self.idx.device)
        return self._sp

    @property
    def meta(self):
        if self._meta is None:
            from .graph import span_meta, settle
            self._meta = span_meta(self.sp, None)           # xi = the element's own index
            settle(self.idx.device)          # built at first use (often inside a backward on a side stream), read on any stream
        return self._meta


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, gp):
        ctx.gp = gp
        return gather_rows(table, gp.idx32)

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        if g.is_cuda:
            return span_gcs("copy", g, None, ctx.gp.meta, ctx.gp.sp), None
        out, _ = _seg_fwd(0, g, None, ctx.gp.plan, ctx.gp.rows, g.shape[1])
        return out, None


def gather(table, gp):
    """table[gp.idx] with autograd (reference models/model_lp.py:131, models/model_search_lp.py:144-145,153-154)."""
    return _Gather.apply(table, gp)


class LazyRows:
    """``table[gp.idx]`` not yet materialised: what the supernet hands to the cell's first stage, whose three compose
    candidates then gather on the fly (compose above).  Anything else that needs the rows calls materialize()."""

    def __init__(self, table, gp):
        self.table, self.gp = table, gp
        self._rows = None

    is_cuda = property(lambda self: self.table.is_cuda)
    device = property(lambda self: self.table.device)
    requires_grad = property(lambda self: self.table.requires_grad)
    shape = property(lambda self: (int(self.gp.idx32.numel()), int(self.table.shape[1])))

    def materialize(self):
        if self._rows is None:
            self._rows = gather(self.table, self.gp)
        return self._rows


def _pair_meta(gp_a, gp_b):
    """Packed metadata of the MUL backward of a fused gather-compose, segments = gp_a's table rows: element e carries
    its own index (the upstream gradient row) and gp_b's table row as the second operand.  Cached on gp_a per partner."""
    cache = gp_a.__dict__.setdefault("_pair", {})
    m = cache.get(id(gp_b))
    if m is None:
        from .graph import settle, span_meta
        m = cache[id(gp_b)] = (span_meta(gp_a.sp, None, gp_b.idx32), gp_b)      # keeps the partner alive: id() stays unique
        settle(gp_a.idx.device)
    return m[0]


class _GatherCompose(torch.autograd.Function):
    """out[i] = ent[ie[i]] (op) rel[ir[i]]  (G + a1: reference models/model_search_lp.py:135-145 + models/operations_lp.py:71-98).
    Backward: gradients of the TABLES, each a balanced segmented sum over the rows that read a table row
    (mrg_span_gcs: COPY / NEGS for sub and add, MUL with the other table's row for mult)."""

    @staticmethod
    def forward(ctx, op, ent, rel, gp_e, gp_r):
        ent, rel = f32c(ent), f32c(rel)
        require_hip(ent, rel)
        rows, D = int(gp_e.idx32.numel()), ent.shape[1]
        if int(gp_r.idx32.numel()) != rows or rel.shape[1] != D:
            raise _lib.MrgnasError("gather-compose: the two index lists / tables do not match")
        out = torch.empty(rows, D, dtype=torch.float32, device=ent.device)
        call("mrg_gather_compose_fwd", (op, ptr(ent), ptr(rel), ptr(gp_e.idx32), ptr(gp_r.idx32), ptr(out), rows, D, stream_of(ent)),
             nbytes=rows * (12 * D + 8))
        ctx.op, ctx.gp = op, (gp_e, gp_r)
        ctx.save_for_backward(*((ent, rel) if op == 0 else ()))
        return out

    @staticmethod
    def backward(ctx, g):
        g = f32c(g)
        gp_e, gp_r = ctx.gp
        g_ent = g_rel = None
        if ctx.op == 0:                                     # mult
            ent, rel = ctx.saved_tensors
            if ctx.needs_input_grad[1]:
                g_ent = span_gcs("mul", g, rel, _pair_meta(gp_e, gp_r), gp_e.sp)
            if ctx.needs_input_grad[2]:
                g_rel = span_gcs("mul", g, ent, _pair_meta(gp_r, gp_e), gp_r.sp)
        else:                                               # sub / add
            if ctx.needs_input_grad[1]:
                g_ent = span_gcs("copy", g, None, gp_e.meta, gp_e.sp)
            if ctx.needs_input_grad[2]:
                g_rel = span_gcs("negs" if ctx.op == 1 else "copy", g, None, gp_r.meta, gp_r.sp)
        return None, g_ent, g_rel, None, None


# ---------------------------------------------------------------------------
# a9: fused gather -> compose -> segmented sum (CompGCN) and its backward
# ---------------------------------------------------------------------------
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
            from .graph import settle
            v = self._c[key] = build()
            settle(self.xi.device)
        return v

    def _span(self, which):
        from .graph import span_plan
        keys = {"seg": (self.seg, self.n_seg), "x": (self.xi, self.n_x), "y": (self.yi, self.n_y)}[which]
        return self._lazy("sp_" + which, lambda: span_plan(*keys))

    def _chunk(self, which):
        from .graph import dst_csr_plan
        keys = {"seg": (self.seg, self.n_seg), "x": (self.xi, self.n_x), "y": (self.yi, self.n_y)}[which]
        return self._lazy("by_" + which, lambda: dst_csr_plan(*keys))

    sp_seg = property(lambda self: self._span("seg"))
    sp_x = property(lambda self: self._span("x"))
    sp_y = property(lambda self: self._span("y"))
    by_seg = property(lambda self: self._chunk("seg"))
    by_x = property(lambda self: self._chunk("x"))
    by_y = property(lambda self: self._chunk("y"))

    def _meta(self, key, plan, a, b, scal):
        from .graph import span_meta
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


# ---------------------------------------------------------------------------
# X: MixedOp epilogue   out = sum_k w_k * ReLU(BatchNorm_k(y_k))
# ---------------------------------------------------------------------------
class _MixCfg:
    """Non-tensor arguments of the epilogue: the BatchNorm modules (running statistics are
    updated in place like torch does), which branches are all-zero, sharding info."""

    def __init__(self, bns, present, group=None, total_rows=None, has_addend=False, rowscale=None, identity=None, gated=None):
        self.bns, self.present, self.group, self.total_rows, self.has_addend = bns, present, group, total_rows, has_addend
        # (k, s, c [rows]): candidate k arrives as its GATE and is recomputed as gate * s * c[r] wherever the kernels read it
        # (include/mrgnas.h: mrg_gated_branch), or None
        self.gated = gated
        self.chain = None              # (StatChain, index): the statistics collectives are shared with other epilogues
        self.identity = identity       # index of the candidate that returns its input unchanged (f_identity), or None
        self.rowscale = rowscale       # per candidate None or (norm [E] | None, edge_rows, scale_edge, scale_self, gated node | None, node): folded into its gradient


def _row_candidate_as_s(cfg, ys):
    """ys with the row-factor candidate's [rows] factor replaced by the tensor the kernels read in its slot: s."""
    if cfg.gated is not None and cfg.gated.get("row_k") is not None:
        ys = list(ys)
        ys[cfg.gated["row_k"]] = cfg.gated["s"]
    return ys


# Test instrumentation (tests/test_configs_gpu.py, "mask replay"): when set, called as MASK_TAP(bns, masks) right after a fused
# epilogue's combine with the ReLU decision of every candidate, [rows, D] bool each, taken from the combine kernel ITSELF (one extra
# launch per candidate with a one-hot weight vector: w_k * relu(bn_k(y_k)) with w_k = 1, so `> 0` is the kernel's own decision, not a
# re-evaluation that could round differently).  None in the product.
MASK_TAP = None


class _MixedEpilogue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, w, *tensors):
        from ._lib import ptr_array
        K_ = len(cfg.bns)
        nz = sum(cfg.present)
        ys_nz = [f32c(t) for t in tensors[:nz]]
        gam, bet = list(tensors[nz:nz + K_]), list(tensors[nz + K_:nz + 2 * K_])
        addend = f32c(tensors[nz + 2 * K_]) if cfg.has_addend else None
        it = iter(ys_nz)
        ys = [next(it) if p else None for p in cfg.present]
        ys = _row_candidate_as_s(cfg, ys)                  # the row-factor candidate's slot holds s; its [rows] factor travels in the descriptor
        ref = next((y for y in ys if y is not None), None)
        if ref is None:
            raise _lib.MrgnasError("mixed epilogue needs at least one non-zero branch to know the row count")
        require_hip(w, addend, *ys_nz, *gam, *bet)
        rows, D = ref.shape
        dev, st = ref.device, stream_of(ref)
        w = f32c(w)
        total = float(cfg.total_rows if cfg.total_rows is not None else rows)
        coef = torch.empty(K_, 4, D, dtype=torch.float32, device=dev)
        ypa = ptr_array(ys)
        gb = _lib.gated_branch(cfg.gated)
        # [rows, D] tensors a pass reads: the stored candidates, the gate of the recomputed one, and s once for every candidate that is a function of it
        nz_rd = len({y.data_ptr() for y in ys if y is not None} | ({cfg.gated["s"].data_ptr()} if cfg.gated is not None else set()))
        bn0 = cfg.bns[0]
        training = bn0.training or not bn0.track_running_stats
        if training:
            ws = _ws(_ws_bytes("mrg_mix_workspace_bytes", K_, D), ref)
            track = bn0.track_running_stats
            rm = ptr_array([b.running_mean if track else None for b in cfg.bns])
            rv = ptr_array([b.running_var if track else None for b in cfg.bns])
            mom = bn0.momentum if bn0.momentum is not None else 0.1
            if cfg.group is None:                          # no collective between statistics and coefficients: two launches, not three
                call("mrg_mix_stats_coef", (ypa, ptr_array(gam), ptr_array(bet), rm, rv, K_, rows, total, D, bn0.eps, mom, ptr(coef), ptr(ws), gb, st),
                     nbytes=4 * D * rows * nz_rd)
            else:
                import torch.distributed as dist
                if cfg.chain is not None and cfg.chain[0].sums is not None:
                    sums = cfg.chain[0].sums[cfg.chain[1]]             # column sums of the whole graph, all-reduced with the other members'
                else:
                    sums = torch.empty(K_, 2, D, dtype=torch.float64, device=dev)
                    call("mrg_mix_colstats", (ypa, K_, rows, D, ptr(sums), ptr(ws), gb, st), nbytes=4 * D * rows * nz_rd)
                    _all_reduce_sum(sums, cfg.group)
                call("mrg_mix_finalize_fwd", (ptr(sums), ptr_array(gam), ptr_array(bet), rm, rv, K_, total, D, bn0.eps, mom, ptr(coef), st))
            if track:                                      # one multi-tensor launch instead of one per BatchNorm
                torch._foreach_add_([b.num_batches_tracked for b in cfg.bns], 1)
        else:   # eval: fixed statistics
            for k, b in enumerate(cfg.bns):
                invstd = torch.rsqrt(b.running_var + b.eps)
                coef[k, 0] = gam[k] * invstd
                coef[k, 1] = bet[k] - b.running_mean * gam[k] * invstd
                coef[k, 2] = invstd
                coef[k, 3] = b.running_mean * invstd
        out = torch.empty(rows, D, dtype=torch.float32, device=dev)
        call("mrg_mix_fwd", (ypa, K_, ptr(coef), ptr(w), ptr(addend), ptr(out), rows, D, gb, st), nbytes=4 * D * rows * (nz_rd + 1 + (addend is not None)))
        if MASK_TAP is not None:
            masks = []
            for k in range(K_):
                one = torch.zeros(K_, dtype=torch.float32, device=dev)
                one[k] = 1.0
                o = torch.empty(rows, D, dtype=torch.float32, device=dev)
                call("mrg_mix_fwd", (ypa, K_, ptr(coef), ptr(one), None, ptr(o), rows, D, gb, st))
                masks.append(o > 0)
            MASK_TAP(cfg.bns, masks)
        ctx.cfg, ctx.training, ctx.total, ctx.nz, ctx.nz_rd = cfg, training, total, nz, nz_rd
        ctx.save_for_backward(w, coef, *ys_nz)
        if cfg.chain is not None:
            cfg.chain[0].register(cfg.chain[1], ctx)
        return out

    @staticmethod
    def _launch_bwd_reduce(ctx, g, red):
        """red[k][0..2] <- this rank's sums of member ctx for upstream gradient g."""
        from ._lib import ptr_array
        w, coef, *ys_nz = ctx.saved_tensors
        cfg, K_ = ctx.cfg, len(ctx.cfg.bns)
        it = iter(ys_nz)
        ys = _row_candidate_as_s(cfg, [next(it) if p else None for p in cfg.present])
        rows, D = g.shape
        ws = _ws(_ws_bytes("mrg_mix_workspace_bytes", K_, D), g)
        call("mrg_mix_bwd_reduce", (ptr(g), ptr_array(ys), K_, ptr(coef), ptr(w), ptr(red), ptr(ws), rows, D, _lib.gated_branch(cfg.gated), stream_of(g)),
             nbytes=4 * D * rows * (ctx.nz_rd + 1))

    @staticmethod
    def backward(ctx, g):
        from ._lib import ptr_array
        w, coef, *ys_nz = ctx.saved_tensors
        cfg, K_, nz = ctx.cfg, len(ctx.cfg.bns), ctx.nz
        g = f32c(g)
        it = iter(ys_nz)
        ys = [next(it) if p else None for p in cfg.present]
        ys = _row_candidate_as_s(cfg, ys)
        rows, D = g.shape
        dev, st = g.device, stream_of(g)
        ypa = ptr_array(ys)
        gb = _lib.gated_branch(cfg.gated)
        shared = None
        if cfg.group is not None and ctx.training and cfg.chain is not None:
            shared = cfg.chain[0].reduced_gradient_sums(cfg.chain[1], g, _MixedEpilogue._launch_bwd_reduce)
        if shared is not None:
            red_local, red = shared
            cfg.chain[0].release(cfg.chain[1])
        else:
            red = torch.empty(K_, 3, D, dtype=torch.float32, device=dev)
            _MixedEpilogue._launch_bwd_reduce(ctx, g, red)
            red_local = red
            if cfg.group is not None and ctx.training:
                red = red.clone()
                _all_reduce_sum(red, cfg.group)
        coef2 = torch.empty(K_, 2, D, dtype=torch.float32, device=dev)
        dw = torch.empty(K_, dtype=torch.float32, device=dev)
        call("mrg_mix_finalize_bwd", (ptr(red), K_, ctx.total, D, ptr(coef2), None, None, ptr(dw), st))
        if not ctx.training:
            coef2.zero_()
        if red_local is not red:                      # sharded: parameter / alpha gradients stay local partial sums
            dw = red_local[:, 2].sum(dim=1)
        need_y = list(ctx.needs_input_grad[2:2 + nz])
        rs = cfg.rowscale
        # f_identity's output IS the operand s of the gated candidate (f_dense_comp) of the same MixedOp: its gradient is added into
        # that candidate's direct term inside the apply kernel instead of being written and re-read by the state's fan-in sum
        add_from = None
        if (FOLD_IDENTITY and cfg.identity is not None and cfg.present[cfg.identity] and rs is not None):
            pos = sum(cfg.present[:cfg.identity])
            for k in range(K_):
                if (rs[k] is not None and rs[k][4] is not None and need_y[pos] and k != cfg.identity
                        and _same_memory(rs[k][4].s, ys[cfg.identity])):
                    add_from = (k, cfg.identity)
                    need_y[pos] = False                    # no gradient tensor of its own: None flows back to the alias
                    break
        gys_nz = [torch.empty_like(y) if nd else None for y, nd in zip(ys_nz, need_y)]
        it = iter(gys_nz)
        gys = [next(it) if p else None for p in cfg.present]
        row_k = cfg.gated.get("row_k") if cfg.gated is not None else None
        row_dq = None
        if row_k is not None:                              # the row-factor candidate: a [rows] gradient w.r.t. its factor, no [rows, D] one
            row_dq = gys[row_k] if gys[row_k] is not None else torch.empty(rows, dtype=torch.float32, device=dev)
            gys[row_k] = None
            rlink = cfg.gated["row_link"]
            if rlink is not None:
                rlink.row_written = row_dq.data_ptr()     # the factor's node checks that THIS buffer reaches it (one reader)
                gb = _lib.gated_branch(dict(cfg.gated, row_h=rlink.row_h, row_uvc=rlink.row_uvc, row_ld=rlink.row_uvc.shape[1]), row_dq)
        n_out = sum(t is not None and t.dim() == 2 for t in gys_nz)
        if rs is not None and any(r is not None for r in rs):
            import ctypes
            on = (ctypes.c_int * K_)(*[int(r is not None) for r in rs])
            rs_ptr = ptr_array([r[0] if r is not None else None for r in rs])
            rs_edge = (ctypes.c_int64 * K_)(*[int(r[1]) if r is not None else 0 for r in rs])
            rs_scale = (ctypes.c_float * K_)(*[float(r[2]) if r is not None else 1.0 for r in rs])
            rs_self = (ctypes.c_float * K_)(*[float(r[3]) if r is not None else 1.0 for r in rs])
            # the same multipliers expanded over all rows (cached per graph): one unconditional load per row in the kernel
            rs_full_t = [_gated_rowscale(r[0], int(r[1]), rows, float(r[2]), float(r[3]), dev) if r is not None else None for r in rs]
            rs_full = ptr_array(rs_full_t)
            # gated consumers (f_dense_comp): their dz AND the direct term of their input gradient are written here; the buffer of
            # the direct term is handed to the consumer's backward node (which runs later, maybe on a side stream)
            gated = [r is not None and r[4] is not None for r in rs]
            for k in range(K_):
                if gated[k]:
                    on[k] = 2
            f_gs = [torch.empty_like(rs[k][4].s) if gated[k] else None for k in range(K_)]
            f_s = ptr_array([rs[k][4].s if gated[k] else None for k in range(K_)])
            f_gate = ptr_array([rs[k][4].gate if gated[k] else None for k in range(K_)])
            for k in range(K_):
                if gated[k]:
                    rs[k][4].gs_direct = f_gs[k]
            f_gs_p = ptr_array(f_gs)
            n_fold = sum(gated)
            f_add = (ctypes.c_int * K_)(*[(add_from[1] if (add_from is not None and k == add_from[0]) else -1) for k in range(K_)])
        else:
            on = rs_ptr = rs_edge = rs_scale = rs_self = rs_full = f_s = f_gate = f_gs_p = f_add = None
            n_fold = 0
        call("mrg_mix_bwd_apply", (ptr(g), ypa, ptr_array(gys), K_, ptr(coef), ptr(coef2), ptr(w), rs_ptr, rs_scale, rs_self, rs_edge, on, rs_full,
                                   f_s, f_gate, f_gs_p, f_add, rows, D, gb, st),
             nbytes=4 * D * rows * (1 + ctx.nz_rd + n_out + n_fold))
        if rs is not None:
            for k in range(K_):                       # the consumer checks that THIS buffer is what reaches it (no second reader of y)
                if rs[k] is not None and gys[k] is not None:
                    rs[k][5].written[rs[k][6]] = gys[k].data_ptr()
        dgam = [red_local[k, 1] for k in range(K_)]
        dbet = [red_local[k, 0] for k in range(K_)]
        return (None, dw, *gys_nz, *dgam, *dbet) + ((g,) if cfg.has_addend else ())       # d out / d addend = identity


FOLD_IDENTITY = os.environ.get("MRG_FOLD_IDENTITY", "1") == "1"     # lab switch: 0 = f_identity's gradient stays a tensor of its own


def mixed_epilogue(ys, bns, w, group=None, total_rows=None, addend=None, fold_row_scales=False, identity=None):
    """addend + sum_k w[k] * relu(bn_k(ys[k]))  (reference models/cell_lp.py:25-33).  ys[k] is None for an
    all-zero operator output (f_zero); bns are the nn.BatchNorm1d modules (affine); addend: the output of the MixedOp this
    one is summed with (the sum of the MixedOps feeding a state, :104-113), accumulated inside the combine kernel."""
    return mixed_epilogue_prepare(ys, bns, group, total_rows, fold_row_scales, identity)(w, addend)


class PreparedEpilogue:
    """A MixedOp epilogue whose candidates are known but which has not run: calling it with (w, addend) runs it.  Exists so that the
    statistics collectives of several epilogues can be issued together (StatChain) before any of them combines."""

    def __init__(self, cfg, cand, bns):
        self.cfg, self.cand, self.bns = cfg, cand, bns

    def __call__(self, w, addend=None):
        self.cfg.has_addend = addend is not None
        tensors = self.cand + [b.weight for b in self.bns] + [b.bias for b in self.bns] + ([addend] if addend is not None else [])
        return _MixedEpilogue.apply(self.cfg, w, *tensors)


def _all_reduce_sum(t, group):
    """In-place sum over the ranks of `group`: a c10d process group, or a communicator of rccl.py (RCCL bound directly -- one
    stream-ordered launch on the current stream, which is what lets a sharded step be captured in a HIP graph)."""
    if getattr(group, "is_direct_rccl", False):
        group.all_reduce(t, "sum")
    else:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)


class StatChain:
    """ONE collective for the BatchNorm statistics of several MixedOp epilogues over row-sharded candidates (VERDICT r2 #4b).
    Forward: the column sums of every member are computed first (they do not depend on one another: an `addend` only enters the
    combine) into one stacked buffer, all-reduced once; each member's forward then finalizes from its slice.  Backward, when the
    members' outputs are summed into one state (`summed=True`: every member receives the SAME upstream gradient): the member whose
    backward runs first launches the gradient reductions of ALL members, all-reduces the stacked result once and leaves each
    member's slice for its own backward.  Values are those of one collective per member (a sum over ranks of the same numbers)."""

    def __init__(self, members, group, summed):
        import torch.distributed as dist
        # Only counts are kept: the members' candidate tensors stay owned by their PreparedEpilogue (advisor r3: a chain that held
        # `members` kept every [rows, D] candidate of every member alive until Python's cyclic collector ran).
        self.group, self.summed = group, summed
        self.ks = [len(m.cfg.bns) for m in members]
        self.ctx = [None] * len(members)
        self.served = [False] * len(members)
        self.bwd = None                                    # (g data_ptr, [red_local_j], [red_global_j])
        self.sums = None
        cfgs = [m.cfg for m in members]
        if not members or not all(self._trains(c) for c in cfgs):
            return                                         # eval mode: fixed statistics, no collective at all
        ks = self.ks
        first = next(y for y in _row_candidate_as_s(cfgs[0], self._ys(members[0])) if y is not None)
        D = first.shape[1]
        sums = torch.empty(sum(ks), 2, D, dtype=torch.float64, device=first.device)
        off = 0
        for j, m in enumerate(members):
            m.cfg.chain = (self, j)
            ys = _row_candidate_as_s(m.cfg, self._ys(m))
            y0 = next(y for y in ys if y is not None)
            rows = y0.shape[0]
            ws = _ws(_ws_bytes("mrg_mix_workspace_bytes", ks[j], D), y0)
            nz_rd = len({y.data_ptr() for y in ys if y is not None} | ({m.cfg.gated["s"].data_ptr()} if m.cfg.gated is not None else set()))
            call("mrg_mix_colstats", (ptr_array(ys), ks[j], rows, D, ptr(sums[off:off + ks[j]]), ptr(ws), _lib.gated_branch(m.cfg.gated), stream_of(y0)),
                 nbytes=4 * D * rows * nz_rd)
            off += ks[j]
        _all_reduce_sum(sums, group)
        self.sums, off = [], 0
        for k_ in ks:
            self.sums.append(sums[off:off + k_])
            off += k_

    @staticmethod
    def _trains(cfg):
        b = cfg.bns[0]
        return b.training or not b.track_running_stats

    @staticmethod
    def _ys(m):
        it = iter(f32c(t) for t in m.cand)
        return [next(it) if p else None for p in m.cfg.present]

    def register(self, j, ctx):
        """Member j's autograd context, needed only when the members share ONE backward reduction (`summed`).  cfg -> chain -> ctx ->
        cfg is a reference cycle: release() breaks it as soon as the last member's backward has taken its slice."""
        if self.summed:
            self.ctx[j] = ctx

    def release(self, j):
        self.served[j] = True
        if all(self.served):
            self.ctx = [None] * len(self.ks)
            self.bwd = None
            self.served = [False] * len(self.ks)

    def reduced_gradient_sums(self, j, g, launch):
        """(red_local, red_global) of member j for upstream gradient g; `launch(ctx, g, red_out)` runs one member's reduction."""
        import torch.distributed as dist
        if not self.summed:
            return None
        if self.bwd is None or self.bwd[0] != g.data_ptr():
            if any(c is None for c in self.ctx):
                return None
            ks = self.ks
            D = g.shape[1]
            red = torch.empty(sum(ks), 3, D, dtype=torch.float32, device=g.device)
            off, loc = 0, []
            for i, c in enumerate(self.ctx):
                launch(c, g, red[off:off + ks[i]])
                loc.append(red[off:off + ks[i]])
                off += ks[i]
            glob = red.clone()
            _all_reduce_sum(glob, self.group)
            off, gl = 0, []
            for k_ in ks:
                gl.append(glob[off:off + k_])
                off += k_
            self.bwd = (g.data_ptr(), loc, gl, g)          # g kept alive: its address identifies the batch
        return self.bwd[1][j], self.bwd[2][j]


def mixed_epilogue_prepare(ys, bns, group=None, total_rows=None, fold_row_scales=False, identity=None):
    """mixed_epilogue without running it: returns a PreparedEpilogue.  ys[k]: None (f_zero), a [rows, D] tensor (a stored
    candidate), or a Candidate from an operator's for_epilogue path (stored with a foldable first backward pass / gate-only /
    row factor)."""
    cands = [y if isinstance(y, Candidate) else None for y in ys]
    ys = [c.y if c is not None else y for c, y in zip(cands, ys)]
    present = [y is not None for y in ys]
    # a candidate whose backward starts with a row scale of its incoming gradient (f_comp: dz = g * c) and whose output feeds
    # ONLY this epilogue gets that scale folded into the epilogue's gradient store; its producer's Link is claimed for it
    rowscale = [None] * len(ys)
    if fold_row_scales:
        for k, c in enumerate(cands):
            if c is not None and c.rowscale is not None and c.link is not None and c.link.claim(c.slot):
                # (norm, b1, scale_edge, scale_self, the Link when the producer is a gated filter -- it holds s / gate and receives
                #  the direct term --, the Link, the producer's output slot)
                rowscale[k] = c.rowscale[:4] + (c.link if c.rowscale[4] else None, c.link, c.slot)
    gated = None
    for k, c in enumerate(cands):
        if c is not None and c.kind == "gate":
            if gated is not None:
                raise _lib.MrgnasError("mixed epilogue: one recomputed (gate-only) candidate at most")
            gated = dict(k=k, s=c.s, c=c.c)
    # the row-factor candidate (f_sparse_comp as fvec [rows]): recomputed as s * fvec[r] by the kernels when the gated candidate of
    # the same rows s is there to receive its gradient w.r.t. s; multiplied out by plain tensor arithmetic otherwise
    ys = list(ys)
    for k, c in enumerate(cands):
        if c is None or c.kind != "rowfactor":
            continue
        y, s_r, rb0, rb1 = c.y, c.s, c.b0, c.b1
        D_ = s_r.shape[1]
        wants_grad = torch.is_grad_enabled() and (y.requires_grad or s_r.requires_grad)
        # one float4 step per lane (KMAX == 1 in mrg_mix_bwd_apply's row dot) needs 16-byte aligned rows of EVERY tensor the kernels
        # touch: an offset view would pass here and fail in the middle of loss.backward() (advisor r3)
        aligned = all(t.data_ptr() % 16 == 0 for t in [s_r] + [t for t in ys if t is not None and t.dim() == 2])
        ok = (gated is not None and "row_k" not in gated and _same_memory(gated["s"], s_r)
              and ((D_ % 4 == 0 and D_ <= 256 and aligned) or D_ <= 64))
        if ok and wants_grad:                              # the gated candidate's folded gradient store is where the gradient w.r.t. s goes
            rs_g = rowscale[gated["k"]]
            ok = rs_g is not None and rs_g[4] is not None and y.requires_grad and c.link is not None
        if ok:
            gated.update(row_k=k, row_f=y, b0=rb0, b1=rb1, row_link=c.link if wants_grad else None)
        else:
            ys[k] = c.materialize()
    cfg = _MixCfg(list(bns), present, group, total_rows, False, rowscale, identity, gated)
    return PreparedEpilogue(cfg, [y for y in ys if y is not None], list(bns))



# ---------------------------------------------------------------------------
# cell zero: the MixedOp over the compose candidates, recomputed from the tables
# ---------------------------------------------------------------------------
CELL_ZERO_FUSED = os.environ.get("MRG_CELL_ZERO_FUSED", "1") == "1"     # lab switch: 0 = three gather-compose launches + the generic epilogue


class _CellZeroMixed(torch.autograd.Function):
    """out = sum_k w[k] * relu(bn_k(ent[ie] (op_k) rel[ir]))  -- Cell_Zero's MixedOp over PRE_OPS (reference models/cell_lp.py:53-68,
    :25-33; models/operations_lp.py:71-98; the gather of models/model_search_lp.py:135-145) without any candidate output:
    statistics, combine and both gradient passes recompute the candidates from the (cache-resident) tables (mrg_zero_*), the
    backward writes the two combined per-row gradients and two balanced span sums turn them into the table gradients."""

    @staticmethod
    def forward(ctx, cfg, w, ent, rel, *gb):
        import ctypes
        ops, bns, gp_e, gp_r, group, total_rows = cfg
        K_ = len(ops)
        ent, rel, w = f32c(ent), f32c(rel), f32c(w)
        gam, bet = list(gb[:K_]), list(gb[K_:])
        require_hip(w, ent, rel, *gam, *bet)
        rows, D = int(gp_e.idx32.numel()), ent.shape[1]
        if int(gp_r.idx32.numel()) != rows or rel.shape[1] != D:
            raise _lib.MrgnasError("cell zero: the two index lists / tables do not match")
        dev, st = ent.device, stream_of(ent)
        opc = (ctypes.c_int * K_)(*[COMPOSE[o] for o in ops])
        total = float(total_rows if total_rows is not None else rows)
        coef = torch.empty(K_, 4, D, dtype=torch.float32, device=dev)
        bn0 = bns[0]
        training = bn0.training or not bn0.track_running_stats
        src = (ptr(ent), ptr(rel), ptr(gp_e.idx32), ptr(gp_r.idx32), opc, K_)
        gathered = rows * (8 * D + 8)
        if training:
            ws = _ws(_ws_bytes("mrg_zero_workspace_bytes", D), ent)
            track = bn0.track_running_stats
            rm = ptr_array([b.running_mean if track else None for b in bns])
            rv = ptr_array([b.running_var if track else None for b in bns])
            mom = bn0.momentum if bn0.momentum is not None else 0.1
            if group is None:
                call("mrg_zero_stats_coef", (*src, ptr_array(gam), ptr_array(bet), rm, rv, rows, total, D, bn0.eps, mom, ptr(coef), ptr(ws), st),
                     nbytes=gathered)
            else:
                import torch.distributed as dist
                sums = torch.empty(K_, 2, D, dtype=torch.float64, device=dev)
                call("mrg_zero_colstats", (*src, rows, D, ptr(sums), ptr(ws), st), nbytes=gathered)
                _all_reduce_sum(sums, group)
                call("mrg_mix_finalize_fwd", (ptr(sums), ptr_array(gam), ptr_array(bet), rm, rv, K_, total, D, bn0.eps, mom, ptr(coef), st))
            if track:
                torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
        else:
            for k, b in enumerate(bns):
                invstd = torch.rsqrt(b.running_var + b.eps)
                coef[k, 0] = gam[k] * invstd
                coef[k, 1] = bet[k] - b.running_mean * gam[k] * invstd
                coef[k, 2] = invstd
                coef[k, 3] = b.running_mean * invstd
        out = torch.empty(rows, D, dtype=torch.float32, device=dev)
        call("mrg_zero_fwd", (*src, ptr(coef), ptr(w), ptr(out), rows, D, st), nbytes=4 * D * rows)
        if MASK_TAP is not None:
            masks = []
            for k in range(K_):
                one = torch.zeros(K_, dtype=torch.float32, device=dev)
                one[k] = 1.0
                o = torch.empty(rows, D, dtype=torch.float32, device=dev)
                call("mrg_zero_fwd", (*src, ptr(coef), ptr(one), ptr(o), rows, D, st))
                masks.append(o > 0)
            MASK_TAP(bns, masks)
        ctx.cfg, ctx.training, ctx.total, ctx.opc = cfg, training, total, opc
        ctx.save_for_backward(w, coef, ent, rel)
        return out

    @staticmethod
    def backward(ctx, g):
        w, coef, ent, rel = ctx.saved_tensors
        ops, bns, gp_e, gp_r, group, _ = ctx.cfg
        K_ = len(ops)
        g = f32c(g)
        rows, D = g.shape
        dev, st = g.device, stream_of(g)
        src = (ptr(ent), ptr(rel), ptr(gp_e.idx32), ptr(gp_r.idx32), ctx.opc, K_)
        ws = _ws(_ws_bytes("mrg_zero_workspace_bytes", D), g)
        red = torch.empty(K_, 3, D, dtype=torch.float32, device=dev)
        call("mrg_zero_bwd_reduce", (ptr(g), *src, ptr(coef), ptr(w), ptr(red), ptr(ws), rows, D, st), nbytes=4 * D * rows)
        red_local = red
        if group is not None and ctx.training:
            import torch.distributed as dist
            red = red.clone()
            _all_reduce_sum(red, group)
        coef2 = torch.empty(K_, 2, D, dtype=torch.float32, device=dev)
        dw = torch.empty(K_, dtype=torch.float32, device=dev)
        call("mrg_mix_finalize_bwd", (ptr(red), K_, ctx.total, D, ptr(coef2), None, None, ptr(dw), st))
        if not ctx.training:
            coef2.zero_()
        if red_local is not red:                      # sharded: parameter / alpha gradients stay local partial sums
            dw = red_local[:, 2].sum(dim=1)
        need_e, need_r = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        ge_rows = torch.empty(rows, D, dtype=torch.float32, device=dev) if need_e else None
        gr_rows = torch.empty(rows, D, dtype=torch.float32, device=dev) if need_r else None
        call("mrg_zero_bwd_apply", (ptr(g), *src, ptr(coef), ptr(coef2), ptr(w), ptr(ge_rows), ptr(gr_rows), rows, D, st),
             nbytes=4 * D * rows * (1 + int(need_e) + int(need_r)))
        g_ent = span_gcs("copy", ge_rows, None, gp_e.meta, gp_e.sp) if need_e else None
        g_rel = span_gcs("copy", gr_rows, None, gp_r.meta, gp_r.sp) if need_r else None
        dgam = [red_local[k, 1] for k in range(K_)]
        dbet = [red_local[k, 0] for k in range(K_)]
        return (None, dw, g_ent, g_rel, *dgam, *dbet)


def cell_zero_mixed(kinds, s, hr, bns, w, group=None, total_rows=None):
    """sum_k w[k] * relu(bn_k(compose(kinds[k], s, hr))) for LazyRows operands (the cell's first stage), nothing [rows, D]-sized
    but the output is written forward; backward two combined per-row gradients."""
    cfg = (tuple(kinds), list(bns), s.gp, hr.gp, group, total_rows)
    return _CellZeroMixed.apply(cfg, w, s.table, hr.table, *[b.weight for b in bns], *[b.bias for b in bns])

# ---------------------------------------------------------------------------
# dense (per-feature) filters on the MFMA row GEMM
# ---------------------------------------------------------------------------
_SIDE_STREAMS = {}
FORK_MIN_ROWS = 1 << 17      # below this many rows an operator is launch-bound: no side streams
SEGMENT_STREAMS = int(os.environ.get("MRG_SEGMENT_STREAMS", "3"))   # streams the direction segments of one operator use


class Fork:
    """Independent kernel chains (the direction segments of one operator write disjoint row ranges) run on
    side HIP streams and join back: the tail of one GEMM (a 272 115-row segment fills 4.15 rounds of the chip,
    the 14 541 self rows a fifth of one) is filled by the workgroups of the next instead of idling.
    Every tensor the chains touch is allocated on the main stream BEFORE the fork."""

    def __init__(self, device, n, tag="segments"):
        if tag == "segments":
            n = min(n, SEGMENT_STREAMS)
        self.main = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), tag)   # one pool per nesting level
        pool = _SIDE_STREAMS.setdefault(key, [])
        while len(pool) < n - 1:
            pool.append(torch.cuda.Stream(device=device))
        self.side = pool[:max(n - 1, 0)]
        from .graph import register_stream
        register_stream(self.main)
        for st in self.side:
            register_stream(st)
            st.wait_stream(self.main)

    def stream(self, i):
        i %= len(self.side) + 1
        return self.main if i == 0 else self.side[i - 1]

    def on(self, i):
        """Context: launch on chain i's stream (nothing to switch when there are no side streams)."""
        return torch.cuda.stream(self.stream(i)) if self.side else contextlib.nullcontext()

    def join(self):
        for st in self.side:
            self.main.wait_stream(st)


FOLD_ROW_SCALE = os.environ.get("MRG_FOLD_ROW_SCALE", "1") == "1"          # lab switch: 0 = f_comp's dz pass stays a launch of its own
GROUPED_SEGMENTS = os.environ.get("MRG_GROUPED_SEGMENTS", "1") == "1"    # lab switch: 0 = one launch per direction segment


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
        ws3 = int(_lib.load().mrg_dense_filter3_workspace_bytes(D, K_)) if (GROUPED_SEGMENTS and all(params[2 * i] is not None for i in range(3))) else 0
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
        fork = Fork(s.device, len(segs) if M >= FORK_MIN_ROWS else 1)
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
        if GROUPED_SEGMENTS and all(params[2 * i] is not None for i in range(3)):
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
        fork = Fork(s.device, len(work) if M >= FORK_MIN_ROWS else 1)
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
    fold = for_epilogue and FOLD_ROW_SCALE and f32c(s).is_cuda and torch.is_grad_enabled()
    link = Link() if fold else None
    y = _DenseFilter.apply(link, kind, s, s_in, norm, int(b0), int(b1), 1.0 / 3.0, float(self_scale),
                           W_in, b_in, W_out, b_out, W_self, b_self)
    if fold and y.requires_grad:
        # the backward begins with an elementwise pass over the incoming gradient (f_comp: dz = g * c; f_dense_comp: dz = g c s gate
        # (1 - gate) and the direct term g c gate; c = norm / 3 on edge rows, self_scale on self rows): a MixedOp epilogue that is
        # the only reader of y writes its gradient in that form (Candidate.rowscale + the Link it claims)
        return Candidate("stored", y, link=link, slot=0, rowscale=(norm, int(b1), 1.0 / 3.0, float(self_scale), kind == 0))
    return y


DENSE_PAIR = os.environ.get("MRG_DENSE_PAIR", "1") == "1"       # lab switch: 0 = f_dense_comp and f_comp of a MixedOp as two autograd nodes


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
    if not (DENSE_PAIR and GROUPED_SEGMENTS):
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


GATED_RECOMPUTE = os.environ.get("MRG_GATED_RECOMPUTE", "1") == "1"     # lab switch: 0 = f_dense_comp's output is stored for the epilogue


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
    fold = for_epilogue and FOLD_ROW_SCALE and s.is_cuda and torch.is_grad_enabled()
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



# ---------------------------------------------------------------------------
# DistMult scoring (the step after the path)
# ---------------------------------------------------------------------------
class ScorePlan:
    """Index structures of a scoring batch of (s, r, o) triples: int32 indices for the forward and
    three span plans (by subject, by object, by relation) whose metadata carries the element id in
    the scale slot, so the backward can take the upstream gradient as an external scale."""

    def __init__(self, triplets, n_ent, n_rel):
        from .graph import span_plan
        t = triplets.long()
        s, r, o = t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous()
        self.T, self.n_ent, self.n_rel = int(t.shape[0]), int(n_ent), int(n_rel)
        self.s32, self.r32, self.o32 = (x.to(torch.int32).contiguous() for x in (s, r, o))

        def packed(plan, xi, yi):
            from .graph import span_meta
            return span_meta(plan, xi, yi, None, w_is_index=True)
        self.by_s, self.by_o, self.by_r = span_plan(s, n_ent), span_plan(o, n_ent), span_plan(r, n_rel)
        self.m_s = packed(self.by_s, o, r)        # g_ent[s] += g_t * ent[o] * rel[r]
        self.m_o = packed(self.by_o, s, r)        # g_ent[o] += g_t * ent[s] * rel[r]
        self.m_r = packed(self.by_r, s, o)        # g_rel[r] += g_t * ent[s] * ent[o]     (Y = ent as well)


class _DistMult(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent, rel, sp):
        ent, rel = f32c(ent), f32c(rel)
        require_hip(ent, rel)
        D = ent.shape[1]
        score = torch.empty(sp.T, dtype=torch.float32, device=ent.device)
        call("mrg_distmult_score", (ptr(ent), ptr(rel), ptr(sp.s32), ptr(sp.r32), ptr(sp.o32), ptr(score), sp.T, D, stream_of(ent)),
             nbytes=sp.T * (12 * D + 16))
        ctx.sp = sp
        ctx.save_for_backward(ent, rel)
        return score

    @staticmethod
    def backward(ctx, g):
        ent, rel = ctx.saved_tensors
        sp = ctx.sp
        g = f32c(g)
        g_ent = span_gcs("mul", ent, rel, sp.m_s, sp.by_s, ext_scal=g)
        g_ent += span_gcs("mul", ent, rel, sp.m_o, sp.by_o, ext_scal=g)
        g_rel = span_gcs("mul", ent, ent, sp.m_r, sp.by_r, ext_scal=g)
        return g_ent, g_rel, None


def distmult_score(ent, rel, sp):
    """sum_c ent[s] * rel[r] * ent[o] per triple (reference models/model_search_lp.py:169-176)."""
    return _DistMult.apply(ent, rel, sp)



# ---------------------------------------------------------------------------
# [B, N] score functions (the step after the path in the fixed-genotype driver)
# ---------------------------------------------------------------------------
def distmult_scores_all(all_ent, sub_emb, rel_emb):
    """sf_DisMult_op (reference models/operations_lp.py:115-127): sigmoid((sub * rel) all_ent^T) as the compose kernel +
    the MFMA row GEMM with a sigmoid epilogue (all_ent [N, D] is the GEMM's weight operand: no transpose, no [B, N]
    pre-activation tensor)."""
    return linear(compose("mult", sub_emb, rel_emb), all_ent, None, "sigmoid")


class _TransE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, all_ent, sub, rel, gamma):
        all_ent, sub, rel = f32c(all_ent), f32c(sub), f32c(rel)
        require_hip(all_ent, sub, rel)
        B, D = sub.shape
        N = all_ent.shape[0]
        score = torch.empty(B, N, dtype=torch.float32, device=sub.device)
        call("mrg_transe_score_fwd", (ptr(all_ent), ptr(sub), ptr(rel), float(gamma), ptr(score), B, N, D, stream_of(sub)),
             nbytes=4 * (B * N + (N + 2 * B) * D))
        ctx.save_for_backward(all_ent, sub, rel, score)
        return score

    @staticmethod
    def backward(ctx, g):
        all_ent, sub, rel, score = ctx.saved_tensors
        g = f32c(g)
        B, D = sub.shape
        N = all_ent.shape[0]
        gent = torch.empty_like(all_ent) if ctx.needs_input_grad[0] else None
        gobj = torch.empty_like(sub) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        call("mrg_transe_score_bwd", (ptr(all_ent), ptr(sub), ptr(rel), ptr(g), ptr(score), ptr(gent), ptr(gobj), B, N, D, stream_of(sub)))
        return gent, gobj, gobj, None


def transe_scores_all(all_ent, sub_emb, rel_emb, gamma):
    """sf_TransE_op (reference models/operations_lp.py:101-112): sigmoid(gamma - ||sub + rel - ent||_1) for all entities."""
    return _TransE.apply(all_ent, sub_emb, rel_emb, gamma)


# ---------------------------------------------------------------------------
# gradient fan-in of a tensor with several readers
# ---------------------------------------------------------------------------
def sum_buffers(xs):
    """Sum of equally-shaped HIP tensors in K-way passes (mrg_sum_buffers), k = 0..K-1 order."""
    xs = [f32c(x) for x in xs]
    require_hip(*xs)
    out = torch.empty_like(xs[0])
    n = out.numel()
    for i in range(0, len(xs), 8):
        part = xs[i:i + 8]
        call("mrg_sum_buffers", (ptr_array(part), len(part), ptr(out), n, int(i > 0), stream_of(out)),
             nbytes=4 * n * (len(part) + 1 + int(i > 0)))
    return out


class _Fanout(torch.autograd.Function):
    """k aliases of one tensor whose gradients are summed in one K-way pass instead of k - 1
    pairwise adds.  Aliases nobody reads cost nothing (their gradient stays None)."""

    @staticmethod
    def forward(ctx, x, k, box):
        ctx.set_materialize_grads(False)
        ctx.box = box                                       # mailbox of this batch of aliases (Fan.take / _AggRows.backward)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        gathered = ctx.box.gathered
        if gathered is not None:                            # a reader (a_sum) left its gradient as a gather of an [N, D] tensor
            gh, gself, graph, ev = gathered
            ctx.box.gathered = None
            cur = torch.cuda.current_stream(gh.device)
            cur.wait_event(ev)                                # the producer's stream may not be this one
            for t in (gh, gself):
                if t is not None:
                    t.record_stream(cur)                      # ... and its allocator must not recycle the blocks under this launch
            E, N, D = graph.num_edges(), graph.number_of_nodes(), gh.shape[1]
            if len(gs) <= 8 and all(g.is_cuda and g.shape == (E + N, D) for g in gs):
                gs = [f32c(g) for g in gs]
                out = torch.empty(E + N, D, dtype=torch.float32, device=gh.device)
                call("mrg_sum_rows_gather", (ptr_array(gs), len(gs), ptr(f32c(gh)), ptr(f32c(gself)), ptr(graph.i32("dst")), E, E + N, D, ptr(out),
                                             stream_of(out)), nbytes=4 * D * (E + N) * (len(gs) + 1))
                return out, None, None
            gx = torch.empty(E + N, D, dtype=torch.float32, device=gh.device)      # shapes the kernel does not take: materialise
            if gself is not None:
                gx[E:] = gself
            else:
                gx[E:].zero_()
            _seg_bwd(0, gh, graph, None, gx, None)
            gs.append(gx)
        if not gs:
            return None, None, None
        if len(gs) == 1:
            return gs[0], None, None
        if not gs[0].is_cuda or any(g.shape != gs[0].shape for g in gs):
            tot = gs[0]
            for g in gs[1:]:
                tot = tot + g
            return tot, None, None
        return sum_buffers(gs), None, None


class _FanBox:
    """Mailbox of one batch of Fan aliases: a reader whose gradient w.r.t. the alias is a gather of a small tensor (a_sum) leaves the
    small tensor here instead of materialising [rows, D]; the batch's fan-in sum (_Fanout.backward) reads it."""
    __slots__ = ("gathered",)

    def __init__(self):
        self.gathered = None


class Fan:
    """Hands out aliases of `x` to its readers: ``fan.take()`` per reader, at most `cap` of them.
    Aliases are created a dozen at a time (a further batch hangs off the last alias of the previous one, so
    its gradients arrive as one pre-summed tensor).  Without autograd (or for tensors that need no
    gradient) the tensor itself is returned."""

    BATCH = 12

    def __init__(self, x, cap):
        self.x = x
        self.left = cap
        self._live = torch.is_grad_enabled() and x.requires_grad and cap > 1
        self._views = []
        self._root = x

    def take(self):
        if not self._live:
            return self.x
        if self.left <= 0:
            raise RuntimeError("Fan: more readers than announced")
        self.left -= 1
        if not self._views:
            self._box = _FanBox()
            views = list(_Fanout.apply(self._root, self.BATCH, self._box))
            self._root = views.pop()                    # source of the next batch, if one is ever needed
            self._views = views
        v = self._views.pop()
        Fan._remember(v, self._box)                     # lets a reader leave its gradient with the fan-in sum (_AggRows.backward)
        return v

    # alias tensor -> the mailbox of the fan-out batch it came from.  A side table keyed by the alias OBJECT's id (tensors compare
    # elementwise, so they cannot key a dict themselves) holding a weak reference that removes the entry when the alias dies: a reader
    # handed anything else -- a copy, a cast that allocates -- simply is not found and materialises its gradient as usual.
    _NODE = {}

    @staticmethod
    def _remember(v, box):
        key = id(v)
        Fan._NODE[key] = (weakref.ref(v, lambda _r, key=key: Fan._NODE.pop(key, None)), box)

    @staticmethod
    def node_of(x):
        hit = Fan._NODE.get(id(x))
        return hit[1] if hit is not None and hit[0]() is x else None
