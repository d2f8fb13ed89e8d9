"""a4 / a5 / a6: destination-segmented reducers (a_sum / a_mean / a_max) and the Linear + ReLU + reduce fusions (csrc/segreduce.hip, linear.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import os

import torch

from .. import _lib
from .._lib import call, f32c, ptr, require_hip, stream_of
from . import switches as SW
from ._base import REDUCE, _cnt, _ws, _ws_bytes
from .gcs import span_gcs


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


ORDERED_BWD = os.environ.get("MRG_SEG_BWD_ORDERED", "1") == "1"     # lab switch: 0 = the aggregators' backward walks the edges in edge-id order (rounds 1-4)
ORDERED_BWD_MIN_BYTES = int(os.environ.get("MRG_SEG_BWD_ORDERED_MIN_BYTES", str(128 << 20)))   # gathered tables smaller than this stay in edge-id order


def _seg_bwd(mode, g, graph, arg, gmsg, gself, relu_src=None, relu_bits=None):
    """gmsg [E, D] (and gself [N, D]) of the destination-segmented reducers' backward (mrg_seg_reduce_bwd_ordered): the edges are walked in
    destination order (the plan's CSR-by-destination list), so the gathered rows of g / arg are re-used from cache."""
    p = graph.plan()
    E, N, D = graph.num_edges(), graph.number_of_nodes(), g.shape[1]
    nb = 4 * D * E * (1 + (relu_src is not None)) + 4 * E + 4 * D * N * (1 + (gself is not None) + (mode == 2))
    # ... where the tables are beyond the caches (C5: 1 GB each, 5.26 -> 3.76 ms for a_max's backward); cache-resident tables (FB15k-237:
    # 11.6 MB) gain nothing from the order and pay its indirection (0.190 -> 0.225 ms): edge-id order there (profiles/r5_stragglers.txt)
    big = 4 * D * N * (2 if mode == 2 else 1) > ORDERED_BWD_MIN_BYTES
    order = p["eid"] if (ORDERED_BWD and g.is_cuda and big) else None
    call("mrg_seg_reduce_bwd_ordered", (mode, ptr(g), ptr(graph.i32("dst")), ptr(p["in_degree"]), ptr(arg), ptr(gmsg), ptr(gself),
                                        ptr(relu_src), ptr(relu_bits), ptr(order), E, N, D, stream_of(g)), nbytes=nb)



def _amax_bwd_sparse(g, mx, arg, W, graph, gy, gx):
    """a_max's backward for the fused forward: gy = the gradient w.r.t. the messages (one non-zero per (node, column): the arg-max
    edge where the maximum is positive) AND gx[:E] = gy W in one pass per edge row, without the dense input-gradient product
    (mrg_segmax_bwd_input).  Returns False when the shape is not taken (W beyond the LDS of a CU: D = 256)."""
    E, N, D = graph.num_edges(), graph.number_of_nodes(), g.shape[1]
    if not (SW.SPARSE_AMAX_BWD and g.is_cuda and _lib.load().mrg_segmax_bwd_input_ok(D, W.shape[1])):
        return False
    p = graph.plan()
    big = 4 * D * N * 3 > ORDERED_BWD_MIN_BYTES            # the three gathered [N, D] tables beyond the caches: walk by destination
    order = p["eid"] if (ORDERED_BWD and big) else None
    call("mrg_segmax_bwd_input", (ptr(g), ptr(mx), ptr(graph.i32("dst")), ptr(arg), ptr(W), ptr(gy), ptr(gx), ptr(order), E, N, D, W.shape[1],
                                  stream_of(g)), nbytes=4 * E * (D + W.shape[1]) + 4 * E + 12 * N * D)
    return True

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
        from .fanin import Fan                     # (fanin imports this module: resolved at call time)
        ctx.fan_node = Fan.node_of(x) if (SW.LAZY_ASUM and mode == 0 and x.is_cuda) else None
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


def aggregate_rows(kind, x, graph, add_self=True, keep=None):
    return _AggRows.apply(REDUCE[kind], x, graph, add_self, keep)


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
        fused_ws = _fused_agg_ws(N, D) if (mode == 2 and SW.FUSED_AMAX and E >= SW.FUSED_AMAX_MIN_ROWS) else 0
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
            if SW.MASK_TAP is not None:                       # test instrumentation: which edge won, and whether the maximum is positive
                SW.MASK_TAP(("a_max", W.data_ptr()), [arg, mx > 0])
            return out
        if (mode == 1 and SW.FUSED_AMEAN and E >= SW.FUSED_AMAX_MIN_ROWS and hasattr(graph, "plan")
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
            if SW.MASK_TAP is not None:                       # test instrumentation: the inner ReLU's decisions, one bit per message element
                SW.MASK_TAP(("a_mean", W.data_ptr()), [bits])
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
            _seg_bwd(1, g, graph, None, gy, gx[E:], relu_bits=bits)
        elif ctx.fused:
            gx[E:] = g
            have_gx = _amax_bwd_sparse(g, mx, arg, W, graph, gy, gx)       # gy and gx[:E] in one pass, no dense product
            if not have_gx:
                _seg_bwd(mode, g * (mx > 0), graph, arg, gy, None)         # the winning message is ReLU-dead iff the maximum is 0
        else:
            _seg_bwd(mode, g, graph, arg, gy, gx[E:], relu_src=y)          # gy masked by ReLU; gx[E:] = g
        work = dict(nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        if not (ctx.fused is True and have_gx):
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
        fused_ws = _fused_agg_ws(N, D) if (mode == 2 and SW.FUSED_AMAX and E >= SW.FUSED_AMAX_MIN_ROWS) else 0
        if fused_ws > 0:                                # one GEMM with the ReLU + segmented-max epilogue, as in _LinReluAgg
            out = torch.empty(N, D, dtype=torch.float32, device=x.device)
            arg = torch.empty(N, D, dtype=torch.int32, device=x.device)
            mx = torch.empty(N, D, dtype=torch.float32, device=x.device)
            call("mrg_linear_relu_segmax_fwd", (ptr(x), ptr(W), ptr(b), ptr(graph.plan()["eid"]), ptr(graph.i32("dst")), None, ptr(out), ptr(arg),
                                                ptr(mx), ptr(_ws(fused_ws, x)), E, N, D, D, st),
                 nbytes=4 * E * D + 8 * E + 4 * D * D + 4 * N * D * 3, flops=2 * E * D * D)
            ctx.mode, ctx.graph, ctx.fused = mode, graph, True
            ctx.save_for_backward(x, W, arg, mx)
            if SW.MASK_TAP is not None:                       # test instrumentation: which edge won, and whether the maximum is positive
                SW.MASK_TAP(("a_max", W.data_ptr()), [arg, mx > 0])
            return out, x[E:].clone()
        if (mode != 2 and SW.FUSED_AMEAN and E >= SW.FUSED_AMAX_MIN_ROWS and _fused_agg_ws(N, D) > 0):
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
            _seg_bwd(0, g, graph, None, gy, None, relu_bits=bits)
        elif ctx.fused:
            have_gx = _amax_bwd_sparse(g, mx, arg, W, graph, gy, gx)    # gy and gx[:E] in one pass, no dense product
            if not have_gx:
                _seg_bwd(mode, g * (mx > 0), graph, arg, gy, None)      # the winning message is ReLU-dead iff the maximum is 0
        else:
            _seg_bwd(mode, g, graph, arg, gy, None, relu_src=y)        # gy masked by ReLU
        work = dict(nbytes=4 * E * 2 * D + 4 * D * D, flops=2 * E * D * D)
        if not (ctx.fused is True and have_gx):
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
