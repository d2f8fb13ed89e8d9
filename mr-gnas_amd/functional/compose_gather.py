"""a1: compose ops (pre_mult / pre_sub / pre_add) and G, the row gather that feeds them (csrc/compose.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib
from .._lib import call, f32c, ptr, require_hip, stream_of
from ._base import COMPOSE
from .gcs import span_gcs
from .reducers import _seg_fwd


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
            from ..graph import dst_csr_plan, settle
            self._plan = dst_csr_plan(self.idx, self.rows)
            settle(self.idx.device)
        return self._plan

    @property
    def sp(self):
        if self._sp is None:
            from ..graph import span_plan, settle
            self._sp = span_plan(self.idx, self.rows)
            settle(self.idx.device)
        return self._sp

    @property
    def meta(self):
        if self._meta is None:
            from ..graph import span_meta, settle
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
        from ..graph import settle, span_meta
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
