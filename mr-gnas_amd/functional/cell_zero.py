"""Cell zero: the MixedOp over the compose candidates, recomputed from the entity / relation tables (csrc/mixedop.hip zero_*).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib
from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of
from . import switches as SW
from ._base import COMPOSE, _ws, _ws_bytes, bump_counters
from .gcs import span_gcs
from .mixed import _all_reduce_sum


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
                sums = torch.empty(K_, 2, D, dtype=torch.float64, device=dev)
                call("mrg_zero_colstats", (*src, rows, D, ptr(sums), ptr(ws), st), nbytes=gathered)
                _all_reduce_sum(sums, group)
                call("mrg_mix_finalize_fwd", (ptr(sums), ptr_array(gam), ptr_array(bet), rm, rv, K_, total, D, bn0.eps, mom, ptr(coef), st))
            if track:
                bump_counters(bns)
        else:
            for k, b in enumerate(bns):
                invstd = torch.rsqrt(b.running_var + b.eps)
                coef[k, 0] = gam[k] * invstd
                coef[k, 1] = bet[k] - b.running_mean * gam[k] * invstd
                coef[k, 2] = invstd
                coef[k, 3] = b.running_mean * invstd
        out = torch.empty(rows, D, dtype=torch.float32, device=dev)
        call("mrg_zero_fwd", (*src, ptr(coef), ptr(w), ptr(out), rows, D, st), nbytes=4 * D * rows)
        if SW.MASK_TAP is not None:
            masks = []
            for k in range(K_):
                one = torch.zeros(K_, dtype=torch.float32, device=dev)
                one[k] = 1.0
                o = torch.empty(rows, D, dtype=torch.float32, device=dev)
                call("mrg_zero_fwd", (*src, ptr(coef), ptr(one), ptr(o), rows, D, st))
                masks.append(o > 0)
            SW.MASK_TAP(bns, masks)
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
