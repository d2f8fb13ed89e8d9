"""X: the MixedOp epilogue  out = sum_k w_k * ReLU(BatchNorm_k(y_k))  (csrc/mixedop.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import torch

from .. import _lib
from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of
from . import switches as SW
from ._base import _same_memory, _ws, _ws_bytes, bump_counters
from .candidates import Candidate
from .dense import _gated_rowscale


class _MixCfg:
    """Non-tensor arguments of the epilogue: the BatchNorm modules (running statistics are
    updated in place like torch does), which branches are all-zero, sharding info."""

    def __init__(self, bns, present, group=None, total_rows=None, has_addend=False, rowscale=None, identity=None, gated=None, act=0):
        self.bns, self.present, self.group, self.total_rows, self.has_addend = bns, present, group, total_rows, has_addend
        self.act = act                 # 0 ReLU, 1 tanh behind the BatchNorm (_lib.ACTS)
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


class _MixedEpilogue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, w, *tensors):
        from .._lib import ptr_array
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
        gb = _lib.gated_branch(cfg.gated, act=cfg.act)
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
                bump_counters(cfg.bns)
        else:   # eval: fixed statistics
            for k, b in enumerate(cfg.bns):
                invstd = torch.rsqrt(b.running_var + b.eps)
                coef[k, 0] = gam[k] * invstd
                coef[k, 1] = bet[k] - b.running_mean * gam[k] * invstd
                coef[k, 2] = invstd
                coef[k, 3] = b.running_mean * invstd
        out = torch.empty(rows, D, dtype=torch.float32, device=dev)
        call("mrg_mix_fwd", (ypa, K_, ptr(coef), ptr(w), ptr(addend), ptr(out), rows, D, gb, st), nbytes=4 * D * rows * (nz_rd + 1 + (addend is not None)))
        if SW.MASK_TAP is not None:
            masks = []
            for k in range(K_):
                one = torch.zeros(K_, dtype=torch.float32, device=dev)
                one[k] = 1.0
                o = torch.empty(rows, D, dtype=torch.float32, device=dev)
                call("mrg_mix_fwd", (ypa, K_, ptr(coef), ptr(one), None, ptr(o), rows, D, gb, st))
                masks.append(o > 0)
            SW.MASK_TAP(cfg.bns, masks)
        ctx.cfg, ctx.training, ctx.total, ctx.nz, ctx.nz_rd = cfg, training, total, nz, nz_rd
        ctx.save_for_backward(w, coef, *ys_nz)
        if cfg.chain is not None:
            cfg.chain[0].register(cfg.chain[1], ctx)
        return out

    @staticmethod
    def _launch_bwd_reduce(ctx, g, red):
        """red[k][0..2] <- this rank's sums of member ctx for upstream gradient g."""
        from .._lib import ptr_array
        w, coef, *ys_nz = ctx.saved_tensors
        cfg, K_ = ctx.cfg, len(ctx.cfg.bns)
        it = iter(ys_nz)
        ys = _row_candidate_as_s(cfg, [next(it) if p else None for p in cfg.present])
        rows, D = g.shape
        ws = _ws(_ws_bytes("mrg_mix_workspace_bytes", K_, D), g)
        call("mrg_mix_bwd_reduce", (ptr(g), ptr_array(ys), K_, ptr(coef), ptr(w), ptr(red), ptr(ws), rows, D, _lib.gated_branch(cfg.gated, act=cfg.act), stream_of(g)),
             nbytes=4 * D * rows * (ctx.nz_rd + 1))

    @staticmethod
    def backward(ctx, g):
        from .._lib import ptr_array
        w, coef, *ys_nz = ctx.saved_tensors
        cfg, K_, nz = ctx.cfg, len(ctx.cfg.bns), ctx.nz
        g = f32c(g)
        it = iter(ys_nz)
        ys = [next(it) if p else None for p in cfg.present]
        ys = _row_candidate_as_s(cfg, ys)
        rows, D = g.shape
        dev, st = g.device, stream_of(g)
        ypa = ptr_array(ys)
        gb = _lib.gated_branch(cfg.gated, act=cfg.act)
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
        if (SW.FOLD_IDENTITY and cfg.identity is not None and cfg.present[cfg.identity] and rs is not None):
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
                gb = _lib.gated_branch(dict(cfg.gated, row_h=rlink.row_h, row_uvc=rlink.row_uvc, row_ld=rlink.row_uvc.shape[1]), row_dq, act=cfg.act)
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


def mixed_epilogue(ys, bns, w, group=None, total_rows=None, addend=None, fold_row_scales=False, identity=None, act="relu"):
    """addend + sum_k w[k] * relu(bn_k(ys[k]))  (reference models/cell_lp.py:25-33).  ys[k] is None for an
    all-zero operator output (f_zero); bns are the nn.BatchNorm1d modules (affine); addend: the output of the MixedOp this
    one is summed with (the sum of the MixedOps feeding a state, :104-113), accumulated inside the combine kernel.
    act: "relu" (the MixedOp) or "tanh" (CompGraphConv's BatchNorm -> tanh tail, reference models/compgcn.py:100-111)."""
    return mixed_epilogue_prepare(ys, bns, group, total_rows, fold_row_scales, identity, act)(w, addend)


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
            call("mrg_mix_colstats", (ptr_array(ys), ks[j], rows, D, ptr(sums[off:off + ks[j]]), ptr(ws), _lib.gated_branch(m.cfg.gated, act=m.cfg.act), stream_of(y0)),
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


def mixed_epilogue_prepare(ys, bns, group=None, total_rows=None, fold_row_scales=False, identity=None, act="relu"):
    """mixed_epilogue without running it: returns a PreparedEpilogue.  ys[k]: None (f_zero), a [rows, D] tensor (a stored
    candidate), or a Candidate from an operator's for_epilogue path (stored with a foldable first backward pass / gate-only /
    row factor)."""
    from ..lazy import real as _real_tensor
    ys = [_real_tensor(y) for y in ys]                     # an operator's lazy handle (lazy.py) handed over directly: its value
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
    cfg = _MixCfg(list(bns), present, group, total_rows, False, rowscale, identity, gated, _lib.ACTS[act])
    return PreparedEpilogue(cfg, [y for y in ys if y is not None], list(bns))
