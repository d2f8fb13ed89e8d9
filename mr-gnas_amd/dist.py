"""Relation-block sharding of one supernet step over the GPUs of a node (RCCL over xGMI).

One process per GPU (``torch.distributed``, backend "nccl" == RCCL on ROCm; the CPU
tests run the same code over "gloo").  Everything row-shaped is partitioned, nothing
edge- or node-shaped is replicated:

* directed edges, sorted by (relation, dst), are cut into P contiguous *relation
  blocks* balanced by edge count; a cut is snapped to a (relation, dst) group
  boundary so one destination's edges of one relation stay together, and a relation
  bigger than E/P is split by dst range (WN18RR: 2 of 11 relations hold ~3/4 of the
  edges).  Relation ids < R are original-direction edges, so a contiguous slice of
  the relation-sorted list is still [in-edges | out-edges]: the shard only carries
  its local split (b0, b1) instead of a per-edge direction flag;
* node rows (self-loop rows, the [N, D] stage of the cell, the scoring triples) are
  cut into P contiguous ranges;
* tables ([N_all, D] entities, [R', D] relations), parameters and alphas are
  replicated.

Exchange steps (the only collectives):
  1. per middle MixedOp: ONE reduce-scatter (max) of a_max's per-node partial
     accumulators and ONE reduce-scatter (sum) of [a_sum | a_mean]'s, [N, D] / [N, 2D]
     built from the local edges, over EQUAL node chunks of ceil(N / P) rows (the
     tensor is padded by < P rows): every rank receives the reduction of its own
     node rows only -- half the bytes of an all-reduce.  Launched asynchronously as
     soon as the partials exist, so the exchange of MixedOp i overlaps the
     edge-parallel GEMMs of MixedOp i + 1.  Backward = the adjoint all-gather;
  2. BatchNorm statistics: all-reduce of [K, 2, D] (sum x, sum x^2) forward and of
     [K, 3, D] (sum g, sum g*xhat, ...) backward -- BN normalises over all M rows
     (reference models/cell_lp.py:21), so shards must share statistics.  One
     collective per MixedOp covers its K candidates; the MixedOps feeding ONE state
     (reference :104-113) share one collective forward and one backward
     (functional.StatChain): per cell 7 forward + 8 backward instead of 12 + 12;
  3. per layer: all_gather_into_tensor of the [N, D] node embeddings over the same
     equal chunks (next layer's gather and the scorer read rows of other ranks);
     backward = reduce-scatter (sum);
  4. per step: one flat all-reduce (sum) of all parameter / alpha gradients.
Every collective is an autograd Function whose backward is its adjoint collective,
so ``loss.backward()`` on each rank yields exact partial gradients.

gloo (the CPU tests) has no reduce_scatter_tensor / all_gather_into_tensor: the two
primitives are emulated there (all-reduce + slice, list all-gather) BELOW the
equal-chunk padding code, so the CPU tests run the very same chunk arithmetic the
RCCL path runs (world 2 / 3 / 8, N not divisible by the world size).
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import functional as K
from . import operations_lp as OPS
from .graph import RelGraph, cached_on

import os
# one all-reduce for the BatchNorm statistics of the MixedOps feeding a state, forward and backward (functional.StatChain); 0 = one per MixedOp
BATCH_STATS = os.environ.get("MRG_BATCH_STATS", "1") == "1"


# ---------------------------------------------------------------------------
# partitioning
# ---------------------------------------------------------------------------
def relation_block_cuts(etype_sorted, dst_sorted, parts):
    """Cut positions (len parts+1) into a (relation, dst)-sorted edge list: the k-th cut is
    the (relation, dst) group boundary nearest to k*E/parts."""
    E = len(etype_sorted)
    if E == 0:
        return [0] * (parts + 1)
    key = etype_sorted.astype(np.int64) * (int(dst_sorted.max()) + 1) + dst_sorted
    bounds = np.concatenate(([0], np.nonzero(np.diff(key))[0] + 1, [E]))     # group starts + E
    cuts = [0]
    for k in range(1, parts):
        ideal = k * E / parts
        j = int(np.argmin(np.abs(bounds - ideal)))
        cuts.append(max(int(bounds[j]), cuts[-1]))
    cuts.append(E)
    return cuts


def node_ranges(n, parts):
    base, extra = divmod(n, parts)
    lo = [r * base + min(r, extra) for r in range(parts + 1)]
    return lo


def node_chunks(n, parts):
    """Row ranges of EQUAL size ceil(n / parts) (the last ones shorter or empty): what reduce_scatter_tensor /
    all_gather_into_tensor move without per-rank padding tricks.  Returns (cuts [parts + 1], chunk)."""
    chunk = (n + parts - 1) // parts if n > 0 else 0
    return [min(r * chunk, n) for r in range(parts + 1)], chunk


class EdgeShard(RelGraph):
    """Rank-local view of a step graph: a relation block of edges + a node range."""

    def __init__(self, n, src, dst, etype, norm, num_rels, rank, world, device):
        if torch.is_tensor(src) and src.is_cuda:
            self._init_on_device(n, src, dst, etype, norm, num_rels, rank, world, device)
            return
        src, dst, etype = (np.asarray(t.cpu() if torch.is_tensor(t) else t).astype(np.int64) for t in (src, dst, etype))
        norm = np.asarray(norm.cpu() if torch.is_tensor(norm) else norm, dtype=np.float32).reshape(-1)
        order = np.lexsort((np.arange(len(src)), dst, etype))          # (relation, dst, caller id)
        cuts = relation_block_cuts(etype[order], dst[order], world)
        mine = order[cuts[rank]:cuts[rank + 1]]
        super().__init__(n, src[mine], dst[mine], etype[mine], norm[mine], device=device)
        self.rank, self.world = rank, world
        self.global_edge_ids = torch.from_numpy(mine).to(device)       # caller-order ids of the local rows
        self.E_global = len(src)
        self._b0 = int((etype[mine] < num_rels).sum())
        lo, self.node_chunk = node_chunks(n, world)             # equal chunks: one reduce-scatter / all-gather per exchange
        self.node_lo, self.node_hi = lo[rank], lo[rank + 1]
        self.node_cuts = lo
        self.global_in_degree = torch.from_numpy(np.bincount(dst, minlength=n)).to(device)
        self.cuts = cuts

    def _init_on_device(self, n, src, dst, etype, norm, num_rels, rank, world, device):
        """The same partition from device tensors (the host np.lexsort of a 544 230-edge list took longer than the
        rank's step): ONE stable device sort by the key relation * n + dst (stable = ties in caller order, what the lexsort's third
        key does), group boundaries by a difference + nonzero, and the cut arithmetic of relation_block_cuts on the boundary list
        (a few thousand integers on the host).  Same cuts, same local edge order as the host path (tests/test_dist_cpu.py)."""
        src, dst, etype = (t.to(device).long() for t in (src, dst, etype))
        norm = norm.to(device).reshape(-1).float()
        E = int(src.numel())
        key = etype * int(n) + dst
        skey, order = torch.sort(key, stable=True)
        if E:
            starts = torch.nonzero(skey[1:] != skey[:-1]).view(-1) + 1
            bounds = np.concatenate(([0], starts.cpu().numpy(), [E]))
            cuts = [0]
            for k in range(1, world):
                j = int(np.argmin(np.abs(bounds - k * E / world)))
                cuts.append(max(int(bounds[j]), cuts[-1]))
            cuts.append(E)
        else:
            cuts = [0] * (world + 1)
        mine = order[cuts[rank]:cuts[rank + 1]]
        RelGraph.__init__(self, n, src[mine], dst[mine], etype[mine], norm[mine], device=device)
        self.rank, self.world = rank, world
        self.global_edge_ids = mine
        self.E_global = E
        self._b0 = int((etype[mine] < num_rels).sum())
        lo, self.node_chunk = node_chunks(n, world)
        self.node_lo, self.node_hi = lo[rank], lo[rank + 1]
        self.node_cuts = lo
        self.global_in_degree = torch.bincount(dst, minlength=n)
        self.cuts = cuts

    def bounds(self):
        return self._b0, self.num_edges()

    @property
    def n_own(self):
        return self.node_hi - self.node_lo


# ---------------------------------------------------------------------------
# collectives with adjoint backward
# ---------------------------------------------------------------------------
def _is_direct(group):
    """A communicator of rccl.py (RCCL bound directly: stream-ordered launches, HIP-graph capturable) instead of a c10d group."""
    return getattr(group, "is_direct_rccl", False)


def _is_gloo(group):
    return not _is_direct(group) and dist.get_backend(group) == "gloo"


_OPNAME = {dist.ReduceOp.SUM: "sum", dist.ReduceOp.MAX: "max"}


def all_reduce(t, op, group):
    """In-place all-reduce of `t` over `group` (a c10d process group, None = the default one, or a direct RCCL communicator)."""
    if _is_direct(group):
        group.all_reduce(t, _OPNAME[op])
    else:
        dist.all_reduce(t, op=op, group=group)


def reduce_scatter_tensor(out, padded, op, group, async_op=False):
    """dist.reduce_scatter_tensor over equal chunks; on gloo (no such primitive) all-reduce + slice of the SAME padded
    tensor, so callers run identical chunk arithmetic on both backends.  Returns the async work handle or None (a direct RCCL
    communicator launches on the current stream: ordered by the stream itself, nothing to wait for)."""
    if _is_direct(group):
        group.reduce_scatter_tensor(out, padded, _OPNAME[op])
        return None
    if not _is_gloo(group):
        return dist.reduce_scatter_tensor(out, padded, op=op, group=group, async_op=async_op)
    tmp = padded.clone()
    dist.all_reduce(tmp, op=op, group=group)
    chunk, rank = out.shape[0], dist.get_rank(group)
    out.copy_(tmp[rank * chunk:(rank + 1) * chunk])
    return None


def all_gather_into_tensor(full, mine, group):
    """dist.all_gather_into_tensor over equal chunks; on gloo the list form into the same output tensor."""
    if _is_direct(group):
        group.all_gather_into_tensor(full, mine)
        return
    if not _is_gloo(group):
        dist.all_gather_into_tensor(full, mine, group=group)
        return
    world = dist.get_world_size(group)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine.contiguous(), group=group)
    chunk = mine.shape[0]
    for r, p in enumerate(parts):
        full[r * chunk:(r + 1) * chunk] = p


class _AllReduceSum(torch.autograd.Function):
    """inplace=True reduces into x itself (a temporary nobody else reads, e.g. a fresh segment-sum partial)."""

    @staticmethod
    def forward(ctx, x, group, inplace=False):
        ctx.group = group
        y = x if inplace else x.clone()
        if inplace:
            ctx.mark_dirty(x)
        all_reduce(y, dist.ReduceOp.SUM, group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        all_reduce(g, dist.ReduceOp.SUM, ctx.group)
        return g, None, None


class Exchange:
    """An aggregator exchange in flight: `start` launches the collective on RCCL's own stream (async_op), `finish`
    makes the current stream wait for it.  Between the two the caller enqueues independent work (the partials of the
    next MixedOp), which then overlaps the transfer."""

    def __init__(self):
        self.work = None

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


class _ReduceScatterRows(torch.autograd.Function):
    """part [N, C] on every rank (partial over the rank's local edges)  ->  own rows [n_own, C] of the reduction over
    ranks (sum | max): ONE reduce_scatter_tensor over equal node chunks (half the bytes of the round-1 all-reduce +
    slice; emulated below the padding code on gloo).  Backward = the adjoint: an all-gather of the
    owners' gradient rows (max: only the ranks that attain the maximum keep it -- ties only occur at 0 behind a ReLU,
    where the ReLU mask removes the gradient anyway)."""

    @staticmethod
    def forward(ctx, part, shard_info, op, group, ex):
        cuts, chunk, rank = shard_info
        world = len(cuts) - 1
        N, C = part.shape
        red = dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM
        ctx.info, ctx.op, ctx.group = shard_info, op, group
        if chunk * world != N:                                       # pad to equal chunks (fewer than `world` rows)
            padded = part.new_zeros(chunk * world, C) if op != "max" else part.new_full((chunk * world, C), float("-inf"))
            padded[:N] = part
        else:
            padded = part
        out = part.new_empty(chunk, C)
        work = reduce_scatter_tensor(out, padded, red, group, async_op=ex is not None)
        if ex is not None:
            ex.work = work
        own = out[: cuts[rank + 1] - cuts[rank]]
        if op == "max":
            ctx.save_for_backward(part, own)
        return own

    @staticmethod
    def backward(ctx, g):
        cuts, chunk, rank = ctx.info
        world = len(cuts) - 1
        N = cuts[-1]
        C = g.shape[1]
        g = g.contiguous()
        n_own = cuts[rank + 1] - cuts[rank]
        if ctx.op == "max":
            part, own = ctx.saved_tensors
            payload = torch.cat((g, own), dim=1)                     # the owners' gradient AND maximum travel together
        else:
            payload = g
        W = payload.shape[1]
        mine = payload if n_own == chunk else torch.cat((payload, payload.new_zeros(chunk - n_own, W)))
        full = payload.new_empty(chunk * world, W)
        all_gather_into_tensor(full, mine.contiguous(), ctx.group)
        full = full[:N]
        if ctx.op == "max":
            return full[:, :C] * (part == full[:, C:]), None, None, None, None
        return full, None, None, None, None


def reduce_scatter_rows(part, shard, op, group, ex=None):
    return _ReduceScatterRows.apply(part, (shard.node_cuts, shard.node_chunk, shard.rank), op, group, ex)


class _AllGatherRows(torch.autograd.Function):
    """[n_own, D] per rank -> [N, D] everywhere (equal node chunks: one all_gather_into_tensor, no per-rank padding
    lists, no torch.cat); backward = the sum over ranks of the gradient rows this rank owns (reduce-scatter)."""

    @staticmethod
    def forward(ctx, x, shard_info, group):
        cuts, chunk, rank = shard_info
        world = len(cuts) - 1
        N, D = cuts[-1], x.shape[1]
        ctx.info, ctx.group = shard_info, group
        mine = x if x.shape[0] == chunk else torch.cat((x, x.new_zeros(chunk - x.shape[0], D)))
        full = x.new_empty(chunk * world, D)
        all_gather_into_tensor(full, mine.contiguous(), group)
        return full[:N]

    @staticmethod
    def backward(ctx, g):
        cuts, chunk, rank = ctx.info
        world = len(cuts) - 1
        N, D = cuts[-1], g.shape[1]
        n_own = cuts[rank + 1] - cuts[rank]
        padded = g.contiguous() if chunk * world == N else torch.cat((g, g.new_zeros(chunk * world - N, D)))
        out = g.new_empty(chunk, D)
        reduce_scatter_tensor(out, padded, dist.ReduceOp.SUM, ctx.group)
        return out[:n_own], None, None


class _SyncBatchNorm(torch.autograd.Function):
    """Training-mode BatchNorm1d over rows that are spread over the ranks."""

    @staticmethod
    def forward(ctx, x, weight, bias, total_rows, eps, group):
        stats = torch.stack((x.sum(0), (x * x).sum(0)))
        all_reduce(stats, dist.ReduceOp.SUM, group)
        mean = stats[0] / total_rows
        var = (stats[1] / total_rows - mean * mean).clamp_(min=0)
        invstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * invstd
        ctx.save_for_backward(xhat, weight, invstd)
        ctx.total, ctx.group = total_rows, group
        ctx.mark_non_differentiable(mean, var)
        return xhat * weight + bias, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        xhat, weight, invstd = ctx.saved_tensors
        gw_local, gb_local = (g * xhat).sum(0), g.sum(0)
        red = torch.stack((gw_local, gb_local))
        all_reduce(red, dist.ReduceOp.SUM, ctx.group)
        gx = (g - red[1] / ctx.total - xhat * (red[0] / ctx.total)) * (weight * invstd)
        return gx, gw_local, gb_local, None, None, None


def sync_batch_norm(x, bn, total_rows, group):
    """F.batch_norm(training=True) semantics of module `bn` (nn.BatchNorm1d) over rows that are
    partitioned across `group`; updates bn.running_* like torch does (unbiased variance).
    In eval mode (running statistics present) it normalises with them, like nn.BatchNorm1d: no collective."""
    if not bn.training and bn.track_running_stats and bn.running_mean is not None:
        return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    y, mean, var = _SyncBatchNorm.apply(x, bn.weight, bn.bias, total_rows, bn.eps, group)
    if bn.training and bn.track_running_stats:
        with torch.no_grad():
            m = bn.momentum if bn.momentum is not None else 0.1
            unbiased = var * (total_rows / max(total_rows - 1, 1))
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(unbiased, alpha=m)
            bn.num_batches_tracked += 1
    return y


# ---------------------------------------------------------------------------
# the sharded supernet step
# ---------------------------------------------------------------------------
class ShardedSupernet:
    """Forward of supernet.SearchNetwork on one relation block (same parameters, same
    arithmetic, rows partitioned).  `kernels` is the namespace providing gather / seg_reduce /
    linear / GatherPlan -- the HIP `functional` module in the product; the CPU tests pass an
    oracle-backed stand-in to exercise the collectives over gloo."""

    def __init__(self, model, shard, node_id, group=None, kernels=K):
        self.m, self.s, self.group, self.k = model, shard, group, kernels
        s, dev = shard, shard.device
        src, _, _ = s.edges(form="all")
        own = torch.arange(s.node_lo, s.node_hi, device=dev)
        node_id = node_id.view(-1).long().to(dev)
        self.rows_total = s.E_global + s.number_of_nodes()            # M of the whole step graph
        self.p_ent0 = kernels.GatherPlan(node_id[torch.cat((src, own))], model._num_ent)
        self.p_ent = kernels.GatherPlan(torch.cat((src, own)), s.number_of_nodes())
        rel_idx = torch.cat((s.edata["e_type"], torch.full((s.n_own,), model._num_rel - 1, dtype=torch.long, device=dev)))
        self.p_rel = kernels.GatherPlan(rel_idx, model._num_rel)

    # -- pieces ---------------------------------------------------------------------------
    def _mixed(self, mixed_op, w, h, h_in, total_rows, addend=None):
        if (h.x if isinstance(h, K.Fan) else h).is_cuda:   # fused HIP epilogue with the statistics all-reduced in between
            return mixed_op(w, self.s, h, h_in, group=self._stat_group(), total_rows=total_rows, addend=addend)
        h, h_in = (t.x if isinstance(t, K.Fan) else t for t in (h, h_in))
        out = 0 if addend is None else addend
        for wk, (op, bn, act) in zip(w, mixed_op._ops):
            out = out + wk * act(sync_batch_norm(op(self.s, h, h_in).float(), bn, total_rows, self.group))
        return out

    def _mixed_sum(self, ops, ws, hs, h_in, total_rows):
        """sum_j MixedOp_j(hs[j], h_in) -- the MixedOps feeding one state (reference models/cell_lp.py:104-113).  On the GPU their
        BatchNorm statistics travel in ONE all-reduce forward and ONE backward (functional.StatChain) instead of one each."""
        first = hs[0].x if isinstance(hs[0], K.Fan) else hs[0]
        if not (first.is_cuda and BATCH_STATS and len(ops) > 1):
            sN = None
            for op, w, h in zip(ops, ws, hs):
                sN = self._mixed(op, w, h, h_in, total_rows, addend=sN)
            return sN
        preps = [op(w, self.s, h, h_in, group=self._stat_group(), total_rows=total_rows, prepare_only=True) for op, w, h in zip(ops, ws, hs)]
        K.StatChain(preps, self._stat_group(), summed=True)
        sN = None
        for prep, w in zip(preps, ws):
            sN = prep(w, sN)
        return sN

    def _partials(self, mixed_op, take):
        """The rank-local halves of a middle MixedOp (reference models/operations_lp.py:223-264): per candidate the
        partial [N, D] over the LOCAL in-edges and the residual self rows; a_sum's and a_mean's partials are summed by
        the same collective, so they are concatenated to one [N, 2D] tensor."""
        s, E = self.s, self.s.num_edges()
        out = {}
        for name, (op, _, _) in zip(OPS.MIDDLE_OPS, mixed_op._ops):
            x = take()
            hip = x.is_cuda and getattr(self.k, "linear_relu_partial", None) is not None
            if name == "a_sum":
                part, self_rows = self.k.sum_partial(x, s) if hip else (self.k.seg_reduce("sum", x[:E], None, s), x[E:])
            else:
                kind = "max" if name == "a_max" else "sum"
                if hip:
                    part, self_rows = self.k.linear_relu_partial(kind, x, op.linear.weight, op.linear.bias, s)
                else:
                    m = self.k.linear(x[:E], op.linear.weight, op.linear.bias, act="relu")
                    part, self_rows = self.k.seg_reduce(kind, m, None, s), x[E:]
            out[name] = (part, self_rows)
        return out

    def _exchange_start(self, parts):
        """Launch the two collectives of one middle MixedOp: reduce-scatter(max) of a_max's partial and ONE
        reduce-scatter(sum) of [a_sum | a_mean] (batch the aggregators of a MixedOp).  Asynchronous on RCCL's
        stream when the tensors are on the GPU: the caller goes on with independent work and calls _exchange_finish."""
        s = self.s
        async_ok = parts["a_max"][0].is_cuda
        ex_max, ex_sum = (Exchange(), Exchange()) if async_ok else (None, None)
        h_max = reduce_scatter_rows(parts["a_max"][0], s, "max", self.group, ex_max)
        both = torch.cat((parts["a_sum"][0], parts["a_mean"][0]), dim=1)
        h_both = reduce_scatter_rows(both, s, "sum", self.group, ex_sum)
        return h_max, h_both, (ex_max, ex_sum)

    def _exchange_finish(self, mixed_op, parts, started):
        h_max, h_both, exs = started
        for ex in exs:
            if ex is not None:
                ex.wait()                                  # the current stream waits for RCCL's; no host synchronisation
        D = h_max.shape[1]
        ys = {"a_max": h_max + parts["a_max"][1],
              "a_sum": mixed_op._ops[OPS.MIDDLE_OPS.index("a_sum")][0].drop_sum(h_both[:, :D]) + parts["a_sum"][1],
              "a_mean": h_both[:, D:] * self._inv_degree() + parts["a_mean"][1]}
        return [ys[name] for name in OPS.MIDDLE_OPS]

    def _inv_degree(self):
        """1 / max(in-degree, 1) of the own node rows over the WHOLE graph, [n_own, 1] (computed once)."""
        if getattr(self, "_inv_deg", None) is None:
            s = self.s
            self._inv_deg = (1.0 / s.global_in_degree[s.node_lo:s.node_hi].clamp(min=1).to(torch.float32)).view(-1, 1).contiguous()
        return self._inv_deg

    def _stat_group(self):
        return self.group if self.group is not None else dist.group.WORLD

    def _middle_stage(self, cell, wm, states, total_nodes):
        """Cell_Middle over a shard: the partials of every MixedOp are computed first and each exchange is launched as
        soon as its partials exist, so the collectives of MixedOp i overlap the edge-parallel GEMMs of MixedOp i + 1."""
        started = []
        for i in range(cell.n_first):
            h = states[i]
            take = h.take if isinstance(h, K.Fan) else (lambda h=h: h)
            parts = self._partials(cell.cell_middle._ops[i], take)
            started.append((parts, self._exchange_start(parts)))
        outs = []
        if started and started[0][0]["a_max"][0].is_cuda and BATCH_STATS and len(started) > 1:
            # the two MixedOps' statistics in ONE forward all-reduce (their outputs are separate states: the backward reductions
            # see different upstream gradients at different times and keep a collective each)
            preps = []
            for i, (parts, st) in enumerate(started):
                mixed_op = cell.cell_middle._ops[i]
                ys = self._exchange_finish(mixed_op, parts, st)
                preps.append(K.mixed_epilogue_prepare(ys, [bn for _, bn, _ in mixed_op._ops], self._stat_group(), total_nodes))
            K.StatChain(preps, self._stat_group(), summed=False)
            return [prep(wm[i]) for i, prep in enumerate(preps)]
        for i, (parts, st) in enumerate(started):
            mixed_op = cell.cell_middle._ops[i]
            ys = self._exchange_finish(mixed_op, parts, st)
            bns = [bn for _, bn, _ in mixed_op._ops]
            if ys[0].is_cuda:
                outs.append(K.mixed_epilogue(ys, bns, wm[i], self._stat_group(), total_nodes))
            else:
                out = 0
                for wk, y, (_, bn, act) in zip(wm[i], ys, mixed_op._ops):
                    out = out + wk * act(sync_batch_norm(y, bn, total_nodes, self.group))
                outs.append(out)
        return outs

    def _cell(self, cell, x, hr, wz, wf, wm, wl):
        M, N = self.rows_total, self.s.number_of_nodes()
        fan = cell._fan                                    # one K-way gradient sum per state (supernet.SuperCell._fan)
        h_in = fan(self._mixed(cell.cell_zero._ops[0], wz[0], x, hr, M))
        states, off = [h_in], 0
        for _ in range(cell.n_first):
            n = len(states)
            sN = self._mixed_sum(cell.cell_first._ops[off:off + n], wf[off:off + n], states, h_in, M)
            off += n
            states.append(fan(sN))
        states = [fan(y) for y in self._middle_stage(cell, wm, states[1:], N)]
        off = 0
        for _ in range(cell.n_last):
            n = len(states)
            sN = self._mixed_sum(cell.cell_last._ops[off:off + n], wl[off:off + n], states, h_in, N)
            off += n
            states.append(fan(sN))
        states = [t.take() if isinstance(t, K.Fan) else t for t in states]
        return K.module_linear(cell.concat_weights, torch.cat(states, dim=1))

    # -- the step ---------------------------------------------------------------------------
    def forward(self):
        """Returns (ent [N, D] gathered on every rank, rel [R', D])."""
        with K.deferred_counters():                        # the BatchNorm step counters of all MixedOps: one launch at the end
            return self._forward()

    def _forward(self):
        m, s = self.m, self.s
        ent_all = K.module_linear(m.linear_e, m.embedding_h.weight)
        rel = torch.mm(m.rel_wt, m.embedding_e.weight)
        N = s.number_of_nodes()
        ent = None
        weights = m.row_weights()
        for l, cell in enumerate(m.cells):
            wz, wf, wm, wl = weights[l]
            table, plan = (ent_all, self.p_ent0) if l == 0 else (ent, self.p_ent)
            if table.is_cuda and self.k is K:              # cell zero's compose candidates gather on the fly (supernet.SearchNetwork.forward)
                x, hr = K.LazyRows(table, plan), K.LazyRows(rel, self.p_rel)
            else:
                x, hr = self.k.gather(table, plan), self.k.gather(rel, self.p_rel)
            own = sync_batch_norm(self._cell(cell, x, hr, wz, wf, wm, wl), m.batchnorm_h, N, self.group)
            if l > 0 or m._layers == 1:
                own = F.relu(own)
            own = F.dropout(own, m._dropout, training=m.training)
            ent = _AllGatherRows.apply(own, (s.node_cuts, s.node_chunk, s.rank), self.group)
            rel = torch.matmul(rel, m.w_rel)
        return ent, rel

    def loss(self, ent, rel, samples_local, labels_local, total_samples):
        """This rank's share of the mean BCE (reference models/model_search_lp.py:181-188):
        the ranks' values add up to the reference loss."""
        if ent.is_cuda:
            plan = cached_on(self, "_score_cache", (samples_local,), (ent.shape[0], rel.shape[0]),
                             lambda: K.ScorePlan(samples_local, ent.shape[0], rel.shape[0]))
            score = K.distmult_score(ent, rel, plan)
        else:
            t = samples_local.long()
            score = torch.sum(ent[t[:, 0]] * rel[t[:, 1]] * ent[t[:, 2]], dim=1)
        return F.binary_cross_entropy_with_logits(score, labels_local, reduction="sum") / total_samples


# ---------------------------------------------------------------------------
# the sharded fixed-genotype step (reference models/model_lp.py:77-150; BASELINE C5: 10 M edges, 1 M nodes)
# ---------------------------------------------------------------------------
class ShardedFixedNet:
    """Forward of supernet.FixedNetwork on one relation block.  Same cell code (supernet.FixedCell.forward with its two hooks),
    rows partitioned like the supernet's, and -- what the supernet's replicated tables do not need at FB15k-237 size but the
    1 M-node table of C5 does -- the NODE TABLES ROW-SHARDED:

    * the initial table: every rank owns the rows [node_lo, node_hi) of embedding_h as its own leaf (`emb_own`, taken out of
      the replicated parameter list: no gradient exchange, 1/P of the optimiser state), projects them and all-gathers the
      [N, D] result once per step;
    * the last cell's output is NOT all-gathered: the [B, N] scorer (reference models/operations_lp.py:115-127) is
      evaluated on the own columns [B, n_own] against the labels' own columns, and the B subject rows travel in one [B, D]
      all-reduce (each rank contributes the subjects it owns).  At C5 that is 256 KB instead of a 1 GB all-gather.

    Per cell: an aggregator node = partial over the local edges -> reduce-scatter over node chunks (dist._ReduceScatterRows);
    every node's BatchNorm over ALL ranks' rows (statistics all-reduce inside the fused epilogue on the GPU, sync_batch_norm on
    CPU); between cells one all-gather of the [n_own, D] output.  `kernels`: as for ShardedSupernet."""

    def __init__(self, model, shard, group=None, kernels=K):
        self.m, self.s, self.group, self.k = model, shard, group, kernels
        s, dev = shard, shard.device
        src, _, _ = s.edges(form="all")
        own = torch.arange(s.node_lo, s.node_hi, device=dev)
        N = s.number_of_nodes()
        self.rows_total = s.E_global + N
        self.p_ent = kernels.GatherPlan(torch.cat((src, own)), N)
        rel_idx = torch.cat((s.edata["e_type"].long(), torch.full((s.n_own,), model._num_rel - 1, dtype=torch.long, device=dev)))
        self.p_rel = kernels.GatherPlan(rel_idx, model._num_rel)
        self.emb_own = torch.nn.Parameter(model.embedding_h.weight.detach()[s.node_lo:s.node_hi].clone())
        self._info = (s.node_cuts, s.node_chunk, s.rank)

    def replicated_parameters(self):
        """The parameters every rank holds in full (their gradients are all-reduced): all but the row-sharded initial table."""
        skip = self.m.embedding_h.weight
        return [p for p in self.m.parameters() if p is not skip]

    def parameters(self):
        return self.replicated_parameters() + [self.emb_own]

    def _stat_group(self):
        return self.group if self.group is not None else dist.group.WORLD

    def _inv_degree(self):
        if getattr(self, "_inv_deg", None) is None:
            s = self.s
            self._inv_deg = (1.0 / s.global_in_degree[s.node_lo:s.node_hi].clamp(min=1).to(torch.float32)).view(-1, 1).contiguous()
        return self._inv_deg

    def _bn_relu(self, y, bn, total, one):
        if y.is_cuda:
            one = one if one.device == y.device else one.to(y.device)
            return K.mixed_epilogue([y], [bn], one, self._stat_group(), total)
        return F.relu(sync_batch_norm(y.float(), bn, total, self.group))

    def _aggregate(self, op, name, x):
        """a_max / a_sum / a_mean over a relation block: partial over the LOCAL in-edges, reduce-scatter over the node chunks,
        residual self rows (reference models/operations_lp.py:223-264)."""
        s, E = self.s, self.s.num_edges()
        hip = x.is_cuda and getattr(self.k, "linear_relu_partial", None) is not None
        kind = "max" if name == "a_max" else "sum"
        if name == "a_sum":
            part, self_rows = self.k.sum_partial(x, s) if hip else (self.k.seg_reduce("sum", x[:E], None, s), x[E:])
        elif hip:
            part, self_rows = self.k.linear_relu_partial(kind, x, op.linear.weight, op.linear.bias, s)
        else:
            part, self_rows = self.k.seg_reduce(kind, self.k.linear(x[:E], op.linear.weight, op.linear.bias, act="relu"), None, s), x[E:]
        own = reduce_scatter_rows(part, s, kind, self.group)
        if name == "a_sum":
            own = op.drop_sum(own)
        elif name == "a_mean":
            own = own * self._inv_degree()
        return own + self_rows

    def _node_rows(self, mod):
        """Does this op module's output have one row per NODE (True) or per edge-and-self row (False)?  f_zero / f_identity are
        in FIRST_OPS and LAST_OPS alike: they take the kind of the other ops feeding the same cell node, else of their input."""
        kinds = getattr(self, "_kinds", None)
        if kinds is None:
            kinds = self._kinds = {}
            node_ops = set(OPS.MIDDLE_OPS) | (set(OPS.LAST_OPS) - set(OPS.FIRST_OPS))
            edge_ops = set(OPS.PRE_OPS) | (set(OPS.FIRST_OPS) - set(OPS.LAST_OPS))
            for cell in self.m.cells:
                state = [False, False]                       # state 0 (gathered rows) and the zero node's output: edge rows
                kinds[id(cell._ops[0][0][0])] = False
                for n in range(1, cell._nb):
                    mods = [(i, cell._ops[n][i][0]) for i in range(n + 1) if len(cell._ops[n][i])]
                    names = {m_.op_name for _, m_ in mods}
                    kind = True if names & node_ops else False if names & edge_ops else state[mods[0][0]]
                    for _, m_ in mods:
                        kinds[id(m_)] = kind
                    state.append(kind)
        return kinds[id(mod)]

    def _apply(self, mod, h, h_in):
        """One node of the fixed cell (supernet.OpModule: op -> BN -> ReLU, 'pre_mult' bare) on this rank's rows."""
        name = mod.op_name
        N = self.s.number_of_nodes()
        if isinstance(h, K.LazyRows):                        # the zero node on un-materialised gathers (supernet.OpModule.forward)
            if name != "pre_mult" and K.switches.CELL_ZERO_FUSED and isinstance(mod.op, OPS._PreOp):
                one = mod._one if mod._one.device == h.device else mod._one.to(h.device)
                return K.cell_zero_mixed([mod.op.kind], h, h_in, [mod.batchnorm_h], one, self._stat_group(), self.rows_total)
            h, h_in = h.materialize(), h_in.materialize()
        if name in OPS.MIDDLE_OPS:
            y, total = self._aggregate(mod.op, name, h), N
        else:
            from . import cell_lp as _cell_lp
            y = _cell_lp._run(mod.op, self.s, h, h_in) if h.is_cuda else mod.op(self.s, h, h_in)
            total = N if self._node_rows(mod) else self.rows_total
        if name == "pre_mult":
            return y
        return self._bn_relu(y, mod.batchnorm_h, total, mod._one)

    def forward(self, subj, rel):
        """Returns pred_own [B, n_own]: this rank's columns of the reference's [B, N] prediction."""
        with K.deferred_counters():
            return self._forward(subj, rel)

    def _forward(self, subj, rel):
        m, s = self.m, self.s
        N = s.number_of_nodes()
        own = K.module_linear(m.linear_e, self.emb_own)                      # [n_own, D]: the own rows of the projected table
        ent = _AllGatherRows.apply(own, self._info, self.group)
        rel_emb = torch.mm(m.rel_wt, m.embedding_e.weight)
        for ci, cell in enumerate(m.cells):
            if ent.is_cuda and self.k is K:
                x, hr = K.LazyRows(ent, self.p_ent), K.LazyRows(rel_emb, self.p_rel)
            else:
                x, hr = self.k.gather(ent, self.p_ent), self.k.gather(rel_emb, self.p_rel)
            own = cell(s, x, hr, apply=self._apply, finish=lambda h, c=cell: self._bn_relu(h, c.batchnorm_h, N, c._one))
            own = F.dropout(own, m._dropout, training=m.training)
            if ci + 1 < len(m.cells):
                ent = _AllGatherRows.apply(own, self._info, self.group)
            rel_emb = torch.matmul(rel_emb, m.w_rel)
        # the B subject rows: each rank contributes the subjects it owns, one [B, D] all-reduce
        subj = subj.view(-1).long()
        mine = ((subj >= s.node_lo) & (subj < s.node_hi)).view(-1, 1).to(own.dtype)
        local = own[(subj - s.node_lo).clamp(0, max(s.n_own - 1, 0))] * mine if s.n_own else own.new_zeros(subj.numel(), own.shape[1])
        sub = _AllReduceSum.apply(local, self.group, True)
        r = rel_emb[rel.view(-1).long()]
        if not own.is_cuda and hasattr(self.k, "score_all"):                  # the gloo tests' stand-in (tests/cpu_kernels.py)
            return self.k.score_all(m.score_func, own, sub, r)
        return m.score_func(own, sub, r)

    def loss(self, pred_own, label):
        """This rank's share of nn.BCELoss over [B, N] (reference train/mr_lp_train.py:139-141): the ranks' values add up."""
        s = self.s
        B, N = label.shape
        return F.binary_cross_entropy(pred_own, label[:, s.node_lo:s.node_hi], reduction="sum") / (B * N)


def all_reduce_gradients(tensors, group=None):
    """One flat all-reduce (sum) over the gradients of `tensors` (parameters and alphas);
    tensors that received no gradient on this rank contribute zeros."""
    tensors = [t for t in tensors if t.requires_grad]
    flat = torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in tensors])
    all_reduce(flat, dist.ReduceOp.SUM, group)
    off = 0
    for t in tensors:                                   # views of the one reduced buffer: no per-parameter copy
        n = t.numel()
        t.grad = flat[off:off + n].view_as(t)
        off += n


class ShardedStep:
    """bench.py's step on N GPUs: same graph, same model, same optimiser as the single-GPU
    Step; edges in relation blocks, one process per GPU."""

    def __init__(self, args, device, inputs, rank, world, group=None):
        from . import graph as G, supernet as S
        N, R, node_id, gtri, samples, labels = inputs
        torch.manual_seed(args.seed)                      # identical parameters on every rank
        g = G.build_search_graph(len(node_id), R, gtri)
        if torch.device(device).type == "cuda":
            g = g.to(device)                              # the partition then runs on the device too (EdgeShard._init_on_device)
        src, dst, _ = g.edges(form="all")
        self.g = EdgeShard(len(node_id), src, dst, g.edata["e_type"], g.edata["norm"], R, rank, world, device)
        self.E, self.E_global = self.g.num_edges(), self.g.E_global
        self.model = S.SearchNetwork(device, N, R, 2, 1, 2, 2, args.dim, 100, 2 * R + 1, 40.0, 0.3, 0.1).to(device)
        S.xavier_init_(self.model)
        self.model.train()
        self.net = ShardedSupernet(self.model, self.g, torch.from_numpy(node_id), group)
        lo = node_ranges(len(samples), world)
        self.samples = torch.from_numpy(samples[lo[rank]:lo[rank + 1]]).to(device)
        self.labels = torch.from_numpy(labels[lo[rank]:lo[rank + 1]]).to(device)
        self.total_samples = len(samples)
        self.params, self.arch = list(self.model.parameters()), list(self.model.arch_parameters())
        self.clip, self.group, self.last_loss = 5.0, group, None
        self.fused_opt = os.environ.get("MRG_TORCH_OPTIM", "0") != "1" and torch.device(device).type == "cuda"
        if self.fused_opt:                                  # clip + SGD(momentum) in three launches (optim.ClippedSGD)
            from .optim import ClippedSGD
            self.opt = ClippedSGD(self.params, 1e-3, momentum=0.9, weight_decay=0.0, max_norm=self.clip)
        else:
            self.opt = torch.optim.SGD(self.params, 1e-3, momentum=0.9, weight_decay=0.0)
        torch.manual_seed(args.seed + 1000 + rank)         # dropout masks differ per rank (disjoint rows)

    def __call__(self):
        ent, rel = self.net.forward()
        loss = self.net.loss(ent, rel, self.samples, self.labels, self.total_samples)
        loss.backward()
        all_reduce_gradients(self.params + self.arch[:4], self.group)
        if not self.fused_opt:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        for a in self.arch:
            a.grad = None
        total = loss.detach().clone()
        all_reduce(total, dist.ReduceOp.SUM, self.group)
        self.last_loss = total


class ShardedFixedStep:
    """bench.py's fixed-genotype step (C1 / C5 shape) on N GPUs: README genotype, Adam, labels built on the device; edges in
    relation blocks, node tables row-sharded (ShardedFixedNet)."""

    def __init__(self, args, device, shape, rank, world, group=None, dim=256, init_dim=100, nbase=100, batch=256):
        from . import graph as G, sampler as SM, supernet as S, synth
        N, R, T = synth.SHAPES[shape]
        tri = synth.synth_kg(N, R, T, args.seed)
        torch.manual_seed(args.seed)                      # identical parameters on every rank
        g = G.build_train_graph(N, R, tri, device=device)
        src, dst, _ = g.edges(form="all")
        self.g = EdgeShard(N, src, dst, g.edata["e_type"], g.edata["norm"], R, rank, world, device)
        del g
        self.E, self.E_global = self.g.num_edges(), self.g.E_global
        geno = [S.Genotype(alpha_cell=[('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2), ('a_max', 4, 2), ('a_max', 5, 3),
                                       ('f_sparse_last', 6, 5), ('f_sparse_last', 7, 5)], concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]
        self.model = S.FixedNetwork(device, geno, N, R, dim, init_dim, nbase, dropout_cell=0.3, drop_aggr=0.1).to(device)
        S.xavier_init_(self.model)
        self.model.train()
        self.net = ShardedFixedNet(self.model, self.g, group)
        self.replicated = self.net.replicated_parameters()
        self.opt = torch.optim.Adam(self.net.parameters(), 1e-3, capturable=True)
        self.idx = SM.LabelIndex(tri, R, N, device)
        rng = np.random.default_rng(args.seed + 3)
        pick = rng.integers(0, T, batch)
        self.subj = torch.from_numpy(tri[pick, 0]).to(device)
        self.rel = torch.from_numpy(tri[pick, 1]).to(device)
        self.samples = self.subj
        self.group, self.last_loss = group, None
        torch.manual_seed(args.seed + 1000 + rank)         # dropout masks differ per rank (disjoint rows)

    def __call__(self):
        labels = self.idx.labels(self.subj, self.rel, 0.1)
        pred = self.net.forward(self.subj, self.rel)
        loss = self.net.loss(pred, labels)
        loss.backward()
        all_reduce_gradients(self.replicated, self.group)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        total = loss.detach().clone()
        all_reduce(total, dist.ReduceOp.SUM, self.group)
        self.last_loss = total
