"""Drop-in for the reference's ``models/cell_lp.py``: the same class names, constructor arguments, forward signatures
and module / parameter names (so the reference's ``state_dict`` keys load unchanged), built on this package's operator
registry and -- when handed CUDA tensors -- on the fused HIP MixedOp path.

    reference                                              here
    MixedOp(feature_dim, drop_aggr, operations)            same      (models/cell_lp.py:12-33)
    MixedOp_SF(gamma, operations)                          same      (:36-50)
    Cell_Zero(nodes, feature_dim, drop_aggr)               same      (:53-68)    forward(g, h, hr, weights)
    Cell_Final(gamma) / Cell_SF(gamma)                     same      (:71-86, :191-200)
    Cell_First(nodes, feature_dim, drop_aggr)              same      (:89-109)   forward(g, states, h_in, weights)
    Cell_Middle(nodes, feature_dim, drop_aggr)             same      (:112-127)
    Cell_Last(in_nodes, nodes, feature_dim, drop_aggr)     same      (:130-152)
    Cell(nb_zero_nodes, nb_first_nodes, nb_last_nodes, feature_dim, dropout_aggr)        (:155-188)
        forward(g, src_emb, hr, weights_zero, weights_first, weights_middle, weights_last)

A maintainer swaps ``models.cell_lp`` for this module next to ``models.operations_lp``: the
reference's ``model_search_lp.Network`` then calls ``Cell`` with the plain ``[M, D]`` tensors it gathered
(models/model_search_lp.py:144-145,153-154) and gets the fused epilogue, the paired dense filters, the gate-only / row-factor
candidates and the K-way gradient fan-in sums; what it does not get is the un-materialised gather of cell zero (that needs
``functional.LazyRows`` operands, which only ``supernet.SearchNetwork`` hands over).

``CALLER`` (environment ``MRG_CALLER``): ``"fused"`` (default) or ``"reference"`` -- the literal formulation of
models/cell_lp.py:25-33 on the HIP operators (one operator call, ``nn.BatchNorm1d``, ``ReLU`` and a scaled add per candidate,
Python ``sum`` over candidates and over the MixedOps feeding a state).  ``bench.py --caller reference`` times it: it is what
the reference's UNCHANGED ``cell_lp`` gets from the operator swap alone.
"""
import os

import torch
import torch.nn as nn

from . import functional as K
from . import lazy as LZ
from . import operations_lp as OPS
from .operations_lp import PRE_OPS, FIRST_OPS, MIDDLE_OPS, LAST_OPS, MIXED_OPS, MIXED_OPS_sf, SF_OPS   # noqa: F401  (the reference module re-exports them)

CALLER = os.environ.get("MRG_CALLER", "fused")

# HIP streams the candidates of a MixedOp are spread over.  Rounds 1-2 ran them on four: the one-wave row GEMM owns a CU's whole
# register file, and other candidates' kernels filled its tails (71.4 -> 69.9 ms/step).  With the LDS-weight row GEMM (two
# workgroups per CU) the step is the same on one stream as on four (62.3 / 62.8 vs 62.3 / 62.8 ms, profiles/r3_streams.txt), so the
# default is ONE: no event traffic, no cross-stream allocator bookkeeping; 2-4 remain available.
MIXED_STREAMS = int(os.environ.get("MRG_MIXED_STREAMS", "1"))


def _literal(h):
    """True when the reference's own formulation has to run: CPU tensors (test registries), or CALLER == 'reference'."""
    x = h.x if isinstance(h, K.Fan) else h
    return CALLER == "reference" or not (x.table if isinstance(x, K.LazyRows) else x).is_cuda


def _tensor(y):
    """The autograd tensor of a candidate (a functional.Candidate wraps it)."""
    return y.y if isinstance(y, K.Candidate) else y


def _plain(h):
    if isinstance(h, K.Fan):
        h = h.take()
    return h.materialize() if isinstance(h, K.LazyRows) else h


def _identity_index(ops):
    ids = [k for k, op in enumerate(ops) if type(op) is OPS.f_identity_op]
    return ids[0] if len(ids) == 1 else None


def _dense_pair(ops, x):
    """(index of f_dense_comp, index of f_comp) when both are candidates of this MixedOp and may share a node, else None."""
    if not x.is_cuda:
        return None
    d = [k for k, op in enumerate(ops) if type(op) is OPS.f_dense_op_comp]
    c = [k for k, op in enumerate(ops) if type(op) is OPS.f_comp_op]
    if len(d) != 1 or len(c) != 1:
        return None
    return (d[0], c[0])


def _run(op, g, a, b, for_epilogue=False):
    """op's result NOW (a tensor, or a functional.Candidate with for_epilogue): `op.run`, or -- when somebody registered hooks on the
    operator module -- the module call itself, so that the hooks fire (its lazy handle is evaluated at once)."""
    if op._forward_hooks or op._forward_pre_hooks or not hasattr(op, "run"):
        return LZ.real(op(g, a, b, for_epilogue=True) if for_epilogue else op(g, a, b))
    return op.run(g, a, b, for_epilogue=for_epilogue)


def fused_candidates(ops, bns, weights, g, h, h_in, addend=None, group=None, total_rows=None, prepare_only=False):
    """addend + sum_k weights[k] * ReLU(BatchNorm_k(ops[k](g, h, h_in)))  on the fused HIP path: ONE epilogue (statistics pass +
    combine pass) for all candidates instead of BN / ReLU / scale / add launches per candidate; f_zero contributes
    w * ReLU(beta) without materialising its all-zero output; f_dense_comp and f_comp share one autograd node; f_dense_comp arrives
    as its gate and f_sparse_comp as a row factor (recomputed by the epilogue); the operands' gradients are K-way sums.

    ops: the operator modules; bns: their nn.BatchNorm1d modules (or lazy.BatchNormView); weights [K]; h / h_in: tensors,
    functional.Fan (a state with its reader bookkeeping) or functional.LazyRows (cell zero).  `group` / `total_rows`: rows sharded
    over ranks; `addend`: the sum of the MixedOps that feed the same state so far (accumulated inside the combine kernel).
    Called by MixedOp.forward below and by the lazy handles (lazy.py) that the reference's own MixedOp produces."""
    if isinstance(h, K.LazyRows):                           # cell zero: the compose candidates gather on the fly
        if (K.switches.CELL_ZERO_FUSED and isinstance(h_in, K.LazyRows) and addend is None and len(ops) <= 3
                and all(isinstance(op, OPS._PreOp) for op in ops)):
            # ... and are never stored: statistics, combine and gradients recompute them from the two tables
            return K.cell_zero_mixed([op.kind for op in ops], h, h_in, bns, weights, group, total_rows)
        ys = [_run(op, g, h, h_in) for op in ops]
        return K.mixed_epilogue(ys, bns, weights, group, total_rows, addend, fold_row_scales=True)
    # every candidate reads h (and most read h_in): hand out aliases whose gradients are summed in
    # one K-way pass; a caller that already tracks the readers of a state passes its Fan.
    n = len(ops)
    fh = h if isinstance(h, K.Fan) else K.Fan(h, n)
    fi = h_in if isinstance(h_in, K.Fan) else K.Fan(h_in, n)
    # f_dense_comp and f_comp read the same (h, h_in): one autograd node whose backward leaves ONE gradient per operand
    pair = _dense_pair(ops, fh.x)
    paired = {}
    # f_sparse_comp as a row factor: only next to the gate-only f_dense_comp, whose folded gradient store receives its gradient
    row_ok = pair is not None and K.switches.GATED_RECOMPUTE and K.switches.FOLD_ROW_SCALE
    # The candidates are independent: they may run round-robin on a few HIP streams so that the tail of one
    # kernel is filled by another candidate's kernels.  Autograd replays each candidate's backward on the stream its
    # forward ran on.  (Launch-bound step graphs gain nothing from it and pay the event traffic: one stream below 128k rows.)
    dev = fh.x.device
    nstreams = min(MIXED_STREAMS, n) if fh.x.shape[0] >= K.switches.FORK_MIN_ROWS else 1
    if nstreams <= 1:
        ys = []
        for k, op in enumerate(ops):
            if isinstance(op, OPS.f_zero_op):
                ys.append(None)
            elif pair is not None and k in pair:
                if not paired:
                    paired[pair[0]], paired[pair[1]] = OPS.dense_pair_forward(ops[pair[0]], ops[pair[1]], g, fh.take(), fi.take(), for_epilogue=True)
                ys.append(paired[k])
            elif (row_ok and type(op) is OPS.f_sparse_op_comp) or type(op) in (OPS.f_dense_op_comp, OPS.f_comp_op):
                ys.append(_run(op, g, fh.take(), fi.take(), for_epilogue=True))      # a functional.Candidate: consumed by the epilogue only
            else:
                ys.append(_run(op, g, fh.take(), fi.take()))
        prep = K.mixed_epilogue_prepare(ys, bns, group, total_rows, True, _identity_index(ops))
        # prepare_only (dist.py): the caller issues the statistics collective of several MixedOps at once (functional.StatChain)
        return prep if prepare_only else prep(weights, addend)
    fork = K.Fork(dev, nstreams, tag="candidates")
    ys = []
    for k, op in enumerate(ops):
        if isinstance(op, OPS.f_zero_op):
            ys.append(None)
            continue
        if pair is not None and k in pair and paired:
            ys.append(paired[k])                   # computed with its partner
            continue
        side = fork.stream(k)
        a, b = fh.take(), fi.take()
        if side is not fork.main:                  # h / h_in live in main-stream blocks and are read (forward and,
            a.record_stream(side)                  # through the saved tensors, backward) on the side stream: the
            b.record_stream(side)                  # allocator must not recycle them before that stream is done
        with torch.cuda.stream(side):
            if pair is not None and k in pair:
                paired[pair[0]], paired[pair[1]] = OPS.dense_pair_forward(ops[pair[0]], ops[pair[1]], g, a, b, for_epilogue=True)
                y = paired[k]
                _tensor(paired[pair[0] + pair[1] - k]).record_stream(fork.main)
            elif (row_ok and type(op) is OPS.f_sparse_op_comp) or type(op) in (OPS.f_dense_op_comp, OPS.f_comp_op):
                y = _run(op, g, a, b, for_epilogue=True)
            else:
                y = _run(op, g, a, b)
        _tensor(y).record_stream(fork.main)        # consumed by the epilogue on the main stream
        ys.append(y)
    fork.join()
    prep = K.mixed_epilogue_prepare(ys, bns, group, total_rows, True, _identity_index(ops))
    return prep if prepare_only else prep(weights, addend)


class MixedOp(nn.Module):
    """sum_k w_k * ReLU(BN_k(op_k(g, h, h_in)))   (reference models/cell_lp.py:12-33)."""

    def __init__(self, feature_dim, drop_aggr, operations, registry=None):
        super().__init__()
        registry = MIXED_OPS if registry is None else registry
        self._feature_dim, self._operations, self._drop_aggr = feature_dim, operations, drop_aggr
        self._args = {'feature_dim': feature_dim, 'drop_aggr': drop_aggr}
        self._ops = nn.ModuleList(nn.ModuleList([registry[name](self._args), nn.BatchNorm1d(feature_dim), nn.ReLU()])
                                  for name in operations)

    def op_forward(self, op, g, h, h_in):
        nh = op[0](g, h, h_in)
        for i in range(1, len(op)):
            nh = op[i](nh.float())
        return nh

    def forward(self, weights, g, h, h_in, group=None, total_rows=None, addend=None, prepare_only=False):
        """CALLER == "reference" (or CPU tensors): the reference's own lines.  Otherwise fused_candidates above."""
        if _literal(h):
            h, h_in = _plain(h), _plain(h_in)
            total = sum(w * self.op_forward(op, g, h, h_in) for w, op in zip(weights, self._ops))
            return total if addend is None else addend + total
        return fused_candidates([op for op, _, _ in self._ops], [bn for _, bn, _ in self._ops], weights, g, h, h_in, addend, group,
                                total_rows, prepare_only)

    def _identity_index(self):
        return _identity_index([op for op, _, _ in self._ops])

    def _dense_pair(self, x):
        return _dense_pair([op for op, _, _ in self._ops], x)


class MixedOp_SF(nn.Module):
    """sum_k w_k * score_k(all_ent, sub_emb, rel_emb)   (reference models/cell_lp.py:36-50)."""

    def __init__(self, gamma, operations, registry=None):
        super().__init__()
        registry = MIXED_OPS_sf if registry is None else registry
        self._operations, self.gamma = operations, gamma
        self._args = {'gamma': gamma}
        self._ops = nn.ModuleList(nn.ModuleList([registry[name](self._args)]) for name in operations)

    def op_forward(self, op, g, h, h_in):
        return op[0](g, h, h_in)

    def forward(self, weights, g, h, h_in):
        return sum(w * self.op_forward(op, g, h, h_in) for w, op in zip(weights, self._ops))


def _dense_stage(mixed_ops, states, h_in, weights, g, steps, fan):
    """The MixedOps feeding one new state each add onto the previous one's output inside the combine kernel (`addend`);
    the reference adds full-size tensors (models/cell_lp.py:103-107, :146-150).  Appends to `states` as the reference does."""
    off = 0
    for _ in range(steps):
        s = None
        for j, h in enumerate(states):
            s = mixed_ops[off + j](weights[off + j], g, h, h_in, addend=s)
        off += len(states)
        states.append(fan(s))
    return states


def _no_fan(x):
    return x


class Cell_Zero(nn.Module):
    def __init__(self, nodes, feature_dim, drop_aggr, registry=None):
        super().__init__()
        self._feature_dim = feature_dim
        self._ops = nn.ModuleList([MixedOp(feature_dim, drop_aggr, PRE_OPS, registry)])

    def forward(self, g, h, hr, weights):
        return self._ops[0](weights[0], g, h, hr)


class Cell_Final(nn.Module):
    def __init__(self, gamma):
        super().__init__()
        self._ops = nn.ModuleList([MixedOp_SF(gamma, SF_OPS)])

    def forward(self, all_ent, sub_emb, rel_emb, weights):
        return self._ops[0](weights[0], all_ent, sub_emb, rel_emb)


class Cell_First(nn.Module):
    def __init__(self, nodes, feature_dim, drop_aggr, registry=None):
        super().__init__()
        self._nodes, self._feature_dim = nodes, feature_dim
        self._ops = nn.ModuleList(MixedOp(feature_dim, drop_aggr, FIRST_OPS, registry) for i in range(nodes) for _ in range(i + 1))

    def forward(self, g, states, h_in, weights, fan=_no_fan):
        return _dense_stage(self._ops, states, h_in, weights, g, self._nodes, fan)[1:]


class Cell_Middle(nn.Module):
    def __init__(self, nodes, feature_dim, drop_aggr, registry=None):
        super().__init__()
        self._nodes, self._feature_dim = nodes, feature_dim
        self._ops = nn.ModuleList(MixedOp(feature_dim, drop_aggr, MIDDLE_OPS, registry) for _ in range(nodes))

    def forward(self, g, states, h_in, weights, fan=_no_fan):
        return [fan(self._ops[i](weights[i], g, states[i], h_in)) for i in range(self._nodes)]


class Cell_Last(nn.Module):
    def __init__(self, in_nodes, nodes, feature_dim, drop_aggr, registry=None):
        super().__init__()
        self._in_nodes, self._nodes, self._feature_dim = in_nodes, nodes, feature_dim
        self._ops = nn.ModuleList(MixedOp(feature_dim, drop_aggr, LAST_OPS, registry) for i in range(nodes) for _ in range(i + in_nodes))

    def forward(self, g, states, h_in, weights, fan=_no_fan):
        return _dense_stage(self._ops, states, h_in, weights, g, self._nodes, fan)


class Cell(nn.Module):
    """Zero -> First -> Middle -> Last stages and the concat linear (reference models/cell_lp.py:155-188)."""

    def __init__(self, nb_zero_nodes, nb_first_nodes, nb_last_nodes, feature_dim, dropout_aggr, registry=None):
        super().__init__()
        self._nb_zero_nodes, self._nb_first_nodes, self._nb_last_nodes = nb_zero_nodes, nb_first_nodes, nb_last_nodes
        self._feature_dim = feature_dim
        self.n_first, self.n_last = nb_first_nodes, nb_last_nodes            # what dist.ShardedSupernet reads
        self.cell_zero = Cell_Zero(nb_zero_nodes, feature_dim, dropout_aggr, registry)
        self.cell_first = Cell_First(nb_first_nodes, feature_dim, dropout_aggr, registry)
        self.cell_middle = Cell_Middle(nb_first_nodes, feature_dim, dropout_aggr, registry)
        self.cell_last = Cell_Last(nb_first_nodes, nb_last_nodes, feature_dim, dropout_aggr, registry)
        self.concat_weights = nn.Linear((nb_first_nodes + nb_last_nodes) * feature_dim, feature_dim)

    def _fan(self, x):
        """Reader bookkeeping for one state: every candidate of every MixedOp of the cell may read it."""
        if CALLER == "reference" or isinstance(x, K.Fan) or not x.is_cuda:
            return x
        n_mixed = 1 + len(self.cell_first._ops) + len(self.cell_middle._ops) + len(self.cell_last._ops)
        width = max(len(m._ops) for st in (self.cell_first, self.cell_middle, self.cell_last) for m in st._ops)
        return K.Fan(x, 2 * n_mixed * width + 2)

    def forward(self, g, src_emb, hr, weights_zero, weights_first, weights_middle, weights_last):
        h_in = self._fan(self.cell_zero(g, src_emb, hr, weights_zero))
        states = self.cell_first(g, [h_in], h_in, weights_first, self._fan)
        states = self.cell_middle(g, states, h_in, weights_middle, self._fan)
        states = self.cell_last(g, states, h_in, weights_last, self._fan)
        states = [s.take() if isinstance(s, K.Fan) else s for s in states]
        cat = torch.cat(states, dim=1)
        return self.concat_weights(cat) if CALLER == "reference" else K.module_linear(self.concat_weights, cat)


class Cell_SF(nn.Module):
    def __init__(self, gamma):
        super().__init__()
        self.cell_score = Cell_Final(gamma)

    def forward(self, all_ent_emb, sub_emb, rel_emb, weights_sf):
        return self.cell_score(all_ent_emb, sub_emb, rel_emb, weights_sf)
