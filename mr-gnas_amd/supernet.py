"""Callers of the hot path: the DARTS-style mixed-op supernet and the fixed-genotype
network, built on this package's operator registry.

They exist so the benchmark and the parity tests can run a full MR-GNAS step on
the GPU box (the reference's own ``models/cell_lp.py`` / ``models/model_lp.py`` /
``models/model_search_lp.py`` work unchanged on ``mr_gnas_amd.operations_lp`` but
cannot travel).  Module/attribute names follow the reference so that its
``state_dict`` keys load here unchanged:

* supernet   ``cells.{l}.cell_{zero,first,middle,last}._ops.{j}._ops.{k}.{0|1}.*``,
             ``cells.{l}.concat_weights.*``, ``batchnorm_h.*``, ``embedding_h``,
             ``embedding_e``, ``linear_e``, ``rel_wt``, ``w_rel``
             (reference models/cell_lp.py:12-188, models/model_search_lp.py:16-163)
* fixed      ``cells.{l}._ops.{center}.{pre}.0.{op|batchnorm_h}.*``, ``cells.{l}.concat.*``,
             ``cells.{l}.batchnorm_h.*``  (reference models/model_lp.py:13-137)
"""
import collections

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as K
from . import operations_lp as OPS
from .graph import cached_on

Genotype = collections.namedtuple('Genotype', 'alpha_cell concat_node score_func')   # reference configs/genotypes.py:3


def xavier_init_(module):
    """What the reference drivers apply after construction (utils/utils.py:121-125)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
    return module


def _tsum(tensors):
    """Sum of tensors without Python's `0 + first` (which is a full-size elementwise kernel and a copy)."""
    it = iter(tensors)
    total = next(it)
    for t in it:
        total = total + t
    return total


def _xavier_param(*shape):
    p = nn.Parameter(torch.empty(*shape))
    nn.init.xavier_normal_(p, gain=nn.init.calculate_gain('relu'))
    return p


# ---------------------------------------------------------------------------
# supernet
# ---------------------------------------------------------------------------
# HIP streams the candidates of a MixedOp are spread over.  Rounds 1-2 ran them on four: the one-wave row GEMM owns a CU's whole
# register file, and other candidates' kernels filled its tails (71.4 -> 69.9 ms/step).  With the LDS-weight row GEMM (two
# workgroups per CU) the step is the same on one stream as on four (62.3 / 62.8 vs 62.3 / 62.8 ms, profiles/r3_streams.txt), so the
# default is ONE: no event traffic, no cross-stream allocator bookkeeping; 2-4 remain available.
MIXED_STREAMS = int(os.environ.get("MRG_MIXED_STREAMS", "1"))


class MixedOp(nn.Module):
    """sum_k w_k * ReLU(BN_k(op_k(g, h, h_in)))   (reference models/cell_lp.py:12-33)."""

    def __init__(self, feature_dim, drop_aggr, operations, registry=OPS.MIXED_OPS):
        super().__init__()
        args = {'feature_dim': feature_dim, 'drop_aggr': drop_aggr}
        self._ops = nn.ModuleList(nn.ModuleList([registry[name](args), nn.BatchNorm1d(feature_dim), nn.ReLU()])
                                  for name in operations)

    def forward(self, weights, g, h, h_in, group=None, total_rows=None, addend=None, prepare_only=False):
        """One fused HIP epilogue for all branches (statistics pass + combine pass) instead of
        BN / ReLU / scale / add launches per branch.  f_zero contributes w * ReLU(beta) without
        materialising its all-zero output.  `group`/`total_rows`: rows sharded over ranks."""
        if not (h.x if isinstance(h, K.Fan) else h).is_cuda:    # reference formulation (registry of non-HIP test operators)
            if isinstance(h, K.LazyRows):
                h, h_in = h.materialize(), h_in.materialize()
            total = 0 if addend is None else addend
            for w, (op, bn, act) in zip(weights, self._ops):
                total = total + w * act(bn(op(g, h, h_in).float()))
            return total
        if isinstance(h, K.LazyRows):                           # cell zero: the compose candidates gather on the fly
            if (K.CELL_ZERO_FUSED and isinstance(h_in, K.LazyRows) and addend is None and len(self._ops) <= 3
                    and all(isinstance(op, OPS._PreOp) for op, _, _ in self._ops)):
                # ... and are never stored: statistics, combine and gradients recompute them from the two tables
                return K.cell_zero_mixed([op.kind for op, _, _ in self._ops], h, h_in, [bn for _, bn, _ in self._ops], weights,
                                         group, total_rows)
            ys = [op(g, h, h_in) for op, _, _ in self._ops]
            return K.mixed_epilogue(ys, [bn for _, bn, _ in self._ops], weights, group, total_rows, addend, fold_row_scales=True)
        # every candidate reads h (and most read h_in): hand out aliases whose gradients are summed in
        # one K-way pass; a caller that already tracks the readers of a state passes its Fan.
        n = len(self._ops)
        fh = h if isinstance(h, K.Fan) else K.Fan(h, n)
        fi = h_in if isinstance(h_in, K.Fan) else K.Fan(h_in, n)
        # f_dense_comp and f_comp read the same (h, h_in): one autograd node whose backward leaves ONE gradient per operand
        pair = self._dense_pair(fh.x)
        paired = {}
        # f_sparse_comp as a row factor: only next to the gate-only f_dense_comp, whose folded gradient store receives its gradient
        row_ok = pair is not None and K.GATED_RECOMPUTE and K.FOLD_ROW_SCALE
        # The candidates are independent: they run round-robin on a few HIP streams so that the tail of one
        # kernel (a GEMM workgroup owns a whole CU) is filled by another candidate's kernels.  Autograd replays
        # each candidate's backward on the stream its forward ran on.
        dev = fh.x.device
        # (launch-bound step graphs gain nothing from it and pay the event traffic: one stream below 128k rows)
        nstreams = min(MIXED_STREAMS, n) if fh.x.shape[0] >= K.FORK_MIN_ROWS else 1
        if nstreams <= 1:
            ys = []
            for k, (op, _, _) in enumerate(self._ops):
                if isinstance(op, OPS.f_zero_op):
                    ys.append(None)
                elif pair is not None and k in pair:
                    if not paired:
                        paired[pair[0]], paired[pair[1]] = OPS.dense_pair_forward(self._ops[pair[0]][0], self._ops[pair[1]][0], g, fh.take(), fi.take(), for_epilogue=True)
                    ys.append(paired[k])
                elif row_ok and type(op) is OPS.f_sparse_op_comp:
                    ys.append(op(g, fh.take(), fi.take(), for_epilogue=True))
                else:
                    ys.append(op(g, fh.take(), fi.take()))
            prep = K.mixed_epilogue_prepare(ys, [bn for _, bn, _ in self._ops], group, total_rows, True, self._identity_index())
            # prepare_only (dist.py): the caller issues the statistics collective of several MixedOps at once (functional.StatChain)
            return prep if prepare_only else prep(weights, addend)
        fork = K.Fork(dev, nstreams, tag="candidates")
        ys = []
        for k, (op, _, _) in enumerate(self._ops):
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
                    paired[pair[0]], paired[pair[1]] = OPS.dense_pair_forward(self._ops[pair[0]][0], self._ops[pair[1]][0], g, a, b, for_epilogue=True)
                    y = paired[k]
                    paired[pair[0] + pair[1] - k].record_stream(fork.main)
                elif row_ok and type(op) is OPS.f_sparse_op_comp:
                    y = op(g, a, b, for_epilogue=True)
                else:
                    y = op(g, a, b)
            y.record_stream(fork.main)                 # consumed by the epilogue on the main stream
            ys.append(y)
        fork.join()
        prep = K.mixed_epilogue_prepare(ys, [bn for _, bn, _ in self._ops], group, total_rows, True, self._identity_index())
        return prep if prepare_only else prep(weights, addend)


    def _identity_index(self):
        ids = [k for k, (op, _, _) in enumerate(self._ops) if type(op) is OPS.f_identity_op]
        return ids[0] if len(ids) == 1 else None

    def _dense_pair(self, x):
        """(index of f_dense_comp, index of f_comp) when both are candidates of this MixedOp and may share a node, else None."""
        if not x.is_cuda:
            return None
        d = [k for k, (op, _, _) in enumerate(self._ops) if type(op) is OPS.f_dense_op_comp]
        c = [k for k, (op, _, _) in enumerate(self._ops) if type(op) is OPS.f_comp_op]
        if len(d) != 1 or len(c) != 1:
            return None
        return (d[0], c[0])



class _Stage(nn.Module):
    """A list of MixedOps under the attribute name ``_ops`` (the reference's Cell_* classes)."""

    def __init__(self, count, feature_dim, drop_aggr, operations, registry):
        super().__init__()
        self._ops = nn.ModuleList(MixedOp(feature_dim, drop_aggr, operations, registry) for _ in range(count))


class SuperCell(nn.Module):
    """Zero -> First -> Middle -> Last stages and the concat linear
    (reference models/cell_lp.py:53-188)."""

    def __init__(self, n_zero, n_first, n_last, feature_dim, drop_aggr, registry=OPS.MIXED_OPS):
        super().__init__()
        self.n_first, self.n_last = n_first, n_last
        mk = lambda cnt, names: _Stage(cnt, feature_dim, drop_aggr, names, registry)
        self.cell_zero = mk(1, OPS.PRE_OPS)
        self.cell_first = mk(sum(i + 1 for i in range(n_first)), OPS.FIRST_OPS)
        self.cell_middle = mk(n_first, OPS.MIDDLE_OPS)
        self.cell_last = mk(sum(n_first + i for i in range(n_last)), OPS.LAST_OPS)
        self.concat_weights = nn.Linear((n_first + n_last) * feature_dim, feature_dim)

    def _fan(self, x):
        """Reader bookkeeping for one state: every candidate of every MixedOp of the cell may read it."""
        if not x.is_cuda:
            return x
        n_mixed = 1 + len(self.cell_first._ops) + len(self.cell_middle._ops) + len(self.cell_last._ops)
        width = max(len(m._ops) for st in (self.cell_first, self.cell_middle, self.cell_last) for m in st._ops)
        return K.Fan(x, 2 * n_mixed * width + 2)

    def _dense_stage(self, stage, states, weights, g, h_in, steps):
        off = 0
        for _ in range(steps):
            s = None                                   # the MixedOps feeding one state: each adds onto the previous one's output
            for j, h in enumerate(states):
                s = stage._ops[off + j](weights[off + j], g, h, h_in, addend=s)
            off += len(states)
            states.append(self._fan(s))
        return states

    def forward(self, g, src_emb, hr, w_zero, w_first, w_middle, w_last):
        h_in = self._fan(self.cell_zero._ops[0](w_zero[0], g, src_emb, hr))
        states = self._dense_stage(self.cell_first, [h_in], w_first, g, h_in, self.n_first)[1:]
        states = [self._fan(self.cell_middle._ops[i](w_middle[i], g, states[i], h_in)) for i in range(self.n_first)]
        states = self._dense_stage(self.cell_last, states, w_last, g, h_in, self.n_last)
        states = [s.take() if isinstance(s, K.Fan) else s for s in states]
        return self.concat_weights(torch.cat(states, dim=1))


class SearchNetwork(nn.Module):
    """The mixed-op supernet (reference models/model_search_lp.py:16-194)."""

    def __init__(self, device, number_of_nodes, num_rels, layers, zero_nodes, first_nodes, last_nodes, feature_dim,
                 init_fea_dim, num_base_r, gamma, dropout_cell, drop_aggr, registry=OPS.MIXED_OPS):
        super().__init__()
        self._device = device
        self._layers, self._num_ent, self._num_rel = layers, number_of_nodes, num_rels * 2 + 1
        self._feature_dim, self._dropout, self.gamma = feature_dim, dropout_cell, gamma
        self.nz, self.nf, self.nl = zero_nodes, first_nodes, last_nodes
        self.n_first_edges = sum(zero_nodes + i for i in range(first_nodes))
        self.n_last_edges = sum(first_nodes + i for i in range(last_nodes))
        self.embedding_h = nn.Embedding(number_of_nodes, init_fea_dim)
        self.embedding_e = nn.Embedding(num_base_r, feature_dim)
        self.linear_e = nn.Linear(init_fea_dim, feature_dim)
        self.rel_wt = _xavier_param(self._num_rel, num_base_r)
        self.w_rel = _xavier_param(feature_dim, feature_dim)
        self.cells = nn.ModuleList(SuperCell(zero_nodes, first_nodes, last_nodes, feature_dim, drop_aggr, registry)
                                   for _ in range(layers))
        self.batchnorm_h = nn.BatchNorm1d(feature_dim)
        # architecture parameters: not part of state_dict, exactly as in the reference (:99-129)
        mk = lambda rows, cols: (1e-3 * torch.randn(rows * layers, cols, device=device)).requires_grad_(True)
        self._arch_parameters = [mk(zero_nodes, len(OPS.PRE_OPS)), mk(self.n_first_edges, len(OPS.FIRST_OPS)),
                                 mk(first_nodes, len(OPS.MIDDLE_OPS)), mk(self.n_last_edges, len(OPS.LAST_OPS)),
                                 (1e-3 * torch.randn(1, len(OPS.SF_OPS), device=device)).requires_grad_(True)]

    def arch_parameters(self):
        return self._arch_parameters

    def load_alpha(self, alphas):
        for x, y in zip(self._arch_parameters, alphas):
            x.data.copy_(y.data)

    def layer_weights(self, l):
        a = self._arch_parameters
        sl = lambda t, n: F.softmax(t[l * n:(l + 1) * n], dim=1)
        return sl(a[0], self.nz), sl(a[1], self.n_first_edges), sl(a[2], self.nf), sl(a[3], self.n_last_edges)

    def row_weights(self):
        """layer_weights for every layer, each matrix as a tuple of its [K] rows: ONE softmax and ONE unbind per architecture
        parameter and step.  (Slicing a layer and indexing a row per MixedOp costs a select, a zero-filled gradient, a copy and
        an accumulate each: ~65 tiny launches per step, which a 30 000-edge step notices.)"""
        ns = (self.nz, self.n_first_edges, self.nf, self.n_last_edges)
        rows = [F.softmax(t, dim=1).unbind(0) for t in self._arch_parameters[:4]]
        return [tuple(r[l * n:(l + 1) * n] for r, n in zip(rows, ns)) for l in range(self._layers)]

    def prepare(self, g, node_id, src_in, edge_type):
        """Gather indices of a step graph: (ent plan, rel plan, layer>=2 plan).  Cached on the graph object
        for as long as the SAME index tensors are passed unmodified (identity + in-place version); anything
        else rebuilds, so a reused graph object with new node_id / src_in / edge_type never sees stale plans."""
        def build():
            n = g.number_of_nodes()
            dev = src_in.device
            src_in_f = torch.cat((src_in.long(), torch.arange(n, device=dev)))
            ent_idx = node_id.view(-1).long()[src_in_f]
            rel_idx = torch.cat((edge_type.long(), torch.full((n,), self._num_rel - 1, dtype=torch.long, device=dev)))
            # layer >= 2 reads cat(ent[src_in], ent): one gather with the index cat(src_in, arange(n)), no torch.cat
            return (K.GatherPlan(ent_idx, self._num_ent), K.GatherPlan(rel_idx, self._num_rel), K.GatherPlan(src_in_f, n))
        return cached_on(g, "_mrg_search_plans", (node_id, src_in, edge_type), (self._num_ent, self._num_rel, g.number_of_nodes()), build)

    _plans = prepare

    def forward(self, g_train, node_id, src_in, edge_type):
        ent_all = self.linear_e(self.embedding_h.weight)
        rel = torch.mm(self.rel_wt, self.embedding_e.weight)
        p_ent, p_rel, p_in = self._plans(g_train, node_id, src_in, edge_type)
        ent = None
        weights = self.row_weights()
        for l, cell in enumerate(self.cells):
            wz, wf, wm, wl = weights[l]
            # the gather G (reference :135-145, :153-154) is not materialised: the cell's first stage gathers inside its compose kernels
            x = K.LazyRows(ent_all, p_ent) if l == 0 else K.LazyRows(ent, p_in)
            ent = self.batchnorm_h(cell(g_train, x, K.LazyRows(rel, p_rel), wz, wf, wm, wl))
            if l > 0 or self._layers == 1:
                ent = F.relu(ent)
            ent = F.dropout(ent, self._dropout, training=self.training)
            rel = torch.matmul(rel, self.w_rel)
        return ent, rel

    def _score_plan(self, ent, rel, triplets):
        """Index plan of a scoring batch.  Reused only for the SAME triplets tensor object at the same in-place
        version (the cache holds the tensor, so its address cannot be recycled under it); a new batch -- the
        reference driver builds one per epoch, search/mr_lp_search.py:187-255 -- always gets a new plan."""
        return cached_on(self, "_score_cache", (triplets,), (ent.shape[0], rel.shape[0]),
                         lambda: K.ScorePlan(triplets, ent.shape[0], rel.shape[0]))

    def calc_score(self, ent, rel, triplets, plan=None):
        """DistMult (reference models/model_search_lp.py:169-176) in one fused HIP kernel: no [T, D]
        gathers are materialised; the backward is three balanced segmented-sum launches.  `plan`: an explicit
        functional.ScorePlan of `triplets` (callers that keep a batch across steps may build it once)."""
        if not ent.is_cuda:
            t = triplets.long()
            return torch.sum(ent[t[:, 0]] * rel[t[:, 1]] * ent[t[:, 2]], dim=1)
        return K.distmult_score(ent, rel, plan if plan is not None else self._score_plan(ent, rel, triplets))

    def get_loss(self, g_train, ent, rel, triplets, labels):
        return F.binary_cross_entropy_with_logits(self.calc_score(ent, rel, triplets), labels)

    def _loss(self, g_train, node_id, src_in, edge_type, triplets, labels):
        """What the reference's Architect calls (models/model_search_lp.py:190-194, models/architect_lp.py:50)."""
        ent, rel = self.forward(g_train, node_id, src_in, edge_type)
        return F.binary_cross_entropy_with_logits(self.calc_score(ent, rel, triplets), labels)

    def show_genotype(self, l):
        """Arg-max decoding of one layer (reference models/model_search_lp.py:215-311)."""
        wz, wf, wm, wl = (w.detach().cpu() for w in self.layer_weights(l))
        gene, nz, nf, nl = [], self.nz, self.nf, self.nl
        for n in range(nz):
            gene.append((OPS.PRE_OPS[int(wz[n].argmax())], n + 1, n))
        top = nz                                           # == max(pre_nodes) after the zero stage

        def best_edge(W, names, count):
            skip = names.index('f_zero')
            cols = [k for k in range(len(names)) if k != skip]
            j = max(range(count), key=lambda x: (max(float(W[x][k]) for k in cols), -x))
            k = max(cols, key=lambda c: (float(W[j][c]), -c))
            return j, k

        start = 0
        for n in range(1, nf + 1):
            j, k = best_edge(wf[start:start + n], OPS.FIRST_OPS, n)
            gene.append((OPS.FIRST_OPS[k], top + n, top + j))
            start += n
        concat, middle = [], list(range(2, 2 + nf))
        for n in range(nf):
            new = max(middle) + 1
            gene.append((OPS.MIDDLE_OPS[int(wm[n].argmax())], new, middle[n]))
            concat.append(new)
            middle[n] = new
        start = 0
        for n in range(nl):
            cnt = nf + n
            j, k = best_edge(wl[start:start + cnt], OPS.LAST_OPS, cnt)
            node = n + max(middle) + 1
            gene.append((OPS.LAST_OPS[k], node, middle[j] if j < nf else j - nf + max(middle) + 1))
            concat.append(node)
            start += cnt
        return Genotype(alpha_cell=gene, concat_node=concat, score_func=None)

    def show_genotypes(self):
        return [self.show_genotype(l) for l in range(self._layers)]


# ---------------------------------------------------------------------------
# fixed genotype
# ---------------------------------------------------------------------------
class OpModule(nn.Module):
    """op -> BN -> ReLU; only 'pre_mult' skips BN/ReLU (the reference's condition at
    models/model_lp.py:31 reduces to ``op_name != 'pre_mult'``; :34 discards its dropout)."""

    def __init__(self, feature_dim, drop_aggr, name, registry=OPS.MIXED_OPS):
        super().__init__()
        self.op = registry[name]({'feature_dim': feature_dim, 'drop_aggr': drop_aggr})
        self.op_name = name
        self.batchnorm_h = nn.BatchNorm1d(feature_dim)

    def forward(self, g, h, h_in):
        h = self.op(g, h, h_in)
        if self.op_name == 'pre_mult':
            return h
        if h.is_cuda:
            # BN + ReLU through the MixedOp epilogue kernels with one branch of weight 1 (statistics pass + combine pass; the
            # backward one reduction + one apply pass): torch's BatchNorm1d over [11 M, 256] rows took 9/10 of the C5 cell step
            one = getattr(self, "_one", None)
            if one is None or one.device != h.device:
                one = self._one = torch.ones(1, dtype=torch.float32, device=h.device)
            return K.mixed_epilogue([h], [self.batchnorm_h], one)
        return F.relu(self.batchnorm_h(h))


class FixedCell(nn.Module):
    def __init__(self, feature_dim, drop_aggr, genotype, registry=OPS.MIXED_OPS):
        super().__init__()
        nb = len({c for _, c, _ in genotype.alpha_cell})
        self._nb = nb
        self._concat_node = list(range(1, 1 + nb)) if genotype.concat_node is None else list(genotype.concat_node)
        self._ops = nn.ModuleList(nn.ModuleList(nn.ModuleList() for _ in range(n)) for n in range(1, 1 + nb))
        for name, center, pre in genotype.alpha_cell:
            self._ops[center - 1][pre].append(OpModule(feature_dim, drop_aggr, name, registry))
        self.concat = nn.Linear(len(self._concat_node) * feature_dim, feature_dim)
        self.batchnorm_h = nn.BatchNorm1d(feature_dim)

    def forward(self, g, src_emb, hr):
        zero_out = self._ops[0][0][0](g, src_emb, hr)
        states = [src_emb, zero_out]
        for n in range(1, self._nb):
            states.append(_tsum(self._ops[n][i][0](g, states[i], zero_out) for i in range(n + 1) if len(self._ops[n][i])))
        h = self.concat(torch.cat([states[i] for i in self._concat_node], dim=1))
        return F.relu(self.batchnorm_h(h))


class FixedNetwork(nn.Module):
    """Fixed-genotype link-prediction network (reference models/model_lp.py:77-150)."""

    def __init__(self, device, genotype, number_of_nodes, num_rels, feature_dim, init_fea_dim, num_base_r,
                 criterion=None, dropout_cell=0.0, drop_aggr=0.0, score_args=None, registry=OPS.MIXED_OPS):
        super().__init__()
        self._device, self._num_ent, self._num_rel = device, number_of_nodes, num_rels * 2 + 1
        self._dropout, self.criterion = dropout_cell, criterion or nn.BCELoss()
        self.embedding_h = nn.Embedding(number_of_nodes, init_fea_dim)
        self.embedding_e = nn.Embedding(num_base_r, feature_dim)
        self.linear_e = nn.Linear(init_fea_dim, feature_dim)
        self.rel_wt = _xavier_param(self._num_rel, num_base_r)
        self.cells = nn.ModuleList(FixedCell(feature_dim, drop_aggr, gt, registry) for gt in genotype)
        self.score_func = OPS.MIXED_OPS_sf[genotype[-1].score_func](score_args or {})
        self.w_rel = _xavier_param(feature_dim, feature_dim)

    def _plans(self, g):
        src, _, _ = g.edges(form='all')
        etype = g.edata['e_type']

        def build():
            n, dev = g.number_of_nodes(), src.device
            ent_idx = torch.cat((src, torch.arange(n, device=dev)))
            rel_idx = torch.cat((etype.long(), torch.full((n,), self._num_rel - 1, dtype=torch.long, device=dev)))
            return (K.GatherPlan(ent_idx, n), K.GatherPlan(rel_idx, self._num_rel))
        return cached_on(g, "_mrg_fixed_plans", (src, etype), (self._num_rel, g.number_of_nodes()), build)

    def forward(self, g, subj, rel):
        ent = self.linear_e(self.embedding_h.weight)
        rel_emb = torch.mm(self.rel_wt, self.embedding_e.weight)
        p_ent, p_rel = self._plans(g)
        for cell in self.cells:
            ent = cell(g, K.gather(ent, p_ent), K.gather(rel_emb, p_rel))
            ent = F.dropout(ent, self._dropout, training=self.training)
            rel_emb = torch.matmul(rel_emb, self.w_rel)
        return self.score_func(ent, ent[subj], rel_emb[rel])

    def _loss(self, g, subj, rel, label):
        return self.criterion(self.forward(g, subj, rel), label)
