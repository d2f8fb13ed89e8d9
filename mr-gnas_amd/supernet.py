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
# The cell and its MixedOps live in cell_lp.py under the reference's own class names (the module a maintainer swaps in for
# models/cell_lp.py); the names below are what rounds 1-3 called them.
from .cell_lp import MixedOp, Cell as SuperCell            # noqa: E402,F401
from . import cell_lp as _cell_lp                          # noqa: E402


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

    # ---- static step graphs (round 5) ------------------------------------------------------------------------------------------
    # The reference's search loop draws a new step graph every step (search/mr_lp_search.py:187-214).  sampler.static_step pads it to a
    # host-known node capacity and keeps the draw's node count on the device; static_rows hands those counts to the MixedOp kernels
    # (mrg_set_dynamic_rows), after which a forward / backward over the padded graph computes exactly the unpadded step's values on the
    # valid rows and keeps the padding rows zero -- with every tensor shape fixed, so the whole step (sampler, graph build, index
    # plans, forward, loss, backward, optimizer) is ONE capturable, replayable HIP graph.
    static_counts = None                                   # (n_rows [1] int32, n_nodes [1] int32) device tensors, or None

    def static_rows(self, n_rows, n_nodes):
        """Switch the static (capacity-padded) step on -- device counts of the valid [M, D] rows (edges + nodes) and node rows -- or
        off (None, None)."""
        from . import _lib, graph as G
        self.static_counts = None if n_rows is None else (n_rows, n_nodes)
        G.STATIC_SHAPES = self.static_counts is not None
        if self.static_counts is None:
            _lib.load().mrg_set_dynamic_rows(-1, None, -1, None)

    def _layer_norm_static(self, h, relu):
        """batchnorm_h (+ ReLU) over the VALID node rows of a capacity-padded [N, D] tensor, padding rows left zero: the MixedOp
        epilogue kernels with one candidate (statistics over *n_nodes rows: mrg_set_dynamic_rows).  Without the ReLU the kernels'
        own ReLU is cancelled exactly: z = relu(z) - relu(-z), two candidates over the same rows with (gamma, beta) and
        (-gamma, -beta) and weights (+1, -1)."""
        from .lazy import BatchNormView
        bn = self.batchnorm_h
        if relu:
            return K.mixed_epilogue([h], [bn], self._ones(h.device)[:1])
        neg = BatchNormView(None, None, -bn.weight, -bn.bias, True, bn.momentum, bn.eps)
        neg.track_running_stats = bn.track_running_stats      # (bns[0] decides; this view has no running statistics of its own)
        return K.mixed_epilogue([h, h], [bn, neg], self._ones(h.device) * self._pm(h.device))

    def _ones(self, dev):
        t = getattr(self, "_ones2", None)
        if t is None or t.device != dev:
            t = self._ones2 = torch.ones(2, dtype=torch.float32, device=dev)
            self._pm2 = torch.tensor([1.0, -1.0], dtype=torch.float32, device=dev)
        return t

    def _pm(self, dev):
        self._ones(dev)
        return self._pm2

    def forward(self, g_train, node_id, src_in, edge_type):
        with K.deferred_counters():                        # the BatchNorm step counters of all MixedOps: one launch at the end
            return self._forward(g_train, node_id, src_in, edge_type)

    def _forward(self, g_train, node_id, src_in, edge_type):
        if _cell_lp.CALLER == "reference" and self.embedding_h.weight.is_cuda:
            return self._forward_reference(g_train, node_id, src_in, edge_type)
        ent_all = K.module_linear(self.linear_e, self.embedding_h.weight)
        rel = torch.mm(self.rel_wt, self.embedding_e.weight)
        p_ent, p_rel, p_in = self._plans(g_train, node_id, src_in, edge_type)
        static = self.static_counts is not None and ent_all.is_cuda
        if static:                                          # the kernels read the counts at run time: registering costs nothing per step
            from . import _lib
            n_cap = int(g_train.number_of_nodes())
            rc = _lib.load().mrg_set_dynamic_rows(int(g_train.num_edges()) + n_cap, _lib.ptr(self.static_counts[0]), n_cap, _lib.ptr(self.static_counts[1]))
            if rc != 0:
                raise _lib.MrgnasError(f"mrg_set_dynamic_rows failed ({rc})")
        ent = None
        weights = self.row_weights()
        for l, cell in enumerate(self.cells):
            wz, wf, wm, wl = weights[l]
            # the gather G (reference :135-145, :153-154) is not materialised: the cell's first stage gathers inside its compose kernels
            x = K.LazyRows(ent_all, p_ent) if l == 0 else K.LazyRows(ent, p_in)
            h = cell(g_train, x, K.LazyRows(rel, p_rel), wz, wf, wm, wl)
            if static and self.training:
                ent = self._layer_norm_static(h, relu=(l > 0 or self._layers == 1))
                ent = F.dropout(ent, self._dropout, training=self.training)
                rel = torch.matmul(rel, self.w_rel)
                continue
            ent = self.batchnorm_h(h)
            if l > 0 or self._layers == 1:
                if K.switches.MASK_TAP is not None and ent.is_cuda:           # test instrumentation (functional.switches.MASK_TAP)
                    K.switches.MASK_TAP(("net", l), [ent > 0])
                ent = F.relu(ent)
            ent = F.dropout(ent, self._dropout, training=self.training)
            rel = torch.matmul(rel, self.w_rel)
        return ent, rel

    def _forward_reference(self, g_train, node_id, src_in, edge_type):
        """CALLER == "reference": the reference's own lines (models/model_search_lp.py:131-163), written as the reference writes them
        -- index tensors built inside every forward, the gathers as plain tensor indexing, torch's Linear / BatchNorm / softmax
        slicing -- so that `bench.py --caller reference` and the tests measure what the UNCHANGED caller gets on this package's
        operators: with lazy handles (mr_gnas_amd/lazy.py, the default) the fused path; with MRG_LAZY=0 the operator swap alone."""
        dev = self.embedding_h.weight.device
        all_ent_emb = self.linear_e(self.embedding_h(torch.arange(self._num_ent, device=dev)))
        rel_embed = torch.mm(self.rel_wt, self.embedding_e(torch.arange(self.embedding_e.num_embeddings, device=dev)))
        src_in = src_in.long()
        src_in_final = torch.cat((src_in, g_train.nodes()), dim=0)
        edge_self = (torch.ones(g_train.nodes().shape) * (self._num_rel - 1)).long().to(dev)
        src_id_final = node_id[src_in_final].squeeze()
        edge_type_final = torch.cat((edge_type.long(), edge_self), dim=0)
        ent_emb = None
        for i, cell in enumerate(self.cells):
            W_zero, W_first, W_middle, W_last = self.layer_weights(i)
            if i == 0:
                ent_emb_in = all_ent_emb[src_id_final]
                ent_emb = cell(g_train, ent_emb_in, rel_embed[edge_type_final], W_zero, W_first, W_middle, W_last)
                ent_emb = self.batchnorm_h(ent_emb)
                if len(self.cells) == 1:
                    ent_emb = F.relu(ent_emb)
            else:
                ent_emb_in = torch.cat((ent_emb[src_in], ent_emb), dim=0)
                ent_emb = cell(g_train, ent_emb_in, rel_embed[edge_type_final], W_zero, W_first, W_middle, W_last)
                ent_emb = F.relu(self.batchnorm_h(ent_emb))
            ent_emb = F.dropout(ent_emb, self._dropout, training=self.training)
            rel_embed = torch.matmul(rel_embed, self.w_rel)
        return ent_emb, rel_embed

    def _score_plan(self, ent, rel, triplets):
        """Index plan of a scoring batch.  Reused only for the SAME triplets tensor object at the same in-place
        version (the cache holds the tensor, so its address cannot be recycled under it); a new batch -- the
        reference driver builds one per epoch, search/mr_lp_search.py:187-255 -- always gets a new plan."""
        return cached_on(self, "_score_cache", (triplets,), (ent.shape[0], rel.shape[0]),
                         lambda: K.ScorePlan(triplets, ent.shape[0], rel.shape[0]))

    def calc_score(self, ent, rel, triplets, plan=None):
        """DistMult (reference models/model_search_lp.py:169-176) in one fused HIP kernel: no [T, D]
        gathers are materialised; the backward is three balanced segmented-sum launches.  `plan`: an explicit
        functional.ScorePlan of `triplets` (callers that keep a batch across steps may build it once).
        CALLER == "reference" (and CPU tensors): the reference's own four lines -- three gathers, a product, a row sum."""
        if not ent.is_cuda or _cell_lp.CALLER == "reference":
            if not ent.is_cuda:
                triplets = triplets.long()
            s = ent[triplets[:, 0]]
            r = rel[triplets[:, 1]]
            o = ent[triplets[:, 2]]
            return torch.sum(s * r * o, dim=1)
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
        self.register_buffer("_one", torch.ones(1), persistent=False)      # the one-branch epilogue's weight (not in state_dict; moves with .to())

    def forward(self, g, h, h_in):
        if isinstance(h, K.LazyRows):
            # the cell's zero node on un-materialised gathers (FixedNetwork._forward): pre_sub / pre_add -> BN -> ReLU recomputed from
            # the two tables (functional.cell_zero_mixed with one branch of weight 1): no [M, D] gather, compose or BatchNorm tensor
            if (self.op_name != 'pre_mult' and K.switches.CELL_ZERO_FUSED and isinstance(h_in, K.LazyRows)
                    and isinstance(self.op, OPS._PreOp) and not (self.op._forward_hooks or self.op._forward_pre_hooks)):
                one = self._one if self._one.device == h.device else self._one.to(h.device)
                return K.cell_zero_mixed([self.op.kind], h, h_in, [self.batchnorm_h], one)
            h = h.materialize()
        if isinstance(h_in, K.LazyRows):
            h_in = h_in.materialize()
        h = _cell_lp._run(self.op, g, h, h_in)             # the operator's value now (module call when it carries hooks)
        if self.op_name == 'pre_mult':
            return h
        if h.is_cuda:
            # BN + ReLU through the MixedOp epilogue kernels with one branch of weight 1 (statistics pass + combine pass; the
            # backward one reduction + one apply pass): torch's BatchNorm1d over [11 M, 256] rows took 9/10 of the C5 cell step
            one = self._one if self._one.device == h.device else self._one.to(h.device)
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
        self.register_buffer("_one", torch.ones(1), persistent=False)

    def _caps(self):
        """Readers per state (state 0 = src_emb, 1 = the zero node's output, ...): the ops that take it as their first operand, the
        concat Linear, and -- state 1 only -- every op of the later nodes, which take it as ``src_emb_in``."""
        caps = getattr(self, "_reader_caps", None)
        if caps is None:
            caps = [0] * (self._nb + 1)
            caps[0] = 1                                  # the zero node's op
            for n in range(1, self._nb):
                for i in range(n + 1):
                    if len(self._ops[n][i]):
                        caps[i] += 1
                        caps[1] += 1
            for i in self._concat_node:
                caps[i] += 1
            self._reader_caps = caps
        return caps

    def forward(self, g, src_emb, hr, apply=None, finish=None):
        """apply(op_module, h, h_in) -> the node's candidate, finish(concat output) -> the cell's output: the hooks through which
        dist.ShardedFixedNet runs the same cell on a relation block (aggregator exchanges, statistics over all ranks' rows)."""
        caps = self._caps()
        apply = apply or (lambda mod, h, h_in: mod(g, h, h_in))
        # a state with several readers hands out aliases (functional.Fan): its gradient is ONE K-way sum of the readers' gradients
        # instead of autograd's chain of pairwise adds (K - 1 launches of three [rows, D] passes each: 12.8 ms of the C5 step)
        fan = lambda x, i: K.Fan(x, caps[i]) if (torch.is_tensor(x) and x.is_cuda and caps[i] > 1) else None
        take = lambda i: fans[i].take() if fans[i] is not None else states[i]
        states = [src_emb]
        fans = [fan(src_emb, 0)]
        states.append(apply(self._ops[0][0][0], take(0), hr))
        fans.append(fan(states[1], 1))
        for n in range(1, self._nb):
            states.append(_tsum(apply(self._ops[n][i][0], take(i), take(1)) for i in range(n + 1) if len(self._ops[n][i])))
            fans.append(fan(states[-1], n + 1))
        h = K.module_linear(self.concat, torch.cat([take(i) for i in self._concat_node], dim=1))
        if finish is not None:
            return finish(h)
        if h.is_cuda:                                   # BN + ReLU on the MixedOp epilogue kernels with one branch of weight 1 (OpModule.forward)
            one = self._one if self._one.device == h.device else self._one.to(h.device)
            return K.mixed_epilogue([h], [self.batchnorm_h], one)
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
        with K.deferred_counters():
            return self._forward(g, subj, rel)

    def _forward(self, g, subj, rel):
        ent = K.module_linear(self.linear_e, self.embedding_h.weight)
        rel_emb = torch.mm(self.rel_wt, self.embedding_e.weight)
        p_ent, p_rel = self._plans(g)
        for cell in self.cells:
            if ent.is_cuda:              # the gathers stay un-materialised: only the cell's zero node reads them (FixedCell._caps: caps[0] == 1)
                ent = cell(g, K.LazyRows(ent, p_ent), K.LazyRows(rel_emb, p_rel))
            else:
                ent = cell(g, K.gather(ent, p_ent), K.gather(rel_emb, p_rel))
            ent = F.dropout(ent, self._dropout, training=self.training)
            rel_emb = torch.matmul(rel_emb, self.w_rel)
        return self.score_func(ent, ent[subj], rel_emb[rel])

    def _loss(self, g, subj, rel, label):
        return self.criterion(self.forward(g, subj, rel), label)
