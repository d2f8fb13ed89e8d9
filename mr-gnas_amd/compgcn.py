"""CompGCN layer and stack on the fused HIP kernel, importable in place of the reference's
``models/compgcn.py`` (``CompGraphConv`` :12-113, ``CompGCN`` :116-185; the ConvE scorer
``CompGCN_ConvE`` :188-269 is outside the hot path and not provided).

Same constructors, same ``forward`` signatures and return values, same parameter names
(``W_O, W_I, W_S, W_R, loop_rel, bn``; ``basis, weights | rel_embds, n_embds, layers, dropouts``).
``g`` is a ``RelGraph`` carrying ``edata['etype'|'norm'|'in_edges_mask'|'out_edges_mask']``.

What changes is the schedule: W_O and W_I are linear and applied before a *sum*, so they
commute with it.  The layer aggregates phi(h_src, r*norm) per (destination, direction) with ONE
fused gather->compose->segmented-sum launch (no [E, D] tensor exists), then applies
[W_O | W_I] to the [N, 2*Din] result on the MFMA pipe; the biases enter through the
per-node edge counts.  max|delta| against the reference's per-edge order is ~1e-6 (tested).
"""
import torch
import torch.nn as nn

from . import functional as K
from .graph import cached_on


def _layer_plans(g, n_rel_rows):
    """Index structures of a graph for CompGraphConv, cached on the graph object for as long as the edge
    tensors they derive from (masks, norm, etype) are the same unmodified objects."""
    src, dst, _ = g.edges(form='all')
    deps = (src, dst, g.edata['in_edges_mask'], g.edata['out_edges_mask'], g.edata['norm'], g.edata['etype'])
    return cached_on(g, "_mrg_compgcn_plans", deps, n_rel_rows, lambda: _build_layer_plans(g, n_rel_rows))


def _build_layer_plans(g, n_rel_rows):
    src, dst, _ = g.edges(form='all')
    N, dev = g.number_of_nodes(), src.device
    in_m = g.edata['in_edges_mask'].bool()
    out_m = g.edata['out_edges_mask'].bool()
    keep = torch.nonzero(in_m | out_m).view(-1)                  # edges in neither mask contribute 0 (reference :80-82)
    direction = in_m[keep].long()                                # an edge in both masks is transformed by W_I (:82 overwrites :81)
    seg = dst[keep] * 2 + direction                              # segment = (destination, direction); 0 = out (W_O), 1 = in (W_I)
    norm = g.edata['norm'].reshape(-1).float()[keep]
    edges = K.ComposePlan(src[keep], g.edata['etype'].long()[keep], seg, norm, N, n_rel_rows, 2 * N)
    counts = torch.bincount(seg, minlength=2 * N).view(N, 2).float()
    ar = torch.arange(N, device=dev)
    loop = K.ComposePlan(ar, torch.full((N,), n_rel_rows - 1, dtype=torch.long, device=dev), ar, None, N, n_rel_rows, N)
    return {"n_rel_rows": n_rel_rows, "edges": edges, "counts": counts, "loop": loop}


class CompGraphConv(nn.Module):
    """One layer of CompGCN."""

    def __init__(self, in_dim, out_dim, comp_fn='sub', batchnorm=True, dropout=0.1):
        super().__init__()
        self.in_dim, self.out_dim, self.comp_fn, self.batchnorm = in_dim, out_dim, comp_fn, batchnorm
        self.actvation = torch.tanh
        self.dropout = nn.Dropout(dropout)
        if batchnorm:
            self.bn = nn.BatchNorm1d(out_dim)
        self.W_O = nn.Linear(in_dim, out_dim)
        self.W_I = nn.Linear(in_dim, out_dim)
        self.W_S = nn.Linear(in_dim, out_dim)
        self.W_R = nn.Linear(in_dim, out_dim)
        self.loop_rel = nn.Parameter(torch.empty(1, in_dim))
        self.register_buffer("_one", torch.ones(1), persistent=False)      # the one-branch epilogue's weight (not in state_dict)
        nn.init.xavier_normal_(self.loop_rel)

    def forward(self, g, n_in_feats, r_feats):
        if self.comp_fn not in ('sub', 'mul', 'ccorr'):
            raise Exception('Only supports sub, mul, and ccorr')
        r_plus = torch.cat((r_feats, self.loop_rel), 0)
        P = _layer_plans(g, r_plus.shape[0])
        N = g.number_of_nodes()
        # steps 1-3: sum over in-edges of phi(h_src, r*norm), per direction  -> [N, 2*Din]
        A = K.compose_aggregate(self.comp_fn, n_in_feats, r_plus, P["edges"]).view(N, 2 * self.in_dim)
        W_cat = torch.cat((self.W_O.weight, self.W_I.weight), dim=1)
        cnt = P["counts"]                                   # the biases enter through the per-node edge counts ([N, 2] x [2, D] as two broadcasts:
        comp_edge = K.linear(A, W_cat) + (cnt[:, :1] * self.W_O.bias + cnt[:, 1:] * self.W_I.bias)   # a K = 2 product costs a 66 us vendor GEMM)
        # step 4: self-loop composition with loop_rel
        comp_s = K.compose_aggregate(self.comp_fn, n_in_feats, r_plus, P["loop"])
        n_out = (K.linear(comp_s, self.W_S.weight, self.W_S.bias) + self.dropout(comp_edge)) * (1 / 3)
        r_out = K.linear(r_plus, self.W_R.weight, self.W_R.bias)
        if K.switches.COMPGCN_TAIL and self.batchnorm and self.actvation is torch.tanh and n_out.is_cuda and not (self.bn._forward_hooks or self.bn._forward_pre_hooks):
            # BatchNorm -> tanh on the MixedOp epilogue kernels with one branch of weight 1 and the tanh activation (statistics
            # pass + combine pass; backward one reduction + one apply pass) instead of torch's four BatchNorm kernels + tanh
            n_out = K.mixed_epilogue([n_out], [self.bn], self._one if self._one.device == n_out.device else self._one.to(n_out.device), act="tanh")
            return n_out, r_out[:-1]
        if self.batchnorm:
            n_out = self.bn(n_out)
        if self.actvation is not None:
            n_out = self.actvation(n_out)
        return n_out, r_out[:-1]


class CompGCN(nn.Module):
    def __init__(self, num_bases, num_rel, num_ent, in_dim=100, layer_size=[200], comp_fn='sub', batchnorm=True,
                 dropout=0.1, layer_dropout=[0.3]):
        super().__init__()
        self.num_bases, self.num_rel, self.num_ent = num_bases, num_rel, num_ent
        self.in_dim, self.layer_size, self.comp_fn = in_dim, layer_size, comp_fn
        self.batchnorm, self.dropout, self.layer_dropout = batchnorm, dropout, layer_dropout
        self.num_layer = len(layer_size)
        dims = [in_dim] + list(layer_size)
        self.layers = nn.ModuleList(CompGraphConv(dims[i], dims[i + 1], comp_fn=comp_fn, batchnorm=batchnorm, dropout=dropout)
                                    for i in range(self.num_layer))
        if num_bases > 0:
            self.basis = nn.Parameter(torch.empty(num_bases, in_dim))
            self.weights = nn.Parameter(torch.empty(num_rel, num_bases))
            nn.init.xavier_normal_(self.basis)
            nn.init.xavier_normal_(self.weights)
        else:
            self.rel_embds = nn.Parameter(torch.empty(num_rel, in_dim))
            nn.init.xavier_normal_(self.rel_embds)
        self.n_embds = nn.Parameter(torch.empty(num_ent, in_dim))
        nn.init.xavier_normal_(self.n_embds)
        self.dropouts = nn.ModuleList(nn.Dropout(layer_dropout[i]) for i in range(self.num_layer))

    def forward(self, graph):
        n_feats = self.n_embds
        r_feats = torch.mm(self.weights, self.basis) if self.num_bases > 0 else self.rel_embds
        for layer, drop in zip(self.layers, self.dropouts):
            n_feats, r_feats = layer(graph, n_feats, r_feats)
            n_feats = drop(n_feats)
        return n_feats, r_feats
