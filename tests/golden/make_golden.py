#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Test infrastructure, never shipped and never imported by the product.  Run it
in the build container only (it needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_golden.py

What it does
------------
* imports the reference's Python modules from /root/reference *as they are*
  (``models.operations_lp``, ``models.cell_lp``, ``models.model_lp``,
  ``models.model_search_lp``, ``models.compgcn``, ``utils.utils_rgcn``),
* supplies the two things the reference needs and this image lacks:
  a stand-in ``dgl`` package (DGL 0.5.3 is pinned by the reference's README and
  is not installable here) and the module ``utils.gpu_memory_log`` that the
  reference imports but does not ship,
* runs every hot-path operator forward + backward on small seeded inputs and
  stores inputs, parameters, outputs and gradients as ``.npz`` fixtures.

The arithmetic of every ``nn.Module`` in the fixtures is the reference's own
code.  The arithmetic of DGL's reducers (``update_all(copy_e, max|sum|mean)``,
``apply_edges(u_sub_e|u_mul_e)``) is supplied by the stand-in below, following
DGL's documented semantics: destination rows without in-edges are 0, mean is
sum / in-degree, the gradient of max goes to ONE arg-max edge (we pin "lowest
edge id wins").  The reference holds no test pinning those, so that part of
the parity is *unpinned* (see oracle/__init__.py and DESIGN.md).

``ccorr`` (reference utils/utils.py:285-301, models/operations_lp.py:58-59)
calls ``torch.rfft``/``torch.irfft`` which no longer exist; its fixtures come
from the direct definition out[k] = sum_i a[i] * b[(i+k) % D] evaluated in
float64 (unpinned as well).
"""
import os
import sys
import types
import contextlib

import numpy as np
import torch

REF = "/root/reference"
OUT = os.environ.get("MRG_GOLDEN_OUT", os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import grad_sample_index, seeded, seeded_param  # noqa: E402  (shared with the tests: big inputs are rebuilt, not stored)


# --------------------------------------------------------------------------
# stand-in dgl
# --------------------------------------------------------------------------
class _SegMaxFirst(torch.autograd.Function):
    """h[v] = max over in-edges of m[e]; rows with no in-edge are 0; the
    gradient is routed to the lowest-numbered arg-max edge."""

    @staticmethod
    def forward(ctx, m, dst, n):
        E, D = m.shape
        h = torch.zeros(n, D, dtype=m.dtype)
        idx = dst.view(-1, 1).expand(E, D)
        h = h.scatter_reduce(0, idx, m, reduce="amax", include_self=False)
        eid = torch.arange(E).view(-1, 1).expand(E, D)
        cand = torch.where(m == h[dst], eid, torch.full_like(eid, E))
        arg = torch.full((n, D), E, dtype=torch.long)
        arg = arg.scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
        ctx.save_for_backward(arg)
        ctx.E = E
        return h

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        E = ctx.E
        n, D = arg.shape
        gm = torch.zeros(E + 1, D, dtype=g.dtype)
        gm.scatter_(0, arg, g)          # rows with arg == E fall into the spare row
        return gm[:E], None, None


class _Frame(dict):
    pass


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.ndata.items()}
        self.dst = {k: v[g._dst] for k, v in g.ndata.items()}
        self.data = g.edata


class FakeGraph:
    """The slice of the DGLGraph protocol the reference touches."""

    def __init__(self, n=0, src=None, dst=None):
        self._n = n
        self._src = torch.zeros(0, dtype=torch.long) if src is None else torch.as_tensor(src, dtype=torch.long)
        self._dst = torch.zeros(0, dtype=torch.long) if dst is None else torch.as_tensor(dst, dtype=torch.long)
        self.ndata = _Frame()
        self.edata = _Frame()

    # construction
    def add_nodes(self, n):
        self._n += n

    def add_edges(self, s, d):
        self._src = torch.cat([self._src, torch.as_tensor(np.asarray(s), dtype=torch.long)])
        self._dst = torch.cat([self._dst, torch.as_tensor(np.asarray(d), dtype=torch.long)])

    # queries
    def number_of_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.numel())

    number_of_edges = num_edges

    def nodes(self):
        return torch.arange(self._n)

    def edges(self, form="uv"):
        if form == "all":
            return self._src, self._dst, torch.arange(self.num_edges())
        return self._src, self._dst

    def in_degrees(self, v=None):
        deg = torch.bincount(self._dst, minlength=self._n)
        return deg if v is None else deg[torch.as_tensor(list(v))]

    def local_var(self):
        g = FakeGraph(self._n, self._src, self._dst)
        g.ndata.update(self.ndata)
        g.edata.update(self.edata)
        return g

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata.clear(); self.ndata.update(nd)
            self.edata.clear(); self.edata.update(ed)

    def to(self, device):
        return self

    @property
    def srcdata(self):
        return self.ndata

    @property
    def dstdata(self):
        return self.ndata

    # message passing
    def apply_edges(self, f):
        if callable(f) and not isinstance(f, tuple):
            self.edata.update(f(_EdgeBatch(self)))
            return
        kind, uf, ef, out = f
        u, e = self.ndata[uf][self._src], self.edata[ef]
        self.edata[out] = u - e if kind == "u_sub_e" else u * e

    def update_all(self, msg, red):
        _, ef, _ = msg
        kind, _, out = red
        m = self.edata[ef]
        if kind == "sum" or kind == "mean":
            h = torch.zeros(self._n, m.shape[1], dtype=m.dtype).index_add(0, self._dst, m)
            if kind == "mean":
                deg = torch.bincount(self._dst, minlength=self._n).clamp(min=1).to(m.dtype)
                h = h / deg.view(-1, 1)
        elif kind == "max":
            h = _SegMaxFirst.apply(m, self._dst, self._n)
        else:
            raise NotImplementedError(kind)
        self.ndata[out] = h


def _install_standins():
    dgl = types.ModuleType("dgl")
    fn = types.ModuleType("dgl.function")
    fn.copy_edge = fn.copy_e = lambda e, out: ("copy_e", e, out)
    fn.max = lambda m, out: ("max", m, out)
    fn.sum = lambda m, out: ("sum", m, out)
    fn.mean = lambda m, out: ("mean", m, out)
    fn.u_sub_e = lambda u, e, out: ("u_sub_e", u, e, out)
    fn.u_mul_e = lambda u, e, out: ("u_mul_e", u, e, out)
    dgl.function = fn
    dgl.DGLGraph = FakeGraph
    dgl.graph = lambda data: FakeGraph(0, *data)
    data = types.ModuleType("dgl.data")
    rdf = types.ModuleType("dgl.data.rdf")
    for nm in ("AIFBDataset", "MUTAGDataset", "BGSDataset", "AMDataset"):
        setattr(rdf, nm, None)
    data.rdf = rdf
    dgl.data = data
    sys.modules.update({"dgl": dgl, "dgl.function": fn, "dgl.data": data, "dgl.data.rdf": rdf})
    sys.path.insert(0, REF)
    import utils  # the reference's namespace package
    gml = types.ModuleType("utils.gpu_memory_log")
    gml.gpu_memory_log = lambda *a, **k: None
    sys.modules["utils.gpu_memory_log"] = gml
    utils.gpu_memory_log = gml


# --------------------------------------------------------------------------
# small seeded knowledge graphs
# --------------------------------------------------------------------------
def make_triples(N, T, R, rng, hub=True, isolated=2, dup=3):
    """T (s, r, o) triples over N nodes / R relations: skewed popularity, one
    hub node, a few nodes that never appear, a few duplicated triples."""
    live = N - isolated
    p = 1.0 / np.arange(1, live + 1) ** 0.75
    p /= p.sum()
    s = rng.choice(live, size=T, p=p)
    o = rng.choice(live, size=T, p=p)
    if hub:
        o[: T // 5] = 0
    r = rng.integers(0, R, size=T)
    for i in range(dup):                       # multi-edges: identical triples
        s[T - 1 - i], r[T - 1 - i], o[T - 1 - i] = s[i], r[i], o[i]
    return np.stack([s, r, o], axis=1).astype(np.int64)


def graph_train_order(N, R, tri):
    """Reference train/mr_lp_train.py:77-89 executed through the stand-in."""
    sys.argv = ["x"]
    g = FakeGraph()
    g.add_nodes(N)
    g.add_edges(tri[:, 0], tri[:, 2])
    g.add_edges(tri[:, 2], tri[:, 0])
    in_deg = g.in_degrees(range(g.number_of_nodes())).float().numpy()
    with np.errstate(divide="ignore"):
        norm = in_deg ** -0.5
    norm[np.isinf(norm)] = 0
    g.ndata["n_norm"] = torch.tensor(norm)
    g.apply_edges(lambda edges: {"norm": edges.dst["n_norm"] * edges.src["n_norm"]})
    g.edata["e_type"] = torch.tensor(np.concatenate([tri[:, 1], tri[:, 1] + R]))
    del g.ndata["n_norm"]
    return g


def graph_search_order(N, R, tri):
    """Reference utils/utils_rgcn.py:129-158 + search/mr_lp_search.py:30-36."""
    import utils.utils_rgcn as ur
    with np.errstate(divide="ignore"):
        g, src_o, rel, node_norm = ur.build_graph_from_triplets(N, R, (tri[:, 0], tri[:, 1], tri[:, 2]))
    g2 = g.local_var()
    g2.ndata["norm"] = torch.from_numpy(node_norm).view(-1, 1)
    g2.apply_edges(lambda edges: {"norm": edges.dst["norm"] * edges.src["norm"]})
    g.edata["norm"] = g2.edata["norm"]            # [E,1], as the search driver stores it
    g.edata["e_type"] = torch.from_numpy(rel)
    return g


def npify(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


# --------------------------------------------------------------------------
# per-operator fixtures
# --------------------------------------------------------------------------
GRAPH_OPS = ["f_zero", "f_identity", "f_dense", "f_dense_comp", "f_comp", "f_sparse", "f_sparse_comp"]
NODE_OPS = ["f_zero", "f_identity", "f_dense_last", "f_sparse_last"]
AGG_OPS = ["a_max", "a_sum", "a_mean"]
PRE = ["pre_mult", "pre_sub", "pre_add"]


def run_op(store, tag, op, g, a, b, gout):
    """forward + backward of one reference operator; everything into store."""
    a = a.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    out = op(g, a, b)
    out.backward(gout)
    store[f"{tag}/out"] = out
    store[f"{tag}/ga"] = a.grad if a.grad is not None else torch.zeros_like(a)
    store[f"{tag}/gb"] = b.grad if b.grad is not None else torch.zeros_like(b)
    for n, p in op.named_parameters():
        store[f"{tag}/param/{n}"] = p
        store[f"{tag}/gparam/{n}"] = p.grad if p.grad is not None else torch.zeros_like(p)


STAR = {"pre_sub", "pre_mult", "f_sparse_comp", "f_sparse_last", "a_max", "a_sum", "a_mean"}


def case_ops(name, N, T, R, D, order, seed, star_only=False, seeded_inputs=False, skip=()):
    """seeded_inputs: the [M, D] inputs / upstream gradients come from conftest.seeded and are NOT stored
    (the tests rebuild them), so a planned SURVEY 8(c) shape such as (300, 2000, 11, 64) stays a few MB."""
    import models.operations_lp as O
    keep = (lambda nm: nm in STAR and nm not in skip) if star_only else (lambda nm: nm not in skip)
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    tri = make_triples(N, T, R, rng)
    g = (graph_train_order if order == "train" else graph_search_order)(N, R, tri)
    src, dst, _ = g.edges(form="all")
    E = g.num_edges()
    M = E + N
    st = {"N": N, "R": R, "D": D, "src": src, "dst": dst, "etype": g.edata["e_type"],
          "norm": g.edata["norm"], "triples": tri}
    if seeded_inputs:
        mk = lambda nm, rows: seeded(nm, (rows, D), seed)
        x, x_in, hr, xn, gM, gN = mk("x", M), mk("x_in", M), mk("hr", M), mk("xn", N), mk("gM", M), mk("gN", N)
        st["input_seed"] = seed
    else:
        x, x_in, hr = torch.randn(M, D), torch.randn(M, D), torch.randn(M, D)
        xn = torch.randn(N, D)
        gM, gN = torch.randn(M, D), torch.randn(N, D)
        st.update(x=x, x_in=x_in, hr=hr, xn=xn, gM=gM, gN=gN)
    args = {"feature_dim": D, "drop_aggr": 0.0}
    for nm in filter(keep, PRE):
        run_op(st, nm, O.MIXED_OPS[nm](args), g, x, hr, gM)
    for nm in filter(keep, GRAPH_OPS):
        op = O.MIXED_OPS[nm](args)
        for p in op.parameters():                 # biases away from 0 so they are exercised
            torch.nn.init.normal_(p, std=0.3)
        run_op(st, nm, op, g, x, x_in, gM)
    for nm in filter(keep, NODE_OPS):
        op = O.MIXED_OPS[nm](args)
        for p in op.parameters():
            torch.nn.init.normal_(p, std=0.3)
        run_op(st, nm + "@node", op, g, xn, xn.clone(), gN)
    for nm in filter(keep, AGG_OPS):
        op = O.MIXED_OPS[nm](args)
        for p in op.parameters():
            torch.nn.init.normal_(p, std=0.3)
        run_op(st, nm, op, g, x, x_in, gN)
    # score functions (reference models/operations_lp.py:101-127)
    B = 9
    sub, rel = torch.randn(B, D), torch.randn(B, D)
    for nm in (() if star_only else ("sf_DisMult", "sf_TransE")):
        op = O.MIXED_OPS_sf[nm]({"gamma": 9.0})
        ent = xn.clone().requires_grad_(True)
        s_ = sub.clone().requires_grad_(True)
        r_ = rel.clone().requires_grad_(True)
        out = op(ent, s_, r_)
        gs = torch.randn(B, N, generator=torch.Generator().manual_seed(seed + 5))
        out.backward(gs)
        st.update({f"{nm}/sub": sub, f"{nm}/rel": rel, f"{nm}/out": out, f"{nm}/gout": gs,
                   f"{nm}/gent": ent.grad, f"{nm}/gsub": s_.grad, f"{nm}/grel": r_.grad})
    np.savez_compressed(os.path.join(OUT, f"ops_{name}.npz"), **npify(st))
    print("wrote ops_%s: N=%d E=%d R=%d D=%d order=%s" % (name, N, E, R, D, order))
    return tri


# --------------------------------------------------------------------------
# CompGCN fixtures (reference models/compgcn.py)
# --------------------------------------------------------------------------
def ccorr_direct(a, b):
    """out[k] = sum_i a[i] * b[(i+k) % D]  (what reference utils/utils.py:285-301 computes)."""
    a = a.double(); b = b.double()
    D = a.shape[-1]
    idx = (torch.arange(D).view(-1, 1) + torch.arange(D).view(1, -1)) % D    # [i, k] -> (i+k)%D
    return torch.einsum("...i,...ik->...k", a, b[..., idx]).float()


def case_compgcn(name, N, T, R, Din, Dout, seed):
    import utils.utils as UU
    UU.ccorr = ccorr_direct                      # torch.rfft is gone: see module docstring
    import models.compgcn as C
    C.ccorr = ccorr_direct
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    tri = make_triples(N, T, R, rng)
    g = graph_train_order(N, R, tri)
    E = g.num_edges()
    g.edata["etype"] = g.edata.pop("e_type")
    m = torch.zeros(E, dtype=torch.bool)
    m[: E // 2] = True
    g.edata["in_edges_mask"] = m
    g.edata["out_edges_mask"] = ~m
    src, dst, _ = g.edges(form="all")
    st = {"N": N, "R": R, "Din": Din, "Dout": Dout, "src": src, "dst": dst, "etype": g.edata["etype"],
          "norm": g.edata["norm"], "in_edges_mask": m}
    n_in, r_in = torch.randn(N, Din), torch.randn(2 * R, Din)
    gn, gr = torch.randn(N, Dout), torch.randn(2 * R, Dout)
    st.update(n_in=n_in, r_in=r_in, gn=gn, gr=gr)
    for fn_ in ("sub", "mul", "ccorr"):
        for bnorm in (True, False):
            tag = f"conv_{fn_}_{'bn' if bnorm else 'nobn'}"
            layer = C.CompGraphConv(Din, Dout, comp_fn=fn_, batchnorm=bnorm, dropout=0.0)
            for p in layer.parameters():
                if p.dim() == 1 and bnorm is False:
                    torch.nn.init.normal_(p, std=0.3)
            layer.train()
            a = n_in.clone().requires_grad_(True)
            b = r_in.clone().requires_grad_(True)
            no, ro = layer(g, a, b)
            (no * gn).sum().add((ro * gr).sum()).backward()
            st.update({f"{tag}/n_out": no, f"{tag}/r_out": ro, f"{tag}/gn_in": a.grad, f"{tag}/gr_in": b.grad})
            for n, p in layer.named_parameters():
                st[f"{tag}/param/{n}"] = p
                st[f"{tag}/gparam/{n}"] = p.grad
            if bnorm:
                st[f"{tag}/bn_running_mean"] = layer.bn.running_mean
                st[f"{tag}/bn_running_var"] = layer.bn.running_var
    # two-layer CompGCN, basis-decomposed relations (reference models/compgcn.py:116-185)
    for fn_, nb in (("sub", 3), ("mul", 0)):
        tag = f"net_{fn_}_b{nb}"
        net = C.CompGCN(nb, 2 * R, N, in_dim=Din, layer_size=[Dout, Din], comp_fn=fn_, batchnorm=True,
                        dropout=0.0, layer_dropout=[0.0, 0.0])
        net.train()
        no, ro = net(g)
        go_n = torch.randn(N, Din, generator=torch.Generator().manual_seed(seed + 1))
        go_r = torch.randn(2 * R, Din, generator=torch.Generator().manual_seed(seed + 2))
        (no * go_n).sum().add((ro * go_r).sum()).backward()
        st.update({f"{tag}/n_out": no, f"{tag}/r_out": ro, f"{tag}/go_n": go_n, f"{tag}/go_r": go_r})
        for n, p in net.named_parameters():
            st[f"{tag}/param/{n}"] = p
            st[f"{tag}/gparam/{n}"] = p.grad
    np.savez_compressed(os.path.join(OUT, f"compgcn_{name}.npz"), **npify(st))
    print("wrote compgcn_%s: N=%d E=%d" % (name, N, E))


# --------------------------------------------------------------------------
# whole-network fixtures (the callers of the hot path, reference models/model_lp.py,
# models/cell_lp.py, models/model_search_lp.py)
# --------------------------------------------------------------------------
README_GENOTYPE = ("[Genotype(alpha_cell=[('pre_sub', 1, 0), ('f_sparse_comp', 2, 1), ('f_sparse_comp', 3, 2), "
                   "('a_max', 4, 2), ('a_max', 5, 3), ('f_sparse_last', 6, 5), ('f_sparse_last', 7, 5)], "
                   "concat_node=[4, 5, 6, 7], score_func='sf_DisMult')]")


def case_fixed_net(name, N, T, R, D, D0, nbase, seed):
    from configs.genotypes import Genotype  # noqa: F401  (used by eval)
    import models.model_lp as ML
    import utils.utils as UU
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    tri = make_triples(N, T, R, rng, dup=0)
    g = graph_train_order(N, R, tri)
    genotype = eval(README_GENOTYPE)
    args = types.SimpleNamespace(feature_dim=D, drop_aggr=0.0, drop_op=0.0, gamma=9.0, embed_dim=D,
                                 conve_hid_drop=0.0, feat_drop=0.0, num_filt=4, ker_sz=3, k_w=2, k_h=D // 2)
    net = ML.Network("cpu", genotype, N, R, D, D0, nbase, torch.nn.BCELoss(), 0.0, args)
    net.apply(UU.weights_init)
    net.train()
    B = 7
    subj = torch.from_numpy(rng.integers(0, N, size=B))
    rel = torch.from_numpy(rng.integers(0, 2 * R, size=B))
    label = (torch.rand(B, N) < 0.1).float()
    pred = net(g, subj, rel)
    loss = net.criterion(pred, label)
    loss.backward()
    src, dst, _ = g.edges(form="all")
    st = {"N": N, "R": R, "D": D, "D0": D0, "nbase": nbase, "src": src, "dst": dst,
          "etype": g.edata["e_type"], "norm": g.edata["norm"], "subj": subj, "rel": rel, "label": label,
          "pred": pred, "loss": loss}
    for n, p in net.named_parameters():
        st[f"param/{n}"] = p
        st[f"gparam/{n}"] = p.grad if p.grad is not None else torch.zeros_like(p)
    for n, b in net.named_buffers():
        st[f"buffer/{n}"] = b
    np.savez_compressed(os.path.join(OUT, f"fixednet_{name}.npz"), **npify(st))
    print("wrote fixednet_%s loss=%.6f" % (name, float(loss)))


def case_supernet(name, Nall, T, R, D, D0, nbase, layers, sample, seed, seeded_params=False):
    """One search step's forward/backward (reference search/mr_lp_search.py:187-245
    without the optimiser), on a sub-sampled graph in the search driver's edge order.
    seeded_params: parameters come from conftest.seeded_param (rebuilt by the tests, not stored) and the
    gradients of big parameters are stored as a seeded 4096-element sample + (sum, sum of squares): the
    D = 200 default-size search step of SURVEY 8(c) then fits in ~1 MB instead of ~50 MB."""
    import models.model_search_lp as MS
    import utils.utils as UU
    import utils.utils_rgcn as ur
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    tri = make_triples(Nall, T, R, rng, dup=0)
    adj, deg = ur.get_adj_and_degrees(Nall, tri)
    with np.errstate(divide="ignore"):
        g, node_id, src_in, edge_type, node_norm, data, labels = ur.generate_sampled_graph_and_labels(
            tri, sample, 0.5, R, adj, deg, 2, "uniform")
    g2 = g.local_var()
    g2.ndata["norm"] = torch.from_numpy(node_norm).view(-1, 1)
    g2.apply_edges(lambda edges: {"norm": edges.dst["norm"] * edges.src["norm"]})
    g.edata["norm"] = g2.edata["norm"]
    net = MS.Network("cpu", Nall, R, layers, 1, 2, 2, D, D0, nbase, 9.0, 0.0, 0.0)
    net.apply(UU.weights_init)
    if seeded_params:
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(seeded_param(n, tuple(p.shape), seed))
    net.train()
    node_id_t = torch.from_numpy(node_id).view(-1, 1).long()
    src_in_t = torch.from_numpy(src_in)
    et_t = torch.from_numpy(edge_type)
    data_t, labels_t = torch.from_numpy(data), torch.from_numpy(labels)
    ent, relo = net(g, node_id_t, src_in_t, et_t)
    loss = net.get_loss(g, ent, relo, data_t, labels_t)
    loss.backward()
    src, dst, _ = g.edges(form="all")
    st = {"Nall": Nall, "R": R, "D": D, "D0": D0, "nbase": nbase, "layers": layers,
          "src": src, "dst": dst, "norm": g.edata["norm"], "node_id": node_id, "src_in": src_in,
          "edge_type": edge_type, "data": data, "labels": labels, "ent": ent, "rel_out": relo, "loss": loss}
    for n, p in net.named_parameters():
        gp = p.grad if p.grad is not None else torch.zeros_like(p)
        if seeded_params:
            st[f"pshape/{n}"] = np.asarray(p.shape, dtype=np.int64)
            if p.numel() > 8192:
                st[f"gsample/{n}"] = gp.reshape(-1)[grad_sample_index(n, p.numel(), seed)]
                st[f"gsums/{n}"] = np.asarray([float(gp.double().sum()), float((gp.double() ** 2).sum())])
            else:
                st[f"gparam/{n}"] = gp
        else:
            st[f"param/{n}"] = p
            st[f"gparam/{n}"] = gp
    if seeded_params:
        st["param_seed"] = seed
    else:
        for n, b in net.named_buffers():
            st[f"buffer/{n}"] = b
    for i, a in enumerate(net.arch_parameters()):
        st[f"alpha/{i}"] = a
        st[f"galpha/{i}"] = a.grad if a.grad is not None else torch.zeros_like(a)
    st["genotype0"] = np.array(repr(net.show_genotype(0)))
    np.savez_compressed(os.path.join(OUT, f"supernet_{name}.npz"), **npify(st))
    zdeg = int((torch.bincount(dst, minlength=len(node_id)) == 0).sum())
    print("wrote supernet_%s: n=%d E=%d zero-in-degree=%d loss=%.6f" % (name, len(node_id), len(src), zdeg, float(loss)))


def case_graph_build(name, N, T, R, seed):
    """Edge order, edge types and norms produced by the reference's two graph builders."""
    rng = np.random.default_rng(seed)
    tri = make_triples(N, T, R, rng)
    st = {"N": N, "R": R, "triples": tri}
    for order, fn_ in (("train", graph_train_order), ("search", graph_search_order)):
        g = fn_(N, R, tri)
        s, d, _ = g.edges(form="all")
        st.update({f"{order}/src": s, f"{order}/dst": d, f"{order}/etype": g.edata["e_type"], f"{order}/norm": g.edata["norm"]})
    np.savez_compressed(os.path.join(OUT, f"graph_{name}.npz"), **npify(st))
    print("wrote graph_%s" % name)


# --------------------------------------------------------------------------
# data preparation either side of the hot path: sampler, negative sampling, labels, ranking
# --------------------------------------------------------------------------
def case_sampling(name, Nall, T, R, sample, neg, seed):
    """Reference utils/utils_rgcn.py:79-118 and :191-204 with numpy's global generator seeded; the four random draws
    are replayed (same seed, same call order) and stored so the device functions can be checked draw for draw."""
    import utils.utils_rgcn as ur
    rng = np.random.default_rng(seed)
    tri = make_triples(Nall, T, R, rng, dup=0)
    adj, deg = ur.get_adj_and_degrees(Nall, tri)
    np.random.seed(seed)
    with np.errstate(divide="ignore"):
        g, uniq_v, src_o, rel, node_norm, samples, labels = ur.generate_sampled_graph_and_labels(tri, sample, 0.5, R, adj, deg, neg, "uniform")
    np.random.seed(seed)                                   # replay: choice(edges) -> randint(values) -> uniform(choices) -> choice(split)
    edges = np.random.choice(np.arange(len(tri)), sample, replace=False)
    values = np.random.randint(len(uniq_v), size=sample * neg)
    choices = np.random.uniform(size=sample * neg)
    split = np.random.choice(np.arange(sample), size=int(sample * 0.5), replace=False)
    e = tri[edges]
    uv, inv = np.unique((e[:, 0], e[:, 2]), return_inverse=True)
    assert np.array_equal(uv, uniq_v) and np.array_equal(np.stack((inv.reshape(2, -1)[0], e[:, 1], inv.reshape(2, -1)[1])).T, samples[:sample])
    s_, d_, _ = g.edges(form="all")
    st = {"Nall": Nall, "R": R, "sample": sample, "neg": neg, "triples": tri, "draw_edges": edges, "draw_values": values,
          "draw_choices": choices, "draw_split": split, "uniq_v": uniq_v, "src_o": src_o, "rel": rel, "node_norm": node_norm,
          "samples": samples, "labels": labels, "g_src": s_, "g_dst": d_}
    # negative_sampling on its own, another rate
    pos = samples[:sample]
    np.random.seed(seed + 1)
    s2, l2 = ur.negative_sampling(pos, len(uniq_v), 3)
    np.random.seed(seed + 1)
    st.update({"ns_pos": pos, "ns_values": np.random.randint(len(uniq_v), size=sample * 3), "ns_choices": np.random.uniform(size=sample * 3),
               "ns_samples": s2, "ns_labels": l2, "ns_num_entity": len(uniq_v)})
    np.savez_compressed(os.path.join(OUT, f"sampling_{name}.npz"), **npify(st))
    print("wrote sampling_%s: n=%d samples=%d" % (name, len(uniq_v), len(samples)))


def case_sampling_neighbor(name, Nall, T, R, sample, neg, seed):
    """Reference utils/utils_rgcn.py:30-71 (`sampler="neighbor"`) with numpy's global generator seeded.  Its draws are adaptive
    (a rejection loop per pick), so they are recorded by REPLAYING the loop with the same numpy calls after re-seeding: the
    uniform behind every np.random.choice(n, p=...) (legacy choice draws one random_sample() and searches the normalised cumsum)
    and every adjacency slot np.random.choice(np.arange(m)) tried (legacy choice without p = randint(0, m)).  The replay is
    asserted to reproduce the reference's edges, then the remaining three draws are read off the same stream."""
    import utils.utils_rgcn as ur
    rng = np.random.default_rng(seed)
    tri = make_triples(Nall, T, R, rng, dup=0)
    adj, deg = ur.get_adj_and_degrees(Nall, tri)
    np.random.seed(seed)
    ref_edges = ur.sample_edge_neighborhood(adj, deg, len(tri), sample)
    np.random.seed(seed)
    with np.errstate(divide="ignore"):
        g, uniq_v, src_o, rel, node_norm, samples, labels = ur.generate_sampled_graph_and_labels(tri, sample, 0.5, R, adj, deg, neg, "neighbor")
    # ---- replay with recorded draws
    np.random.seed(seed)
    counts, picked, seen = deg.copy(), np.zeros(len(tri), bool), np.zeros(Nall, bool)
    u_vertex, tries, edges = [], [], []
    for i in range(sample):
        w = counts * seen
        if w.sum() == 0:
            w = np.ones_like(w)
            w[np.where(counts == 0)] = 0
        p = w / np.sum(w)
        u = np.random.random_sample()
        cdf = p.cumsum()
        cdf /= cdf[-1]
        v = int(cdf.searchsorted(u, side="right"))
        u_vertex.append(u)
        seen[v] = True
        while True:
            t = int(np.random.randint(0, adj[v].shape[0]))
            tries.append(t)
            e, o = adj[v][t]
            if not picked[e]:
                break
        edges.append(e)
        picked[e] = True
        counts[v] -= 1
        counts[o] -= 1
        seen[o] = True
    assert np.array_equal(np.asarray(edges), ref_edges), "the recorded replay does not reproduce the reference's neighbourhood sample"
    values = np.random.randint(len(uniq_v), size=sample * neg)
    choices = np.random.uniform(size=sample * neg)
    split = np.random.choice(np.arange(sample), size=int(sample * 0.5), replace=False)
    s_, d_, _ = g.edges(form="all")
    st = {"Nall": Nall, "R": R, "sample": sample, "neg": neg, "triples": tri, "edges": ref_edges, "draw_u_vertex": np.asarray(u_vertex),
          "draw_tries": np.asarray(tries, dtype=np.int64), "draw_values": values, "draw_choices": choices, "draw_split": split,
          "uniq_v": uniq_v, "src_o": src_o, "rel": rel, "node_norm": node_norm, "samples": samples, "labels": labels, "g_src": s_, "g_dst": d_}
    np.savez_compressed(os.path.join(OUT, f"sampling_neighbor_{name}.npz"), **npify(st))
    print("wrote sampling_neighbor_%s: n=%d picks=%d tries=%d" % (name, len(uniq_v), sample, len(tries)))


def _install_train_driver_standins():
    """train/mr_lp_train.py imports three modules this image (and, for `dataloader`, the reference itself) lacks;
    none of them is touched by predict()."""
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    contrib = types.ModuleType("dgl.contrib")
    cdata = types.ModuleType("dgl.contrib.data")
    cdata.load_data = None
    contrib.data = cdata
    dl = types.ModuleType("dataloader")
    dl.get_dataset = None
    sys.modules.update({"tensorboardX": tbx, "dgl.contrib": contrib, "dgl.contrib.data": cdata, "dataloader": dl})
    sys.modules["dgl"].contrib = contrib
    sys.path.insert(0, os.path.join(REF, "train"))


def case_labels_and_ranking(name, N, T, R, B, seed):
    """process() + TrainDataset / TestDataset labels (reference utils/process_data.py:4-31, utils/data_set.py) and the
    filtered ranking of predict() (reference train/mr_lp_train.py:269-314) run as they are, with a stand-in model that
    returns prepared scores."""
    from utils.process_data import process
    from utils.data_set import TestDataset, TrainDataset
    _install_train_driver_standins()
    import mr_lp_train as TR
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    tri = make_triples(N, T, R, rng, dup=2)
    n_tr, n_va = int(T * 0.7), int(T * 0.15)
    ds = {"train": tri[:n_tr], "valid": tri[n_tr:n_tr + n_va], "test": tri[n_tr + n_va:]}
    trip = process(ds, R)
    params = types.SimpleNamespace(lbl_smooth=0.1)
    tr = TrainDataset(trip["train"], N, params)
    te = TestDataset(trip["test_tail"] + trip["test_head"], N, params)
    st = {"N": N, "R": R, "train": ds["train"], "valid": ds["valid"], "test": ds["test"]}
    idx = rng.choice(len(tr), size=min(B, len(tr)), replace=False)
    st["train_triples"] = torch.stack([tr[i][0] for i in idx])
    st["train_labels"] = torch.stack([tr[i][1] for i in idx])                 # label-smoothed, float32
    tidx = rng.choice(len(te), size=min(B, len(te)), replace=False)
    t_trip = torch.stack([te[i][0] for i in tidx])
    t_lab = torch.stack([te[i][1] for i in tidx])
    st["test_triples"], st["test_labels"] = t_trip, t_lab
    # predict(): prepared scores in (0, 1) without ties among the unfiltered entities
    pred = torch.sigmoid(torch.randn(len(tidx), N) * 2)
    st["pred"] = pred

    class FakeModel:
        def eval(self):
            pass

        def __call__(self, g, subj, rel):
            return self.rows.pop(0)

    model = FakeModel()
    bs = 16
    loader = [(t_trip[i:i + bs], t_lab[i:i + bs]) for i in range(0, len(tidx), bs)]
    model.rows = [pred[i:i + bs].clone() for i in range(0, len(tidx), bs)]
    res, loss = TR.predict(loader, None, model, "cpu")
    for k, v in res.items():
        st["res/" + k] = np.float64(v)
    st["res_loss"] = np.float64(loss)
    ranks = []
    for i in range(len(tidx)):                                               # one row per call: results['mr'] is that row's rank
        model.rows = [pred[i:i + 1].clone()]
        r1, _ = TR.predict([(t_trip[i:i + 1], t_lab[i:i + 1])], None, model, "cpu")
        ranks.append(int(r1["mr"]))
    st["ranks"] = np.asarray(ranks, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"labels_ranking_{name}.npz"), **npify(st))
    print("wrote labels_ranking_%s: train pairs=%d test rows=%d mrr=%.4f" % (name, len(tr), len(te), res["mrr"] / res["count"]))


def main():
    _install_standins()
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    case_ops("tiny_train", 37, 101, 5, 8, "train", 1)
    case_ops("small_search", 50, 200, 7, 16, "search", 2)
    case_ops("mid_train", 120, 330, 11, 40, "train", 3, star_only=True)
    case_ops("d100_search", 60, 45, 6, 100, "search", 4, star_only=True)
    case_ops("odd_train", 41, 90, 4, 10, "train", 5)          # D not a multiple of 4
    case_compgcn("small", 45, 160, 6, 12, 20, 11)
    case_fixed_net("tiny", 37, 101, 5, 8, 6, 4, 21)
    case_fixed_net("d64", 50, 180, 7, 64, 16, 5, 22)
    case_supernet("tiny", 80, 400, 5, 8, 6, 11, 2, 60, 31)
    case_supernet("d24", 150, 900, 9, 24, 12, 19, 2, 120, 32)
    case_graph_build("small", 64, 300, 6, 41)
    # the two shapes SURVEY 8(c) planned and round 1 skipped (inputs / parameters rebuilt from seeds, not stored)
    case_ops("r300_d64_search", 300, 2000, 11, 64, "search", 6, star_only=True, seeded_inputs=True, skip=("pre_mult", "pre_sub"))
    case_supernet("d200_sampled", 14541, 60000, 237, 200, 100, 475, 2, 300, 33, seeded_params=True)
    case_sampling("small", 400, 3000, 7, 200, 10, 51)
    case_sampling_neighbor("small", 400, 3000, 7, 200, 10, 53)
    case_sampling_neighbor("dense", 60, 900, 5, 600, 2, 54)        # two thirds of all triples: long rejection runs, exhausted vertices
    case_labels_and_ranking("small", 150, 1200, 5, 64, 52)


if __name__ == "__main__":
    main()
