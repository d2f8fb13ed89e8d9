"""Oracle operators: functional float32 restatement of reference
models/operations_lp.py (CPU, test-only; see oracle/__init__.py).

Every graph operator has the shape ``op(g, P, a, b) -> Tensor`` where ``g`` is
an ``oracle.graph.OGraph``, ``P`` maps the reference's parameter names
(``"W_in.weight"``, ``"a_in.weight"``, ``"linear.bias"`` ...) to tensors, ``a``
is the reference's ``src_emb`` and ``b`` its ``src_emb_in`` (``hr`` for the
pre-ops).  Row layout of every [M, D] tensor, M = E + N: rows [0, E/2) original
direction edges, [E/2, E) inverse edges, [E, M) one self-loop row per node.
"""
import torch
import torch.nn.functional as F

from .graph import seg_max, seg_mean, seg_sum

# reference models/operations_lp.py:32-37 -- order defines alpha columns
PRE_OPS = ["pre_mult", "pre_sub", "pre_add"]
FIRST_OPS = ["f_zero", "f_identity", "f_dense_comp", "f_sparse_comp", "f_comp"]
MIDDLE_OPS = ["a_max", "a_sum", "a_mean"]
LAST_OPS = ["f_zero", "f_identity", "f_dense_last", "f_sparse_last"]
SF_OPS = ["sf_TransE", "sf_DisMult"]


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _thirds(g, x_edges_in, x_edges_out, x_self, self_scale=1 / 3):
    # reference models/operations_lp.py:339-342 (same tail in :386-389; f_comp's
    # tail :285-287 leaves the self rows UNscaled)
    e = torch.cat((1 / 3 * x_edges_in, 1 / 3 * x_edges_out), dim=0) * g.norm.view(-1, 1)
    return torch.cat((e, self_scale * x_self), dim=0)


def _split(g, a, b):
    h, E = g.E // 2, g.E
    return (a[:h], b[:h]), (a[h:E], b[h:E]), (a[E:], b[E:])


# --- a1: compose ops, reference models/operations_lp.py:71-98 ---------------
def pre_mult(g, P, a, hr):
    return a * hr


def pre_sub(g, P, a, hr):
    return a - hr


def pre_add(g, P, a, hr):
    return a + hr


# --- filters on [M, D] -------------------------------------------------------
def f_zero(g, P, a, b):
    # reference models/operations_lp.py:214-220
    return 0 * a


def f_identity(g, P, a, b):
    # reference models/operations_lp.py:204-210
    return a


def f_sparse_comp(g, P, a, b):
    """a2 -- reference models/operations_lp.py:304-343: per direction a scalar
    gate sigmoid(a_x(W_x[s ; s_in])) times s, times 1/3, edge rows times norm."""
    outs = []
    for (s, s_in), x in zip(_split(g, a, b), ("in", "out", "self")):
        gate = _lin(P, "a_" + x, _lin(P, "W_" + x, torch.cat([s, s_in], dim=1)))
        outs.append(torch.sigmoid(gate) * s)
    return _thirds(g, *outs)


def f_dense_comp(g, P, a, b):
    """reference models/operations_lp.py:356-390: as f_sparse_comp with a
    per-feature gate sigmoid(W_x[s ; s_in])."""
    outs = []
    for (s, s_in), x in zip(_split(g, a, b), ("in", "out", "self")):
        outs.append(torch.sigmoid(_lin(P, "W_" + x, torch.cat([s, s_in], dim=1))) * s)
    return _thirds(g, *outs)


def f_comp(g, P, a, b):
    """reference models/operations_lp.py:266-288: W_x[s ; s_in] (no bias, no gate);
    edge rows * 1/3 * norm, self rows as they are (:285-287)."""
    outs = [_lin(P, "W_" + x, torch.cat([s, s_in], dim=1))
            for (s, s_in), x in zip(_split(g, a, b), ("in", "out", "self"))]
    return _thirds(g, *outs, self_scale=1.0)


def f_sparse(g, P, a, b):
    # reference models/operations_lp.py:290-301 (registered, not in any op list)
    return torch.sigmoid(_lin(P, "a", _lin(P, "W", torch.cat([a, b], dim=1)))) * a


def f_dense(g, P, a, b):
    # reference models/operations_lp.py:345-354 (registered, not in any op list)
    return torch.sigmoid(_lin(P, "W", torch.cat([a, b], dim=1))) * a


# --- filters on [N, D] -------------------------------------------------------
def f_sparse_last(g, P, a, b):
    # a3 -- reference models/operations_lp.py:405-416 (src_emb_in ignored)
    return torch.sigmoid(_lin(P, "a", _lin(P, "W", a))) * a


def f_dense_last(g, P, a, b):
    # reference models/operations_lp.py:392-401
    return torch.sigmoid(_lin(P, "W", a)) * a


# --- aggregators [M, D] -> [N, D] ---------------------------------------------
# Test instrumentation ("mask replay", tests/test_configs_gpu.py): AGG_HOOK(kind, P, lin, g) -> aggregated [n, D] tensor or None
# lets a float64 run take another run's decisions inside a_max / a_mean (which edge wins the maximum, which messages the inner
# ReLU passes) instead of its own.  None (the default) = the reference's arithmetic.
AGG_HOOK = None


def a_max(g, P, a, b):
    # a4 -- reference models/operations_lp.py:223-235
    lin = _lin(P, "linear", a[: g.E])
    if AGG_HOOK is not None:
        h = AGG_HOOK("a_max", P, lin, g)
        if h is not None:
            return h + a[g.E:]
    return seg_max(F.relu(lin), g.dst, g.n) + a[g.E:]


def a_mean(g, P, a, b):
    # a6 -- reference models/operations_lp.py:238-250
    lin = _lin(P, "linear", a[: g.E])
    if AGG_HOOK is not None:
        h = AGG_HOOK("a_mean", P, lin, g)
        if h is not None:
            return h + a[g.E:]
    return seg_mean(F.relu(lin), g.dst, g.n) + a[g.E:]


def a_sum(g, P, a, b, drop_aggr=0.0, training=True):
    # a5 -- reference models/operations_lp.py:252-264
    h = seg_sum(a[: g.E], g.dst, g.n)
    return F.dropout(h, drop_aggr, training) + a[g.E:]


# --- score functions ----------------------------------------------------------
def sf_TransE(all_ent, sub, rel, gamma=40.0):
    # reference models/operations_lp.py:101-112
    obj = sub + rel
    return torch.sigmoid(gamma - torch.norm(obj.unsqueeze(1) - all_ent, p=1, dim=2))


def sf_DisMult(all_ent, sub, rel, gamma=None):
    # reference models/operations_lp.py:115-127
    return torch.sigmoid(torch.mm(sub * rel, all_ent.t()))


OPS = {
    "pre_mult": pre_mult, "pre_sub": pre_sub, "pre_add": pre_add,
    "f_zero": f_zero, "f_identity": f_identity, "f_dense": f_dense, "f_dense_comp": f_dense_comp,
    "f_comp": f_comp, "f_sparse": f_sparse, "f_sparse_comp": f_sparse_comp,
    "f_dense_last": f_dense_last, "f_sparse_last": f_sparse_last,
    "a_max": a_max, "a_mean": a_mean, "a_sum": a_sum,
}
SF = {"sf_TransE": sf_TransE, "sf_DisMult": sf_DisMult}


def param_shapes(name, D):
    """Parameter names/shapes of each operator (reference constructors,
    models/operations_lp.py:225-228, 267-272, 305-315, 357-365, 393-396, 406-410)."""
    lin = lambda n, i, o, bias=True: {n + ".weight": (o, i), **({n + ".bias": (o,)} if bias else {})}
    if name in ("f_sparse_comp",):
        d = {}
        for x in ("in", "out", "self"):
            d.update(lin("W_" + x, 2 * D, D)); d.update(lin("a_" + x, D, 1, False))
        return d
    if name == "f_dense_comp":
        d = {}
        for x in ("in", "out", "self"):
            d.update(lin("W_" + x, 2 * D, D))
        return d
    if name == "f_comp":
        d = {}
        for x in ("in", "out", "self"):
            d.update(lin("W_" + x, 2 * D, D, False))
        return d
    if name == "f_sparse":
        return {**lin("W", 2 * D, D), **lin("a", D, 1, False)}
    if name == "f_dense":
        return lin("W", 2 * D, D)
    if name == "f_sparse_last":
        return {**lin("W", D, D), **lin("a", D, 1, False)}
    if name == "f_dense_last":
        return lin("W", D, D)
    if name in ("a_max", "a_mean"):
        return lin("linear", D, D)
    return {}


def init_params(name, D, gen, xavier=True):
    """Reference callers apply utils/utils.py:121-125 (xavier-normal weights,
    zero bias) over default nn.Linear init."""
    P = {}
    for k, shp in param_shapes(name, D).items():
        if k.endswith(".bias"):
            P[k] = torch.zeros(shp)
        else:
            std = (2.0 / (shp[0] + shp[1])) ** 0.5
            P[k] = torch.randn(shp, generator=gen) * std
    return P
