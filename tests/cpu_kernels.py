"""CPU stand-ins for the HIP kernel namespace, for the multi-process (gloo) tests of the
sharding / collective logic ONLY.  Test infrastructure: never imported by the product.
The modules follow the oracle's arithmetic but honour a shard's local (b0, b1) split."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from mr_gnas_amd.functional import GatherPlan  # pure tensor code, device-agnostic  # noqa: F401
from oracle import graph as OG


def gather(table, plan):
    return table[plan.idx32.long()]


def linear(x, W, b=None, act=None):
    y = F.linear(x, W, b)
    return F.relu(y) if act == "relu" else y


def score_all(score_func, all_ent, sub, rel):
    """sf_DisMult_op on CPU rows (reference models/operations_lp.py:115-127): sigmoid((sub * rel) all_ent^T)."""
    assert type(score_func).__name__ == "sf_DisMult_op"
    return torch.sigmoid(torch.mm(sub * rel, all_ent.t()))


def seg_reduce(kind, msg, self_rows, graph):
    _, dst, _ = graph.edges(form="all")
    n = graph.number_of_nodes()
    h = {"sum": OG.seg_sum, "mean": OG.seg_mean, "max": OG.seg_max}[kind](msg, dst, n)
    return h if self_rows is None else h + self_rows


def _split(g, a, b):
    b0, b1 = g.bounds()
    return (a[:b0], b[:b0]), (a[b0:b1], b[b0:b1]), (a[b1:], b[b1:])


def _tail(g, e_in, e_out, x_self, self_scale):
    e = torch.cat((e_in, e_out), 0) * (1 / 3) * g.norm_flat().view(-1, 1)
    return torch.cat((e, x_self * self_scale), 0)


class _Pre(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f

    def forward(self, g, a, hr):
        return self.f(a, hr)


class _Zero(nn.Module):
    def forward(self, g, a, b):
        return 0 * a


class _Id(nn.Module):
    def forward(self, g, a, b):
        return a


class _Comp(nn.Module):
    def __init__(self, D, kind):
        super().__init__()
        self.kind = kind
        for x in ("in", "out", "self"):
            setattr(self, "W_" + x, nn.Linear(2 * D, D, bias=kind != "comp"))
            if kind == "sparse":
                setattr(self, "a_" + x, nn.Linear(D, 1, bias=False))

    def forward(self, g, a, b):
        outs = []
        for (s, s_in), x in zip(_split(g, a, b), ("in", "out", "self")):
            z = getattr(self, "W_" + x)(torch.cat([s, s_in], 1))
            if self.kind == "sparse":
                z = getattr(self, "a_" + x)(z)
            outs.append(z if self.kind == "comp" else torch.sigmoid(z) * s)
        return _tail(g, *outs, 1.0 if self.kind == "comp" else 1 / 3)


class _Last(nn.Module):
    def __init__(self, D, sparse):
        super().__init__()
        self.W = nn.Linear(D, D)
        if sparse:
            self.a = nn.Linear(D, 1, bias=False)

    def forward(self, g, a, b):
        z = self.W(a)
        if hasattr(self, "a"):
            z = self.a(z)
        return torch.sigmoid(z) * a


class _Agg(nn.Module):
    def __init__(self, D, kind, p):
        super().__init__()
        self.kind = kind
        if kind == "sum":
            self.drop_sum = nn.Dropout(p)
        else:
            self.linear = nn.Linear(D, D)

    def forward(self, g, a, b):
        E = g.num_edges()
        m = a[:E] if self.kind == "sum" else F.relu(self.linear(a[:E]))
        h = seg_reduce(self.kind, m, None, g)
        return (self.drop_sum(h) if self.kind == "sum" else h) + a[E:]


def registry():
    D = lambda a: a.get("feature_dim", 100)
    return {
        "pre_mult": lambda a: _Pre(torch.mul), "pre_sub": lambda a: _Pre(torch.sub), "pre_add": lambda a: _Pre(torch.add),
        "f_zero": lambda a: _Zero(), "f_identity": lambda a: _Id(),
        "f_dense_comp": lambda a: _Comp(D(a), "dense"), "f_sparse_comp": lambda a: _Comp(D(a), "sparse"),
        "f_comp": lambda a: _Comp(D(a), "comp"),
        "f_dense_last": lambda a: _Last(D(a), False), "f_sparse_last": lambda a: _Last(D(a), True),
        "a_max": lambda a: _Agg(D(a), "max", 0.0), "a_mean": lambda a: _Agg(D(a), "mean", 0.0),
        "a_sum": lambda a: _Agg(D(a), "sum", a.get("drop_aggr", 0.0)),
    }
