"""Oracle graph container, graph builders and DGL-reducer semantics (CPU, test-only).

See oracle/__init__.py for what is pinned and what is not.
"""
import numpy as np
import torch


class OGraph:
    """Plain edge-list graph: what the reference's operators read off a DGLGraph.

    ``src``/``dst``/``etype`` are int64 [E] in the CALLER's edge order (first
    E/2 original-direction edges, then E/2 inverse edges); ``norm`` is float32
    [E] or [E,1].
    """

    def __init__(self, n, src, dst, etype, norm):
        self.n = int(n)
        self.src = torch.as_tensor(src, dtype=torch.long)
        self.dst = torch.as_tensor(dst, dtype=torch.long)
        self.etype = torch.as_tensor(etype, dtype=torch.long)
        self.norm = torch.as_tensor(norm, dtype=torch.float32)

    def to(self, device=None, dtype=None):
        """Copy on another device / with the norm in another float type: the GPU tests run this same
        restatement in float64 on the device as the full-size checker (tests only)."""
        g = OGraph.__new__(OGraph)
        g.n = self.n
        g.src, g.dst, g.etype = (t.to(device) for t in (self.src, self.dst, self.etype))
        g.norm = self.norm.to(device=device, dtype=dtype or self.norm.dtype)
        return g

    @property
    def E(self):
        return int(self.src.numel())

    def in_degree(self):
        return torch.bincount(self.dst, minlength=self.n)


def _deg_norm(in_deg):
    # reference train/mr_lp_train.py:81-83 and utils/utils_rgcn.py:120-127:
    # norm = in_deg ** -0.5 with inf -> 0, computed in float32 by numpy
    with np.errstate(divide="ignore"):
        norm = in_deg.astype(np.float32) ** np.float32(-0.5)
    norm[np.isinf(norm)] = 0
    return norm.astype(np.float32)


def build_train_graph(n, num_rels, triples):
    """Reference train/mr_lp_train.py:77-89: both directions, un-sorted halves,
    edge norm = d_in(dst)^-1/2 * d_in(src)^-1/2 as a 1-D [E] tensor."""
    t = np.asarray(triples, dtype=np.int64)
    src = np.concatenate([t[:, 0], t[:, 2]])
    dst = np.concatenate([t[:, 2], t[:, 0]])
    etype = np.concatenate([t[:, 1], t[:, 1] + num_rels])
    nn_ = _deg_norm(np.bincount(dst, minlength=n))
    norm = nn_[dst] * nn_[src]
    return OGraph(n, src, dst, etype, norm)


def build_search_graph(n, num_rels, triples):
    """Reference utils/utils_rgcn.py:129-158 (inverse edges appended, then
    ``sorted(zip(rel, dst, src))``) and search/mr_lp_search.py:30-36
    (edge norm = node_norm[dst] * node_norm[src], shape [E,1])."""
    t = np.asarray(triples, dtype=np.int64)
    src = np.concatenate([t[:, 0], t[:, 2]])
    dst = np.concatenate([t[:, 2], t[:, 0]])
    rel = np.concatenate([t[:, 1], t[:, 1] + num_rels])
    order = np.lexsort((src, dst, rel))          # primary key rel, then dst, then src
    src, dst, rel = src[order], dst[order], rel[order]
    nn_ = _deg_norm(np.bincount(dst, minlength=n))
    norm = (nn_[dst] * nn_[src]).reshape(-1, 1)
    return OGraph(n, src, dst, rel, norm)


# ---------------------------------------------------------------------------
# DGL reducers (third-party, restated from documented semantics -- UNPINNED)
# call sites: reference models/operations_lp.py:233,248,262; models/compgcn.py:87
# ---------------------------------------------------------------------------
class _SegMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, dst, n):
        E, D = m.shape
        idx = dst.view(-1, 1).expand(E, D)
        h = torch.zeros(n, D, dtype=m.dtype, device=m.device).scatter_reduce(0, idx, m, reduce="amax", include_self=False)
        eid = torch.arange(E, device=m.device).view(-1, 1).expand(E, D)
        cand = torch.where(m == h[dst], eid, torch.full_like(eid, E))
        arg = torch.full((n, D), E, dtype=torch.long, device=m.device).scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
        ctx.save_for_backward(arg)
        ctx.E = E
        ctx.mark_non_differentiable(arg)
        return h, arg

    @staticmethod
    def backward(ctx, g, _):
        (arg,) = ctx.saved_tensors
        gm = torch.zeros(ctx.E + 1, arg.shape[1], dtype=g.dtype, device=g.device)
        gm.scatter_(0, arg, g)
        return gm[: ctx.E], None, None


def seg_max(m, dst, n, return_arg=False):
    """h[v] = max_{e: dst(e)=v} m[e]; rows without in-edges are 0; backward to
    the lowest-numbered arg-max edge.  arg == E marks 'no in-edge'."""
    h, arg = _SegMax.apply(m, dst, n)
    return (h, arg) if return_arg else h


def seg_sum(m, dst, n):
    return torch.zeros(n, m.shape[1], dtype=m.dtype, device=m.device).index_add(0, dst, m)


def seg_mean(m, dst, n):
    deg = torch.bincount(dst, minlength=n).clamp(min=1).to(m.dtype)
    return seg_sum(m, dst, n) / deg.view(-1, 1)
