"""Drop-in operator zoo for MR-GNAS link prediction, MI355X-native.

Importable in place of the reference's ``models/operations_lp.py``: the same
registries ``MIXED_OPS`` / ``MIXED_OPS_sf`` (``name -> constructor(args dict)``,
reference :8-30), the same op-name lists in the same order (reference :32-37;
the order defines the alpha columns and the genotype decoding), the same
``forward(g, src_emb, src_emb_in)`` signatures and the same parameter names, so
reference checkpoints load and ``cell_lp.MixedOp`` / ``model_lp.OpModule`` call
into it unchanged.  ``g`` is a ``mr_gnas_amd.RelGraph`` (DGL is not available on
ROCm).

Hot-path operators (compose, sparse gates, aggregators) run in the HIP kernels
of libmrgnas_hip.so through ``functional``; there is no CPU fallback.  The
reference's side effect of leaving ``msg_e`` / ``h`` on the graph object
(reference :232-233) is dropped on purpose: the fused path never materialises
them.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as K
from . import lazy as LZ

LZ.install_indexing()      # table[idx] of the callers (the gather G feeding the path) -> Gather handles; MRG_FAST_INDEX=0 / MRG_LAZY=0 opt out

PRE_OPS = ['pre_mult', 'pre_sub', 'pre_add']
FIRST_OPS = ['f_zero', 'f_identity', 'f_dense_comp', 'f_sparse_comp', 'f_comp']
MIDDLE_OPS = ['a_max', 'a_sum', 'a_mean']
LAST_OPS = ['f_zero', 'f_identity', 'f_dense_last', 'f_sparse_last']
SF_OPS = ['sf_TransE', 'sf_DisMult']


def _bounds(g):
    if hasattr(g, "bounds"):
        return g.bounds()
    E = g.num_edges()
    return E // 2, E


class _Operator(nn.Module):
    """Base of the graph operators.  ``forward(g, src_emb, src_emb_in)`` -- what the reference's callers invoke
    (models/cell_lp.py:30, models/model_lp.py:28) -- returns a LAZY HANDLE for HIP operands (mr_gnas_amd/lazy.py: a torch.Tensor
    subclass that is evaluated when something other than the reference's BatchNorm -> ReLU -> w * . -> sum chain touches it, and
    that lets that chain run as the fused MixedOp path); ``run(...)`` computes the result now (this package's own fused callers,
    and the handle's evaluation).  With MRG_LAZY=0, or for CPU tensors, ``forward`` is ``run``."""

    def out_shape(self, g, src_emb):
        return tuple(src_emb.shape)

    def forward(self, g, src_emb, src_emb_in, for_epilogue=False):
        if for_epilogue:
            return self.run(g, src_emb, src_emb_in, for_epilogue=True)
        if LZ.wanted(src_emb):
            return LZ.defer(self, g, src_emb, src_emb_in, self.out_shape(g, src_emb))
        return self.run(g, src_emb, src_emb_in)


# ---- a1: compose -------------------------------------------------------------
class _PreOp(_Operator):
    kind = None

    def run(self, g, src_emb, hr, for_epilogue=False):
        return K.compose(self.kind, LZ.real(src_emb), LZ.real(hr))


class pre_mult_op(_PreOp):
    kind = "mult"


class pre_sub_op(_PreOp):
    kind = "sub"


class pre_add_op(_PreOp):
    kind = "add"


# ---- trivial filters -----------------------------------------------------------
class f_identity_op(_Operator):
    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        return LZ.real(src_emb)


class f_zero_op(_Operator):
    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        return 0 * LZ.real(src_emb)


# ---- a2 / a3: sparse (scalar-gate) filters ---------------------------------------
class f_sparse_op_comp(_Operator):
    """Per-direction scalar gate sigmoid(a_x(W_x[s ; s_in])) * s * 1/3 (* norm on
    edge rows); one fused HIP pass instead of the reference's ~22 launches."""

    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        for x in ("in", "out", "self"):
            setattr(self, "W_" + x, nn.Linear(2 * D, D, bias=True))
            setattr(self, "a_" + x, nn.Linear(D, 1, bias=False))

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        """for_epilogue: the result goes to functional.mixed_epilogue and nowhere else -- it may then be the gate as a ROW FACTOR
        fvec [rows] (the candidate is src_emb * fvec[:, None], which the epilogue recomputes instead of reading it back)."""
        src_emb, src_emb_in = LZ.real(src_emb), LZ.real(src_emb_in)
        b0, b1 = _bounds(g)
        p = []
        for x in ("in", "out", "self"):
            W, a = getattr(self, "W_" + x), getattr(self, "a_" + x)
            p += [W.weight, W.bias, a.weight]
        if for_epilogue and K.switches.ROW_FACTOR and src_emb.is_cuda:
            return K.gate_comp_row_factor(src_emb, src_emb_in, g.norm_flat(), b0, b1, *p)
        return K.gate_comp(src_emb, src_emb_in, g.norm_flat(), b0, b1, *p)


class f_sparse_op_last(_Operator):
    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W = nn.Linear(D, D, bias=True)
        self.a = nn.Linear(D, 1, bias=False)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        return K.gate_last(LZ.real(src_emb), self.W.weight, self.W.bias, self.a.weight)


class f_sparse_op(_Operator):
    """Registered by the reference but in none of its op lists (:290-301)."""

    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W = nn.Linear(2 * D, D, bias=True)
        self.a = nn.Linear(D, 1, bias=False)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        src_emb, src_emb_in = LZ.real(src_emb), LZ.real(src_emb_in)
        none3 = [None, None, None]
        return K._Gate.apply(src_emb, src_emb_in, None, 0, 0, 1.0, *none3, *none3, self.W.weight, self.W.bias, self.a.weight)


# ---- dense (per-feature) filters: MFMA row GEMM with the gate / scale fused in its epilogue -----
class f_dense_op_comp(_Operator):
    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W_in = nn.Linear(2 * D, D, bias=True)
        self.W_out = nn.Linear(2 * D, D, bias=True)
        self.W_self = nn.Linear(2 * D, D, bias=True)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        """for_epilogue: the result goes to functional.mixed_epilogue and nowhere else -- a functional.Candidate whose Link lets the
        epilogue's gradient store perform this operator's first backward pass (cell_lp.MixedOp asks for it)."""
        src_emb, src_emb_in = LZ.real(src_emb), LZ.real(src_emb_in)
        b0, b1 = _bounds(g)
        return K.dense_filter_comp(0, src_emb, src_emb_in, g.norm_flat(), b0, b1, self.W_in.weight, self.W_in.bias,
                                   self.W_out.weight, self.W_out.bias, self.W_self.weight, self.W_self.bias, 1.0 / 3.0,
                                   for_epilogue=for_epilogue and src_emb.is_cuda)


class f_comp_op(_Operator):
    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W_in = nn.Linear(2 * D, D, bias=False)
        self.W_out = nn.Linear(2 * D, D, bias=False)
        self.W_self = nn.Linear(2 * D, D, bias=False)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        src_emb, src_emb_in = LZ.real(src_emb), LZ.real(src_emb_in)
        b0, b1 = _bounds(g)       # self rows are NOT scaled (reference :285-287)
        return K.dense_filter_comp(1, src_emb, src_emb_in, g.norm_flat(), b0, b1, self.W_in.weight, None,
                                   self.W_out.weight, None, self.W_self.weight, None, 1.0, for_epilogue=for_epilogue and src_emb.is_cuda)


def dense_pair_forward(op_dense, op_comp, g, src_emb, src_emb_in, for_epilogue=False):
    """(f_dense_comp(g, src_emb, src_emb_in), f_comp(g, src_emb, src_emb_in)) of one MixedOp as one autograd node when the
    shapes allow (functional.dense_filter_pair), else the two operators on their own.  for_epilogue: both results go to
    functional.mixed_epilogue and nowhere else -- two functional.Candidate values; f_dense_comp's may then be its GATE, which the
    epilogue recomputes the output from (functional.switches.GATED_RECOMPUTE)."""
    src_emb, src_emb_in = LZ.real(src_emb), LZ.real(src_emb_in)
    D = src_emb.shape[1]
    tied = src_emb_in is not None and K.same_rows(src_emb, src_emb_in)
    if not (src_emb.is_cuda and K.dense_pair_available(D, tied)):
        return op_dense.run(g, src_emb, src_emb_in, for_epilogue=for_epilogue), op_comp.run(g, src_emb, src_emb_in, for_epilogue=for_epilogue)
    b0, b1 = _bounds(g)
    dp = (op_dense.W_in.weight, op_dense.W_in.bias, op_dense.W_out.weight, op_dense.W_out.bias, op_dense.W_self.weight, op_dense.W_self.bias)
    cw = (op_comp.W_in.weight, op_comp.W_out.weight, op_comp.W_self.weight)
    return K.dense_filter_pair(src_emb, src_emb_in, g.norm_flat(), b0, b1, dp, cw, gate_only=for_epilogue and K.switches.GATED_RECOMPUTE,
                               for_epilogue=for_epilogue)


class f_dense_op(_Operator):
    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W = nn.Linear(2 * D, D, bias=True)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        return K.dense_filter_single(LZ.real(src_emb), LZ.real(src_emb_in), self.W.weight, self.W.bias)


class f_dense_op_last(_Operator):
    def __init__(self, args):
        super().__init__()
        D = self._feature_dim = args.get('feature_dim', 100)
        self.W = nn.Linear(D, D, bias=True)

    def run(self, g, src_emb, src_emb_in, for_epilogue=False):
        return K.dense_filter_single(LZ.real(src_emb), None, self.W.weight, self.W.bias)


# ---- a4 / a5 / a6: aggregators ---------------------------------------------------------
class _LinReluAgg(_Operator):
    kind = None

    def __init__(self, args):
        super().__init__()
        D = args.get('feature_dim', 100)
        self.linear = nn.Linear(D, D)

    def out_shape(self, block, src_emb):
        return (block.number_of_nodes(), int(src_emb.shape[1]))

    def run(self, block, src_emb, src_emb_in, for_epilogue=False):
        return K.linear_relu_aggregate(self.kind, LZ.real(src_emb), self.linear.weight, self.linear.bias, block)


class a_max_op(_LinReluAgg):
    kind = "max"


class a_mean_op(_LinReluAgg):
    kind = "mean"


class a_sum_op(_Operator):
    def __init__(self, args):
        super().__init__()
        self.drop_aggr = args.get('drop_aggr', 0.1)
        self.drop_sum = nn.Dropout(self.drop_aggr)

    def out_shape(self, block, src_emb):
        return (block.number_of_nodes(), int(src_emb.shape[1]))

    def run(self, block, src_emb, src_emb_in, for_epilogue=False):
        src_emb = LZ.real(src_emb)
        keep = None
        if self.training and self.drop_aggr > 0:      # nn.Dropout semantics: keep-mask scaled by 1 / (1 - p), on the [N, D] sums
            N, D = block.number_of_nodes(), src_emb.shape[1]
            keep = self.drop_sum(torch.ones(N, D, dtype=torch.float32, device=src_emb.device))
        return K.aggregate_rows("sum", src_emb, block, add_self=True, keep=keep)


# ---- score functions (the step after the path): HIP kernels as well (compose + MFMA GEMM with a sigmoid epilogue; L1 kernel) ----
class sf_TransE_op(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.gamma = args.get('gamma', 40)

    def forward(self, all_ent, sub_emb, rel_emb):
        return K.transe_scores_all(all_ent, sub_emb, rel_emb, self.gamma)


class sf_DisMult_op(nn.Module):
    def __init__(self, args):
        super().__init__()

    def forward(self, all_ent, sub_emb, rel_emb):
        return K.distmult_scores_all(all_ent, sub_emb, rel_emb)


class sf_ConvE_op(nn.Module):
    """ConvE scorer: 2-D conv over the stacked (subject, relation) embedding."""

    def __init__(self, args):
        super().__init__()
        g = args.get
        self.embed_dim = g('embed_dim', 200)
        self.conve_hid_drop, self.feat_drop = g('conve_hid_drop', 0.3), g('feat_drop', 0.3)
        self.num_filt, self.ker_sz, self.k_w, self.k_h = g('num_filt', 200), g('ker_sz', 7), g('k_w', 10), g('k_h', 20)
        self.bn0, self.bn1, self.bn2 = nn.BatchNorm2d(1), nn.BatchNorm2d(self.num_filt), nn.BatchNorm1d(self.embed_dim)
        self.feature_drop, self.hidden_drop = nn.Dropout(self.feat_drop), nn.Dropout(self.conve_hid_drop)
        self.conv2d = nn.Conv2d(1, self.num_filt, (self.ker_sz, self.ker_sz), stride=1, padding=0, bias=True)
        self.flat_sz = (2 * self.k_h - self.ker_sz + 1) * (self.k_w - self.ker_sz + 1) * self.num_filt
        self.fc = nn.Linear(self.flat_sz, self.embed_dim)

    def forward(self, all_ent, sub_emb, rel_emb):
        if self.embed_dim != self.k_h * self.k_w:
            raise AssertionError("embed_dim must equal k_h * k_w")
        x = torch.stack([sub_emb, rel_emb], dim=1).reshape(-1, 1, 2 * self.k_h, self.k_w)
        x = self.feature_drop(F.relu(self.bn1(self.conv2d(self.bn0(x)))))
        x = F.relu(self.bn2(self.hidden_drop(self.fc(x.flatten(1)))))
        return torch.sigmoid(x @ all_ent.t())


MIXED_OPS = {
    'pre_mult': lambda args: pre_mult_op(),
    'pre_sub': lambda args: pre_sub_op(),
    'pre_add': lambda args: pre_add_op(),
    'f_zero': lambda args: f_zero_op(),
    'f_identity': lambda args: f_identity_op(),
    'f_dense': lambda args: f_dense_op(args),
    'f_dense_comp': lambda args: f_dense_op_comp(args),
    'f_comp': lambda args: f_comp_op(args),
    'f_sparse': lambda args: f_sparse_op(args),
    'f_sparse_comp': lambda args: f_sparse_op_comp(args),
    'f_dense_last': lambda args: f_dense_op_last(args),
    'f_sparse_last': lambda args: f_sparse_op_last(args),
    'a_max': lambda args: a_max_op(args),
    'a_mean': lambda args: a_mean_op(args),
    'a_sum': lambda args: a_sum_op(args),
}

MIXED_OPS_sf = {
    'sf_TransE': lambda args: sf_TransE_op(args),
    'sf_DisMult': lambda args: sf_DisMult_op(args),
    'sf_ConvE': lambda args: sf_ConvE_op(args),
}
