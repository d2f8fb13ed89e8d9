"""Lazy candidate handles: what lets the reference's UNCHANGED ``models/cell_lp.py`` reach the fused MixedOp path.

The reference's MixedOp is written operator by operator (models/cell_lp.py:25-33)::

    output = sum(w * self.op_forward(op, g, h, h_in) for w, op in zip(weights, self._ops))
    def op_forward(self, op, g, h, h_in):
        nh = op[0](g, h, h_in)              # the operator (this package's)
        nh = op[1](nh.float())              # nn.BatchNorm1d
        nh = op[2](nh)                      # nn.ReLU

With only the operator registry swapped that is one operator launch, a torch BatchNorm, a ReLU, a scale and an add of ``[M, D]``
tensors per candidate, and no sharing between the candidates of a MixedOp: 262 ms per full-graph step against 49 ms for this
package's own ``cell_lp`` (VERDICT r4 "missing" #1).  SURVEY section 7 named the way out -- "a lazy edge-tensor handle".  This is it:

* an operator called through its ``forward`` returns a :class:`Lazy` -- a ``torch.Tensor`` SUBCLASS that carries the shape, dtype
  and device of the result and a description of how to compute it (``Op``: operator module + its arguments), but no storage;
* ``__torch_function__`` sees every torch call that receives one.  The calls of the reference's formulation only extend the
  description: ``.float()`` (f32 already: the handle itself), ``F.batch_norm`` -> ``Bn``, ``F.relu`` -> ``Act``, ``w * x`` ->
  ``Term``, ``0 + x`` / ``x + y`` (Python's ``sum``) -> ``Sum``; shape / dtype / device queries are answered from the metadata;
* ANY other call (``torch.cat`` of a cell's states, ``.backward()``, indexing, printing, a user's own arithmetic ...) first
  MATERIALISES the handles among its arguments and then runs on the real tensors -- the handle is observationally a tensor;
* materialising a ``Sum`` groups its terms by the operands their operators were called with: the terms of one MixedOp (same
  ``g``, ``h``, ``h_in``) run as ``cell_lp.fused_candidates`` -- the paired dense filters, the gate-only / row-factor candidates,
  one statistics pass and one combine pass for all of them, K-way gradient fan-in of the operands (``functional.Fan``), and the
  MixedOps that feed one state (models/cell_lp.py:103-107: ``s = sum(self._ops[...](...) ...)``) chained through the combine
  kernel's ``addend``; whatever does not fit the pattern (a foreign term, a BatchNorm without affine parameters ...) is evaluated
  literally, in the reference's own order.

* the gather G that feeds the path (``all_ent_emb[src_id_final]``, ``rel_embed[edge_type_final]``,
  ``torch.cat((ent_emb[src_in], ent_emb), dim=0)``: models/model_search_lp.py:144-145,153-154, models/model_lp.py:131) is plain
  tensor indexing in the caller, and torch's backward of it is a sort-based ``index_put_(accumulate=True)`` -- 11 ms per ``[M, D]``
  gather at the FB15k-237 shape, 67 ms of a step.  ``install_indexing()`` (run when ``operations_lp`` is imported; ``MRG_FAST_INDEX=0``
  opts out) wraps ``torch.Tensor.__getitem__``: ``table[idx]`` with a float32 ``[rows, D]`` HIP table and a 1-D integer HIP index of at
  least ``MIN_GATHER_ROWS`` entries returns a ``Gather`` handle; every other indexing expression goes to torch unchanged.  Cell zero's
  MixedOp reads such handles as ``functional.LazyRows`` (the three compose candidates, their statistics and gradients recomputed from
  the two tables: no ``[M, D]`` gather is ever written), ``torch.cat((table[idx], table), 0)`` stays one handle, DistMult's
  ``torch.sum(s * r * o, dim=1)`` over three handles (models/model_search_lp.py:169-176) runs as ``functional.distmult_score``,
  and anything else materialises the handle with this library's gather kernel (bit-exact rows) whose backward is a balanced,
  deterministic segmented sum instead of the sort-based accumulate.

Values: the fused path is the one ``tests/test_nets_gpu.py`` holds against the reference's golden outputs and gradients; the
handle adds no arithmetic of its own.  ``ENABLED`` (environment ``MRG_LAZY``, default on) switches the operators back to eager
results; CPU tensors are always eager (the CPU test registries have no fused path).

Reference counterparts: models/cell_lp.py:12-33 (MixedOp), :89-152 (the stages' ``sum`` over MixedOps), models/model_lp.py:27-35
(OpModule: the same operator -> BatchNorm -> ReLU chain with one candidate and weight 1).
"""
import os

import weakref

import torch
import torch.nn.functional as F

ENABLED = os.environ.get("MRG_LAZY", "1") == "1"
FAST_INDEX = os.environ.get("MRG_FAST_INDEX", "1") == "1"      # table[idx] -> Gather handles (install_indexing)
MIN_GATHER_ROWS = int(os.environ.get("MRG_MIN_GATHER_ROWS", "1024"))   # shorter index lists stay with torch
# Test switch (tests/test_dropin_cpu.py): hand out handles for CPU tensors too.  There is no fused path on the CPU, so every handle is
# evaluated literally, in the caller's own order -- which is exactly what the test wants: the REFERENCE's unchanged cell_lp.py /
# model_search_lp.py driving the handle protocol (every torch call they make on a handle), compared bit for bit with eager operators.
FORCE_CPU = False

# answered from the handle's metadata, never a reason to compute: methods ...
_META = {"dim", "size", "numel", "is_floating_point", "is_complex", "ndimension", "nelement", "element_size", "__len__", "get_device"}
# ... and property getters (func is the descriptor's __get__; the property's name is on its __self__)
_META_PROPS = {"shape", "dtype", "device", "ndim", "layout", "is_cuda", "is_cpu", "is_sparse", "is_quantized", "is_meta", "is_leaf", "requires_grad",
               "names", "is_nested", "is_mkldnn", "is_xpu", "is_mps", "is_xla", "is_ipu", "is_maia", "is_mtia", "is_vulkan", "is_sparse_csr"}


class Op:
    """operator(g, a, b) not yet run."""
    __slots__ = ("op", "g", "a", "b")

    def __init__(self, op, g, a, b):
        self.op, self.g, self.a, self.b = op, g, a, b


class Bn:
    """F.batch_norm(src, ...) not yet run; `view` carries what nn.BatchNorm1d.forward handed to F.batch_norm."""
    __slots__ = ("src", "view")

    def __init__(self, src, view):
        self.src, self.view = src, view


class Act:
    __slots__ = ("src",)

    def __init__(self, src):
        self.src = src


class Term:
    __slots__ = ("w", "src")

    def __init__(self, w, src):
        self.w, self.src = w, src


class Sum:
    __slots__ = ("parts",)

    def __init__(self, parts):
        self.parts = parts


class Gather:
    """table[idx] not yet run (the gather G feeding the path).  `then_table`: the caller wrote torch.cat((table[short], table), 0) and
    `idx` is cat(short, arange(rows)) -- kept so that a literal evaluation can do exactly what the caller wrote."""
    __slots__ = ("table", "idx", "then_table")

    def __init__(self, table, idx, then_table=None):
        self.table, self.idx, self.then_table = table, idx, then_table


class Prod:
    """Elementwise product of unmaterialised handles of one shape (DistMult's s * r * o)."""
    __slots__ = ("factors",)

    def __init__(self, factors):
        self.factors = factors


class Drop:
    """F.dropout(src, p, training=True) not yet run: the fixed-genotype OpModule computes one and throws it away
    (models/model_lp.py:34) -- as a handle it costs nothing unless somebody reads it."""
    __slots__ = ("src", "p")

    def __init__(self, src, p):
        self.src, self.p = src, p


class BatchNormView:
    """The arguments of one F.batch_norm call under the attribute names the fused epilogue reads from an nn.BatchNorm1d.  The
    module's own forward has already advanced ``num_batches_tracked`` and resolved its momentum before it called F.batch_norm, so
    the view has no counter (functional.bump_counters skips it) and `momentum` is the resolved averaging factor."""
    __slots__ = ("running_mean", "running_var", "weight", "bias", "training", "momentum", "eps", "track_running_stats", "num_batches_tracked")

    def __init__(self, running_mean, running_var, weight, bias, training, momentum, eps):
        self.running_mean, self.running_var, self.weight, self.bias = running_mean, running_var, weight, bias
        self.training, self.momentum, self.eps = bool(training), momentum, eps
        self.track_running_stats = running_mean is not None and running_var is not None
        self.num_batches_tracked = None


# A training-mode BatchNorm call has a side effect -- the running statistics' momentum update -- and two calls on the SAME buffers do
# not commute.  The reference uses every BatchNorm module once per forward, so the order of evaluation cannot show; for callers that
# reuse one, a new Bn handle first evaluates the pending handle that would update the same buffers: the updates keep the call order.
# (A BatchNorm whose output NOBODY ever reads is never evaluated and does not update its statistics: the one observable difference
# from eager evaluation.)
_PENDING_BN = {}


def _bn_handle(x, view):
    h = Lazy(Bn(x, view), x.shape, x)
    rm = view.running_mean
    if view.training and rm is not None:
        key = id(rm)
        prev = _PENDING_BN.get(key)
        if prev is not None:
            buf, old = prev[0](), prev[1]()
            if buf is rm and old is not None and old._value is None:
                old.materialize()
        _PENDING_BN[key] = (weakref.ref(rm, lambda _r, key=key: _PENDING_BN.pop(key, None)), weakref.ref(h))
    return h


class Lazy(torch.Tensor):
    """A tensor that has not been computed (module docstring)."""

    @staticmethod
    def __new__(cls, node, shape, like):
        return torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=like.device,
                                                   requires_grad=torch.is_grad_enabled())

    def __init__(self, node, shape, like):
        self.node = node
        self._grad_mode = torch.is_grad_enabled()   # the mode the eager call would have run in: a consumer inside no_grad must not decide it
        self._value = None          # the real tensor once materialised
        self._fan = None            # functional.Fan over the real tensor: the readers of a state share one K-way gradient sum
        self._rows = None           # Gather handles: the functional.LazyRows form (one per handle: its readers share the materialised rows)

    # the dispatcher is never reached: __torch_function__ answers or materialises first
    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        raise RuntimeError(f"mr_gnas_amd.lazy: an unmaterialised handle reached the dispatcher ({func}); please report the call")

    def __repr__(self):
        return f"Lazy({type(self.node).__name__}, shape={tuple(self.shape)}, materialised={self._value is not None})"

    def materialize(self):
        if self._value is None:
            # evaluated in the autograd mode of the CALL that made the handle, whoever looks first: a `with torch.no_grad():` consumer
            # would otherwise cache a value without a graph and every later reader would lose its gradient (found by
            # tests/test_host_cpu.py::test_lazy_handles_random_programs_equal_eager)
            with torch.set_grad_enabled(self._grad_mode):
                self._value = _evaluate(self)
            self.node = None        # the description's operands are no longer needed: let them go
        return self._value

    def lazy_rows(self):
        """A Gather handle as functional.LazyRows (what cell zero's MixedOp and the compose kernels gather from on the fly)."""
        if self._rows is None:
            from . import functional as K
            n = self.node
            self._rows = K.LazyRows(n.table, _gather_plan(n.idx, int(n.table.shape[0])))
        return self._rows

    def fan(self):
        """Aliases of the materialised value for its readers (functional.Fan: one K-way gradient sum instead of pairwise adds)."""
        if self._fan is None:
            from . import functional as K
            self._fan = K.Fan(self.materialize(), 1 << 20)
        return self._fan

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", None)
        if name in _META or (name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", None) in _META_PROPS):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        x = args[0] if args else None
        if isinstance(x, Lazy) and x._value is None:
            if name in ("float", "contiguous") and not kwargs and len(args) == 1:
                return x
            if (func is F.dropout or name == "dropout") and not kwargs.get("inplace", False):
                p = kwargs.get("p", args[1] if len(args) > 1 else 0.5)
                training = kwargs.get("training", args[2] if len(args) > 2 else True)
                if not training or p == 0:
                    return x                                        # identity
                return Lazy(Drop(x, p), x.shape, x)
            if func is F.batch_norm and isinstance(x.node, Op) and len(args) >= 3:
                view = BatchNormView(args[1], args[2], kwargs.get("weight", args[3] if len(args) > 3 else None),
                                     kwargs.get("bias", args[4] if len(args) > 4 else None),
                                     kwargs.get("training", args[5] if len(args) > 5 else False),
                                     kwargs.get("momentum", args[6] if len(args) > 6 else 0.1),
                                     kwargs.get("eps", args[7] if len(args) > 7 else 1e-5))
                return _bn_handle(x, view)
            if (func is F.relu or name == "relu") and isinstance(x.node, Bn) and not kwargs.get("inplace", False) and len(args) == 1:
                return Lazy(Act(x), x.shape, x)
            if name in ("add", "__add__", "__radd__") and len(args) == 2 and not kwargs:
                other = args[1]
                if isinstance(other, (int, float)) and not isinstance(other, bool) and other == 0 and isinstance(x.node, (Term, Sum, Act)):
                    return x if isinstance(x.node, Sum) else Lazy(Sum([x]), x.shape, x)     # Python's sum() starts from 0
                if isinstance(other, Lazy) and other._value is None and isinstance(other.node, (Term, Sum, Act)) and isinstance(x.node, (Term, Sum, Act)) \
                        and other.shape == x.shape:
                    # the caller's association is kept (a literal evaluation adds in the caller's order): a MixedOp's sum collects its
                    # terms; the sum over MixedOps (models/cell_lp.py:103-107) collects the MixedOps' sums as parts
                    if isinstance(other.node, Sum) and isinstance(x.node, Sum) and not any(isinstance(p.node, Sum) for p in x.node.parts if p._value is None):
                        return Lazy(Sum([x, other]), x.shape, x)
                    return Lazy(Sum(_parts(x) + [other]), x.shape, x)
        if name == "cat" and args and isinstance(args[0], (list, tuple)) and len(args[0]) == 2:
            first, second = args[0]
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            if (dim == 0 and isinstance(first, Lazy) and first._value is None and isinstance(first.node, Gather) and second is first.node.table
                    and set(kwargs) <= {"dim"}):
                # torch.cat((ent[src_in], ent), dim=0) (models/model_search_lp.py:153): rows idx of the table, then the table itself -- one
                # gather with the index cat(idx, arange(rows)), never written
                t, idx = first.node.table, first.node.idx
                from .graph import cached_on
                full = cached_on(idx, "_mrg_idx_and_self", (idx,), int(t.shape[0]),
                                 lambda: torch.cat((idx.long(), torch.arange(t.shape[0], device=idx.device))))
                return Lazy(Gather(t, full, then_table=idx), (full.numel(), t.shape[1]), t)
        if name in ("mul", "__mul__", "__rmul__") and len(args) == 2 and not kwargs:
            a, b = args
            if (isinstance(a, Lazy) and isinstance(b, Lazy) and a._value is None and b._value is None and a.shape == b.shape
                    and isinstance(a.node, (Gather, Prod)) and isinstance(b.node, (Gather, Prod))):
                fa = list(a.node.factors) if isinstance(a.node, Prod) else [a]
                fb = list(b.node.factors) if isinstance(b.node, Prod) else [b]
                return Lazy(Prod(fa + fb), a.shape, a)
            if isinstance(b, Lazy) and not isinstance(a, Lazy):
                a, b = b, a
            if (isinstance(a, Lazy) and a._value is None and isinstance(a.node, Act) and isinstance(b, torch.Tensor) and not isinstance(b, Lazy)
                    and b.dim() == 0 and b.device == a.device):
                return Lazy(Term(b, a), a.shape, a)
        if name == "sum" and isinstance(x, Lazy) and x._value is None and isinstance(x.node, Prod):
            dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
            if dim in (1, -1) and set(kwargs) <= {"dim"} and len(args) <= 2:
                score = _distmult(x.node.factors)
                if score is not None:
                    return score
        # anything else: the handles become real tensors and the call proceeds on them
        args, kwargs = _real(args), _real(kwargs)        # (evaluated with torch-function dispatch still ON: evaluation may itself meet handles)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _parts(x):
    return list(x.node.parts) if isinstance(x.node, Sum) else [x]


def _real(obj):
    if isinstance(obj, Lazy):
        return obj.materialize()
    if isinstance(obj, (list, tuple)):
        out = [_real(o) for o in obj]
        return type(obj)(out) if not hasattr(obj, "_fields") else type(obj)(*out)
    if isinstance(obj, dict):
        return {k: _real(v) for k, v in obj.items()}
    return obj


def real(x):
    """x as a real tensor (a Lazy is materialised; anything else is returned as it is)."""
    return x.materialize() if isinstance(x, Lazy) else x


def defer(op, g, a, b, shape):
    """What an operator's forward returns when handles are enabled: the call, not yet run."""
    like = a.table if hasattr(a, "table") else (a if isinstance(a, torch.Tensor) else b)
    return Lazy(Op(op, g, a, b), shape, like)


def wanted(a):
    """Do the operators hand out handles for this operand?  (CUDA float32 rows only: the CPU registries of the tests are eager.)"""
    if not ENABLED or not torch._C._is_torch_function_enabled():
        return False
    if isinstance(a, Lazy):
        return True
    t = a.table if hasattr(a, "table") else a
    return isinstance(t, torch.Tensor) and (t.is_cuda or FORCE_CPU) and t.dtype == torch.float32


# ---- evaluation ------------------------------------------------------------------------------------------------------------------
def _evaluate(x):
    n = x.node
    if isinstance(n, Op):
        return n.op.run(n.g, real(n.a), real(n.b))        # (the module call that produced this handle has already fired its hooks)
    if isinstance(n, Bn):
        v = n.view
        return F.batch_norm(real(n.src), v.running_mean, v.running_var, v.weight, v.bias, v.training, v.momentum, v.eps)
    if isinstance(n, Act):
        return _evaluate_sum([x]) if _chain(x) is not None else F.relu(real(n.src))
    if isinstance(n, Drop):
        return F.dropout(real(n.src), n.p, True)
    if isinstance(n, Gather):
        if n.table.is_cuda:
            return x.lazy_rows().materialize()
        if n.then_table is not None:                          # (CPU test mode: the caller's own expression)
            return torch.cat((_ORIG_GETITEM(n.table, n.then_table), n.table), dim=0)
        return _ORIG_GETITEM(n.table, n.idx)
    if isinstance(n, Prod):
        out = real(n.factors[0])
        for f in n.factors[1:]:
            out = out * real(f)
        return out
    if isinstance(n, Term):
        return _evaluate_sum([x])
    if isinstance(n, Sum):
        return _evaluate_sum(n.parts)
    raise TypeError(f"unknown lazy node {type(n).__name__}")


def _chain(t):
    """(w, bn view, Op node) of a term w * relu(batch_norm(op(...))) -- or relu(batch_norm(op(...))), w = None: the fixed-genotype
    OpModule's chain, weight one -- that nobody has materialised in part, else None."""
    if not (isinstance(t, Lazy) and t._value is None and isinstance(t.node, (Term, Act))):
        return None
    w, act = (t.node.w, t.node.src) if isinstance(t.node, Term) else (None, t)
    if act._value is not None or not isinstance(act.node, Act):
        return None
    bn = act.node.src
    if bn._value is not None or not isinstance(bn.node, Bn):
        return None
    opl = bn.node.src
    if opl._value is not None or not isinstance(opl.node, Op):
        return None
    v = bn.node.view
    if v.weight is None or v.bias is None:
        return None
    a = opl.node.a
    if not (a.table if hasattr(a, "table") else a).is_cuda or not hasattr(opl.node.op, "run"):
        return None                                          # the fused path is HIP only
    return w, v, opl.node


_ONES = {}


def _one(node):
    """The weight of a candidate that was not multiplied by anything (a 0-d float32 one on the operands' device)."""
    a = node.a
    dev = (a.table if hasattr(a, "table") else a).device
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def _operand(x):
    """An operator's operand for the fused path: the LazyRows form of a gather nobody has materialised (cell zero gathers on the
    fly), a Fan over a state that is itself a handle (its readers share one gradient sum), the object itself otherwise."""
    if isinstance(x, Lazy):
        if x._value is None and isinstance(x.node, Gather) and x.node.table.is_cuda:
            return x.lazy_rows()
        return x.fan()
    return x


# ---- the gather G: table[idx] as a handle ------------------------------------------------------------------------------------------
def _gather_plan(idx, rows):
    """functional.GatherPlan of an index tensor, cached on the tensor OBJECT (identity + in-place version: graph.cached_on)."""
    from . import functional as K
    from .graph import cached_on
    return cached_on(idx, "_mrg_gather_plan", (idx,), rows, lambda: K.GatherPlan(idx if idx.dtype == torch.int64 else idx.long(), rows))


def _gatherable(table, idx):
    return (isinstance(idx, torch.Tensor) and not isinstance(idx, Lazy) and idx.dim() == 1 and idx.dtype in (torch.int64, torch.int32)
            and (idx.is_cuda or FORCE_CPU) and idx.numel() >= MIN_GATHER_ROWS and type(table) in _PLAIN and table.dim() == 2
            and table.device == idx.device and table.dtype == torch.float32 and table.is_contiguous() and table.shape[0] > 0)


_PLAIN = (torch.Tensor, torch.nn.Parameter)
_ORIG_GETITEM = None


def _getitem(self, idx):
    # (no handles where nobody would see them: under DisableTorchFunctionSubclass a handle would reach the dispatcher)
    if ENABLED and FAST_INDEX and _gatherable(self, idx) and torch._C._is_torch_function_enabled():
        # the library's kernels do not wrap negative indices and do not bound-check: trap (asynchronously) what torch would have
        # wrapped or refused
        torch._assert_async(((idx >= 0) & (idx < self.shape[0])).all())
        return Lazy(Gather(self, idx), (idx.numel(), self.shape[1]), self)
    return _ORIG_GETITEM(self, idx)


def install_indexing():
    """Wrap torch.Tensor.__getitem__ (idempotent): see the module docstring.  ``uninstall_indexing()`` restores torch's own."""
    global _ORIG_GETITEM
    if _ORIG_GETITEM is None:
        _ORIG_GETITEM = torch.Tensor.__getitem__
        torch.Tensor.__getitem__ = _getitem


def uninstall_indexing():
    global _ORIG_GETITEM
    if _ORIG_GETITEM is not None:
        torch.Tensor.__getitem__ = _ORIG_GETITEM
        _ORIG_GETITEM = None


def _distmult(factors):
    """torch.sum(ent[s] * rel[r] * ent[o], dim=1) over three Gather handles (models/model_search_lp.py:169-176) as ONE fused
    scoring kernel (functional.distmult_score: no [T, D] gather is written; backward three balanced segmented sums), or None."""
    if len(factors) != 3 or not all(isinstance(f, Lazy) and f._value is None and isinstance(f.node, Gather) and f.node.table.is_cuda for f in factors):
        return None
    a, b, c = (f.node for f in factors)
    if a.table is c.table and b.table is not a.table:
        s, r, o = a, b, c
    elif a.table is b.table and c.table is not a.table:
        s, o, r = a, b, c
    elif b.table is c.table and a.table is not b.table:
        r, s, o = a, b, c
    else:
        return None
    from . import functional as K
    from .graph import cached_on
    n_ent, n_rel = int(s.table.shape[0]), int(r.table.shape[0])
    base = s.idx._base
    if (base is not None and r.idx._base is base and o.idx._base is base and base.dim() == 2 and base.shape[1] == 3 and base.is_contiguous()
            and (s.idx.storage_offset(), r.idx.storage_offset(), o.idx.storage_offset()) == (base.storage_offset(), base.storage_offset() + 1,
                                                                                             base.storage_offset() + 2)
            and s.idx.stride() == r.idx.stride() == o.idx.stride() == (3,)):
        # triplets[:, 0], triplets[:, 1], triplets[:, 2] of one [T, 3] tensor: the plan is cached on THAT tensor (a resident batch)
        plan = cached_on(base, "_mrg_score_plan", (base,), (n_ent, n_rel), lambda: K.ScorePlan(base, n_ent, n_rel))
    else:
        plan = K.ScorePlan(torch.stack((s.idx.long(), r.idx.long(), o.idx.long()), 1), n_ent, n_rel)
    return K.distmult_score(s.table, r.table, plan)


def _leaves(parts):
    out = []
    for t in parts:
        if isinstance(t, Lazy) and t._value is None and isinstance(t.node, Sum):
            out += _leaves(t.node.parts)
        else:
            out.append(t)
    return out


def _literal_sum(parts):
    """The sum as the caller wrote it: parts left to right, a nested sum evaluated as one value first."""
    total = None
    for t in parts:
        if isinstance(t, Lazy) and t._value is None and isinstance(t.node, Sum):
            t._value = v = _literal_sum(t.node.parts)
            t.node = None
        elif isinstance(t, Lazy) and t._value is None and isinstance(t.node, Term):
            v = t.node.w * real(t.node.src)
        elif isinstance(t, Lazy) and t._value is None and isinstance(t.node, Act):
            v = F.relu(real(t.node.src))
        else:
            v = real(t)
        total = v if total is None else total + v
    return total


def _evaluate_sum(parts):
    from . import cell_lp
    leaves = _leaves(parts)
    chains = [_chain(t) for t in leaves]
    if not any(c is not None for c in chains):
        return _literal_sum(parts)
    groups, order, rest = {}, [], []
    for t, c in zip(leaves, chains):
        if c is None:
            rest.append(t)
            continue
        key = (id(c[2].g), id(c[2].a), id(c[2].b))
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(c)
    total = None
    for key in order:
        grp = groups[key]
        node0 = grp[0][2]
        ops = [c[2].op for c in grp]
        bns = [c[1] for c in grp]
        w = torch.stack([c[0] if c[0] is not None else _one(node0) for c in grp])
        total = cell_lp.fused_candidates(ops, bns, w, node0.g, _operand(node0.a), _operand(node0.b), addend=total)
    if rest:                                                 # foreign terms, in the order they were written
        v = _literal_sum(rest)
        total = v if total is None else total + v
    return total
