"""Gradient fan-in of a tensor with several readers: K-way sums and the Fan alias bookkeeping (csrc/accum.hip).

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import weakref

import torch

from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of
from .reducers import _seg_bwd


def sum_buffers(xs):
    """Sum of equally-shaped HIP tensors in K-way passes (mrg_sum_buffers), k = 0..K-1 order."""
    xs = [f32c(x) for x in xs]
    require_hip(*xs)
    out = torch.empty_like(xs[0])
    n = out.numel()
    for i in range(0, len(xs), 8):
        part = xs[i:i + 8]
        call("mrg_sum_buffers", (ptr_array(part), len(part), ptr(out), n, int(i > 0), stream_of(out)),
             nbytes=4 * n * (len(part) + 1 + int(i > 0)))
    return out


class _Fanout(torch.autograd.Function):
    """k aliases of one tensor whose gradients are summed in one K-way pass instead of k - 1
    pairwise adds.  Aliases nobody reads cost nothing (their gradient stays None)."""

    @staticmethod
    def forward(ctx, x, k, box):
        ctx.set_materialize_grads(False)
        ctx.box = box                                       # mailbox of this batch of aliases (Fan.take / _AggRows.backward)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        gathered = ctx.box.gathered
        if gathered is not None:                            # a reader (a_sum) left its gradient as a gather of an [N, D] tensor
            gh, gself, graph, ev = gathered
            ctx.box.gathered = None
            cur = torch.cuda.current_stream(gh.device)
            cur.wait_event(ev)                                # the producer's stream may not be this one
            for t in (gh, gself):
                if t is not None:
                    t.record_stream(cur)                      # ... and its allocator must not recycle the blocks under this launch
            E, N, D = graph.num_edges(), graph.number_of_nodes(), gh.shape[1]
            if len(gs) <= 8 and all(g.is_cuda and g.shape == (E + N, D) for g in gs):
                gs = [f32c(g) for g in gs]
                out = torch.empty(E + N, D, dtype=torch.float32, device=gh.device)
                call("mrg_sum_rows_gather", (ptr_array(gs), len(gs), ptr(f32c(gh)), ptr(f32c(gself)), ptr(graph.i32("dst")), E, E + N, D, ptr(out),
                                             stream_of(out)), nbytes=4 * D * (E + N) * (len(gs) + 1))
                return out, None, None
            gx = torch.empty(E + N, D, dtype=torch.float32, device=gh.device)      # shapes the kernel does not take: materialise
            if gself is not None:
                gx[E:] = gself
            else:
                gx[E:].zero_()
            _seg_bwd(0, gh, graph, None, gx, None)
            gs.append(gx)
        if not gs:
            return None, None, None
        if len(gs) == 1:
            return gs[0], None, None
        if not gs[0].is_cuda or any(g.shape != gs[0].shape for g in gs):
            tot = gs[0]
            for g in gs[1:]:
                tot = tot + g
            return tot, None, None
        return sum_buffers(gs), None, None


class _FanBox:
    """Mailbox of one batch of Fan aliases: a reader whose gradient w.r.t. the alias is a gather of a small tensor (a_sum) leaves the
    small tensor here instead of materialising [rows, D]; the batch's fan-in sum (_Fanout.backward) reads it."""
    __slots__ = ("gathered",)

    def __init__(self):
        self.gathered = None


class Fan:
    """Hands out aliases of `x` to its readers: ``fan.take()`` per reader, at most `cap` of them.
    Aliases are created a dozen at a time (a further batch hangs off the last alias of the previous one, so
    its gradients arrive as one pre-summed tensor).  Without autograd (or for tensors that need no
    gradient) the tensor itself is returned."""

    BATCH = 12

    def __init__(self, x, cap):
        self.x = x
        self.left = cap
        self._live = torch.is_grad_enabled() and x.requires_grad and cap > 1
        self._views = []
        self._root = x

    def take(self):
        if not self._live:
            return self.x
        if self.left <= 0:
            raise RuntimeError("Fan: more readers than announced")
        self.left -= 1
        if not self._views:
            self._box = _FanBox()
            views = list(_Fanout.apply(self._root, self.BATCH, self._box))
            self._root = views.pop()                    # source of the next batch, if one is ever needed
            self._views = views
        v = self._views.pop()
        Fan._remember(v, self._box)                     # lets a reader leave its gradient with the fan-in sum (_AggRows.backward)
        return v

    # alias tensor -> the mailbox of the fan-out batch it came from.  A side table keyed by the alias OBJECT's id (tensors compare
    # elementwise, so they cannot key a dict themselves) holding a weak reference that removes the entry when the alias dies: a reader
    # handed anything else -- a copy, a cast that allocates -- simply is not found and materialises its gradient as usual.
    _NODE = {}

    @staticmethod
    def _remember(v, box):
        key = id(v)
        Fan._NODE[key] = (weakref.ref(v, lambda _r, key=key: Fan._NODE.pop(key, None)), box)

    @staticmethod
    def node_of(x):
        hit = Fan._NODE.get(id(x))
        return hit[1] if hit is not None and hit[0]() is x else None
