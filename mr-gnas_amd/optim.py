"""clip_grad_norm_ + SGD(momentum) as ONE call over every parameter tensor (mrg_clip_sgd_step: three launches).

The reference's search step ends with ``torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm)`` and
``optimizer.step()`` of ``torch.optim.SGD(lr, momentum, weight_decay)`` (search/mr_lp_search.py:118-119,243-245).  On ~300
parameter tensors torch issues ~30 ``multi_tensor_apply`` launches for the pair; ``ClippedSGD.step()`` is the same arithmetic
(same clip coefficient, same momentum recurrence, buffers starting at zero = torch's ``buf = grad`` first step) through a device
table of pointers.  The product path has no CPU form: CPU parameters raise.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_of


class ClippedSGD:
    CAPTURE_TABLES = 4          # step() calls that may be captured into HIP graphs over the optimiser's life (one pinned table each)

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, max_norm=0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("ClippedSGD: no parameters")
        dev = self.params[0].device
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise _lib.MrgnasError("ClippedSGD: parameters must be contiguous float32 tensors on one HIP device")
        self.lr, self.momentum, self.weight_decay, self.max_norm = float(lr), float(momentum), float(weight_decay), float(max_norm)
        chunk = int(_lib.load().mrg_optim_chunk())
        sizes = [p.numel() for p in self.params]
        self._flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)         # the momentum buffers, one allocation
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n
        self.bufs = [self._flat[a:a + n].view_as(p) for a, n, p in zip(offs, sizes, self.params)]
        t_idx, c_off, c_len = [], [], []
        for t, n in enumerate(sizes):
            for a in range(0, n, chunk):
                t_idx.append(t)
                c_off.append(a)
                c_len.append(min(chunk, n - a))
        self.n_chunks = len(t_idx)
        self._chunk_tensor = torch.tensor(t_idx, dtype=torch.int32, device=dev)
        self._chunk_off = torch.tensor(c_off, dtype=torch.int64, device=dev)
        self._chunk_len = torch.tensor(c_len, dtype=torch.int32, device=dev)
        self._p_ptr_host = [p.data_ptr() for p in self.params]
        self._p_ptrs = torch.tensor(self._p_ptr_host, dtype=torch.int64, device=dev)
        self._b_ptrs = torch.tensor([b.data_ptr() for b in self.bufs], dtype=torch.int64, device=dev)
        # pinned staging for the gradient pointers: two buffers used in turn, each guarded by the event of the copy that last read it
        # (the host may run a whole step ahead of the device)
        self._g_host = [torch.zeros(len(self.params), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._g_np = [h.numpy() for h in self._g_host]      # the same pinned memory: one vectorised write per step
        self._g_event = [None, None]
        self._turn = 0
        self._captured = []
        self._spare = [torch.zeros(len(self.params), dtype=torch.int64).pin_memory() for _ in range(self.CAPTURE_TABLES)]
        self._g_ptrs = torch.zeros(len(self.params), dtype=torch.int64, device=dev)
        self._partial = torch.empty(max(self.n_chunks, 1), dtype=torch.float64, device=dev)
        self.norm_coef = torch.zeros(2, dtype=torch.float32, device=dev)                # [total gradient norm, clip coefficient] of the last step

    def step(self):
        """Clip (when max_norm > 0) and update.  Gradients are read where autograd left them; a parameter without a gradient, or
        whose gradient is not a contiguous float32 tensor, is handled as torch does (skipped / made contiguous)."""
        grads = []
        for i, p in enumerate(self.params):
            if p.data_ptr() != self._p_ptr_host[i]:
                raise _lib.MrgnasError("ClippedSGD: a parameter's storage moved since construction (build the optimiser after .to(device))")
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            grads.append(g)
        ptrs = [0 if g is None else g.data_ptr() for g in grads]
        if torch.cuda.is_current_stream_capturing():
            # the captured copy node reads its host buffer again on every replay: a buffer of its own, never rewritten (the gradients
            # of a captured step live at fixed addresses of the graph's memory pool)
            if not self._spare:                              # (pinning allocates: not allowed while a capture is open)
                raise _lib.MrgnasError("ClippedSGD: more captures than spare pinned pointer tables (ClippedSGD.CAPTURE_TABLES)")
            host = self._spare.pop()
            host.numpy()[:] = ptrs
            self._captured.append(host)
            self._g_ptrs.copy_(host, non_blocking=True)
        else:
            k = self._turn
            self._turn ^= 1
            if self._g_event[k] is not None:
                self._g_event[k].synchronize()              # normally long complete
            else:
                self._g_event[k] = torch.cuda.Event()
            self._g_np[k][:] = ptrs
            self._g_ptrs.copy_(self._g_host[k], non_blocking=True)
            self._g_event[k].record()
        call("mrg_clip_sgd_step", (ptr(self._p_ptrs), ptr(self._g_ptrs), ptr(self._b_ptrs), ptr(self._chunk_tensor), ptr(self._chunk_off),
                                   ptr(self._chunk_len), self.n_chunks, ptr(self._partial), ptr(self.norm_coef), self.max_norm, self.lr,
                                   self.momentum, self.weight_decay, stream_of(self._flat)))
        # (the gradients -- and a converted copy of one -- are still referenced here: the launches that read them are enqueued, and the
        #  caching allocator hands their memory out again in stream order)
        torch.autograd.graph.increment_version(self.params)   # the kernels wrote the parameters behind autograd's back: say so
        return self.norm_coef

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        return {"momentum_buffers": [b.clone() for b in self.bufs], "lr": self.lr, "momentum": self.momentum,
                "weight_decay": self.weight_decay, "max_norm": self.max_norm}

    def load_state_dict(self, state):
        for b, s in zip(self.bufs, state["momentum_buffers"]):
            b.copy_(s)
        self.lr, self.momentum = float(state["lr"]), float(state["momentum"])
        self.weight_decay, self.max_norm = float(state["weight_decay"]), float(state["max_norm"])
