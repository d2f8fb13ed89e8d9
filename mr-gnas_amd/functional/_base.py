"""Shared plumbing of the autograd Functions: op / reduce / activation codes, workspace helpers, plan counts, side streams.

Part of ``mr_gnas_amd.functional`` (autograd Functions over the C ABI, include/mrgnas.h): every Function enqueues HIP kernels of
libmrgnas_hip.so on torch's current stream through ctypes; every call site states the algorithmic bytes / flops of the launch."""
import contextlib

import torch

from .. import _lib
from . import switches as SW


COMPOSE = {"mult": 0, "sub": 1, "add": 2}
REDUCE = {"sum": 0, "mean": 1, "max": 2}
ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2}


def gate_ld(D):
    return 2 * D + 4          # MRG_GATE_LD


def same_rows(a, b):
    """True when the two operands of an operator are the SAME rows (one tensor, or two aliases of one storage --
    e.g. Fan views of the cell's h_in, which the first two MixedOps of a cell receive as both `src_emb` and
    `src_emb_in`, reference models/cell_lp.py:95-104).  Then W [s ; s_in] = (W[:, :D] + W[:, D:]) s: the
    operators fold the weight halves and run at half the inner dimension; the whole input gradient is returned
    through the first operand."""
    if a is None or b is None:
        return False
    if a is b:
        return True
    # aliases fold only when the whole input gradient may leave through the first operand: both need it or neither does
    # (a detached alias next to a tracked one keeps the untied path, which routes each gradient to its own operand);
    # empty tensors all share data_ptr 0 and are never aliases of each other
    return (a.numel() > 0 and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
            and a.dtype == b.dtype and a.requires_grad == b.requires_grad)


def _same_memory(a, b):
    """Do two tensors denote the same [rows, D] block of device memory (e.g. two Fan aliases of one state)?"""
    return a is not None and b is not None and a.numel() > 0 and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()


def _cnt(plan, name):
    """Launch-side count of a plan: the host-known capacity when the plan has one (HIP-built plans pad with -1 and the
    kernels skip the padding: no device-to-host read), else the exact number."""
    cap = "cap_" + name
    if cap in plan and not (hasattr(plan, "try_resolve") and plan.try_resolve()):
        return plan[cap]                         # exact sizes still on their way to the host: launch over the padded capacity
    return plan["n_" + name]


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)
_WS_BYTES = {}


def _ws_bytes(name, *args):
    """Workspace size queries are pure functions of the shape: asked once per (entry point, shape)."""
    key = (name, args)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = getattr(_lib.load(), name)(*args)
    return n
_SIDE_STREAMS = {}


class Fork:
    """Independent kernel chains (the direction segments of one operator write disjoint row ranges) run on
    side HIP streams and join back: the tail of one GEMM (a 272 115-row segment fills 4.15 rounds of the chip,
    the 14 541 self rows a fifth of one) is filled by the workgroups of the next instead of idling.
    Every tensor the chains touch is allocated on the main stream BEFORE the fork."""

    def __init__(self, device, n, tag="segments"):
        if tag == "segments":
            n = min(n, SW.SEGMENT_STREAMS)
        self.main = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), tag)   # one pool per nesting level
        pool = _SIDE_STREAMS.setdefault(key, [])
        while len(pool) < n - 1:
            pool.append(torch.cuda.Stream(device=device))
        self.side = pool[:max(n - 1, 0)]
        from ..graph import register_stream
        register_stream(self.main)
        for st in self.side:
            register_stream(st)
            st.wait_stream(self.main)

    def stream(self, i):
        i %= len(self.side) + 1
        return self.main if i == 0 else self.side[i - 1]

    def on(self, i):
        """Context: launch on chain i's stream (nothing to switch when there are no side streams)."""
        return torch.cuda.stream(self.stream(i)) if self.side else contextlib.nullcontext()

    def join(self):
        for st in self.side:
            self.main.wait_stream(st)


# ---- BatchNorm step counters -------------------------------------------------------------------------------------------------
# nn.BatchNorm1d counts its training forwards in `num_batches_tracked` (a device scalar per module).  The fused epilogues update the
# running statistics inside their kernels; the counters are left to torch -- one multi-tensor launch per MixedOp, 22 per supernet
# step.  Inside `deferred_counters()` (the networks' forward) they are collected and bumped by ONE launch at the end.
_PENDING_COUNTERS = None


def bump_counters(bns):
    """num_batches_tracked += 1 for the BatchNorm modules of one fused epilogue (now, or at the end of the enclosing deferred_counters())."""
    counters = [b.num_batches_tracked for b in bns if b.num_batches_tracked is not None]
    if not counters:
        return
    if _PENDING_COUNTERS is not None:
        for c in counters:                       # a module applied twice inside the block counts twice -- as ONE entry of the flush
            slot = _PENDING_COUNTERS.setdefault(id(c), [c, 0])       # (a multi-tensor add must not see the same tensor twice)
            slot[1] += 1
    else:
        torch._foreach_add_(counters, 1)


@contextlib.contextmanager
def deferred_counters():
    """Collect the BatchNorm step counters of every fused epilogue inside the block and add 1 to all of them in one launch at its end
    (re-entrant: an inner block leaves the flush to the outermost one)."""
    global _PENDING_COUNTERS
    if _PENDING_COUNTERS is not None:
        yield
        return
    _PENDING_COUNTERS = {}
    try:
        yield
    finally:
        pending, _PENDING_COUNTERS = _PENDING_COUNTERS, None
        by_count = {}
        for c, n in pending.values():
            by_count.setdefault(n, []).append(c)
        for n, cs in by_count.items():           # one launch in the usual case (every module applied once)
            torch._foreach_add_(cs, n)

