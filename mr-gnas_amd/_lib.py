"""ctypes binding of libmrgnas_hip.so (C ABI: include/mrgnas.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C mr-gnas_amd/csrc``.  Nothing here falls back to another
implementation: a missing library raises ``MrgnasLibraryError``.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRG_LIB_PATH") or os.path.join(_HERE, "lib", "libmrgnas_hip.so")     # MRG_LIB_PATH: lab builds of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mrgnas.h")

ABI_VERSION = 14


class MrgnasLibraryError(RuntimeError):
    pass


class MrgnasError(RuntimeError):
    """A C-ABI call returned a non-zero code."""


_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); must list every function declared in include/mrgnas.h
SIGNATURES = {
    "mrg_abi_version": (_I, []),
    "mrg_set_stream_blocks": (_I, [_I]),
    "mrg_error_string": (ctypes.c_char_p, [_I]),
    "mrg_target_arch": (ctypes.c_char_p, []),
    "mrg_compose_fwd": (_I, [_I, _P, _P, _P, _L, _I, _P]),
    "mrg_compose_bwd": (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_gather_compose_fwd": (_I, [_I, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_gate_collapse": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "mrg_gate_collapse3": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mrg_gate_param_grad3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mrg_fold_halves3": (_I, [_P, _P, _I, _P]),
    "mrg_unfold_halves3": (_I, [_P, _P, _I, _P]),
    "mrg_gate_fwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _F, _P]),
    "mrg_gate_bwd_workspace_bytes": (_L, [_L, _I]),
    "mrg_gate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _F, _P]),
    "mrg_gate_param_grad": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mrg_seg_reduce_workspace_bytes": (_L, [_L, _I]),
    "mrg_seg_reduce_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _L, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_seg_reduce_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "mrg_fused_gcs": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _L, _P, _P, _P, _L, _I, _P]),
    "mrg_span_gcs": (_I, [_I, _P, _P, _P, _P, _L, _I, _P, _P, _L, _P, _P, _P, _L, _L, _P, _P, _P, _L, _I, _P]),
    "mrg_sum_buffers": (_I, [_P, _I, _P, _L, _I, _P]),
    "mrg_sum_rows_gather": (_I, [_P, _I, _P, _P, _P, _L, _L, _I, _P, _P]),
    "mrg_distmult_score": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_gate_row_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _F, _P]),
    "mrg_gate_row_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P]),
    "mrg_mix_workspace_bytes": (_L, [_I, _I]),
    "mrg_mix_colstats": (_I, [_P, _I, _L, _I, _P, _P, _P, _P]),
    "mrg_mix_finalize_fwd": (_I, [_P, _P, _P, _P, _P, _I, ctypes.c_double, _I, _F, _F, _P, _P]),
    "mrg_mix_stats_coef": (_I, [_P, _P, _P, _P, _P, _I, _L, ctypes.c_double, _I, _F, _F, _P, _P, _P, _P]),
    "mrg_zero_workspace_bytes": (_L, [_I]),
    "mrg_zero_colstats": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P]),
    "mrg_zero_stats_coef": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, ctypes.c_double, _I, _F, _F, _P, _P, _P]),
    "mrg_zero_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _P]),
    "mrg_zero_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_zero_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P]),
    "mrg_mix_fwd": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _P, _P]),
    "mrg_mix_bwd_reduce": (_I, [_P, _P, _I, _P, _P, _P, _P, _L, _I, _P, _P]),
    "mrg_mix_finalize_bwd": (_I, [_P, _I, ctypes.c_double, _I, _P, _P, _P, _P, _P]),
    "mrg_mix_bwd_apply": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "mrg_dense_filter_fwd": (_I, [_I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _P]),
    "mrg_dense_filter_dz": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _L, _I, _P]),
    "mrg_gemm_workspace_bytes": (_L, [_I, _I]),
    "mrg_gemm_set_mode": (_I, [_I]),
    "mrg_gemm_set_epilogue": (_I, [_I]),
    "mrg_gemm_set_wide8": (_I, [_I]),
    "mrg_gemm_set_q": (_I, [_I]),
    "mrg_set_dynamic_rows": (_I, [_L, _P, _L, _P]),
    "mrg_gemm_set_small": (_I, [_I]),
    "mrg_wgrad_set_share": (_I, [_I]),
    "mrg_segmax_bwd_input_ok": (_I, [_I, _I]),
    "mrg_segmax_bwd_input": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P]),
    "mrg_optim_chunk": (_I, []),
    "mrg_clip_sgd_step": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _P]),
    "mrg_act_grad_transpose": (_I, [_P, _P, _P, _L, _L, _I, _P]),
    "mrg_wgrad_set_variant": (_I, [_I]),
    "mrg_linear_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "mrg_dense_filter3_workspace_bytes": (_L, [_I, _I]),
    "mrg_dense_filter_fwd3": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _L, _L, _L, _I, _P]),
    "mrg_dense_filter_dz3": (_I, [_I, _P, _P, _P, _P, _F, _F, _P, _P, _L, _L, _I, _P]),
    "mrg_linear_bwd_input3_workspace_bytes": (_L, [_I, _I]),
    "mrg_linear_bwd_input3": (_I, [_P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _P]),
    "mrg_linear_bwd_input3_pair_workspace_bytes": (_L, [_I, _I]),
    "mrg_linear_bwd_input3_pair": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _P]),
    "mrg_linear_bwd_weight3_workspace_bytes": (_L, [_L, _L, _L, _I, _I, _I]),
    "mrg_linear_bwd_weight3": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _I, _I, _P]),
    "mrg_linear_relu_segsum_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "mrg_seg_reduce_heads_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _L, _P, _P, _P, _L, _I, _P]),
    "mrg_seg_reduce_bwd_bits": (_I, [_I, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "mrg_seg_reduce_bwd_ordered": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "mrg_linear_relu_segmax_workspace_bytes": (_L, [_L, _I, _I]),
    "mrg_linear_relu_segmax_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _P]),
    "mrg_linear_bwd_input_workspace_bytes": (_L, [_I, _I]),
    "mrg_linear_bwd_input": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "mrg_linear_bwd_weight_workspace_bytes": (_L, [_L, _I, _I]),
    "mrg_linear_bwd_weight": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "mrg_build_graph_workspace_bytes": (_L, [_L]),
    "mrg_build_graph": (_I, [_P, _L, _L, _I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mrg_plan_workspace_bytes": (_L, [_L, _L, _I]),
    "mrg_span_plan_build": (_I, [_P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mrg_span_meta_pack": (_I, [_P, _P, _P, _P, _P, _I, _P, _L, _P]),
    "mrg_chunk_plan_workspace_bytes": (_L, [_L, _L]),
    "mrg_chunk_plan_build": (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "mrg_negative_sampling": (_I, [_P, _L, _I, _P, _P, _P, _P, _P]),
    "mrg_sample_neighborhood_workspace_bytes": (_L, [_L, _L]),
    "mrg_sample_edge_neighborhood": (_I, [_P, _P, _P, _P, _L, _L, _L, _P, _P, _L, _P, _P, _P, _P, _L, _P]),
    "mrg_relabel_workspace_bytes": (_L, [_L]),
    "mrg_relabel_nodes": (_I, [_P, _P, _L, _L, _P, _P, _P, _P, _P, _L, _P]),
    "mrg_multi_hot_labels": (_I, [_P, _P, _P, _P, _L, _L, _L, _F, _F, _P, _P]),
    "mrg_rank_filtered": (_I, [_P, _P, _P, _L, _L, _P, _P]),
    "mrg_transe_score_fwd": (_I, [_P, _P, _P, _F, _P, _L, _L, _I, _P]),
    "mrg_transe_score_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
}

_lib = None


def declared_symbols():
    """Function names declared in include/mrgnas.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mrg_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrgnasLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mr-gnas_amd/csrc`). mr_gnas_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, do not hide it
        fn.restype, fn.argtypes = res, args
    if lib.mrg_abi_version() != ABI_VERSION:
        raise MrgnasLibraryError(f"ABI version {lib.mrg_abi_version()} != expected {ABI_VERSION}; rebuild the library")
    if os.environ.get("MRG_STREAM_BLOCKS"):          # lab: grid bound of the streaming kernels (mrg_set_stream_blocks)
        if lib.mrg_set_stream_blocks(int(os.environ["MRG_STREAM_BLOCKS"])) != 0:
            raise MrgnasLibraryError("MRG_STREAM_BLOCKS must be 64..4096")
    if os.environ.get("MRG_WGRAD_VARIANT"):          # lab: 0 = every wave splits the fragments it multiplies (mrg_wgrad_set_variant)
        lib.mrg_wgrad_set_variant(int(os.environ["MRG_WGRAD_VARIANT"]))
    if os.environ.get("MRG_GEMM_EPILOGUE"):          # lab: 0 = accumulator-order stores of the row GEMM (mrg_gemm_set_epilogue)
        lib.mrg_gemm_set_epilogue(int(os.environ["MRG_GEMM_EPILOGUE"]))
    _lib = lib
    return lib


class KernelMeter:
    """Optional per-entry-point timing with HIP events on the launch stream.

    ``meter.start(names)`` brackets every call of the named C-ABI functions with a
    pair of events recorded on torch's current stream (the stream the kernels are
    enqueued on); ``meter.stop()`` synchronises and returns, per name, the number
    of launches, the summed device time and the summed algorithmic bytes / flops
    the call sites declared.  Used by bench.py for the roofline figures."""

    def __init__(self):
        self.names = None
        self.records = {}

    def start(self, names=None):
        self.names = set(names) if names is not None else True
        self.records = {}

    def active(self, name):
        return self.names is not None and (self.names is True or name in self.names)

    def stop(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _, _ in recs)
            out[name] = {"launches": len(recs), "ms": ms, "bytes": sum(r[2] for r in recs), "flops": sum(r[3] for r in recs)}
        self.names, self.records = None, {}
        return out


meter = KernelMeter()


_FN = {}
TRACE = os.environ.get("MRG_TRACE") == "1"     # debugging aid: log every C-ABI call and synchronise after it,
                                                # so a faulting kernel is the last line of the log


def _trace(name, args):
    import sys
    shown = [hex(a.value) if isinstance(a, ctypes.c_void_p) and a.value else ("NULL" if isinstance(a, ctypes.c_void_p) or a is None else a)
             for a in args]
    print(f"[mrg] {name} {shown}", file=sys.stderr, flush=True)


def call(name, args, nbytes=0, flops=0):
    """Invoke C-ABI function `name`; raise on a non-zero return code."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    if TRACE:
        _trace(name, args)
        code = fn(*args)
        torch.cuda.synchronize()
        if code != 0:
            check(code, name)
        return
    if meter.names is not None and meter.active(name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        code = fn(*args)
        b.record()
        meter.records.setdefault(name, []).append((a, b, nbytes, flops))
    else:
        code = fn(*args)
    if code != 0:
        check(code, name)


def check(code, what=""):
    if code != 0:
        msg = load().mrg_error_string(code).decode()
        raise MrgnasError(f"{what}: {msg} (code {code})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """HOST array of device pointers (NULL for None) for the *_host arguments."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class GatedBranch(ctypes.Structure):
    """include/mrgnas.h: mrg_gated_branch."""
    _fields_ = [("k", ctypes.c_int32), ("s", ctypes.c_void_p), ("rowscale", ctypes.c_void_p),
                ("row_k", ctypes.c_int32), ("row_f", ctypes.c_void_p), ("row_h", ctypes.c_void_p), ("row_uvc", ctypes.c_void_p),
                ("row_ld", ctypes.c_int32), ("b0", ctypes.c_int64), ("b1", ctypes.c_int64), ("row_dq", ctypes.c_void_p),
                ("act", ctypes.c_int32)]


ACTS = {"relu": 0, "tanh": 1}


def gated_branch(spec, row_dq=None, act=0):
    """HOST mrg_gated_branch for spec = dict(k, s, c) and / or dict(row_k, s, row_f, row_h, row_uvc, row_ld, b0, b1) merged, or None.
    row_dq: the [rows] output of mrg_mix_bwd_apply.  Returns the by-reference argument (which keeps the structure alive for the call)."""
    if spec is None:
        if not act:
            return None
        return ctypes.byref(GatedBranch(-1, None, None, -1, None, None, None, 0, 0, 0, None, int(act)))   # the activation alone
    dp = lambda t: None if t is None else t.data_ptr()
    g = GatedBranch(int(spec.get("k", -1)), dp(spec["s"]), dp(spec.get("c")), int(spec.get("row_k", -1)), dp(spec.get("row_f")),
                    dp(spec.get("row_h")), dp(spec.get("row_uvc")), int(spec.get("row_ld", 0)), int(spec.get("b0", 0)), int(spec.get("b1", 0)),
                    dp(row_dq), int(act))
    return ctypes.byref(g)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("MRG_RAW_STREAM", "1") == "1" else None


def stream_of(t):
    """The hipStream_t torch would launch on for t's device (the raw-handle query: ~10x cheaper than building a Stream object,
    which matters in the launch-bound sampled / sharded steps)."""
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(t.device.index if t.device.index is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_hip(*tensors):
    """The kernels read raw HBM pointers: refuse anything else loudly."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MrgnasError("mr_gnas_amd operators need tensors on a HIP device (no CPU fallback); got " + str(t.device))
        if not t.is_contiguous():
            raise MrgnasError("mr_gnas_amd operators need contiguous tensors")
    load()


def f32c(t):
    """float32 + contiguous (what the reference feeds: `.float()` rows, cell_lp.py:32)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
