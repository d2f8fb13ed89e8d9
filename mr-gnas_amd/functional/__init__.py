"""autograd Functions over the C ABI (include/mrgnas.h), one module per kernel family.

Each Function enqueues HIP kernels of libmrgnas_hip.so on torch's current stream through ctypes; tensors are only used for memory
and stream plumbing.  Inputs are never modified; outputs are fresh tensors.  Every call site states the algorithmic bytes / flops of
the launch so bench.py can price the kernels against the roofline.

This package re-exports every name of its modules, so callers keep writing ``from mr_gnas_amd import functional as K``:

  switches    lab switches (``K.switches.NAME``; each names its environment variable) and the test tap
  _base       Shared plumbing of the autograd Functions: op / reduce / activation codes, workspace helpers, plan counts, side streams.
  candidates  The explicit MixedOp candidate objects (Link, Candidate) the operators hand to the fused epilogue.
  gcs         a9: fused gather -> compose -> segmented sum (CompGCN aggregation, csrc/fused_gcs.hip) and its backward.
  reducers    a4 / a5 / a6: destination-segmented reducers (a_sum / a_mean / a_max) and the Linear + ReLU + reduce fusions (csrc/segreduce.hip, linear.hip).
  fanin       Gradient fan-in of a tensor with several readers: K-way sums and the Fan alias bookkeeping (csrc/accum.hip).
  compose_gather  a1: compose ops (pre_mult / pre_sub / pre_add) and G, the row gather that feeds them (csrc/compose.hip).
  gates       a2 / a3: collapsed scalar gates f_sparse_comp / f_sparse_last (csrc/gate.hip).
  row_linear  Dense linear on rows: the split-bf16 / exact-f32 MFMA row GEMM and its two gradients (csrc/linear.hip).
  dense       Dense (per-feature) filters f_dense_comp / f_comp / f_dense_last on the MFMA row GEMM (csrc/dense.hip, linear.hip).
  mixed       X: the MixedOp epilogue  out = sum_k w_k * ReLU(BatchNorm_k(y_k))  (csrc/mixedop.hip).
  cell_zero   Cell zero: the MixedOp over the compose candidates, recomputed from the entity / relation tables (csrc/mixedop.hip zero_*).
  scoring     The step after the path: DistMult triple scoring and the [B, N] score functions (csrc/scoring.hip).
"""
from . import switches                                   # noqa: F401
from ._base import (  # noqa: F401
    bump_counters, deferred_counters,
    COMPOSE, REDUCE, ACT, gate_ld, same_rows, _same_memory, _cnt, _ws, _WS_BYTES, _ws_bytes, _SIDE_STREAMS, Fork,
)
from .candidates import (  # noqa: F401
    Link, Candidate,
)
from .gcs import (  # noqa: F401
    GCS, fused_gcs, span_gcs, ComposePlan, _ComposeAggregate, compose_aggregate,
)
from .reducers import (  # noqa: F401
    _seg_fwd, _seg_bwd, _SegReduce, seg_reduce, _AggRows, aggregate_rows, _fused_agg_ws, _LinReluAgg, linear_relu_aggregate,
    _LinReluPartial, linear_relu_partial, _SumPartial, sum_partial,
)
from .fanin import (  # noqa: F401
    sum_buffers, _Fanout, _FanBox, Fan,
)
from .compose_gather import (  # noqa: F401
    _Compose, compose, gather_rows, GatherPlan, _Gather, gather, LazyRows, _pair_meta, _GatherCompose,
)
from .gates import (  # noqa: F401
    _Gate, gate_comp, gate_last, _GateRow, gate_comp_row_factor,
)
from .row_linear import (  # noqa: F401
    _Linear, linear, module_linear,
)
from .dense import (  # noqa: F401
    _DenseFilter, _FoldHalves, dense_filter_comp, _DensePair, dense_pair_available, _GATED_C, _gated_rowscale, dense_filter_pair,
    dense_filter_single,
)
from .mixed import (  # noqa: F401
    _MixCfg, _row_candidate_as_s, _MixedEpilogue, mixed_epilogue, PreparedEpilogue, _all_reduce_sum, StatChain,
    mixed_epilogue_prepare,
)
from .cell_zero import (  # noqa: F401
    _CellZeroMixed, cell_zero_mixed,
)
from .scoring import (  # noqa: F401
    ScorePlan, _DistMult, distmult_score, distmult_scores_all, _TransE, transe_scores_all,
)
from .._lib import ptr_array, call, f32c, ptr, require_hip, stream_of   # noqa: F401  (part of the module's historical surface)
