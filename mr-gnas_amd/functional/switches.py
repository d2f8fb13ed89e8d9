"""Lab switches of the autograd layer (each names its environment variable): the shipped defaults are the measured winners;
the other setting of every switch is a tested comparison point (tests/, tools/).  ``MASK_TAP`` is test instrumentation."""
import os

ROW_FACTOR = os.environ.get("MRG_ROW_FACTOR", "1") == "1"     # lab switch: 0 = f_sparse_comp's output is stored for the epilogue
LAZY_ASUM = os.environ.get("MRG_LAZY_ASUM", "1") == "1"     # lab switch: 0 = a_sum's [M, D] input gradient is materialised for the fan-in sum
FUSED_AMAX = os.environ.get("MRG_FUSED_AMAX", "1") == "1"      # lab switch: 0 = linear + segmented max as separate launches
# below this many edges the step is launch-bound and the fused form's extra launches (key memset, unpack pass, mask product)
# cost more than the [E, D] round trip they save: 30 000-edge sampled step 18.4 vs 17.7 ms
FUSED_AMAX_MIN_ROWS = int(os.environ.get("MRG_FUSED_AMAX_MIN_ROWS", "100000"))
FUSED_AMEAN = os.environ.get("MRG_FUSED_AMEAN", "1") == "1"    # lab switch: 0 = linear, then the span reducer over the [E, D] messages
WIDE_BWD_INPUT = os.environ.get("MRG_WIDE_BWD_INPUT", "1") == "1"     # lab switch: 0 = the [B, N] scorer's input gradient on the row GEMM
NODE_LINEAR = os.environ.get("MRG_NODE_LINEAR", "1") == "1"      # lab switch: 0 = the node-level nn.Linear modules stay on torch (Tensile)
# Test instrumentation (tests/test_configs_gpu.py, "mask replay"): when set, called as MASK_TAP(bns, masks) right after a fused
# epilogue's combine with the ReLU decision of every candidate, [rows, D] bool each, taken from the combine kernel ITSELF (one extra
# launch per candidate with a one-hot weight vector: w_k * relu(bn_k(y_k)) with w_k = 1, so `> 0` is the kernel's own decision, not a
# re-evaluation that could round differently).  None in the product.
MASK_TAP = None
FOLD_IDENTITY = os.environ.get("MRG_FOLD_IDENTITY", "1") == "1"     # lab switch: 0 = f_identity's gradient stays a tensor of its own
CELL_ZERO_FUSED = os.environ.get("MRG_CELL_ZERO_FUSED", "1") == "1"     # lab switch: 0 = three gather-compose launches + the generic epilogue
FORK_MIN_ROWS = int(os.environ.get("MRG_FORK_MIN_ROWS", str(1 << 17)))      # below this many rows an operator is launch-bound: no side streams
SEGMENT_STREAMS = int(os.environ.get("MRG_SEGMENT_STREAMS", "3"))   # streams the direction segments of one operator use
FOLD_ROW_SCALE = os.environ.get("MRG_FOLD_ROW_SCALE", "1") == "1"          # lab switch: 0 = f_comp's dz pass stays a launch of its own
GROUPED_SEGMENTS = os.environ.get("MRG_GROUPED_SEGMENTS", "1") == "1"    # lab switch: 0 = one launch per direction segment
DENSE_PAIR = os.environ.get("MRG_DENSE_PAIR", "1") == "1"       # lab switch: 0 = f_dense_comp and f_comp of a MixedOp as two autograd nodes
GATED_RECOMPUTE = os.environ.get("MRG_GATED_RECOMPUTE", "1") == "1"     # lab switch: 0 = f_dense_comp's output is stored for the epilogue
COMPGCN_TAIL = os.environ.get("MRG_COMPGCN_TAIL", "1") == "1"     # lab switch: 0 = CompGraphConv's BatchNorm -> tanh tail on torch kernels
SPARSE_AMAX_BWD = os.environ.get("MRG_SPARSE_AMAX_BWD", "1") == "1"   # lab switch: 0 = a_max's input gradient as seg_bwd_k + the dense row GEMM
