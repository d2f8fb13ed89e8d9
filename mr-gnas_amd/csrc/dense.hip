// Dense (per-feature) filters fused on the MFMA row GEMM: no torch.cat, no [rows, 2D] copy, the
// gate / scaling applied in the GEMM epilogue.
//   f_dense_op_comp  reference models/operations_lp.py:356-390   out = sigmoid(W_x [s ; s_in] + b_x) * s * 1/3 (* norm)
//   f_comp_op        reference models/operations_lp.py:266-288   out = (W_x [s ; s_in]) * 1/3 * norm   (self rows unscaled)
//   f_dense_op_last  reference models/operations_lp.py:392-401   out = sigmoid(W s + b) * s
//   f_dense_op       reference models/operations_lp.py:345-354   out = sigmoid(W [s ; s_in] + b) * s
// One call handles one direction segment (rows with one weight matrix); the caller loops over
// in / out / self.  MFMA-bound: 2*rows*K*D flop, K = 2D (or D for f_dense_last).
#include "gemm_dispatch.hpp"

namespace mrg {

// kind 0 (gate):   dz = g * s * c * gate * (1 - gate);  gs = g * c * gate     (c = scale * rowscale[row])
// kind 1 (linear): dz = g * c
template <int VEC, int LPR, int KMAX, int KIND>
__global__ __launch_bounds__(MRG_BLOCK) void dense_dz_k(const float* __restrict__ g, const float* __restrict__ s,
                                                        const float* __restrict__ gate, const float* __restrict__ rowscale,
                                                        float scale, float* __restrict__ dz, float* __restrict__ gs,
                                                        int64_t rows, int D, int64_t edge_rows, float scale_self) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    // rows [0, edge_rows): c = scale * rowscale[row]; rows behind them (the self rows of a three-segment call): scale_self
    const float c = r < edge_rows ? scale * (rowscale ? rowscale[r] : 1.0f) : scale_self;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int col = sl + k * LPR;
      if (col < dv) {
        Vec<VEC> gv = Vec<VEC>::load(g + r * D + col * VEC), o;
        if (KIND == 0) {
          Vec<VEC> sv = Vec<VEC>::load(s + r * D + col * VEC), ga = Vec<VEC>::load(gate + r * D + col * VEC), o2;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float gc = gv[j] * c;
            o2[j] = gc * ga[j];
            o[j] = gc * sv[j] * ga[j] * (1.0f - ga[j]);
          }
          o2.store(gs + r * D + col * VEC);
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] = gv[j] * c;
        }
        o.store(dz + r * D + col * VEC);
      }
    }
  }
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_dense_filter_fwd(int kind, const float* s, const float* s_in, const float* W, const float* bias,
                                    const float* rowscale, float scale, float* out, float* gate, void* ws, int64_t rows, int D,
                                    void* stream) {
  if (kind != 0 && kind != 1) return MRG_E_ENUM;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!s || !W || !out) return MRG_E_NULLPTR;
  GemmArgs a{};
  a.A1 = s; a.K1 = D;
  a.A2 = s_in; a.K2 = s_in ? D : 0;
  a.B = W;
  a.bias = bias; a.C = out; a.ldc = D; a.N = D; a.rows = rows;
  a.rowscale = rowscale; a.scale = scale;
  if (kind == 0) {
    a.S = s; a.ld_s = D; a.aux = gate;
    return launch_gemm<EPI_GATE>(a, a.K1 + a.K2, 1, ws, (hipStream_t)stream);
  }
  return launch_gemm<EPI_SCALE>(a, a.K1 + a.K2, 1, ws, (hipStream_t)stream);
}

// ---- the three direction segments of one operator in ONE launch each (split core only) -------------------------------------
// rows [0, b0) use W[0] / bias[0], [b0, b1) W[1] / bias[1] (edge rows: c = scale_edge * norm[row]), [b1, M) W[2] / bias[2]
// (self rows: c = scale_self).  One weight-split launch + one grouped row GEMM instead of three of each; a sampled
// 30 000-edge step graph is launch-latency bound, and on the full graph the three row ranges share their tail rounds.
static bool dense3_shape_ok(int D, int K) { return D > 0 && (K == D || K == 2 * D) && K > 48 && D % 4 == 0; }

extern "C" int64_t mrg_dense_filter3_workspace_bytes(int D, int K) {
  if (!dense3_shape_ok(D, K) || gemm_mode() == 1) return 0;  // 0: not available for this shape / split core switched off (use the per-segment entry points)
  return 3 * (int64_t)(((int64_t)bsplit_bytes_any(D, K) + 255) / 256 * 256);
}

extern "C" int mrg_dense_filter_fwd3(int kind, const float* s, const float* s_in, const float* const* W_host, const float* const* bias_host,
                                     const float* norm, float scale_edge, float scale_self, float* out, float* gate, void* ws,
                                     int64_t b0, int64_t b1, int64_t M, int D, void* stream) {
  if (kind != 0 && kind != 1) return MRG_E_ENUM;
  if (D <= 0 || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  const int K = s_in ? 2 * D : D;
  if (!dense3_shape_ok(D, K)) return MRG_E_SHAPE;
  if (M == 0) return MRG_OK;
  if (!s || !W_host) return MRG_E_NULLPTR;
  if (kind == 0 && !gate) return MRG_E_NULLPTR;
  if (!out && kind != 0) return MRG_E_NULLPTR;     // kind 0, out == NULL: gate only (the consumer recomputes gate * s * c, mrg_gated_branch)
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t lo[3] = {0, b0, b1}, hi[3] = {b0, b1, M};
  const size_t each = (size_t)(((int64_t)bsplit_bytes_any(D, K) + 255) / 256 * 256);
  GemmArgs a{};
  a.A1 = s; a.K1 = D; a.A2 = s_in; a.K2 = s_in ? D : 0;
  a.C = out; a.ldc = D; a.N = D; a.rows = M; a.rowscale = norm;
  if (kind == 0) { a.S = s; a.ld_s = D; a.aux = gate; }
  if (!x3_eligible(a)) return MRG_E_SHAPE;
  const float* Bs[3]; void* outs[3];
  a.grp.n = 3;
  a.grp.bp_stride = (int64_t)each;
  for (int i = 0; i < 3; ++i) {
    const bool live = hi[i] > lo[i];
    if (live && !W_host[i]) return MRG_E_NULLPTR;
    Bs[i] = live ? W_host[i] : nullptr;
    outs[i] = (char*)ws + i * each;
    a.grp.lo[i] = lo[i]; a.grp.hi[i] = live ? hi[i] : lo[i];
    a.grp.bias[i] = bias_host ? bias_host[i] : nullptr;
    a.grp.scale[i] = i < 2 ? scale_edge : scale_self;
    a.grp.use_rowscale[i] = (i < 2 && norm) ? 1 : 0;
  }
  launch_bsplit3_any(kind == 0 ? EPI_GATE : EPI_SCALE, M, Bs, K, 1, D, K, outs, st);
  MRG_LAUNCH_CHECK();
  if (kind == 0) return launch_rowgemm_x3_mode<EPI_GATE>(a, outs[0], st);
  return launch_rowgemm_x3_mode<EPI_SCALE>(a, outs[0], st);
}

// dz (and, for the gated kinds, the direct term of the gradient w.r.t. s):
//   kind 0:  dz = g * s * c * gate * (1 - gate),  gs = g * c * gate
//   kind 1:  dz = g * c                            (gs untouched)
// The caller then runs mrg_linear_bwd_input (dz W[:, :D] accumulated into gs, dz W[:, D:] into gs_in) and
// mrg_linear_bwd_weight (dz^T [s | s_in]).
static int dense_dz_launch(int kind, const float* g, const float* s, const float* gate, const float* rowscale, float scale, float* dz,
                           float* gs, int64_t rows, int D, int64_t edge_rows, float scale_self, void* stream);

extern "C" int mrg_dense_filter_dz(int kind, const float* g, const float* s, const float* gate, const float* rowscale,
                                   float scale, float* dz, float* gs, int64_t rows, int D, void* stream) {
  return dense_dz_launch(kind, g, s, gate, rowscale, scale, dz, gs, rows, D, rows, scale, stream);
}

// all M rows of the three direction segments in one launch: rows [0, b1) are edge rows (c = scale_edge * norm[row]),
// rows [b1, M) self rows (c = scale_self)
extern "C" int mrg_dense_filter_dz3(int kind, const float* g, const float* s, const float* gate, const float* norm, float scale_edge,
                                    float scale_self, float* dz, float* gs, int64_t b1, int64_t M, int D, void* stream) {
  if (b1 < 0 || M < b1) return MRG_E_SHAPE;
  return dense_dz_launch(kind, g, s, gate, norm, scale_edge, dz, gs, M, D, b1, scale_self, stream);
}

static int dense_dz_launch(int kind, const float* g, const float* s, const float* gate, const float* rowscale, float scale, float* dz,
                           float* gs, int64_t rows, int D, int64_t edge_rows, float scale_self, void* stream) {
  if (kind != 0 && kind != 1) return MRG_E_ENUM;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!g || !dz) return MRG_E_NULLPTR;
  if (kind == 0 && (!s || !gate || !gs)) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  RowGeom gm = row_geom(D, aligned16(g) && aligned16(s) && aligned16(gate) && aligned16(gs) && aligned16(dz));
  if (!gm.ok) return MRG_E_SHAPE;
#define CALL(V, L, KM)                                                                                                \
  do {                                                                                                                \
    int grid = grid_for(rows, (MRG_BLOCK / L) * 4);                                                                         \
    if (kind == 0) hipLaunchKernelGGL((dense_dz_k<V, L, KM, 0>), dim3(grid), dim3(MRG_BLOCK), 0, st, g, s, gate, rowscale, scale, dz, gs, rows, D, edge_rows, scale_self); \
    else hipLaunchKernelGGL((dense_dz_k<V, L, KM, 1>), dim3(grid), dim3(MRG_BLOCK), 0, st, g, s, gate, rowscale, scale, dz, gs, rows, D, edge_rows, scale_self); \
  } while (0)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
