// Dense (per-feature) filters fused on the MFMA row GEMM: no torch.cat, no [rows, 2D] copy, the
// gate / scaling applied in the GEMM epilogue.
//   f_dense_op_comp  reference models/operations_lp.py:356-390   out = sigmoid(W_x [s ; s_in] + b_x) * s * 1/3 (* norm)
//   f_comp_op        reference models/operations_lp.py:266-288   out = (W_x [s ; s_in]) * 1/3 * norm   (self rows unscaled)
//   f_dense_op_last  reference models/operations_lp.py:392-401   out = sigmoid(W s + b) * s
//   f_dense_op       reference models/operations_lp.py:345-354   out = sigmoid(W [s ; s_in] + b) * s
// One call handles one direction segment (rows with one weight matrix); the caller loops over
// in / out / self.  MFMA-bound: 2*rows*K*D flop, K = 2D (or D for f_dense_last).
#include "gemm_x3p.hpp"

namespace mrg {

// kind 0 (gate):   dz = g * s * c * gate * (1 - gate);  gs = g * c * gate     (c = scale * rowscale[row])
// kind 1 (linear): dz = g * c
template <int VEC, int LPR, int KMAX, int KIND>
__global__ __launch_bounds__(MRG_BLOCK) void dense_dz_k(const float* __restrict__ g, const float* __restrict__ s,
                                                        const float* __restrict__ gate, const float* __restrict__ rowscale,
                                                        float scale, float* __restrict__ dz, float* __restrict__ gs,
                                                        int64_t rows, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = threadIdx.x / LPR;
  const int dv = D / VEC;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float c = scale * (rowscale ? rowscale[r] : 1.0f);
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

// dz (and, for the gated kinds, the direct term of the gradient w.r.t. s):
//   kind 0:  dz = g * s * c * gate * (1 - gate),  gs = g * c * gate
//   kind 1:  dz = g * c                            (gs untouched)
// The caller then runs mrg_linear_bwd_input (dz W[:, :D] accumulated into gs, dz W[:, D:] into gs_in) and
// mrg_linear_bwd_weight (dz^T [s | s_in]).
extern "C" int mrg_dense_filter_dz(int kind, const float* g, const float* s, const float* gate, const float* rowscale,
                                   float scale, float* dz, float* gs, int64_t rows, int D, void* stream) {
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
    int grid = grid_for(rows, (MRG_BLOCK / L) * 4);                                                                   \
    if (kind == 0) hipLaunchKernelGGL((dense_dz_k<V, L, KM, 0>), dim3(grid), dim3(MRG_BLOCK), 0, st, g, s, gate, rowscale, scale, dz, gs, rows, D); \
    else hipLaunchKernelGGL((dense_dz_k<V, L, KM, 1>), dim3(grid), dim3(MRG_BLOCK), 0, st, g, s, gate, rowscale, scale, dz, gs, rows, D); \
  } while (0)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
