// Dense (per-feature) filters fused on the MFMA row GEMM: no torch.cat, no [rows, 2D] copy, the
// gate / scaling applied in the GEMM epilogue.
//   f_dense_op_comp  reference models/operations_lp.py:356-390   out = sigmoid(W_x [s ; s_in] + b_x) * s * 1/3 (* norm)
//   f_comp_op        reference models/operations_lp.py:266-288   out = (W_x [s ; s_in]) * 1/3 * norm   (self rows unscaled)
//   f_dense_op_last  reference models/operations_lp.py:392-401   out = sigmoid(W s + b) * s
//   f_dense_op       reference models/operations_lp.py:345-354   out = sigmoid(W [s ; s_in] + b) * s
// One call handles one direction segment (rows with one weight matrix); the caller loops over
// in / out / self.  MFMA-bound: 2*rows*K*D flop, K = 2D (or D for f_dense_last).
#include "gemm.hpp"

namespace mrg {
int launch_wgrad(const float* gY, const float* X1, const float* X2, int K1, int K2, float* gW, float* gbias, void* ws,
                 int64_t rows, int Nout, hipStream_t st);
int64_t wgrad_workspace_bytes(int64_t rows, int K, int Nout);

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

static int64_t round16(int64_t b) { return (b + 15) & ~(int64_t)15; }

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_dense_filter_fwd(int kind, const float* s, const float* s_in, const float* W, const float* bias,
                                    const float* rowscale, float scale, float* out, float* gate, int64_t rows, int D,
                                    void* stream) {
  if (kind != 0 && kind != 1) return MRG_E_ENUM;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!s || !W || !out) return MRG_E_NULLPTR;
  GemmArgs a{};
  a.A1 = s; a.K1 = D;
  a.A2 = s_in; a.K2 = s_in ? D : 0;
  a.B = W; a.ldb = a.K1 + a.K2;
  a.bias = bias; a.C = out; a.ldc = D; a.N = D; a.rows = rows;
  a.rowscale = rowscale; a.scale = scale;
  if (kind == 0) {
    a.S = s; a.ld_s = D; a.aux = gate;
    return launch_rowgemm<EPI_GATE>(a, (hipStream_t)stream);
  }
  return launch_rowgemm<EPI_SCALE>(a, (hipStream_t)stream);
}

extern "C" int64_t mrg_dense_filter_bwd_workspace_bytes(int64_t rows, int D, int has_in) {
  if (rows < 0 || D <= 0) return 0;
  const int K = has_in ? 2 * D : D;
  return round16((int64_t)rows * D * 4) + round16((int64_t)K * D * 4) + wgrad_workspace_bytes(rows, K, D) + 64;
}

extern "C" int mrg_dense_filter_bwd(int kind, const float* g, const float* s, const float* s_in, const float* W,
                                    const float* gate, const float* rowscale, float scale, float* gs, float* gs_in,
                                    float* gW, float* gbias, void* ws, int64_t rows, int D, void* stream) {
  if (kind != 0 && kind != 1) return MRG_E_ENUM;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (!gW) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  const int K = s_in ? 2 * D : D;
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)D * K, st);
    if (e == hipSuccess && gbias) e = hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)D, st);
    return (int)e;
  }
  if (!g || !s || !W || !gs) return MRG_E_NULLPTR;
  if (kind == 0 && !gate) return MRG_E_NULLPTR;
  if (s_in && !gs_in) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  char* wsb = (char*)ws;
  float* dz = (float*)wsb;
  float* Wt = (float*)(wsb + round16(rows * (int64_t)D * 4));
  void* ws_w = wsb + round16(rows * (int64_t)D * 4) + round16((int64_t)K * D * 4);

  // 1. dz (and the direct term of gs for the gated kinds)
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

  // 2. W^T halves, row-major [D(k-col)][D(out)] each
  launch_transpose(W, Wt, D, D, K, st);                                   // (W[:, :D])^T
  if (s_in) launch_transpose(W + D, Wt + (int64_t)D * D, D, D, K, st);    // (W[:, D:])^T
  MRG_LAUNCH_CHECK();

  // 3. gs (+)= dz * W[:, :D]      4. gs_in = dz * W[:, D:]
  GemmArgs a{};
  a.A1 = dz; a.K1 = D; a.B = Wt; a.ldb = D; a.C = gs; a.ldc = D; a.N = D; a.rows = rows; a.act = MRG_ACT_NONE;
  int rc;
  if (kind == 0) {
    a.Cin = gs; a.ld_cin = D;
    rc = launch_rowgemm<EPI_ACCUM>(a, st);
  } else {
    rc = launch_rowgemm<EPI_BIAS_ACT>(a, st);
  }
  if (rc != MRG_OK) return rc;
  if (s_in) {
    GemmArgs b{};
    b.A1 = dz; b.K1 = D; b.B = Wt + (int64_t)D * D; b.ldb = D; b.C = gs_in; b.ldc = D; b.N = D; b.rows = rows; b.act = MRG_ACT_NONE;
    rc = launch_rowgemm<EPI_BIAS_ACT>(b, st);
    if (rc != MRG_OK) return rc;
  }
  // 5. gW = dz^T [s | s_in], gbias = column sums of dz
  return launch_wgrad(dz, s, s_in, D, s_in ? D : 0, gW, gbias, ws_w, rows, D, st);
}
