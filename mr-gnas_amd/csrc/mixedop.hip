// MixedOp epilogue:  out = sum_k w_k * ReLU(BatchNorm_k(y_k))  over K operator outputs y_k [rows, D]
//   reference models/cell_lp.py:25-33 (MixedOp.forward / op_forward) with nn.BatchNorm1d in
//   training mode (:21) -- in the reference 3 ATen launches per branch + the weighted sum,
//   each a full pass over [rows, D], and as many again in backward.
// Here: one statistics pass (column sums in float64, deterministic two-stage reduction), one
// combine pass (reads every y_k once, writes out once), and in backward one reduction pass
// and one apply pass.  A NULL branch pointer stands for an all-zero operator output (f_zero).
// HBM-bound: algorithmic bytes  stats 4*D*rows*Kt,  fwd 4*D*rows*(Kt+1),
// bwd-reduce 4*D*rows*(Kt+1),  bwd-apply 4*D*rows*(2*Kt+1)   (Kt = non-NULL branches).
#include "common.hpp"

#define MRG_MIX_MAXK 8

namespace mrg {

struct PtrPack { const float* p[MRG_MIX_MAXK]; };
struct MutPack { float* p[MRG_MIX_MAXK]; };
// optional per-candidate row scale applied to the candidate's output gradient as it is written (mix_bwd_apply_k):
// gy_k[r] *= r < edge_rows[k] ? scale[k] * (rs[k] ? rs[k][r] : 1) : self_scale[k]      when on[k]
// on[k] == 2: the gated form (f_dense_comp, mrg_dense_filter_dz kind 0): with gc = gy * c, the candidate's gradient buffer receives
// dz = gc * s * gate * (1 - gate) and gs_out[k] the direct term gc * gate of the gradient w.r.t. s.
// add_from[k] (gated candidates only, -1 = none): the index of ANOTHER candidate whose own gradient gy is ADDED to gs_out[k] instead
// of being stored -- f_identity of the same MixedOp: its output IS the operand s of candidate k, so both are gradients w.r.t. the
// same rows and the state's fan-in sum would add them anyway (one [rows, D] write here and one read there less).
// full[k] (optional): the same multiplier expanded over ALL rows by the caller -- loaded with the candidates' rows at the top of the
// trip instead of conditionally (edge rows only) between the arithmetic and the stores, where its latency was exposed.
struct RowScalePack { const float* rs[MRG_MIX_MAXK]; float scale[MRG_MIX_MAXK]; float self_scale[MRG_MIX_MAXK]; int64_t edge_rows[MRG_MIX_MAXK]; int on[MRG_MIX_MAXK];
                      const float* full[MRG_MIX_MAXK];
                      const float* s[MRG_MIX_MAXK]; const float* gate[MRG_MIX_MAXK]; float* gs_out[MRG_MIX_MAXK]; int add_from[MRG_MIX_MAXK]; };

// A gated candidate that is never stored (f_dense_comp, reference models/operations_lp.py:356-390): ys.p[k] holds its GATE
// sigmoid(W [s ; s_in] + b) and the candidate's value is recomputed here as  gate * s * c_r  -- the expression, and its order, of
// the row GEMM's gate epilogue (gemm.hpp EPI_GATE: g * in * cs), so every statistic, the output and every gradient are the
// stored form's bit for bit.  The gate is needed by the candidate's backward anyway; its [rows, D] output is one write (GEMM) and
// four reads (statistics, combine, backward reduction, backward apply) that do not happen; s is the MixedOp's input state, which
// the f_identity candidate of the same MixedOp reads at the same place (one L2 / L1 hit more, no HBM pass).   k < 0: none.
struct GatedPack { int k; const float* s; const float* c;
                   int pair_k;      // pair_k: the candidate whose stored output is s itself (f_identity), -1 = none  (statistics kernel)
                   // the row-scaled candidate (f_sparse_op_comp, never stored either): y = s * rf[r]; ys.p[rk] == s.   rk < 0: none
                   int rk; const float* rf;
                   // ... its backward (mix_bwd_apply_k only): rh[r] = d pre-activation / d row dot, the collapsed gate vectors
                   // uvc[seg][uld] of the three direction segments [0, b0) [b0, b1) [b1, rows), rdq[r] out (the gradient w.r.t. rf[r])
                   const float* rh; const float* uvc; int uld; int64_t b0, b1; float* rdq;
                   // round 5, "static step graphs" (mrg_set_dynamic_rows): the number of VALID rows lives in device memory; rows at
                   // and beyond it are capacity padding -- left out of every statistic, written as zeros by the passes that write
                   const int32_t* vrows;
                   // the activation behind the BatchNorm: 0 = ReLU (every MixedOp of the search space), 1 = tanh (CompGraphConv's tail,
                   // reference models/compgcn.py:100-111).  A template parameter of the three kernels that apply it (a run-time branch
                   // cost mix_bwd_apply_k 14 %); the tanh instances exist for un-gated launches of at most five candidates.
                   int act; };

// rows that count: min(rows, *vrows) when the launch's row count is a registered capacity, else rows
__device__ __forceinline__ int64_t valid_rows(const int32_t* vrows, int64_t rows) {
  if (vrows == nullptr) return rows;
  const int64_t v = (int64_t)*vrows;
  return v < rows ? (v < 0 ? 0 : v) : rows;
}

// c: the candidate's per-row multiplier for ALL rows (the caller expands scale_edge * norm on edge rows, scale_self on self rows,
// once per graph): an unconditional load.  A conditional one (edge rows only) was compiled into an exec-masked block that waited for
// it -- s_waitcnt vmcnt(0) -- before the candidates' loads were issued: two dependent memory latencies per trip.
__device__ __forceinline__ float gated_rowscale(const GatedPack& gp, int64_t r) { return gp.c[r]; }

// ---- column statistics: sums[k][0][c] = sum_r y_k[r][c], sums[k][1][c] = sum_r y_k[r][c]^2 (float64)
// block reduction of one candidate's per-lane column sums into its per-block partial [2][D] (every thread of the block calls it)
template <int VEC, int LPR, int KMAX>
__device__ __forceinline__ void colstats_flush(const double (&s1)[KMAX][VEC], const double (&s2)[KMAX][VEC], double* red, double* __restrict__ dst,
                                               int D, int sl, int rw) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KMAX; ++q) {
    int c = sl + q * LPR;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[(rw * 2 + 0) * WIDTH + c * VEC + j] = s1[q][j];
      red[(rw * 2 + 1) * WIDTH + c * VEC + j] = s2[q][j];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * D; t += MRG_BLOCK) {
    int which = t / D, c = t - which * D;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < RPB; ++q) acc += red[(q * 2 + which) * WIDTH + c];
    dst[t] = acc;
  }
}

// GATED: candidate gp.k is recomputed from its gate and candidate gp.rk from its row factor (GatedPack); with gp.pair_k >= 0 --
// f_identity of the same MixedOp, whose output IS the multiplicand s -- up to three candidates are functions of the same rows s
// and share ONE sweep (s is read once for all; a sweep per candidate would read it from HBM each time, the tensors being far
// larger than the caches).  Each candidate's sums see the same values in the same order as a sweep of its own.
template <int VEC, int LPR, int KMAX, bool GATED>
__global__ __launch_bounds__(MRG_BLOCK) void mix_colstats_k(PtrPack ys, int K, int64_t rows, int D, double* __restrict__ ws, GatedPack gp) {
  rows = valid_rows(gp.vrows, rows);                        // capacity padding is not part of any statistic
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  __shared__ double red[RPB * 2 * WIDTH];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int64_t step = (int64_t)gridDim.x * RPB;
  int first = MRG_MIX_MAXK;                                // the shared sweep runs at the first of its candidates
  if (GATED) {
    if (gp.k >= 0 && gp.k < first) first = gp.k;
    if (gp.pair_k >= 0 && gp.pair_k < first) first = gp.pair_k;
    if (gp.rk >= 0 && gp.rk < first) first = gp.rk;
  }
  for (int k = 0; k < K; ++k) {
    const float* __restrict__ y = ys.p[k];
    const bool shared = GATED && (k == gp.k || k == gp.pair_k || k == gp.rk);
    if (shared && k != first) continue;
    double s1[KMAX][VEC], s2[KMAX][VEC];
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s1[q][j] = 0.0; s2[q][j] = 0.0; }
    if (shared) {
      // s1 / s2: s itself (candidate pair_k); t1 / t2: gate * s * c_r (candidate gp.k); u1 / u2: s * f_r (candidate gp.rk)
      const float* __restrict__ gate = gp.k >= 0 ? ys.p[gp.k] : gp.s;
      double t1[KMAX][VEC], t2[KMAX][VEC], u1[KMAX][VEC], u2[KMAX][VEC];
#pragma unroll
      for (int q = 0; q < KMAX; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { t1[q][j] = 0.0; t2[q][j] = 0.0; u1[q][j] = 0.0; u2[q][j] = 0.0; }
      int64_t r = (int64_t)blockIdx.x * RPB + rw;
      for (; r + step < rows; r += 2 * step) {              // two rows per trip: four independent 16-byte loads in flight per lane
        const float ck0 = gated_rowscale(gp, r), ck1 = gated_rowscale(gp, r + step);
        const float rf0 = gp.rf[r], rf1 = gp.rf[r + step];
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
          int c = sl + q * LPR;
          if (c < dv) {
            const Vec<VEC> sv0 = Vec<VEC>::load(gp.s + r * D + c * VEC), sv1 = Vec<VEC>::load(gp.s + (r + step) * D + c * VEC);
            const Vec<VEC> ga0 = Vec<VEC>::load(gate + r * D + c * VEC), ga1 = Vec<VEC>::load(gate + (r + step) * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              double d = (double)sv0[j]; s1[q][j] += d; s2[q][j] += d * d;
              d = (double)sv1[j]; s1[q][j] += d; s2[q][j] += d * d;
              d = (double)(ga0[j] * sv0[j] * ck0); t1[q][j] += d; t2[q][j] += d * d;
              d = (double)(ga1[j] * sv1[j] * ck1); t1[q][j] += d; t2[q][j] += d * d;
              d = (double)(sv0[j] * rf0); u1[q][j] += d; u2[q][j] += d * d;
              d = (double)(sv1[j] * rf1); u1[q][j] += d; u2[q][j] += d * d;
            }
          }
        }
      }
      for (; r < rows; r += step) {
        const float ck = gated_rowscale(gp, r), rf = gp.rf[r];
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
          int c = sl + q * LPR;
          if (c < dv) {
            const Vec<VEC> sv = Vec<VEC>::load(gp.s + r * D + c * VEC), ga = Vec<VEC>::load(gate + r * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              double d = (double)sv[j]; s1[q][j] += d; s2[q][j] += d * d;
              d = (double)(ga[j] * sv[j] * ck); t1[q][j] += d; t2[q][j] += d * d;
              d = (double)(sv[j] * rf); u1[q][j] += d; u2[q][j] += d * d;
            }
          }
        }
      }
      if (gp.k >= 0) colstats_flush<VEC, LPR, KMAX>(t1, t2, red, ws + ((int64_t)blockIdx.x * K + gp.k) * 2 * D, D, sl, rw);
      if (gp.pair_k >= 0) colstats_flush<VEC, LPR, KMAX>(s1, s2, red, ws + ((int64_t)blockIdx.x * K + gp.pair_k) * 2 * D, D, sl, rw);
      if (gp.rk >= 0) colstats_flush<VEC, LPR, KMAX>(u1, u2, red, ws + ((int64_t)blockIdx.x * K + gp.rk) * 2 * D, D, sl, rw);
      continue;
    }
    if (y != nullptr) {
      // four rows per trip: four independent loads in flight per lane (one load per trip left HBM latency exposed)
      int64_t r = (int64_t)blockIdx.x * RPB + rw;
      for (; r + 3 * step < rows; r += 4 * step) {
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
          int c = sl + q * LPR;
          if (c < dv) {
            Vec<VEC> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = Vec<VEC>::load(y + (r + u * step) * D + c * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int j = 0; j < VEC; ++j) { double d = (double)v[u][j]; s1[q][j] += d; s2[q][j] += d * d; }
          }
        }
      }
      for (; r < rows; r += step) {
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
          int c = sl + q * LPR;
          if (c < dv) {
            Vec<VEC> v = Vec<VEC>::load(y + r * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { double d = (double)v[j]; s1[q][j] += d; s2[q][j] += d * d; }
          }
        }
      }
    }
    colstats_flush<VEC, LPR, KMAX>(s1, s2, red, ws + ((int64_t)blockIdx.x * K + k) * 2 * D, D, sl, rw);
  }
}

// generic ordered reduction of per-block partial vectors: out[t] = sum_b ws[b*len + t]
template <typename T>
__global__ void mix_reduce_k(const T* __restrict__ ws, T* __restrict__ out, int nblocks, int len) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= len) return;
  T acc = 0;
  for (int b = 0; b < nblocks; ++b) acc += ws[(int64_t)b * len + t];
  out[t] = acc;
}

// ---- finalize forward: statistics -> per-column coefficients, running-stat update
// coef[k][0]=scale=gamma*invstd, [1]=shift=beta-mean*scale, [2]=invstd, [3]=mean*invstd
__global__ void mix_finalize_fwd_k(const double* __restrict__ sums, PtrPack gamma, PtrPack beta, MutPack rmean, MutPack rvar,
                                   int K, double total_rows, int D, float eps, float momentum, float* __restrict__ coef,
                                   const int32_t* __restrict__ vrows) {
  if (vrows) total_rows = *vrows > 0 ? (double)*vrows : 1.0;
  int k = blockIdx.y;
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D || k >= K) return;
  double mean = sums[(k * 2 + 0) * D + c] / total_rows;
  double var = sums[(k * 2 + 1) * D + c] / total_rows - mean * mean;
  if (var < 0) var = 0;
  double invstd = 1.0 / sqrt(var + (double)eps);
  float g = gamma.p[k] ? gamma.p[k][c] : 1.f, b = beta.p[k] ? beta.p[k][c] : 0.f;
  float* o = coef + (int64_t)k * 4 * D;
  o[c] = (float)(g * invstd);
  o[D + c] = (float)(b - mean * g * invstd);
  o[2 * D + c] = (float)invstd;
  o[3 * D + c] = (float)(mean * invstd);
  if (rmean.p[k]) {
    double unbiased = total_rows > 1 ? var * (total_rows / (total_rows - 1.0)) : var;
    rmean.p[k][c] = (1.f - momentum) * rmean.p[k][c] + momentum * (float)mean;
    rvar.p[k][c] = (1.f - momentum) * rvar.p[k][c] + momentum * (float)unbiased;
  }
}

// The ordered reduction of the per-block statistics and the finalize step in ONE launch (single-GPU path: no all-reduce sits
// between them): a 64 x 16 block owns 64 columns of one candidate, reduces their sum and sum of squares over the nb partial
// rows exactly like ordered_reduce_k and turns them into the coefficients.  Bit-identical to the two-launch form.
__global__ void mix_reduce_finalize_fwd_k(const double* __restrict__ ws, int nb, PtrPack gamma, PtrPack beta, MutPack rmean, MutPack rvar,
                                          int K, double total_rows, int D, float eps, float momentum, float* __restrict__ coef,
                                          const int32_t* __restrict__ vrows) {
  if (vrows) total_rows = *vrows > 0 ? (double)*vrows : 1.0;
  __shared__ double part[2][16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx, k = blockIdx.y;
  const int ld = K * 2 * D;
#pragma unroll
  for (int j = 0; j < 2; ++j) part[j][ty][tx] = c < D ? ordered_partial<double>(ws, 0, nb, ld, (k * 2 + j) * D + c, ty) : 0.0;
  __syncthreads();
  if (ty != 0 || c >= D) return;
  double tot[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double t = part[j][0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += part[j][i][tx];
    tot[j] = t;
  }
  double mean = tot[0] / total_rows;
  double var = tot[1] / total_rows - mean * mean;
  if (var < 0) var = 0;
  double invstd = 1.0 / sqrt(var + (double)eps);
  float g = gamma.p[k] ? gamma.p[k][c] : 1.f, b = beta.p[k] ? beta.p[k][c] : 0.f;
  float* o = coef + (int64_t)k * 4 * D;
  o[c] = (float)(g * invstd);
  o[D + c] = (float)(b - mean * g * invstd);
  o[2 * D + c] = (float)invstd;
  o[3 * D + c] = (float)(mean * invstd);
  if (rmean.p[k]) {
    double unbiased = total_rows > 1 ? var * (total_rows / (total_rows - 1.0)) : var;
    rmean.p[k][c] = (1.f - momentum) * rmean.p[k][c] + momentum * (float)mean;
    rvar.p[k][c] = (1.f - momentum) * rvar.p[k][c] + momentum * (float)unbiased;
  }
}

// ---- forward combine
// KB: the candidate slots the kernel is unrolled for (K <= KB): 5 covers every MixedOp of the search space (3, 4 or 5 candidates)
// with 3/8 fewer registers than the general 8 -- one more wave per SIMD.
template <int VEC, int LPR, int KMAX, bool GATED, int KB, int ACT = 0>
__global__ __launch_bounds__(MRG_BLOCK) void mix_fwd_k(PtrPack ys, int K, const float* __restrict__ coef, const float* __restrict__ w,
                                                       float* __restrict__ out, int64_t rows, int D,
                                                       const float* __restrict__ addend, GatedPack gp) {
  extern __shared__ float lds[];                 // [K][2][D] scale, shift
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int t = threadIdx.x; t < K * 2 * D; t += MRG_BLOCK) {
    int k = t / (2 * D), rem = t - k * 2 * D;
    lds[t] = coef[(int64_t)k * 4 * D + rem];
  }
  __syncthreads();
  float wk[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) wk[k] = k < K ? w[k] : 0.f;
  const int64_t nvalid = valid_rows(gp.vrows, rows);
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const bool pad = r >= nvalid;                            // capacity padding: the state keeps zero rows there
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        // `addend`: the output of the MixedOp this one is summed with (reference models/cell_lp.py:104-113: sum of the
        // MixedOps feeding a state) -- added here instead of by a separate full-size kernel
        Vec<VEC> acc = addend ? Vec<VEC>::load(addend + r * D + c * VEC) : Vec<VEC>::fill(0.f);
        // the recomputed candidate's row scale and multiplicand: issued FIRST (loads return in order: the candidates before
        // it can be consumed while the later loads are in flight).  The candidates that ARE s (f_identity, the row-scaled one) still
        // load it themselves: the repeated addresses hit in L1, and taking them from gsv by a select was measured slower (lab: 413 vs
        // 409 us, backward reduction 482 vs 425 us) and let the compiler re-associate the products (1-ulp differences)
        float gck = 1.f, rfv = 1.f;
        Vec<VEC> gsv = Vec<VEC>::fill(0.f);
        if constexpr (GATED) {
          gck = gated_rowscale(gp, r);
          rfv = gp.rf[r];
          gsv = Vec<VEC>::load(gp.s + r * D + c * VEC);
        }
        // phase 1: every branch's load is issued before any is used (a load consumed inside its own `if (k < K)`
        // block leaves one 16-byte load in flight per lane)
        Vec<VEC> vin[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          vin[k] = Vec<VEC>::fill(0.f);
          if (k < K && ys.p[k]) vin[k] = Vec<VEC>::load(ys.p[k] + r * D + c * VEC);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          if (k < K) {
            Vec<VEC> v = vin[k];
            if constexpr (GATED) {                         // the recomputed candidates: gate -> gate * s * c_r, s -> s * 1 * f_r; the others * 1 * 1
              const bool isg = k == gp.k;                  // (selects, not a branch: the loads above stay in flight together)
              const float cm = isg ? gck : (k == gp.rk ? rfv : 1.0f);
#pragma unroll
              for (int j = 0; j < VEC; ++j) v[j] = v[j] * (isg ? gsv[j] : 1.0f) * cm;
            }
            Vec<VEC> sc = Vec<VEC>::load(lds + (k * 2 + 0) * D + c * VEC), sh = Vec<VEC>::load(lds + (k * 2 + 1) * D + c * VEC);
            if constexpr (ACT == 0) {
#pragma unroll
              for (int j = 0; j < VEC; ++j) {
                float z = v[j] * sc[j] + sh[j];
                acc[j] += wk[k] * (z > 0.f ? z : 0.f);
              }
            } else {
#pragma unroll
              for (int j = 0; j < VEC; ++j) acc[j] += wk[k] * tanhf(v[j] * sc[j] + sh[j]);
            }
          }
        }
        if (pad) acc = Vec<VEC>::fill(0.f);
        acc.store(out + r * D + c * VEC);
      }
    }
  }
}

// ---- backward reduce: per branch  red[k][0] = sum gr, [1] = sum gr*xhat, [2] = sum g*relu(z)  (gr = w g [z>0])
// rows outermost: g is read once, every y_k once; per-branch column accumulators live in registers.
template <int VEC, int LPR, int KMAX, int KB, bool GATED, int ACT = 0>
__global__ __launch_bounds__(MRG_BLOCK) void mix_bwd_reduce_k(const float* __restrict__ g, PtrPack ys, int K,
                                                              const float* __restrict__ coef, const float* __restrict__ w,
                                                              float* __restrict__ ws, int64_t rows, int D, GatedPack gp) {
  rows = valid_rows(gp.vrows, rows);
  extern __shared__ float lds[];                 // coef [K][4][D], then the block-reduction buffer
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  float* red = lds + K * 4 * D;                  // [RPB][3][WIDTH]
  for (int t = threadIdx.x; t < K * 4 * D; t += MRG_BLOCK) lds[t] = coef[t];
  __syncthreads();
  float wk[KB];
  Vec<VEC> a0[KB][KMAX], a1[KB][KMAX], a2[KB][KMAX];
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    wk[k] = k < K ? w[k] : 0.f;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) { a0[k][q] = Vec<VEC>::fill(0.f); a1[k][q] = Vec<VEC>::fill(0.f); a2[k][q] = Vec<VEC>::fill(0.f); }
  }
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        float gck = 1.f, rfv = 1.f;                        // the recomputed candidates' row multipliers and multiplicand (see mix_fwd_k)
        Vec<VEC> gsv = Vec<VEC>::fill(0.f);
        if constexpr (GATED) {
          gck = gated_rowscale(gp, r);
          rfv = gp.rf[r];
          gsv = Vec<VEC>::load(gp.s + r * D + c * VEC);
        }
        Vec<VEC> gv = Vec<VEC>::load(g + r * D + c * VEC);
        Vec<VEC> vin[KB];                                  // all loads first (see mix_fwd_k)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          vin[k] = Vec<VEC>::fill(0.f);
          if (k < K && ys.p[k]) vin[k] = Vec<VEC>::load(ys.p[k] + r * D + c * VEC);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          if (k < K) {
            Vec<VEC> v = vin[k];
            if constexpr (GATED) {                         // the recomputed candidates: gate -> gate * s * c_r, s -> s * 1 * f_r; the others * 1 * 1
              const bool isg = k == gp.k;
              const float cm = isg ? gck : (k == gp.rk ? rfv : 1.0f);
#pragma unroll
              for (int j = 0; j < VEC; ++j) v[j] = v[j] * (isg ? gsv[j] : 1.0f) * cm;
            }
            const float* cf = lds + k * 4 * D + c * VEC;
            const Vec<VEC> c0 = Vec<VEC>::load(cf), c1 = Vec<VEC>::load(cf + D), c2 = Vec<VEC>::load(cf + 2 * D), c3 = Vec<VEC>::load(cf + 3 * D);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float z = v[j] * c0[j] + c1[j];
              float xh = v[j] * c2[j] - c3[j];
              float rl, gr;
              if constexpr (ACT == 0) { rl = z > 0.f ? z : 0.f; gr = z > 0.f ? wk[k] * gv[j] : 0.f; }
              else { rl = tanhf(z); gr = wk[k] * gv[j] * (1.f - rl * rl); }
              a0[k][q][j] += gr;
              a1[k][q][j] += gr * xh;
              a2[k][q][j] += gv[j] * rl;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    if (k < K) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < KMAX; ++q) {
        int c = sl + q * LPR;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          red[(rw * 3 + 0) * WIDTH + c * VEC + j] = a0[k][q][j];
          red[(rw * 3 + 1) * WIDTH + c * VEC + j] = a1[k][q][j];
          red[(rw * 3 + 2) * WIDTH + c * VEC + j] = a2[k][q][j];
        }
      }
      __syncthreads();
      float* dst = ws + ((int64_t)blockIdx.x * K + k) * 3 * D;
      for (int t = threadIdx.x; t < 3 * D; t += MRG_BLOCK) {
        int which = t / D, c = t - which * D;
        float acc = 0.f;
#pragma unroll
        for (int q2 = 0; q2 < RPB; ++q2) acc += red[(q2 * 3 + which) * WIDTH + c];
        dst[t] = acc;
      }
    }
  }
}

// ---- finalize backward: red[k][3][D] -> c1 = sum gr / rows, c2 = sum gr*xhat / rows into coef2[k][2][D];
//      dgamma_k = sum gr*xhat, dbeta_k = sum gr, dw[k] = sum_c sum g*relu(z)
__global__ void mix_finalize_bwd_k(const float* __restrict__ red, int K, double total_rows, int D, float* __restrict__ coef2,
                                   MutPack dgamma, MutPack dbeta, float* __restrict__ dw, const int32_t* __restrict__ vrows) {
  if (vrows) total_rows = *vrows > 0 ? (double)*vrows : 1.0;
  __shared__ float part[256];
  int k = blockIdx.x;
  float accw = 0.f;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float s0 = red[(k * 3 + 0) * D + c], s1 = red[(k * 3 + 1) * D + c];
    coef2[(k * 2 + 0) * D + c] = (float)(s0 / total_rows);
    coef2[(k * 2 + 1) * D + c] = (float)(s1 / total_rows);
    if (dgamma.p[k]) dgamma.p[k][c] = s1;
    if (dbeta.p[k]) dbeta.p[k][c] = s0;
    accw += red[(k * 3 + 2) * D + c];
  }
  part[threadIdx.x] = accw;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) tot += part[i];
    dw[k] = tot;
  }
}

// ---- backward apply: gy_k = (gr - c1 - xhat*c2) * scale      (skipped where gy_k is NULL)
// The row-scaled candidate gp.rk (GATED variant; f_sparse_op_comp, y = s * f_r with f_r = sigmoid(u.s + v.s_in + c0) * t_r) has no
// gradient tensor of its own: with gy its gradient w.r.t. y, this kernel forms the row dot q_r = sum_c gy * s (the lanes of the
// row, same order as gate_bwd_k) -> gp.rdq[r], the gradient w.r.t. f_r, from which mrg_gate_row_bwd derives the candidate's
// parameter / s_in gradients; and with dz_r = q_r * h_r (h_r = t_r * gate * (1 - gate), saved by the forward) it ADDS the candidate's
// whole gradient w.r.t. s, gy * f_r + dz_r * u[c], into the gated candidate's direct term gs_out (gradients w.r.t. the same rows s).
template <int VEC, int LPR, int KMAX, bool GATED, int KB, int ACT = 0>
__global__ __launch_bounds__(MRG_BLOCK) void mix_bwd_apply_k(const float* __restrict__ g, PtrPack ys, MutPack gys, int K,
                                                             const float* __restrict__ coef, const float* __restrict__ coef2,
                                                             const float* __restrict__ w, int64_t rows, int D, RowScalePack rsp,
                                                             GatedPack gp) {
  extern __shared__ float lds[];                 // [K][6][D]: scale, shift, invstd, mean*invstd, c1, c2
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int t = threadIdx.x; t < K * 6 * D; t += MRG_BLOCK) {
    int k = t / (6 * D), rem = t - k * 6 * D;
    lds[t] = rem < 4 * D ? coef[(int64_t)k * 4 * D + rem] : coef2[(int64_t)k * 2 * D + rem - 4 * D];
  }
  __syncthreads();
  float wk[KB];
  bool need[KB];                                 // gy_k is wanted: stored, or added into a gated candidate's gs_out
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    wk[k] = k < K ? w[k] : 0.f;
    need[k] = k < K && (gys.p[k] != nullptr || (GATED && k == gp.rk));
  }
#pragma unroll
  for (int k = 0; k < KB; ++k)
    if (k < K && rsp.on[k] == 2 && rsp.add_from[k] >= 0) {
#pragma unroll
      for (int q = 0; q < KB; ++q) if (q == rsp.add_from[k]) need[q] = true;
    }
  const bool hasr = GATED && gp.rk >= 0;
  const int64_t nvalid = valid_rows(gp.vrows, rows);
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float live = r < nvalid ? 1.0f : 0.0f;             // capacity padding: every gradient row written there is zero
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      const int c = sl + q * LPR;
      const bool act = c < dv;
      float gck = 1.f, rfv = 1.f, rhv = 0.f;               // the recomputed candidates' row multipliers and s (see mix_fwd_k)
      Vec<VEC> gsv = Vec<VEC>::fill(0.f), gga = Vec<VEC>::fill(0.f), guv = Vec<VEC>::fill(0.f);
      Vec<VEC> ov[KB];                           // every wanted gy_k first: a gated candidate may add another one's
      Vec<VEC> orv = Vec<VEC>::fill(0.f);                  // gy of the row-scaled candidate
      float dq = 0.f;
      float ckv[KB];                                       // folded row multipliers (RowScalePack.full)
#pragma unroll
      for (int k = 0; k < KB; ++k) ckv[k] = 1.f;
      if (act) {
        if constexpr (GATED) {                             // (unconditional: the host hands safe pointers for what is absent)
          gck = gated_rowscale(gp, r);
          rfv = gp.rf[r];
          rhv = gp.rh[r];
          gsv = Vec<VEC>::load(gp.s + r * D + c * VEC);
          const int seg = (r >= gp.b0) + (r >= gp.b1);
          guv = Vec<VEC>::load(gp.uvc + (int64_t)seg * gp.uld + c * VEC);
        }
        Vec<VEC> gv = Vec<VEC>::load(g + r * D + c * VEC);
        Vec<VEC> vin[KB];                        // all loads first (see mix_fwd_k)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          vin[k] = Vec<VEC>::fill(0.f);
          if (k < K && need[k] && ys.p[k]) vin[k] = Vec<VEC>::load(ys.p[k] + r * D + c * VEC);
          if (k < K && rsp.on[k] && rsp.full[k]) ckv[k] = rsp.full[k][r];
        }
        if constexpr (GATED) {
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            if (k == gp.k) {
              gga = vin[k];
#pragma unroll
              for (int j = 0; j < VEC; ++j) vin[k][j] = vin[k][j] * gsv[j] * gck;
            }
            if (k == gp.rk) {
#pragma unroll
              for (int j = 0; j < VEC; ++j) vin[k][j] = vin[k][j] * rfv;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          ov[k] = Vec<VEC>::fill(0.f);
          if (k < K && need[k]) {
            const Vec<VEC> v = vin[k];
            const float* cf = lds + k * 6 * D + c * VEC;
            const Vec<VEC> c0 = Vec<VEC>::load(cf), c1 = Vec<VEC>::load(cf + D), c2 = Vec<VEC>::load(cf + 2 * D),
                           c3 = Vec<VEC>::load(cf + 3 * D), c4 = Vec<VEC>::load(cf + 4 * D), c5 = Vec<VEC>::load(cf + 5 * D);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float z = v[j] * c0[j] + c1[j];
              float xh = v[j] * c2[j] - c3[j];
              float gr;
              if constexpr (ACT == 0) gr = z > 0.f ? wk[k] * gv[j] : 0.f;
              else { const float th = tanhf(z); gr = wk[k] * gv[j] * (1.f - th * th); }
              ov[k][j] = (gr - c4[j] - xh * c5[j]) * c0[j] * live;
            }
          }
        }
        if (hasr) {
#pragma unroll
          for (int k = 0; k < KB; ++k) if (k == gp.rk) orv = ov[k];
#pragma unroll
          for (int j = 0; j < VEC; ++j) dq += orv[j] * gsv[j];
        }
      }
      float dzr = 0.f;
      if (hasr) {                                          // all lanes of the row (inactive ones carry 0): KMAX == 1 (host-checked)
        const float qr = group_sum<LPR>(dq);                // (one row per wave: DPP + scalar registers, no LDS round trips)
        dzr = qr * rhv;                                    // (used by the active lanes only, which loaded rhv)
        if (sl == 0) gp.rdq[r] = qr;
      }
      if (act) {
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          if (k < K && gys.p[k] != nullptr) {
            Vec<VEC> o = ov[k];
            if (rsp.on[k]) {                               // the consumer's first backward pass (mrg_dense_filter_dz) folded into this store
              const float ck = rsp.full[k] ? ckv[k]
                                           : (r < rsp.edge_rows[k] ? rsp.scale[k] * (rsp.rs[k] ? rsp.rs[k][r] : 1.0f) : rsp.self_scale[k]);
              if (rsp.on[k] == 2) {                        // f_dense_comp: same expressions, same order as dense_dz_k<.., 0>
                // (the recomputed candidate holds both already: the host checks rsp.s[k] == gp.s and rsp.gate[k] == ys.p[k])
                const Vec<VEC> sv = (GATED && k == gp.k) ? gsv : Vec<VEC>::load(rsp.s[k] + r * D + c * VEC);
                const Vec<VEC> ga = (GATED && k == gp.k) ? gga : Vec<VEC>::load(rsp.gate[k] + r * D + c * VEC);
                Vec<VEC> o2;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                  const float gc = o[j] * ck;
                  o2[j] = gc * ga[j];
                  o[j] = gc * sv[j] * ga[j] * (1.0f - ga[j]);
                }
                const int af = rsp.add_from[k];
                if (af >= 0) {                             // + the gradient of the candidate whose output IS this one's operand s
#pragma unroll
                  for (int qq = 0; qq < KB; ++qq)
                    if (qq == af) {
#pragma unroll
                      for (int j = 0; j < VEC; ++j) o2[j] += ov[qq][j];
                    }
                }
                if (hasr && k == gp.k) {                   // + the row-scaled candidate's gradient w.r.t. s: gy * f_r + dz_r * u
#pragma unroll
                  for (int j = 0; j < VEC; ++j) o2[j] += orv[j] * rfv + dzr * guv[j];
                }
                o2.store(rsp.gs_out[k] + r * D + c * VEC);
              } else {                                     // f_comp: dz = g * c
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = o[j] * ck;
              }
            }
            o.store(gys.p[k] + r * D + c * VEC);
          }
        }
      }
    }
  }
}


// =====================================================================================================================
// Cell zero: the MixedOp over the compose candidates WITHOUT their [rows, D] outputs
//   out = sum_k w_k * ReLU(BN_k(ent[ei[r]] (op_k) rel[ri[r]])),   op_k in {mult, sub, add}
//   reference models/cell_lp.py:53-68 (Cell_Zero: one MixedOp over PRE_OPS), :25-33 (MixedOp), models/operations_lp.py:71-98
//   (pre_*_op) and the gather that feeds them, models/model_search_lp.py:135-145.
// Every candidate is an elementwise function of two table rows that stay cache resident (11.6 MB of entities, 0.4 MB of
// relations at FB15k-237), so the statistics / combine / gradient passes RECOMPUTE y_k from the tables instead of reading
// three stored [rows, D] tensors, and the backward emits the two combined per-row gradients (w.r.t. the entity row and
// w.r.t. the relation row) for the table gradients' span sums instead of three gy_k that six span sums would read.
// Per layer (rounds 1-2: 3 gather-compose launches + the generic epilogue on stored candidates): 27 [rows, D] passes -> 6.
// Same values, same summation order per statistic as the stored form (mix_colstats_k / mix_bwd_reduce_k): the
// coefficients, the output and every gy_k are bit-identical; only the association of the table gradients differs.
struct ZeroSrc { const float* ent; const float* rel; const int32_t* ei; const int32_t* ri; int op[4]; int K; const int32_t* vrows; };

template <int VEC>
__device__ __forceinline__ Vec<VEC> zero_val(int op, const Vec<VEC>& a, const Vec<VEC>& b) {
  Vec<VEC> y;
#pragma unroll
  for (int j = 0; j < VEC; ++j) y[j] = op == MRG_COMPOSE_MULT ? a[j] * b[j] : (op == MRG_COMPOSE_SUB ? a[j] - b[j] : a[j] + b[j]);
  return y;
}

constexpr int ZK = 3;      // at most three compose candidates

template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void zero_colstats_k(ZeroSrc z, int64_t rows, int D, double* __restrict__ ws) {
  rows = valid_rows(z.vrows, rows);
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  __shared__ double red[RPB * 2 * WIDTH];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int K = z.K;
  double s1[ZK][KMAX][VEC], s2[ZK][KMAX][VEC];
#pragma unroll
  for (int k = 0; k < ZK; ++k)
#pragma unroll
    for (int q = 0; q < KMAX; ++q)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s1[k][q][j] = 0.0; s2[k][q][j] = 0.0; }
  // the same row sequence per lane as mix_colstats_k (r, r + step, ...: four rows per trip, then the remainder): the sums
  // of a column come out bit-identical with the stored form
  const int64_t step = (int64_t)gridDim.x * RPB;
  int64_t r = (int64_t)blockIdx.x * RPB + rw;
  for (; r + 3 * step < rows; r += 4 * step) {
    const float* a[4]; const float* b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = z.ent + (int64_t)z.ei[r + u * step] * D; b[u] = z.rel + (int64_t)z.ri[r + u * step] * D; }
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        Vec<VEC> va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { va[u] = Vec<VEC>::load(a[u] + c * VEC); vb[u] = Vec<VEC>::load(b[u] + c * VEC); }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < ZK; ++k)
            if (k < K) {
              const Vec<VEC> y = zero_val<VEC>(z.op[k], va[u], vb[u]);
#pragma unroll
              for (int j = 0; j < VEC; ++j) { double d = (double)y[j]; s1[k][q][j] += d; s2[k][q][j] += d * d; }
            }
      }
    }
  }
  for (; r < rows; r += step) {
    const float* a = z.ent + (int64_t)z.ei[r] * D;
    const float* b = z.rel + (int64_t)z.ri[r] * D;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        const Vec<VEC> va = Vec<VEC>::load(a + c * VEC), vb = Vec<VEC>::load(b + c * VEC);
#pragma unroll
        for (int k = 0; k < ZK; ++k)
          if (k < K) {
            const Vec<VEC> y = zero_val<VEC>(z.op[k], va, vb);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { double d = (double)y[j]; s1[k][q][j] += d; s2[k][q][j] += d * d; }
          }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ZK; ++k) {
    if (k < K) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < KMAX; ++q) {
        int c = sl + q * LPR;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          red[(rw * 2 + 0) * WIDTH + c * VEC + j] = s1[k][q][j];
          red[(rw * 2 + 1) * WIDTH + c * VEC + j] = s2[k][q][j];
        }
      }
      __syncthreads();
      double* dst = ws + ((int64_t)blockIdx.x * K + k) * 2 * D;
      for (int t = threadIdx.x; t < 2 * D; t += MRG_BLOCK) {
        int which = t / D, c = t - which * D;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < RPB; ++q) acc += red[(q * 2 + which) * WIDTH + c];
        dst[t] = acc;
      }
    }
  }
}

template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void zero_fwd_k(ZeroSrc z, const float* __restrict__ coef, const float* __restrict__ w,
                                                        float* __restrict__ out, int64_t rows, int D) {
  extern __shared__ float lds[];                 // [K][2][D] scale, shift
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int K = z.K;
  for (int t = threadIdx.x; t < K * 2 * D; t += MRG_BLOCK) {
    int k = t / (2 * D), rem = t - k * 2 * D;
    lds[t] = coef[(int64_t)k * 4 * D + rem];
  }
  __syncthreads();
  float wk[ZK];
#pragma unroll
  for (int k = 0; k < ZK; ++k) wk[k] = k < K ? w[k] : 0.f;
  const int64_t nvalid = valid_rows(z.vrows, rows);
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float* a = z.ent + (int64_t)z.ei[r] * D;
    const float* b = z.rel + (int64_t)z.ri[r] * D;
    const bool pad = r >= nvalid;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        const Vec<VEC> va = Vec<VEC>::load(a + c * VEC), vb = Vec<VEC>::load(b + c * VEC);
        Vec<VEC> acc = Vec<VEC>::fill(0.f);
#pragma unroll
        for (int k = 0; k < ZK; ++k) {
          if (k < K) {
            const Vec<VEC> v = zero_val<VEC>(z.op[k], va, vb);
            Vec<VEC> sc = Vec<VEC>::load(lds + (k * 2 + 0) * D + c * VEC), sh = Vec<VEC>::load(lds + (k * 2 + 1) * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float zz = v[j] * sc[j] + sh[j];
              acc[j] += wk[k] * (zz > 0.f ? zz : 0.f);
            }
          }
        }
        if (pad) acc = Vec<VEC>::fill(0.f);
        acc.store(out + r * D + c * VEC);
      }
    }
  }
}

template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void zero_bwd_reduce_k(const float* __restrict__ g, ZeroSrc z, const float* __restrict__ coef,
                                                               const float* __restrict__ w, float* __restrict__ ws, int64_t rows, int D) {
  rows = valid_rows(z.vrows, rows);
  extern __shared__ float lds[];                 // coef [K][4][D], then the block-reduction buffer
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int K = z.K;
  float* red = lds + K * 4 * D;                  // [RPB][3][WIDTH]
  for (int t = threadIdx.x; t < K * 4 * D; t += MRG_BLOCK) lds[t] = coef[t];
  __syncthreads();
  float wk[ZK];
  Vec<VEC> a0[ZK][KMAX], a1[ZK][KMAX], a2[ZK][KMAX];
#pragma unroll
  for (int k = 0; k < ZK; ++k) {
    wk[k] = k < K ? w[k] : 0.f;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) { a0[k][q] = Vec<VEC>::fill(0.f); a1[k][q] = Vec<VEC>::fill(0.f); a2[k][q] = Vec<VEC>::fill(0.f); }
  }
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float* a = z.ent + (int64_t)z.ei[r] * D;
    const float* b = z.rel + (int64_t)z.ri[r] * D;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        const Vec<VEC> gv = Vec<VEC>::load(g + r * D + c * VEC);
        const Vec<VEC> va = Vec<VEC>::load(a + c * VEC), vb = Vec<VEC>::load(b + c * VEC);
#pragma unroll
        for (int k = 0; k < ZK; ++k) {
          if (k < K) {
            const Vec<VEC> v = zero_val<VEC>(z.op[k], va, vb);
            const float* cf = lds + k * 4 * D + c * VEC;
            const Vec<VEC> c0 = Vec<VEC>::load(cf), c1 = Vec<VEC>::load(cf + D), c2 = Vec<VEC>::load(cf + 2 * D), c3 = Vec<VEC>::load(cf + 3 * D);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float zz = v[j] * c0[j] + c1[j];
              float xh = v[j] * c2[j] - c3[j];
              float rl = zz > 0.f ? zz : 0.f;
              float gr = zz > 0.f ? wk[k] * gv[j] : 0.f;
              a0[k][q][j] += gr;
              a1[k][q][j] += gr * xh;
              a2[k][q][j] += gv[j] * rl;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ZK; ++k) {
    if (k < K) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < KMAX; ++q) {
        int c = sl + q * LPR;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          red[(rw * 3 + 0) * WIDTH + c * VEC + j] = a0[k][q][j];
          red[(rw * 3 + 1) * WIDTH + c * VEC + j] = a1[k][q][j];
          red[(rw * 3 + 2) * WIDTH + c * VEC + j] = a2[k][q][j];
        }
      }
      __syncthreads();
      float* dst = ws + ((int64_t)blockIdx.x * K + k) * 3 * D;
      for (int t = threadIdx.x; t < 3 * D; t += MRG_BLOCK) {
        int which = t / D, c = t - which * D;
        float acc = 0.f;
#pragma unroll
        for (int q2 = 0; q2 < RPB; ++q2) acc += red[(q2 * 3 + which) * WIDTH + c];
        dst[t] = acc;
      }
    }
  }
}

// gy_k = (gr - c1 - xhat * c2) * scale_k as in mix_bwd_apply_k, never stored:
//   g_ent_rows[r] = sum_k gy_k * d y_k / d ent   (mult: rel row, sub / add: 1)
//   g_rel_rows[r] = sum_k gy_k * d y_k / d rel   (mult: ent row, sub: -1, add: 1)          k = 0..K-1 order
template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void zero_bwd_apply_k(const float* __restrict__ g, ZeroSrc z, const float* __restrict__ coef,
                                                              const float* __restrict__ coef2, const float* __restrict__ w,
                                                              float* __restrict__ ge_rows, float* __restrict__ gr_rows, int64_t rows, int D) {
  extern __shared__ float lds[];                 // [K][6][D]: scale, shift, invstd, mean*invstd, c1, c2
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int K = z.K;
  for (int t = threadIdx.x; t < K * 6 * D; t += MRG_BLOCK) {
    int k = t / (6 * D), rem = t - k * 6 * D;
    lds[t] = rem < 4 * D ? coef[(int64_t)k * 4 * D + rem] : coef2[(int64_t)k * 2 * D + rem - 4 * D];
  }
  __syncthreads();
  float wk[ZK];
#pragma unroll
  for (int k = 0; k < ZK; ++k) wk[k] = k < K ? w[k] : 0.f;
  const int64_t nvalid = valid_rows(z.vrows, rows);
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float* a = z.ent + (int64_t)z.ei[r] * D;
    const float* b = z.rel + (int64_t)z.ri[r] * D;
    const float live = r < nvalid ? 1.0f : 0.0f;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      int c = sl + q * LPR;
      if (c < dv) {
        const Vec<VEC> gv = Vec<VEC>::load(g + r * D + c * VEC);
        const Vec<VEC> va = Vec<VEC>::load(a + c * VEC), vb = Vec<VEC>::load(b + c * VEC);
        Vec<VEC> ge = Vec<VEC>::fill(0.f), gr2 = Vec<VEC>::fill(0.f);
#pragma unroll
        for (int k = 0; k < ZK; ++k) {
          if (k < K) {
            const int op = z.op[k];
            const Vec<VEC> v = zero_val<VEC>(op, va, vb);
            const float* cf = lds + k * 6 * D + c * VEC;
            const Vec<VEC> c0 = Vec<VEC>::load(cf), c1 = Vec<VEC>::load(cf + D), c2 = Vec<VEC>::load(cf + 2 * D),
                           c3 = Vec<VEC>::load(cf + 3 * D), c4 = Vec<VEC>::load(cf + 4 * D), c5 = Vec<VEC>::load(cf + 5 * D);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              float zz = v[j] * c0[j] + c1[j];
              float xh = v[j] * c2[j] - c3[j];
              float gr = zz > 0.f ? wk[k] * gv[j] : 0.f;
              const float o = (gr - c4[j] - xh * c5[j]) * c0[j] * live;
              ge[j] += op == MRG_COMPOSE_MULT ? o * vb[j] : o;
              gr2[j] += op == MRG_COMPOSE_MULT ? o * va[j] : (op == MRG_COMPOSE_SUB ? -o : o);
            }
          }
        }
        if (ge_rows) ge.store(ge_rows + r * D + c * VEC);
        if (gr_rows) gr2.store(gr_rows + r * D + c * VEC);
      }
    }
  }
}

static int lab_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// ---- "static step graphs" (round 5): row counts that live in device memory ------------------------------------------------------
// The reference's search loop draws a NEW step graph every step (search/mr_lp_search.py:187-214) whose node count depends on the
// draw; to replay that step from ONE captured HIP graph every tensor must keep its shape, so the step graph is padded to a host-known
// node CAPACITY and the true counts stay in device memory.  A launch of the MixedOp-epilogue / cell-zero kernels whose row count
// equals a registered capacity treats rows at and beyond the device count as padding: left out of the BatchNorm statistics and of
// every gradient reduction, counted out of `total_rows`, and WRITTEN AS ZEROS by the combine and gradient passes -- which keeps the
// padding rows of every state zero, and a zero row yields a zero candidate in every operator of the search space, so no other kernel
// needs to know.  Two slots: the [M, D] edge + node rows and the [N, D] node rows of a step graph.
static struct { int64_t cap[2]; const int32_t* count[2]; } g_dyn = {{-1, -1}, {nullptr, nullptr}};
static const int32_t* dyn_rows_for(int64_t rows) {
  for (int i = 0; i < 2; ++i)
    if (g_dyn.count[i] != nullptr && g_dyn.cap[i] == rows) return g_dyn.count[i];
  return nullptr;
}
// statistics: the flat kernels' bound (512 blocks measured best: profiles/r3_stream_grid.txt)
// FEW ROWS (round 5: a sampled step graph, a rank's node chunk): a wave walks its rows one dependent memory round trip at a time, so
// with `trips` rows per wave a launch over 900 rows is ~30 blocks of 8 trips = a 15 us latency chain on an idle chip.  Until there
// is a block per CU, a block gets ONE row per wave: rows / (rows per trip) blocks, at most 256.
static int64_t row_blocks(int64_t rows, int lpr, int trips) {
  const int64_t per_trip = MRG_BLOCK / lpr;
  int64_t b = (rows + per_trip * trips - 1) / (per_trip * trips);
  if (b < 256) {
    const int64_t b1 = (rows + per_trip - 1) / per_trip;
    b = b1 < 256 ? b1 : 256;
  }
  return b < 1 ? 1 : b;
}
static int mix_grid(int64_t rows, int lpr) {
  int64_t g = row_blocks(rows, lpr, 8);
  if (g > stream_blocks()) g = stream_blocks();
  return (int)(g > 1024 ? 1024 : g);               // partial buffers are sized for 1024 blocks
}
// backward reduction / combine: since the row addressing became scalar these kernels hold 4 / 8 workgroups per CU and gain from
// more blocks than the 512 of the flat kernels (lab: 3.0 -> 2.4 ms and 2.53 -> 2.42 ms per step at 1024)
static int mix_reduce_grid(int64_t rows, int lpr) {
  static const int cap = lab_env_int("MRG_MIX_REDUCE_BLOCKS", 1024);
  int64_t b = row_blocks(rows, lpr, 8);
  const int c = cap > 1024 ? 1024 : (cap < 1 ? 1 : cap);              // partial buffers are sized for 1024 blocks
  return (int)(b < 1 ? 1 : (b > c ? c : b));
}
static int mix_apply_grid(int64_t rows, int lpr) {
  static const int cap = lab_env_int("MRG_MIX_APPLY_BLOCKS", MRG_MAX_GRID);
  int64_t b = row_blocks(rows, lpr, 4);
  const int c = cap > 8192 ? 8192 : (cap < 1 ? 1 : cap);
  return (int)(b < 1 ? 1 : (b > c ? c : b));
}
static int mix_fwd_grid(int64_t rows, int lpr) {
  static const int cap = lab_env_int("MRG_MIX_FWD_BLOCKS", 1024);
  int64_t b = row_blocks(rows, lpr, 4);
  const int c = cap > MRG_MAX_GRID ? MRG_MAX_GRID : (cap < 1 ? 1 : cap);
  return (int)(b < 1 ? 1 : (b > c ? c : b));
}

static bool pack_ok(const void* const* host, int K) { return host != nullptr && K >= 1 && K <= MRG_MIX_MAXK; }

// host descriptor (include/mrgnas.h: mrg_gated_branch) -> kernel argument; *al: every pointer it adds is 16-byte aligned.
// Absent candidates keep their index < 0 and get SAFE pointers (valid [rows] / [rows, D] memory): the kernels load their row
// factors unconditionally and discard them by a select.
static const int32_t* dyn_rows_for(int64_t rows);
static int gated_pack(const mrg_gated_branch* gb, const float* const* y_host, int K, GatedPack* gp, bool* al, bool apply = false, int64_t rows = -1) {
  *gp = GatedPack{};
  gp->k = -1; gp->pair_k = -1; gp->rk = -1;
  gp->vrows = rows >= 0 ? dyn_rows_for(rows) : nullptr;
  if (gb && gb->act != 0 && gb->act != 1) return MRG_E_ENUM;
  gp->act = gb ? gb->act : 0;
  if (gp->act == 1 && (K > 5 || gb->k >= 0 || gb->row_k >= 0)) return MRG_E_SHAPE;    // the tanh instances: un-gated, at most five candidates
  if (!gb || (gb->k < 0 && gb->row_k < 0)) return MRG_OK;
  if (gb->k >= K || gb->row_k >= K || (gb->k >= 0 && gb->k == gb->row_k)) return MRG_E_SHAPE;
  if (!gb->s) return MRG_E_NULLPTR;
  gp->s = gb->s;
  if (gb->k >= 0) {
    if (!gb->rowscale || !y_host[gb->k]) return MRG_E_NULLPTR;       // the gate stands where the candidate's output would
    gp->k = gb->k; gp->c = gb->rowscale;
  }
  if (gb->row_k >= 0) {
    if (!gb->row_f) return MRG_E_NULLPTR;
    if (y_host[gb->row_k] != gb->s) return MRG_E_SHAPE;               // the row-scaled candidate's slot holds s itself
    gp->rk = gb->row_k; gp->rf = gb->row_f;
    if (apply) {
      if (!gb->row_h || !gb->row_uvc || !gb->row_dq) return MRG_E_NULLPTR;
      if (gb->b0 < 0 || gb->b1 < gb->b0 || gb->row_ld <= 0) return MRG_E_SHAPE;
      gp->rh = gb->row_h; gp->uvc = gb->row_uvc; gp->uld = gb->row_ld; gp->b0 = gb->b0; gp->b1 = gb->b1; gp->rdq = gb->row_dq;
      *al = *al && aligned16(gb->row_uvc) && gb->row_ld % 4 == 0;
    }
  }
  if (!gp->c) gp->c = gp->rf;
  if (!gp->rf) gp->rf = gp->c;
  if (!gp->rh) { gp->rh = gp->rf; gp->uvc = gp->s; gp->uld = 0; gp->b0 = gp->b1 = 0; }   // loaded, never used
  for (int q = 0; q < K; ++q)
    if (q != gb->k && q != gb->row_k && y_host[q] == gb->s) { gp->pair_k = q; break; }
  *al = *al && aligned16(gb->s);
  return MRG_OK;
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_set_dynamic_rows(int64_t cap_m, const int32_t* count_m, int64_t cap_n, const int32_t* count_n) {
  if ((count_m && cap_m < 0) || (count_n && cap_n < 0) || (count_m && count_n && cap_m == cap_n)) return MRG_E_SHAPE;
  g_dyn.cap[0] = cap_m; g_dyn.count[0] = count_m;
  g_dyn.cap[1] = cap_n; g_dyn.count[1] = count_n;
  return MRG_OK;
}

extern "C" int64_t mrg_mix_workspace_bytes(int K, int D) {
  if (K < 1 || K > MRG_MIX_MAXK || D <= 0) return 0;
  return (int64_t)1024 * K * 3 * D * sizeof(double);
}

// sums [K][2][D] float64
static int mix_colstats_blocks(const float* const* y_host, int K, int64_t rows, int D, void* ws, hipStream_t st, int* grid_out,
                               const mrg_gated_branch* gated);

// mrg_mix_colstats + mrg_mix_finalize_fwd in two launches instead of three (statistics kernel, then reduction and finalize fused):
// for callers without a collective between the two (the single-GPU step).  Same results bit for bit.
extern "C" int mrg_mix_stats_coef(const float* const* y_host, const float* const* gamma_host, const float* const* beta_host,
                                  float* const* rmean_host, float* const* rvar_host, int K, int64_t rows, double total_rows, int D, float eps,
                                  float momentum, float* coef, void* ws, const mrg_gated_branch* gated, void* stream) {
  if (K < 1 || K > MRG_MIX_MAXK || D <= 0 || total_rows < 0) return MRG_E_SHAPE;
  if (!coef || !gamma_host || !beta_host) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
  const int rc = mix_colstats_blocks(y_host, K, rows, D, ws, st, &grid, gated);
  if (rc != MRG_OK) return rc;
  PtrPack ga{}, be{};
  MutPack rm{}, rv{};
  for (int k = 0; k < K; ++k) {
    ga.p[k] = gamma_host[k]; be.p[k] = beta_host[k];
    rm.p[k] = rmean_host ? rmean_host[k] : nullptr;
    rv.p[k] = rvar_host ? rvar_host[k] : nullptr;
    if ((rm.p[k] == nullptr) != (rv.p[k] == nullptr)) return MRG_E_NULLPTR;
  }
  hipLaunchKernelGGL(mix_reduce_finalize_fwd_k, dim3((D + 63) / 64, K), dim3(1024), 0, st, (const double*)ws, grid, ga, be, rm, rv, K,
                     total_rows > 0 ? total_rows : 1.0, D, eps, momentum, coef, total_rows == (double)rows ? dyn_rows_for(rows) : nullptr);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_mix_colstats(const float* const* y_host, int K, int64_t rows, int D, double* sums, void* ws,
                                const mrg_gated_branch* gated, void* stream) {
  if (!sums) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
  const int rc = mix_colstats_blocks(y_host, K, rows, D, ws, st, &grid, gated);
  if (rc != MRG_OK) return rc;
  int len = K * 2 * D;
  launch_ordered_reduce<double>((const double*)ws, sums, 0, grid, len, len, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// the statistics kernel: per-block partial sums [grid][K][2][D] in ws
static int mix_colstats_blocks(const float* const* y_host, int K, int64_t rows, int D, void* ws, hipStream_t st, int* grid_out,
                               const mrg_gated_branch* gated) {
  if (!pack_ok((const void* const*)y_host, K)) return MRG_E_SHAPE;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (!ws) return MRG_E_WORKSPACE;
  PtrPack ys{};
  bool al = true;
  for (int k = 0; k < K; ++k) { ys.p[k] = y_host[k]; al = al && aligned16(y_host[k]); }
  GatedPack gp;
  const int grc = gated_pack(gated, y_host, K, &gp, &al, false, rows);
  if (grc != MRG_OK) return grc;
  RowGeom g = row_geom(D, al);
  if (!g.ok) return MRG_E_SHAPE;
  int grid = 1;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    grid = mix_grid(rows, L);                                                                             \
    if (gp.k >= 0 || gp.rk >= 0) hipLaunchKernelGGL((mix_colstats_k<V, L, KM, true>), dim3(grid), dim3(MRG_BLOCK), 0, st, ys, K, rows, D, (double*)ws, gp); \
    else hipLaunchKernelGGL((mix_colstats_k<V, L, KM, false>), dim3(grid), dim3(MRG_BLOCK), 0, st, ys, K, rows, D, (double*)ws, gp); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  *grid_out = grid;
  return MRG_OK;
}

// coef [K][4][D]; gamma/beta/running_mean/running_var: host arrays of K device pointers (entries may be NULL)
extern "C" int mrg_mix_finalize_fwd(const double* sums, const float* const* gamma_host, const float* const* beta_host,
                                    float* const* rmean_host, float* const* rvar_host, int K, double total_rows, int D,
                                    float eps, float momentum, float* coef, void* stream) {
  if (K < 1 || K > MRG_MIX_MAXK || D <= 0 || total_rows < 0) return MRG_E_SHAPE;
  if (!sums || !coef || !gamma_host || !beta_host) return MRG_E_NULLPTR;
  PtrPack ga{}, be{};
  MutPack rm{}, rv{};
  for (int k = 0; k < K; ++k) {
    ga.p[k] = gamma_host[k]; be.p[k] = beta_host[k];
    rm.p[k] = rmean_host ? rmean_host[k] : nullptr;
    rv.p[k] = rvar_host ? rvar_host[k] : nullptr;
    if ((rm.p[k] == nullptr) != (rv.p[k] == nullptr)) return MRG_E_NULLPTR;
  }
  hipLaunchKernelGGL(mix_finalize_fwd_k, dim3((D + 127) / 128, K), dim3(128), 0, (hipStream_t)stream, sums, ga, be, rm, rv, K,
                     total_rows > 0 ? total_rows : 1.0, D, eps, momentum, coef, (const int32_t*)nullptr);   // (sharded path: host-known totals)
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_mix_fwd(const float* const* y_host, int K, const float* coef, const float* w, const float* addend, float* out,
                           int64_t rows, int D, const mrg_gated_branch* gated, void* stream) {
  if (!pack_ok((const void* const*)y_host, K) || rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!coef || !w || !out) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  PtrPack ys{};
  bool al = aligned16(out) && aligned16(addend);
  for (int k = 0; k < K; ++k) { ys.p[k] = y_host[k]; al = al && aligned16(y_host[k]); }
  GatedPack gp;
  const int grc = gated_pack(gated, y_host, K, &gp, &al, false, rows);
  if (grc != MRG_OK) return grc;
  RowGeom g = row_geom(D, al);
  if (!g.ok) return MRG_E_SHAPE;
  size_t lds = (size_t)K * 2 * D * sizeof(float);
  if (lds > 64 * 1024) return MRG_E_SHAPE;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    const dim3 grid_(mix_fwd_grid(rows, L));                                                               \
    if (gp.act == 1) {                                                                                    \
      hipLaunchKernelGGL((mix_fwd_k<V, L, KM, false, 5, 1>), grid_, dim3(MRG_BLOCK), lds, st, ys, K, coef, w, out, rows, D, addend, gp); \
    } else if (gp.k >= 0 || gp.rk >= 0) {                                                                        \
      if (K <= 5) hipLaunchKernelGGL((mix_fwd_k<V, L, KM, true, 5>), grid_, dim3(MRG_BLOCK), lds, st, ys, K, coef, w, out, rows, D, addend, gp); \
      else hipLaunchKernelGGL((mix_fwd_k<V, L, KM, true, MRG_MIX_MAXK>), grid_, dim3(MRG_BLOCK), lds, st, ys, K, coef, w, out, rows, D, addend, gp); \
    } else {                                                                                              \
      if (K <= 5) hipLaunchKernelGGL((mix_fwd_k<V, L, KM, false, 5>), grid_, dim3(MRG_BLOCK), lds, st, ys, K, coef, w, out, rows, D, addend, gp); \
      else hipLaunchKernelGGL((mix_fwd_k<V, L, KM, false, MRG_MIX_MAXK>), grid_, dim3(MRG_BLOCK), lds, st, ys, K, coef, w, out, rows, D, addend, gp); \
    }                                                                                                     \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// red [K][3][D] float32
extern "C" int mrg_mix_bwd_reduce(const float* g, const float* const* y_host, int K, const float* coef, const float* w, float* red,
                                  void* ws, int64_t rows, int D, const mrg_gated_branch* gated, void* stream) {
  if (!pack_ok((const void* const*)y_host, K) || rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (!coef || !w || !red || (rows > 0 && !g)) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  PtrPack ys{};
  bool al = aligned16(g);
  for (int k = 0; k < K; ++k) { ys.p[k] = y_host[k]; al = al && aligned16(y_host[k]); }
  GatedPack gp;
  const int grc = gated_pack(gated, y_host, K, &gp, &al, false, rows);
  if (grc != MRG_OK) return grc;
  RowGeom gm = row_geom(D, al);
  if (!gm.ok) return MRG_E_SHAPE;
  int grid = 1;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    grid = mix_reduce_grid(rows, L);                                                                             \
    size_t lds = ((size_t)K * 4 * D + (size_t)(MRG_BLOCK / L) * 3 * (L * KM * V)) * sizeof(float);        \
    if (lds > 64 * 1024) return MRG_E_SHAPE;                                                              \
    if (gp.act == 1) hipLaunchKernelGGL((mix_bwd_reduce_k<V, L, KM, 5, false, 1>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, ys, K, coef, w, (float*)ws, rows, D, gp); \
    else if ((gp.k >= 0 || gp.rk >= 0) && K <= 5) hipLaunchKernelGGL((mix_bwd_reduce_k<V, L, KM, 5, true>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, ys, K, coef, w, (float*)ws, rows, D, gp); \
    else if (gp.k >= 0 || gp.rk >= 0) hipLaunchKernelGGL((mix_bwd_reduce_k<V, L, KM, MRG_MIX_MAXK, true>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, ys, K, coef, w, (float*)ws, rows, D, gp); \
    else if (K <= 4) hipLaunchKernelGGL((mix_bwd_reduce_k<V, L, KM, 4, false>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, ys, K, coef, w, (float*)ws, rows, D, gp); \
    else hipLaunchKernelGGL((mix_bwd_reduce_k<V, L, KM, MRG_MIX_MAXK, false>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, ys, K, coef, w, (float*)ws, rows, D, gp); \
  } while (0)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  int len = K * 3 * D;
  launch_ordered_reduce<float>((const float*)ws, red, 0, grid, len, len, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// coef2 [K][2][D]; dgamma/dbeta: host arrays of K device pointers (NULL entries skipped); dw [K]
extern "C" int mrg_mix_finalize_bwd(const float* red, int K, double total_rows, int D, float* coef2, float* const* dgamma_host,
                                    float* const* dbeta_host, float* dw, void* stream) {
  if (K < 1 || K > MRG_MIX_MAXK || D <= 0) return MRG_E_SHAPE;
  if (!red || !coef2 || !dw) return MRG_E_NULLPTR;
  MutPack dg{}, db{};
  for (int k = 0; k < K; ++k) {
    dg.p[k] = dgamma_host ? dgamma_host[k] : nullptr;
    db.p[k] = dbeta_host ? dbeta_host[k] : nullptr;
  }
  // (total_rows is the launch's own row count unless the rows are sharded over ranks: a registered capacity then names the device count)
  const int32_t* vr = (total_rows >= 0 && total_rows == (double)(int64_t)total_rows) ? dyn_rows_for((int64_t)total_rows) : nullptr;
  hipLaunchKernelGGL(mix_finalize_bwd_k, dim3(K), dim3(256), 0, (hipStream_t)stream, red, K, total_rows > 0 ? total_rows : 1.0, D,
                     coef2, dg, db, dw, vr);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_mix_bwd_apply(const float* g, const float* const* y_host, float* const* gy_host, int K, const float* coef,
                                 const float* coef2, const float* w, const float* const* rs_host, const float* rs_scale_host,
                                 const float* rs_self_host, const int64_t* rs_edge_rows_host, const int* rs_on_host,
                                 const float* const* rs_full_host,
                                 const float* const* fold_s_host, const float* const* fold_gate_host, float* const* fold_gs_host,
                                 const int* fold_add_from_host, int64_t rows, int D, const mrg_gated_branch* gated, void* stream) {
  if (!pack_ok((const void* const*)y_host, K) || !gy_host || rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!g || !coef || !coef2 || !w) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  PtrPack ys{};
  MutPack gys{};
  bool al = aligned16(g);
  bool any = false;
  for (int k = 0; k < K; ++k) {
    ys.p[k] = y_host[k]; gys.p[k] = gy_host[k];
    al = al && aligned16(y_host[k]) && aligned16(gy_host[k]);
    any = any || gy_host[k] != nullptr;
  }
  if (!any) return MRG_OK;
  RowScalePack rsp{};
  for (int k = 0; k < MRG_MIX_MAXK; ++k) rsp.add_from[k] = -1;
  if (rs_on_host) {
    if (!rs_scale_host || !rs_self_host || !rs_edge_rows_host) return MRG_E_NULLPTR;
    for (int k = 0; k < K; ++k) {
      rsp.on[k] = rs_on_host[k];
      rsp.rs[k] = rs_host ? rs_host[k] : nullptr;
      rsp.full[k] = (rs_full_host && rsp.on[k]) ? rs_full_host[k] : nullptr;
      rsp.scale[k] = rs_scale_host[k]; rsp.self_scale[k] = rs_self_host[k]; rsp.edge_rows[k] = rs_edge_rows_host[k];
      if (rsp.on[k] == 2) {
        if (!fold_s_host || !fold_gate_host || !fold_gs_host || !fold_s_host[k] || !fold_gate_host[k] || !fold_gs_host[k]) return MRG_E_NULLPTR;
        rsp.s[k] = fold_s_host[k]; rsp.gate[k] = fold_gate_host[k]; rsp.gs_out[k] = fold_gs_host[k];
        if (fold_add_from_host && fold_add_from_host[k] >= 0) {
          const int q = fold_add_from_host[k];
          if (q >= K || q == k || !y_host[q] || y_host[q] != rsp.s[k]) return MRG_E_SHAPE;      // only a candidate whose OUTPUT is this one's operand s
          rsp.add_from[k] = q;
          any = true;
        }
        al = al && aligned16(rsp.s[k]) && aligned16(rsp.gate[k]) && aligned16(rsp.gs_out[k]);
      } else if (rsp.on[k] != 0 && rsp.on[k] != 1) {
        return MRG_E_ENUM;
      }
    }
  }
  GatedPack gp;
  const int grc = gated_pack(gated, y_host, K, &gp, &al, true, rows);
  if (grc != MRG_OK) return grc;
  // the recomputed candidate's folded gradient store reads s and the gate it already holds: they must be the same tensors
  if (gp.k >= 0 && rsp.on[gp.k] == 2 && (rsp.s[gp.k] != gp.s || rsp.gate[gp.k] != y_host[gp.k])) return MRG_E_SHAPE;
  // the row-scaled candidate has no gradient tensor: its gradient w.r.t. s goes into the gated candidate's direct term
  if (gp.rk >= 0 && (gp.k < 0 || rsp.on[gp.k] != 2 || gy_host[gp.rk] != nullptr)) return MRG_E_SHAPE;
  RowGeom gm = row_geom(D, al);
  if (!gm.ok) return MRG_E_SHAPE;
  if (gp.rk >= 0 && gm.kmax != 1) return MRG_E_SHAPE;               // the row dot is one group_sum over the row's lanes
  size_t lds = (size_t)K * 6 * D * sizeof(float);
  if (lds > 64 * 1024) return MRG_E_SHAPE;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    const dim3 grid_(mix_apply_grid(rows, L));                                                             \
    if (gp.act == 1) {                                                                                    \
      hipLaunchKernelGGL((mix_bwd_apply_k<V, L, KM, false, 5, 1>), grid_, dim3(MRG_BLOCK), lds, st, g, ys, gys, K, coef, coef2, w, rows, D, rsp, gp); \
    } else if (gp.k >= 0 || gp.rk >= 0) {                                                                        \
      if (K <= 5) hipLaunchKernelGGL((mix_bwd_apply_k<V, L, KM, true, 5>), grid_, dim3(MRG_BLOCK), lds, st, g, ys, gys, K, coef, coef2, w, rows, D, rsp, gp); \
      else hipLaunchKernelGGL((mix_bwd_apply_k<V, L, KM, true, MRG_MIX_MAXK>), grid_, dim3(MRG_BLOCK), lds, st, g, ys, gys, K, coef, coef2, w, rows, D, rsp, gp); \
    } else {                                                                                              \
      if (K <= 5) hipLaunchKernelGGL((mix_bwd_apply_k<V, L, KM, false, 5>), grid_, dim3(MRG_BLOCK), lds, st, g, ys, gys, K, coef, coef2, w, rows, D, rsp, gp); \
      else hipLaunchKernelGGL((mix_bwd_apply_k<V, L, KM, false, MRG_MIX_MAXK>), grid_, dim3(MRG_BLOCK), lds, st, g, ys, gys, K, coef, coef2, w, rows, D, rsp, gp); \
    }                                                                                                     \
  } while (0)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}


// ---- cell zero: the MixedOp over compose candidates recomputed from the tables -------------------------------------------------
static int zero_src(ZeroSrc* z, const float* ent, const float* rel, const int32_t* ei, const int32_t* ri, const int* ops_host, int K) {
  if (K < 1 || K > ZK || !ops_host) return MRG_E_SHAPE;
  if (!ent || !rel || !ei || !ri) return MRG_E_NULLPTR;
  z->ent = ent; z->rel = rel; z->ei = ei; z->ri = ri; z->K = K;
  for (int k = 0; k < K; ++k) {
    if (ops_host[k] < 0 || ops_host[k] > 2) return MRG_E_ENUM;
    z->op[k] = ops_host[k];
  }
  return MRG_OK;
}

// The recomputing reductions gather two table rows per row through an index (a dependent load chain per row): they are
// latency-bound per wave and want MORE blocks than the flat statistics kernels (512): up to 2048, which is what the
// workspace (mrg_zero_workspace_bytes) is sized for.  Measured at FB15k-237, D = 200: statistics 307 -> see profiles/r3_cell_zero.txt.
static int zero_grid(int64_t rows, int lpr) {
  int64_t b = row_blocks(rows, lpr, 8);
  return (int)(b > 2048 ? 2048 : b);
}

extern "C" int64_t mrg_zero_workspace_bytes(int D) {
  if (D <= 0) return 0;
  return (int64_t)2048 * ZK * 3 * D * sizeof(double);
}

static int zero_colstats_blocks(const ZeroSrc& z, int64_t rows, int D, void* ws, hipStream_t st, int* grid_out) {
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (!ws) return MRG_E_WORKSPACE;
  RowGeom g = row_geom(D, aligned16(z.ent) && aligned16(z.rel));
  if (!g.ok) return MRG_E_SHAPE;
  int grid = 1;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    grid = zero_grid(rows, L);                                                                             \
    hipLaunchKernelGGL((zero_colstats_k<V, L, KM>), dim3(grid), dim3(MRG_BLOCK), 0, st, z, rows, D, (double*)ws); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  *grid_out = grid;
  return MRG_OK;
}

extern "C" int mrg_zero_colstats(const float* ent, const float* rel, const int32_t* ent_idx, const int32_t* rel_idx, const int* ops_host, int K,
                                 int64_t rows, int D, double* sums, void* ws, void* stream) {
  ZeroSrc z{};
  z.vrows = dyn_rows_for(rows);
  int rc = zero_src(&z, ent, rel, ent_idx, rel_idx, ops_host, K);
  if (rc != MRG_OK) return rc;
  if (!sums) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
  rc = zero_colstats_blocks(z, rows, D, ws, st, &grid);
  if (rc != MRG_OK) return rc;
  const int len = K * 2 * D;
  launch_ordered_reduce<double>((const double*)ws, sums, 0, grid, len, len, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_zero_stats_coef(const float* ent, const float* rel, const int32_t* ent_idx, const int32_t* rel_idx, const int* ops_host, int K,
                                   const float* const* gamma_host, const float* const* beta_host, float* const* rmean_host,
                                   float* const* rvar_host, int64_t rows, double total_rows, int D, float eps, float momentum, float* coef,
                                   void* ws, void* stream) {
  ZeroSrc z{};
  z.vrows = dyn_rows_for(rows);
  int rc = zero_src(&z, ent, rel, ent_idx, rel_idx, ops_host, K);
  if (rc != MRG_OK) return rc;
  if (D <= 0 || total_rows < 0) return MRG_E_SHAPE;
  if (!coef || !gamma_host || !beta_host) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
  rc = zero_colstats_blocks(z, rows, D, ws, st, &grid);
  if (rc != MRG_OK) return rc;
  PtrPack ga{}, be{};
  MutPack rm{}, rv{};
  for (int k = 0; k < K; ++k) {
    ga.p[k] = gamma_host[k]; be.p[k] = beta_host[k];
    rm.p[k] = rmean_host ? rmean_host[k] : nullptr;
    rv.p[k] = rvar_host ? rvar_host[k] : nullptr;
    if ((rm.p[k] == nullptr) != (rv.p[k] == nullptr)) return MRG_E_NULLPTR;
  }
  hipLaunchKernelGGL(mix_reduce_finalize_fwd_k, dim3((D + 63) / 64, K), dim3(1024), 0, st, (const double*)ws, grid, ga, be, rm, rv, K,
                     total_rows > 0 ? total_rows : 1.0, D, eps, momentum, coef, total_rows == (double)rows ? dyn_rows_for(rows) : nullptr);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_zero_fwd(const float* ent, const float* rel, const int32_t* ent_idx, const int32_t* rel_idx, const int* ops_host, int K,
                            const float* coef, const float* w, float* out, int64_t rows, int D, void* stream) {
  ZeroSrc z{};
  z.vrows = dyn_rows_for(rows);
  const int rc = zero_src(&z, ent, rel, ent_idx, rel_idx, ops_host, K);
  if (rc != MRG_OK) return rc;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!coef || !w || !out) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  RowGeom g = row_geom(D, aligned16(ent) && aligned16(rel) && aligned16(out));
  if (!g.ok) return MRG_E_SHAPE;
  const size_t lds = (size_t)K * 2 * D * sizeof(float);
  if (lds > 64 * 1024) return MRG_E_SHAPE;
#define CALL(V, L, KM)                                                                                    \
  hipLaunchKernelGGL((zero_fwd_k<V, L, KM>), dim3(grid_for(rows, (MRG_BLOCK / L) * 4)), dim3(MRG_BLOCK), lds, st, z, coef, w, out, rows, D)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_zero_bwd_reduce(const float* g, const float* ent, const float* rel, const int32_t* ent_idx, const int32_t* rel_idx,
                                   const int* ops_host, int K, const float* coef, const float* w, float* red, void* ws, int64_t rows, int D,
                                   void* stream) {
  ZeroSrc z{};
  z.vrows = dyn_rows_for(rows);
  const int rc = zero_src(&z, ent, rel, ent_idx, rel_idx, ops_host, K);
  if (rc != MRG_OK) return rc;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (!coef || !w || !red || (rows > 0 && !g)) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  RowGeom gm = row_geom(D, aligned16(g) && aligned16(ent) && aligned16(rel));
  if (!gm.ok) return MRG_E_SHAPE;
  int grid = 1;
#define CALL(V, L, KM)                                                                                    \
  do {                                                                                                    \
    grid = zero_grid(rows, L);                                                                             \
    size_t lds = ((size_t)K * 4 * D + (size_t)(MRG_BLOCK / L) * 3 * (L * KM * V)) * sizeof(float);        \
    if (lds > 64 * 1024) return MRG_E_SHAPE;                                                              \
    hipLaunchKernelGGL((zero_bwd_reduce_k<V, L, KM>), dim3(grid), dim3(MRG_BLOCK), lds, st, g, z, coef, w, (float*)ws, rows, D); \
  } while (0)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  const int len = K * 3 * D;
  launch_ordered_reduce<float>((const float*)ws, red, 0, grid, len, len, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_zero_bwd_apply(const float* g, const float* ent, const float* rel, const int32_t* ent_idx, const int32_t* rel_idx,
                                  const int* ops_host, int K, const float* coef, const float* coef2, const float* w, float* g_ent_rows,
                                  float* g_rel_rows, int64_t rows, int D, void* stream) {
  ZeroSrc z{};
  z.vrows = dyn_rows_for(rows);
  const int rc = zero_src(&z, ent, rel, ent_idx, rel_idx, ops_host, K);
  if (rc != MRG_OK) return rc;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0 || (!g_ent_rows && !g_rel_rows)) return MRG_OK;
  if (!g || !coef || !coef2 || !w) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  RowGeom gm = row_geom(D, aligned16(g) && aligned16(ent) && aligned16(rel) && aligned16(g_ent_rows) && aligned16(g_rel_rows));
  if (!gm.ok) return MRG_E_SHAPE;
  const size_t lds = (size_t)K * 6 * D * sizeof(float);
  if (lds > 64 * 1024) return MRG_E_SHAPE;
#define CALL(V, L, KM)                                                                                    \
  hipLaunchKernelGGL((zero_bwd_apply_k<V, L, KM>), dim3(grid_for(rows, (MRG_BLOCK / L) * 4)), dim3(MRG_BLOCK), lds, st, g, z, coef, coef2, w, g_ent_rows, g_rel_rows, rows, D)
  MRG_DISPATCH_GEOM(gm, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
