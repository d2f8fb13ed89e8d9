// a2 / a3: collapsed scalar gates -- one streaming pass per direction segment.
//   f_sparse_op_comp  reference models/operations_lp.py:304-343
//   f_sparse_op_last  reference models/operations_lp.py:405-416
// a_x(W_x[s ; s_in] + b_x) == u_x.s + v_x.s_in + c_x  with [u;v] = W_x^T a_x, c = a_x.b_x,
// so the reference's three [rows,2D]x[2D,D] GEMMs become a 2D-long dot product per row.
// Algorithmic bytes: fwd 12*D per row (+4 per edge row for norm); bwd 20*D per row.
#include "common.hpp"

namespace mrg {

struct SegPlan {
  int64_t lo[3], hi[3];   // row range of each segment
  int blk[4];             // blocks [blk[s], blk[s+1]) work on segment s
};

// rows in flight per lane group and trip of gate_fwd_k / gate_bwd_k when a wave owns a row (D > 128): 2 (default, round 5), 1 = rounds 1-4.
// Lab override: environment MRG_GATE_RPT (read once).
static int gate_rows_per_trip() {
  static int v = [] { const char* e = getenv("MRG_GATE_RPT"); return e ? atoi(e) : 2; }();
  return v;
}

static SegPlan make_plan(int64_t b0, int64_t b1, int64_t M, int grid) {
  SegPlan p;
  p.lo[0] = 0;  p.hi[0] = b0;
  p.lo[1] = b0; p.hi[1] = b1;
  p.lo[2] = b1; p.hi[2] = M;
  int nseg = 0;
  for (int s = 0; s < 3; ++s) nseg += (p.hi[s] > p.lo[s]);
  int spare = grid - nseg;                     // every non-empty segment gets one block first
  p.blk[0] = 0;
  for (int s = 0; s < 3; ++s) {
    int64_t rows = p.hi[s] - p.lo[s];
    int nb = rows > 0 ? 1 + (int)((double)spare * (double)rows / (double)M) : 0;
    p.blk[s + 1] = p.blk[s] + nb;
  }
  return p;                                    // blk[3] <= grid
}

// RPT rows per trip and lane group (round 5): a trip is load -> dot -> group sum -> sigmoid -> scale -> store, one dependent chain per
// row; with ONE row in flight per wave the C5 launches (D = 256: a wave per row) ran at 0.44 of the HBM peak with 39-55 % of the wave
// cycles stalled at issue (profiles/r4_sq_counters_c5.txt).  With RPT = 2 the loads of two rows are issued before the first reduction
// and the two chains interleave.  Same arithmetic per row: bit-identical outputs.
template <int VEC, int LPR, int KMAX, bool HAS_IN, int RPT>
__global__ __launch_bounds__(MRG_BLOCK) void gate_fwd_k(const float* __restrict__ s, const float* __restrict__ sin_,
                                                        const float* __restrict__ norm, const float* __restrict__ uvc,
                                                        float* __restrict__ out, SegPlan p, int D, float scale) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int b = blockIdx.x;
  if (b >= p.blk[3]) return;
  const int seg = (b >= p.blk[1]) + (b >= p.blk[2]);
  const int jb = b - p.blk[seg], nb = p.blk[seg + 1] - p.blk[seg];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const float* u = uvc + (int64_t)seg * MRG_GATE_LD(D);
  const float cc = u[HAS_IN ? 2 * D : D];
  Vec<VEC> uk[KMAX], vk[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
    uk[k] = c < dv ? Vec<VEC>::load(u + c * VEC) : Vec<VEC>::fill(0.f);
    vk[k] = (HAS_IN && c < dv) ? Vec<VEC>::load(u + D + c * VEC) : Vec<VEC>::fill(0.f);
  }
  const bool use_norm = norm != nullptr && seg < 2;
  const int64_t stride = (int64_t)nb * RPB;
  for (int64_t r0 = p.lo[seg] + (int64_t)jb * RPB + rw; r0 < p.hi[seg]; r0 += stride * RPT) {
    Vec<VEC> sk[RPT][KMAX];
    float dot[RPT];
    int64_t rr[RPT];
    bool live[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
      const int64_t r = r0 + t * stride;
      live[t] = r < p.hi[seg];
      rr[t] = live[t] ? r : r0;                              // beyond the range: re-read the first row (never stored)
      dot[t] = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        int c = sl + k * LPR;
        if (c < dv) {
          sk[t][k] = Vec<VEC>::load(s + rr[t] * D + c * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) dot[t] += sk[t][k][j] * uk[k][j];
          if (HAS_IN) {
            Vec<VEC> x = Vec<VEC>::load(sin_ + rr[t] * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) dot[t] += x[j] * vk[k][j];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
      float z = group_sum<LPR>(dot[t]) + cc;
      float f = sigmoidf_fast(z) * scale * (use_norm ? norm[rr[t]] : 1.0f);
      if (live[t]) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (c < dv) {
            Vec<VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = sk[t][k][j] * f;
            o.store(out + rr[t] * D + c * VEC);
          }
        }
      }
    }
  }
}

template <int VEC, int LPR, int KMAX, bool HAS_IN, int RPT>
__global__ __launch_bounds__(MRG_BLOCK) void gate_bwd_k(const float* __restrict__ g, const float* __restrict__ s,
                                                        const float* __restrict__ sin_, const float* __restrict__ norm,
                                                        const float* __restrict__ uvc, float* __restrict__ gs,
                                                        float* __restrict__ gsin, float* __restrict__ ws, SegPlan p,
                                                        int D, float scale) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;                 // >= D
  __shared__ float red[RPB * (2 * WIDTH + 1)];
  const int b = blockIdx.x;
  if (b >= p.blk[3]) return;
  const int seg = (b >= p.blk[1]) + (b >= p.blk[2]);
  const int jb = b - p.blk[seg], nb = p.blk[seg + 1] - p.blk[seg];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int ld = MRG_GATE_LD(D);
  const int cidx = HAS_IN ? 2 * D : D;
  const float* u = uvc + (int64_t)seg * ld;
  const float cc = u[HAS_IN ? 2 * D : D];
  Vec<VEC> uk[KMAX], vk[KMAX], du[KMAX], dvv[KMAX];
  double dc = 0.0;                                        // the sum of mixed-sign dz_r cancels heavily: double (round-5 f32 control)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
    uk[k] = c < dv ? Vec<VEC>::load(u + c * VEC) : Vec<VEC>::fill(0.f);
    vk[k] = (HAS_IN && c < dv) ? Vec<VEC>::load(u + D + c * VEC) : Vec<VEC>::fill(0.f);
    du[k] = Vec<VEC>::fill(0.f);
    dvv[k] = Vec<VEC>::fill(0.f);
  }
  const bool use_norm = norm != nullptr && seg < 2;
  const int64_t stride = (int64_t)nb * RPB;
  // rows of a lane group in ascending order whatever RPT is (trip t0 handles r0, r0 + stride, ...): du / dv / dc receive the same
  // terms in the same order as with one row per trip -- bit-identical partial sums
  for (int64_t r0 = p.lo[seg] + (int64_t)jb * RPB + rw; r0 < p.hi[seg]; r0 += stride * RPT) {
    Vec<VEC> sk[RPT][KMAX], xk[RPT][KMAX], gk[RPT][KMAX];
    float dz_[RPT], dq[RPT];
    int64_t rr[RPT];
    bool live[RPT];
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
      const int64_t r = r0 + t * stride;
      live[t] = r < p.hi[seg];
      rr[t] = live[t] ? r : r0;
      dz_[t] = 0.f; dq[t] = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        int c = sl + k * LPR;
        if (c < dv) {
          sk[t][k] = Vec<VEC>::load(s + rr[t] * D + c * VEC);
          gk[t][k] = Vec<VEC>::load(g + rr[t] * D + c * VEC);
          if (HAS_IN) xk[t][k] = Vec<VEC>::load(sin_ + rr[t] * D + c * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            dz_[t] += sk[t][k][j] * uk[k][j];
            if (HAS_IN) dz_[t] += xk[t][k][j] * vk[k][j];
            dq[t] += gk[t][k][j] * sk[t][k][j];
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < RPT; ++t) {
      float z = group_sum<LPR>(dz_[t]) + cc;
      float q = group_sum<LPR>(dq[t]);
      float gt = sigmoidf_fast(z);
      float tt = scale * (use_norm ? norm[rr[t]] : 1.0f);
      float dz = live[t] ? q * tt * gt * (1.0f - gt) : 0.f;
      float f = gt * tt;
      if (live[t]) {
        dc += (double)dz;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (c < dv) {
            Vec<VEC> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              o[j] = gk[t][k][j] * f + dz * uk[k][j];
              du[k][j] += dz * sk[t][k][j];
            }
            o.store(gs + rr[t] * D + c * VEC);
            if (HAS_IN) {
              Vec<VEC> o2;
#pragma unroll
              for (int j = 0; j < VEC; ++j) {
                o2[j] = dz * vk[k][j];
                dvv[k][j] += dz * xk[t][k][j];
              }
              o2.store(gsin + rr[t] * D + c * VEC);
            }
          }
        }
      }
    }
  }
  // block reduction of (du, dv, dc) over the RPB row groups, fixed order
  float* mine = red + rw * (2 * WIDTH + 1);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mine[c * VEC + j] = du[k][j];
      mine[WIDTH + c * VEC + j] = dvv[k][j];
    }
  }
  if (sl == 0) mine[2 * WIDTH] = (float)dc;
  __syncthreads();
  float* dst = ws + (int64_t)b * ld;
  for (int t = threadIdx.x; t < ld; t += MRG_BLOCK) {
    int srcidx = t < D ? t : ((HAS_IN && t < 2 * D) ? WIDTH + (t - D) : (t == cidx ? 2 * WIDTH : -1));
    double acc = 0.0;
    if (srcidx >= 0) {
#pragma unroll
      for (int q = 0; q < RPB; ++q) acc += (double)red[q * (2 * WIDTH + 1) + srcidx];
    }
    dst[t] = (float)acc;
  }
}

// ---- the gate as a ROW FACTOR (the candidate y = s * f_r is never stored: the MixedOp epilogue recomputes it, mixedop.hip) ----
// fvec[r] = sigmoid(u.s + v.s_in + c0) * t_r   (t_r = scale * norm_r on edge rows, scale on self rows: f of gate_fwd_k, same
// expression, same dot order: y = s * fvec[r] is gate_fwd_k's output bit for bit);  hvec[r] = t_r * gate * (1 - gate), the factor
// between the row dot q_r = sum_c gy * s and the pre-activation gradient dz_r of gate_bwd_k.
template <int VEC, int LPR, int KMAX, bool HAS_IN>
__global__ __launch_bounds__(MRG_BLOCK) void gate_row_fwd_k(const float* __restrict__ s, const float* __restrict__ sin_,
                                                            const float* __restrict__ norm, const float* __restrict__ uvc,
                                                            float* __restrict__ fvec, float* __restrict__ hvec, SegPlan p, int D, float scale) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int b = blockIdx.x;
  if (b >= p.blk[3]) return;
  const int seg = (b >= p.blk[1]) + (b >= p.blk[2]);
  const int jb = b - p.blk[seg], nb = p.blk[seg + 1] - p.blk[seg];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const float* u = uvc + (int64_t)seg * MRG_GATE_LD(D);
  const float cc = u[HAS_IN ? 2 * D : D];
  Vec<VEC> uk[KMAX], vk[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
    uk[k] = c < dv ? Vec<VEC>::load(u + c * VEC) : Vec<VEC>::fill(0.f);
    vk[k] = (HAS_IN && c < dv) ? Vec<VEC>::load(u + D + c * VEC) : Vec<VEC>::fill(0.f);
  }
  const bool use_norm = norm != nullptr && seg < 2;
  for (int64_t r = p.lo[seg] + (int64_t)jb * RPB + rw; r < p.hi[seg]; r += (int64_t)nb * RPB) {
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) {
        const Vec<VEC> sk = Vec<VEC>::load(s + r * D + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dot += sk[j] * uk[k][j];
        if (HAS_IN) {
          Vec<VEC> x = Vec<VEC>::load(sin_ + r * D + c * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) dot += x[j] * vk[k][j];
        }
      }
    }
    const float z = group_sum<LPR>(dot) + cc;
    const float gt = sigmoidf_fast(z);
    const float nv = use_norm ? norm[r] : 1.0f;
    if (sl == 0) {
      fvec[r] = gt * scale * nv;
      const float t = scale * nv;
      hvec[r] = t * gt * (1.0f - gt);
    }
  }
}

// q[r] (the gradient w.r.t. fvec[r], from the epilogue's backward), dz_r = q_r * hvec[r] -> gs_in = dz_r * v (HAS_IN), per-block
// partials of du = sum_r dz_r s_r, dv = sum_r dz_r s_in_r, dc = sum_r dz_r  (the block reduction and its order are gate_bwd_k's).
// The gradient w.r.t. s (gy * f_r + dz_r * u) was added to the epilogue's gs_out.
template <int VEC, int LPR, int KMAX, bool HAS_IN>
__global__ __launch_bounds__(MRG_BLOCK) void gate_row_bwd_k(const float* __restrict__ qv, const float* __restrict__ hv, const float* __restrict__ s,
                                                            const float* __restrict__ sin_, const float* __restrict__ uvc,
                                                            float* __restrict__ gsin, float* __restrict__ ws, SegPlan p, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;                 // >= D
  __shared__ float red[RPB * (2 * WIDTH + 1)];
  const int b = blockIdx.x;
  if (b >= p.blk[3]) return;
  const int seg = (b >= p.blk[1]) + (b >= p.blk[2]);
  const int jb = b - p.blk[seg], nb = p.blk[seg + 1] - p.blk[seg];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  const int ld = MRG_GATE_LD(D);
  const int cidx = HAS_IN ? 2 * D : D;
  const float* u = uvc + (int64_t)seg * ld;
  Vec<VEC> vk[KMAX], du[KMAX], dvv[KMAX];
  double dc = 0.0;                                        // the sum of mixed-sign dz_r cancels heavily: double (round-5 f32 control)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
    vk[k] = (HAS_IN && c < dv) ? Vec<VEC>::load(u + D + c * VEC) : Vec<VEC>::fill(0.f);
    du[k] = Vec<VEC>::fill(0.f);
    dvv[k] = Vec<VEC>::fill(0.f);
  }
  for (int64_t r = p.lo[seg] + (int64_t)jb * RPB + rw; r < p.hi[seg]; r += (int64_t)nb * RPB) {
    const float dz = qv[r] * hv[r];
    dc += (double)dz;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) {
        const Vec<VEC> sk = Vec<VEC>::load(s + r * D + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) du[k][j] += dz * sk[j];
        if (HAS_IN) {
          const Vec<VEC> xk = Vec<VEC>::load(sin_ + r * D + c * VEC);
          Vec<VEC> o2;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            o2[j] = dz * vk[k][j];
            dvv[k][j] += dz * xk[j];
          }
          o2.store(gsin + r * D + c * VEC);
        }
      }
    }
  }
  float* mine = red + rw * (2 * WIDTH + 1);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mine[c * VEC + j] = du[k][j];
      mine[WIDTH + c * VEC + j] = dvv[k][j];
    }
  }
  if (sl == 0) mine[2 * WIDTH] = (float)dc;
  __syncthreads();
  float* dst = ws + (int64_t)b * ld;
  for (int t = threadIdx.x; t < ld; t += MRG_BLOCK) {
    int srcidx = t < D ? t : ((HAS_IN && t < 2 * D) ? WIDTH + (t - D) : (t == cidx ? 2 * WIDTH : -1));
    double acc = 0.0;
    if (srcidx >= 0) {
#pragma unroll
      for (int q = 0; q < RPB; ++q) acc += (double)red[q * (2 * WIDTH + 1) + srcidx];
    }
    dst[t] = (float)acc;
  }
}

// d_uvc[s][t] = sum over the blocks of segment s, in block order
__global__ void gate_reduce_k(const float* __restrict__ ws, float* __restrict__ d_uvc, SegPlan p, int ld) {
  int seg = blockIdx.y;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ld) return;
  double acc = 0.0;                                      // hundreds of block partials of mixed sign, in block order: double
  for (int b = p.blk[seg]; b < p.blk[seg + 1]; ++b) acc += (double)ws[(int64_t)b * ld + t];
  d_uvc[seg * ld + t] = (float)acc;
}

// uvc[k] = sum_j W[j,k] a[j]  (k < in_dim);  uvc[in_dim] = sum_j a[j] b[j].
// 64 x 16 threads per 64 columns: thread row ty sums j = ty, ty+16, ... (coalesced along k).
__global__ void gate_collapse_k(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ a,
                                float* __restrict__ uvc, int D, int in_dim) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (k < in_dim) {
    for (int j = ty; j < D; j += 16) acc += W[(int64_t)j * in_dim + k] * a[j];
  } else if (k == in_dim && b) {
    for (int j = ty; j < D; j += 16) acc += a[j] * b[j];
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && k <= in_dim) {
    float tot = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += part[i][tx];
    uvc[k] = tot;
  }
}

// one block per output row j of W
__global__ void gate_param_grad_k(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ a,
                                  const float* __restrict__ d, float* __restrict__ gW, float* __restrict__ gb,
                                  float* __restrict__ ga, int D, int in_dim) {
  __shared__ float part[MRG_BLOCK / MRG_WAVE];
  const int j = blockIdx.x;
  const float aj = a[j];
  float acc = 0.f;
  for (int k = threadIdx.x; k < in_dim; k += blockDim.x) {
    float dk = d[k];
    gW[(int64_t)j * in_dim + k] = aj * dk;
    acc += W[(int64_t)j * in_dim + k] * dk;
  }
  acc = group_sum<64>(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < MRG_BLOCK / MRG_WAVE; ++w) tot += part[w];
    float dc = d[in_dim];
    ga[j] = tot + (b ? b[j] * dc : 0.f);
    if (gb) gb[j] = aj * dc;
  }
}

}  // namespace mrg

using namespace mrg;

// ---- batched forms: the three direction segments of an operator in ONE launch each, with the "both operands are the same
// rows" fold built in (fold != 0: the nn.Linear parameters have in_dim = 2D, the gate sees u + v and c at index D) --------
struct Ptr3 { const float* p[3]; };
struct MPtr3 { float* p[3]; };

__global__ void gate_collapse3_k(Ptr3 W, Ptr3 b, Ptr3 a, float* __restrict__ uvc, int D, int in_dim, int fold) {
  __shared__ float part[16][64];
  const int seg = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + tx;
  const int width = fold ? D : in_dim;                       // columns of the collapsed vector before the constant
  float* out = uvc + (int64_t)seg * MRG_GATE_LD(D);
  const float* Ws = W.p[seg];
  const float* as = a.p[seg];
  const float* bs = b.p[seg];
  float acc = 0.f;
  if (Ws != nullptr) {
    if (k < width) {
      for (int j = ty; j < D; j += 16) {
        float w = Ws[(int64_t)j * in_dim + k];
        if (fold) w += Ws[(int64_t)j * in_dim + k + D];
        acc += w * as[j];
      }
    } else if (k == width && bs) {
      for (int j = ty; j < D; j += 16) acc += as[j] * bs[j];
    }
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && k <= width) {
    float tot = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += part[i][tx];
    out[k] = tot;
  }
}

// one block per (output row j of W, segment)
__global__ void gate_param_grad3_k(Ptr3 W, Ptr3 b, Ptr3 a, const float* __restrict__ d_uvc, MPtr3 gW, MPtr3 gb, MPtr3 ga, int D, int in_dim,
                                   int fold) {
  __shared__ float part[MRG_BLOCK / MRG_WAVE];
  const int seg = blockIdx.y, j = blockIdx.x;
  const float* Ws = W.p[seg];
  if (Ws == nullptr) return;
  const float* d = d_uvc + (int64_t)seg * MRG_GATE_LD(D);
  const float aj = a.p[seg][j];
  float acc = 0.f;
  for (int k = threadIdx.x; k < in_dim; k += blockDim.x) {
    const float dk = d[fold ? (k < D ? k : k - D) : k];
    gW.p[seg][(int64_t)j * in_dim + k] = aj * dk;
    acc += Ws[(int64_t)j * in_dim + k] * dk;
  }
  acc = group_sum<64>(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < MRG_BLOCK / MRG_WAVE; ++w) tot += part[w];
    const float dc = d[fold ? D : in_dim];
    ga.p[seg][j] = tot + (b.p[seg] ? b.p[seg][j] * dc : 0.f);
    if (gb.p[seg]) gb.p[seg][j] = aj * dc;
  }
}

// Wt_i = W_i[:, :D] + W_i[:, D:]   /   gW_i = [gWt_i | gWt_i]   (dense filters whose two operands are the same rows)
__global__ void fold_halves3_k(Ptr3 W, float* __restrict__ Wt, int D) {
  const int seg = blockIdx.y;
  const float* Ws = W.p[seg];
  if (Ws == nullptr) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D * D; i += gridDim.x * blockDim.x) {
    const int r = i / D, c = i - r * D;
    Wt[(int64_t)seg * D * D + i] = Ws[(int64_t)r * 2 * D + c] + Ws[(int64_t)r * 2 * D + D + c];
  }
}
__global__ void unfold_halves3_k(Ptr3 gWt, MPtr3 gW, int D) {
  const int seg = blockIdx.y;
  float* g = gW.p[seg];
  const float* src = gWt.p[seg];
  if (g == nullptr) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D * 2 * D; i += gridDim.x * blockDim.x) {
    const int r = i / (2 * D), c = i - r * 2 * D;
    g[i] = src ? src[(int64_t)r * D + (c < D ? c : c - D)] : 0.f;
  }
}

extern "C" int mrg_gate_collapse(const float* W, const float* b, const float* a, float* uvc, int D, int in_dim, void* stream) {
  if (!W || !a || !uvc) return MRG_E_NULLPTR;
  if (D <= 0 || in_dim <= 0) return MRG_E_SHAPE;
  hipLaunchKernelGGL(gate_collapse_k, dim3((in_dim + 1 + 63) / 64), dim3(1024), 0, (hipStream_t)stream, W, b, a, uvc, D, in_dim);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_gate_param_grad(const float* W, const float* b, const float* a, const float* d_uvc, float* gW, float* gb,
                                   float* ga, int D, int in_dim, void* stream) {
  if (!W || !a || !d_uvc || !gW || !ga) return MRG_E_NULLPTR;
  if (D <= 0 || in_dim <= 0) return MRG_E_SHAPE;
  hipLaunchKernelGGL(gate_param_grad_k, dim3(D), dim3(MRG_BLOCK), 0, (hipStream_t)stream, W, b, a, d_uvc, gW, gb, ga, D, in_dim);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

static Ptr3 ptr3(const float* const* h) { Ptr3 p; for (int i = 0; i < 3; ++i) p.p[i] = h ? h[i] : nullptr; return p; }
static MPtr3 mptr3(float* const* h) { MPtr3 p; for (int i = 0; i < 3; ++i) p.p[i] = h ? h[i] : nullptr; return p; }

extern "C" int mrg_gate_collapse3(const float* const* W_host, const float* const* b_host, const float* const* a_host, float* uvc, int D,
                                  int in_dim, int fold, void* stream) {
  if (!W_host || !a_host || !uvc) return MRG_E_NULLPTR;
  if (D <= 0 || in_dim <= 0 || (fold && in_dim != 2 * D)) return MRG_E_SHAPE;
  for (int i = 0; i < 3; ++i)
    if (W_host[i] && !a_host[i]) return MRG_E_NULLPTR;
  const int width = fold ? D : in_dim;
  hipLaunchKernelGGL(gate_collapse3_k, dim3((width + 1 + 63) / 64, 3), dim3(1024), 0, (hipStream_t)stream, ptr3(W_host), ptr3(b_host), ptr3(a_host),
                     uvc, D, in_dim, fold);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_gate_param_grad3(const float* const* W_host, const float* const* b_host, const float* const* a_host, const float* d_uvc,
                                    float* const* gW_host, float* const* gb_host, float* const* ga_host, int D, int in_dim, int fold, void* stream) {
  if (!W_host || !a_host || !d_uvc || !gW_host || !ga_host) return MRG_E_NULLPTR;
  if (D <= 0 || in_dim <= 0 || (fold && in_dim != 2 * D)) return MRG_E_SHAPE;
  for (int i = 0; i < 3; ++i)
    if (W_host[i] && (!a_host[i] || !gW_host[i] || !ga_host[i])) return MRG_E_NULLPTR;
  hipLaunchKernelGGL(gate_param_grad3_k, dim3(D, 3), dim3(MRG_BLOCK), 0, (hipStream_t)stream, ptr3(W_host), ptr3(b_host), ptr3(a_host), d_uvc,
                     mptr3(gW_host), mptr3(gb_host), mptr3(ga_host), D, in_dim, fold);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_fold_halves3(const float* const* W_host, float* Wt, int D, void* stream) {
  if (!W_host || !Wt) return MRG_E_NULLPTR;
  if (D <= 0) return MRG_E_SHAPE;
  hipLaunchKernelGGL(fold_halves3_k, dim3((D * D + 255) / 256 < 64 ? (D * D + 255) / 256 : 64, 3), dim3(256), 0, (hipStream_t)stream, ptr3(W_host), Wt, D);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_unfold_halves3(const float* const* gWt_host, float* const* gW_host, int D, void* stream) {
  if (!gWt_host || !gW_host) return MRG_E_NULLPTR;
  if (D <= 0) return MRG_E_SHAPE;
  hipLaunchKernelGGL(unfold_halves3_k, dim3((2 * D * D + 255) / 256 < 128 ? (2 * D * D + 255) / 256 : 128, 3), dim3(256), 0, (hipStream_t)stream, ptr3(gWt_host),
                     mptr3(gW_host), D);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_gate_fwd(const float* s, const float* s_in, const float* norm, const float* uvc, float* out, int64_t b0,
                            int64_t b1, int64_t M, int D, float scale, void* stream) {
  if (D <= 0 || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (M == 0) return MRG_OK;
  if (!s || !uvc || !out) return MRG_E_NULLPTR;
  RowGeom g = row_geom(D, aligned16(s) && aligned16(s_in) && aligned16(out) && aligned16(uvc));
  if (!g.ok) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
#define CALL(V, L, K)                                                                                                   \
  do {                                                                                                                  \
    grid = grid_for(M, (MRG_BLOCK / L) * 4);                                                                                   \
    if (grid < 3) grid = 3;                                                                                             \
    SegPlan p = make_plan(b0, b1, M, grid);                                                                             \
    if (gate_rows_per_trip() == 2 && L == 64) {                                                                          \
      if (s_in) hipLaunchKernelGGL((gate_fwd_k<V, L, K, true, 2>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, out, p, D, scale); \
      else hipLaunchKernelGGL((gate_fwd_k<V, L, K, false, 2>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, out, p, D, scale); \
    } else {                                                                                                            \
      if (s_in) hipLaunchKernelGGL((gate_fwd_k<V, L, K, true, 1>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, out, p, D, scale); \
      else hipLaunchKernelGGL((gate_fwd_k<V, L, K, false, 1>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, out, p, D, scale); \
    }                                                                                                                   \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int64_t mrg_gate_bwd_workspace_bytes(int64_t M, int D) {
  (void)M;
  return (int64_t)(MRG_MAX_GRID + 3) * MRG_GATE_LD((int64_t)D) * sizeof(float);
}

extern "C" int mrg_gate_bwd(const float* gout, const float* s, const float* s_in, const float* norm, const float* uvc,
                            float* gs, float* gs_in, float* d_uvc, void* ws, int64_t b0, int64_t b1, int64_t M, int D,
                            float scale, void* stream) {
  if (D <= 0 || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (!uvc || !d_uvc) return MRG_E_NULLPTR;
  if (M > 0 && (!gout || !s || !gs)) return MRG_E_NULLPTR;
  if (s_in && !gs_in) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int ld = MRG_GATE_LD(D);
  RowGeom g = row_geom(D, aligned16(gout) && aligned16(s) && aligned16(s_in) && aligned16(gs) && aligned16(gs_in) && aligned16(uvc));
  if (!g.ok) return MRG_E_SHAPE;
  SegPlan p{};
#define CALL(V, L, K)                                                                                                   \
  do {                                                                                                                  \
    int grid = grid_for(M, (MRG_BLOCK / L) * 4);                                                                               \
    if (grid < 3) grid = 3;                                                                                             \
    p = make_plan(b0, b1, M, grid);                                                                                     \
    if (M > 0) {                                                                                                        \
      if (gate_rows_per_trip() == 2 && L == 64) {                                                                        \
        if (s_in) hipLaunchKernelGGL((gate_bwd_k<V, L, K, true, 2>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, gout, s, s_in, norm, uvc, gs, gs_in, (float*)ws, p, D, scale); \
        else hipLaunchKernelGGL((gate_bwd_k<V, L, K, false, 2>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, gout, s, s_in, norm, uvc, gs, gs_in, (float*)ws, p, D, scale); \
      } else {                                                                                                          \
        if (s_in) hipLaunchKernelGGL((gate_bwd_k<V, L, K, true, 1>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, gout, s, s_in, norm, uvc, gs, gs_in, (float*)ws, p, D, scale); \
        else hipLaunchKernelGGL((gate_bwd_k<V, L, K, false, 1>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, gout, s, s_in, norm, uvc, gs, gs_in, (float*)ws, p, D, scale); \
      }                                                                                                                 \
    }                                                                                                                   \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  launch_ordered_reduce_ranges<float>((const float*)ws, d_uvc, p.blk, 3, ld, ld, ld, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// f_sparse_op_comp as a row factor (see gate_row_fwd_k): fvec / hvec [M].
extern "C" int mrg_gate_row_fwd(const float* s, const float* s_in, const float* norm, const float* uvc, float* fvec, float* hvec,
                                int64_t b0, int64_t b1, int64_t M, int D, float scale, void* stream) {
  if (D <= 0 || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (M == 0) return MRG_OK;
  if (!s || !uvc || !fvec || !hvec) return MRG_E_NULLPTR;
  RowGeom g = row_geom(D, aligned16(s) && aligned16(s_in) && aligned16(uvc));
  if (!g.ok) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define CALL(V, L, K)                                                                                                   \
  do {                                                                                                                  \
    int grid = grid_for(M, (MRG_BLOCK / L) * 4);                                                                        \
    if (grid < 3) grid = 3;                                                                                             \
    SegPlan p = make_plan(b0, b1, M, grid);                                                                             \
    if (s_in) hipLaunchKernelGGL((gate_row_fwd_k<V, L, K, true>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, fvec, hvec, p, D, scale); \
    else hipLaunchKernelGGL((gate_row_fwd_k<V, L, K, false>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, s, s_in, norm, uvc, fvec, hvec, p, D, scale); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// q [M] (written by mrg_mix_bwd_apply, mrg_gated_branch.row_dq), hvec [M] (mrg_gate_row_fwd) -> gs_in [M, D] (s_in != NULL) and
// d_uvc [3][MRG_GATE_LD(D)] (then mrg_gate_param_grad3).  ws: mrg_gate_bwd_workspace_bytes(M, D).
extern "C" int mrg_gate_row_bwd(const float* q, const float* hvec, const float* s, const float* s_in, const float* uvc, float* gs_in, float* d_uvc,
                                void* ws, int64_t b0, int64_t b1, int64_t M, int D, void* stream) {
  if (D <= 0 || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (!uvc || !d_uvc) return MRG_E_NULLPTR;
  if (M > 0 && (!q || !hvec || !s)) return MRG_E_NULLPTR;
  if (s_in && !gs_in) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int ld = MRG_GATE_LD(D);
  RowGeom g = row_geom(D, aligned16(s) && aligned16(s_in) && aligned16(gs_in) && aligned16(uvc));
  if (!g.ok) return MRG_E_SHAPE;
  SegPlan p{};
#define CALL(V, L, K)                                                                                                   \
  do {                                                                                                                  \
    int grid = grid_for(M, (MRG_BLOCK / L) * 4);                                                                        \
    if (grid < 3) grid = 3;                                                                                             \
    p = make_plan(b0, b1, M, grid);                                                                                     \
    if (M > 0) {                                                                                                        \
      if (s_in) hipLaunchKernelGGL((gate_row_bwd_k<V, L, K, true>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, q, hvec, s, s_in, uvc, gs_in, (float*)ws, p, D); \
      else hipLaunchKernelGGL((gate_row_bwd_k<V, L, K, false>), dim3(p.blk[3]), dim3(MRG_BLOCK), 0, st, q, hvec, s, s_in, uvc, gs_in, (float*)ws, p, D); \
    }                                                                                                                   \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  launch_ordered_reduce_ranges<float>((const float*)ws, d_uvc, p.blk, 3, ld, ld, ld, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
