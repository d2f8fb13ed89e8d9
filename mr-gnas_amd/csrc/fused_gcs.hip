// The fused per-relation kernel: gather -> compose(node row, relation row) -> segmented sum.
//   out[seg,:] = sum over e in list(seg) of  combine(X[xi[e],:], Y[yi[e],:], s[e])
// It replaces, for CompGraphConv (reference models/compgcn.py:58-87):
//   r_feats[etype] * norm  (:58)  ->  apply_edges(u_sub_e | u_mul_e | ccorr)  (:62-67)
//   -> mask + scatter into new_comp_h (:74-82) -> update_all(copy_e, sum) (:87)
// without materialising any [E, D] tensor (the per-direction linears W_O / W_I commute with
// the sum and are applied to the [N, D] result afterwards), and the same kernel with other
// index arrays is its backward (gradients w.r.t. node rows: segments keyed by src; w.r.t.
// relation rows: segments keyed by etype) and the backward of plain gathers.
//
// Work decomposition = segreduce.hip: one LPR-lane group per chunk of a segment's list, whole
// rows as float4 per lane, hub lists split and recombined in order (no float atomics).
// Algorithmic bytes per launch: E*(8 + 4*D) + 4*(nseg+1) + 4*D*(rows(Y) + nseg):
// the gathered X row is counted once per edge, index + scalar 8 B per edge.
#include "segcommon.hpp"

namespace mrg {

template <int MODE> struct NeedsY { static constexpr bool value = MODE == MRG_GCS_SUB || MODE == MRG_GCS_MUL; };

// elementwise modes: SUB, MUL, COPY, NEGS
template <int VEC, int LPR, int KMAX, int MODE>
__global__ __launch_bounds__(MRG_BLOCK) void gcs_k(const float* __restrict__ X, const int32_t* __restrict__ xi,
                                                   const float* __restrict__ Y, const int32_t* __restrict__ yi,
                                                   const float* __restrict__ scal, const int32_t* __restrict__ eid,
                                                   const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ chunk_start,
                                                   const int32_t* __restrict__ chunk_end, const int32_t* __restrict__ chunk_slot,
                                                   int64_t n_chunks, float* __restrict__ out, float* __restrict__ ws_val, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int U = 4;
  constexpr bool NY = NeedsY<MODE>::value;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t ch = (int64_t)blockIdx.x * RPB + rw; ch < n_chunks; ch += (int64_t)gridDim.x * RPB) {
    const int v = chunk_node[ch];
    if (v < 0) continue;                                    // padding beyond the plan's real chunks
    const int j0 = chunk_start[ch], j1 = chunk_end[ch];
    const int slot = chunk_slot[ch];
    Vec<VEC> acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = Vec<VEC>::fill(0.f);
    for (int j = j0; j < j1; j += U) {
      int e[U], ex[U], ey[U];
      float s[U];
      Vec<VEC> x[U][KMAX], y[U][KMAX];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        e[q] = (j + q < j1) ? eid[j + q] : -1;
        ex[q] = e[q] >= 0 ? xi[e[q]] : 0;
        ey[q] = (NY && e[q] >= 0) ? yi[e[q]] : 0;
        s[q] = (scal != nullptr && e[q] >= 0) ? scal[e[q]] : 1.0f;
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (e[q] >= 0 && c < dv) {
            x[q][k] = Vec<VEC>::load(X + (int64_t)ex[q] * D + c * VEC);
            if (NY) y[q][k] = Vec<VEC>::load(Y + (int64_t)ey[q] * D + c * VEC);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (e[q] >= 0) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            if (sl + k * LPR < dv) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) {
                float xv = x[q][k][i];
                float val;
                if (MODE == MRG_GCS_SUB) val = xv - y[q][k][i] * s[q];
                else if (MODE == MRG_GCS_MUL) val = xv * (y[q][k][i] * s[q]);
                else if (MODE == MRG_GCS_COPY) val = xv * s[q];
                else val = -(xv * s[q]);
                acc[k][i] += val;
              }
            }
          }
        }
      }
    }
    float* dst = slot < 0 ? out + (int64_t)v * D : ws_val + (int64_t)slot * D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) acc[k].store(dst + c * VEC);
    }
  }
}

// circular correlation / convolution modes: per edge O(D^2) through an LDS copy of both rows
//   CCORR: val[k] = sum_i x[i] * (y[(i+k) % D] * s)      CCONV: val[k] = s * sum_i x[i] * y[(k-i) % D]
// VEC == 4 (D % 4 == 0), round 4: a lane owns four consecutive outputs k0 .. k0+3 and slides a REGISTER window over y.  The y row
// is stored twice in LDS (z | z: no modulo), for CCONV index-reversed (z[m] = y[(D - m) % D]: conv(x, y)[k] = corr(x, z)[(D - k) % D], so
// the lane's accumulators belong to the mirrored outputs and only the final store permutes).  Four steps of i cost ONE ds_read_b128
// of the next four window values + ONE broadcast ds_read_b128 of x for sixteen multiply-adds; round 3's form read x[t] and four
// y[(t + k) % D] words from LDS for every four multiply-adds and was bound by LDS issue (10.5 TF/s = 0.067 of the vector peak, 4.1 ms
// per launch at the FB15k-237 shape).  VEC == 1 (odd D) keeps the element-wise form.  Sums run over i ascending in both.
template <int VEC, int LPR, int KMAX, int MODE>
__global__ __launch_bounds__(MRG_BLOCK) void gcs_corr_k(const float* __restrict__ X, const int32_t* __restrict__ xi,
                                                        const float* __restrict__ Y, const int32_t* __restrict__ yi,
                                                        const float* __restrict__ scal, const int32_t* __restrict__ eid,
                                                        const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ chunk_start,
                                                        const int32_t* __restrict__ chunk_end, const int32_t* __restrict__ chunk_slot,
                                                        int64_t n_chunks, float* __restrict__ out, float* __restrict__ ws_val, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  constexpr int ROWF = VEC == 4 ? 3 * WIDTH : 2 * WIDTH;     // x | z | z   or   x | y
  __shared__ __align__(16) float lds[RPB * ROWF];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  float* lx = lds + rw * ROWF;
  float* ly = lx + WIDTH;
  for (int64_t ch = (int64_t)blockIdx.x * RPB + rw; ch < n_chunks; ch += (int64_t)gridDim.x * RPB) {
    const int v = chunk_node[ch];
    if (v < 0) continue;                                    // padding beyond the plan's real chunks
    const int j0 = chunk_start[ch], j1 = chunk_end[ch];
    const int slot = chunk_slot[ch];
    Vec<VEC> acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = Vec<VEC>::fill(0.f);
    for (int j = j0; j < j1; ++j) {
      const int e = eid[j];
      const float s = scal != nullptr ? scal[e] : 1.0f;
      const float* xr = X + (int64_t)xi[e] * D;
      const float* yr = Y + (int64_t)yi[e] * D;
      __threadfence_block();                    // earlier reads of lx/ly by this wave are done
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        int c = sl + k * LPR;
        if (c < dv) {
          Vec<VEC> a = Vec<VEC>::load(xr + c * VEC), b = Vec<VEC>::load(yr + c * VEC);
          if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(lx + c * 4) = a.v;
            if (MODE == MRG_GCS_CCORR) {
              *reinterpret_cast<float4*>(ly + c * 4) = b.v;
              *reinterpret_cast<float4*>(ly + D + c * 4) = b.v;
            } else {                            // z[m] = y[(D - m) % D]: y[4c + i] lands at m = (D - 4c - i) % D, in both copies
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int m = (c * 4 + i) == 0 ? 0 : D - (c * 4 + i);
                ly[m] = b[i];
                ly[D + m] = b[i];
              }
            }
          } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              lx[c * VEC + i] = a[i];
              ly[c * VEC + i] = b[i];
            }
          }
        }
      }
      __threadfence_block();                    // the group's rows are in LDS before anyone reads them
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        int c = sl + k * LPR;
        if (c < dv) {
          if constexpr (VEC == 4) {
            // outputs k0 + jj (jj = 0..3) of corr(x, z): sum_i x[i] z[i + k0 + jj]; window w = z[i + k0 .. i + k0 + 3], wn the next four
            const float* zp = ly + c * 4;
            float4 w = *reinterpret_cast<const float4*>(zp);
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll 10
            for (int i = 0; i < D; i += 4) {                // (unrolled: the window rotates through registers without the four moves per step)
              const float4 xv = *reinterpret_cast<const float4*>(lx + i);
              const float4 wn = *reinterpret_cast<const float4*>(zp + i + 4);
              p0 += xv.x * w.x;  p1 += xv.x * w.y;  p2 += xv.x * w.z;  p3 += xv.x * w.w;
              p0 += xv.y * w.y;  p1 += xv.y * w.z;  p2 += xv.y * w.w;  p3 += xv.y * wn.x;
              p0 += xv.z * w.z;  p1 += xv.z * w.w;  p2 += xv.z * wn.x; p3 += xv.z * wn.y;
              p0 += xv.w * w.w;  p1 += xv.w * wn.x; p2 += xv.w * wn.y; p3 += xv.w * wn.z;
              w = wn;
            }
            acc[k][0] += p0 * s; acc[k][1] += p1 * s; acc[k][2] += p2 * s; acc[k][3] += p3 * s;
          } else {
            float part[VEC];
            int idx[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) { part[i] = 0.f; idx[i] = c * VEC + i; }
            for (int t = 0; t < D; ++t) {
              const float xv = lx[t];
#pragma unroll
              for (int i = 0; i < VEC; ++i) {
                part[i] += xv * ly[idx[i]];
                if (MODE == MRG_GCS_CCORR) { idx[i] = idx[i] + 1 == D ? 0 : idx[i] + 1; }
                else { idx[i] = idx[i] == 0 ? D - 1 : idx[i] - 1; }
              }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[k][i] += part[i] * s;
          }
        }
      }
    }
    float* dst = slot < 0 ? out + (int64_t)v * D : ws_val + (int64_t)slot * D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) {
        if constexpr (VEC == 4 && MODE != MRG_GCS_CCORR) {
          // the accumulators are corr(x, z)[k'] for k' = 4c .. 4c+3 = conv(x, y)[(D - k') % D]
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int kp = c * 4 + i;
            dst[kp == 0 ? 0 : D - kp] = acc[k][i];
          }
        } else {
          acc[k].store(dst + c * VEC);
        }
      }
    }
  }
}

// ---- circular correlation / convolution with EIGHT outputs per lane (round 5) ---------------------------------------------------------
// gcs_corr_k above is bound by LDS issue: per four steps of i a lane reads two ds_read_b128 (the next four window values and the
// broadcast x) for sixteen multiply-adds, and the four SIMDs of a CU share one LDS pipe -- 0.25 of the vector peak.  Here a lane owns
// EIGHT consecutive outputs k0 .. k0+7: the register window is twelve values (w0 | w1 | wn), one step of four i still costs two
// ds_read_b128 but feeds THIRTY-TWO multiply-adds.  A row needs D / 8 lanes, so a wave carries two chunks (LPR = 32).  Every output's
// sum runs over i ascending as in gcs_corr_k; the two agree to rounding (3e-7: the compiler contracts the multiply-adds differently).
// D % 8 == 0, D <= 256; MRG_CORR8=0 keeps the four-output kernel.
template <int MODE, int OUT>
__global__ __launch_bounds__(MRG_BLOCK) void gcs_corr8_k(const float* __restrict__ X, const int32_t* __restrict__ xi,
                                                         const float* __restrict__ Y, const int32_t* __restrict__ yi,
                                                         const float* __restrict__ scal, const int32_t* __restrict__ eid,
                                                         const int32_t* __restrict__ chunk_node, const int32_t* __restrict__ chunk_start,
                                                         const int32_t* __restrict__ chunk_end, const int32_t* __restrict__ chunk_slot,
                                                         int64_t n_chunks, float* __restrict__ out, float* __restrict__ ws_val, int D) {
  // OUT outputs per lane (8 or 16): LPR = 256 / OUT lanes per row cover up to 256 outputs; the last lane of a row may own outputs
  // beyond D (computed from the zero padding behind z | z, never stored)
  constexpr int LPR = 256 / OUT, RPB = MRG_BLOCK / LPR;
  constexpr int XW = 256, ZW = 2 * 256 + 32;               // x | z z + the window's read-ahead / the overhang of a partial last lane
  __shared__ __align__(16) float lds[RPB * (XW + ZW)];
  const int sl = threadIdx.x % LPR, rw = threadIdx.x / LPR;
  const int d4 = D >> 2, dl = (D + OUT - 1) / OUT;         // lanes of a row that own outputs
  float* lx = lds + rw * (XW + ZW);
  float* lz = lx + XW;
  for (int64_t ch = (int64_t)blockIdx.x * RPB + rw; ch < n_chunks; ch += (int64_t)gridDim.x * RPB) {
    const int v = chunk_node[ch];
    if (v < 0) continue;                                    // padding beyond the plan's real chunks (the whole lane group skips together)
    const int j0 = chunk_start[ch], j1 = chunk_end[ch];
    const int slot = chunk_slot[ch];
    float acc[OUT];
#pragma unroll
    for (int q = 0; q < OUT; ++q) acc[q] = 0.f;
    for (int j = j0; j < j1; ++j) {
      const int e = eid[j];
      const float s = scal != nullptr ? scal[e] : 1.0f;
      const float* xr = X + (int64_t)xi[e] * D;
      const float* yr = Y + (int64_t)yi[e] * D;
      __threadfence_block();                                // earlier reads of lx / lz by this lane group are done
      __builtin_amdgcn_wave_barrier();
      for (int c = sl; c < d4; c += LPR) {
        const float4 a = *reinterpret_cast<const float4*>(xr + c * 4), b = *reinterpret_cast<const float4*>(yr + c * 4);
        *reinterpret_cast<float4*>(lx + c * 4) = a;
        if (MODE == MRG_GCS_CCORR) {
          *reinterpret_cast<float4*>(lz + c * 4) = b;
          *reinterpret_cast<float4*>(lz + D + c * 4) = b;
        } else {                                            // z[m] = y[(D - m) % D] in both copies
          const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = (c * 4 + i) == 0 ? 0 : D - (c * 4 + i);
            lz[m] = bb[i];
            lz[D + m] = bb[i];
          }
        }
      }
      for (int c = sl; c < 8; c += LPR) *reinterpret_cast<float4*>(lz + 2 * D + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);   // behind z | z: finite
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      if (sl < dl) {
        const float* zp = lz + sl * OUT;
        float w[OUT + 4];
#pragma unroll
        for (int q = 0; q < OUT; q += 4) {
          const float4 t = *reinterpret_cast<const float4*>(zp + q);
          w[q] = t.x; w[q + 1] = t.y; w[q + 2] = t.z; w[q + 3] = t.w;
        }
        float p[OUT];
#pragma unroll
        for (int q = 0; q < OUT; ++q) p[q] = 0.f;
#pragma unroll 10
        for (int i = 0; i < D; i += 4) {                    // (ten steps per trip: the window rotates through its registers without moves)
          const float4 xv = *reinterpret_cast<const float4*>(lx + i);
          const float4 wn = *reinterpret_cast<const float4*>(zp + i + OUT);
          w[OUT] = wn.x; w[OUT + 1] = wn.y; w[OUT + 2] = wn.z; w[OUT + 3] = wn.w;
          const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < OUT; ++q) p[q] += xs[t] * w[t + q];
#pragma unroll
          for (int q = 0; q < OUT; ++q) w[q] = w[q + 4];
        }
#pragma unroll
        for (int q = 0; q < OUT; ++q) acc[q] += p[q] * s;
      }
    }
    float* dst = slot < 0 ? out + (int64_t)v * D : ws_val + (int64_t)slot * D;
    if (sl < dl) {
#pragma unroll
      for (int q = 0; q < OUT; q += 4) {
        const int kp = sl * OUT + q;
        if (kp < D) {                                       // D % 4 == 0: whole float4s
          if (MODE == MRG_GCS_CCORR) {
            *reinterpret_cast<float4*>(dst + kp) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
          } else {                                          // corr(x, z)[k'] = conv(x, y)[(D - k') % D]
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[(kp + u) == 0 ? 0 : D - (kp + u)] = acc[q + u];
          }
        }
      }
    }
  }
}

}  // namespace mrg

using namespace mrg;

// lab switch: MRG_CORR8=0 keeps the four-outputs-per-lane kernel
static int corr8_on() { static const int v = [] { const char* e = getenv("MRG_CORR8"); return e ? atoi(e) : 8; }(); return v; }

extern "C" int mrg_fused_gcs(int mode, const float* X, const int32_t* xi, const float* Y, const int32_t* yi,
                             const float* scal, const int32_t* eid, const int32_t* chunk_node, const int32_t* chunk_start,
                             const int32_t* chunk_end, const int32_t* chunk_slot, int64_t n_chunks, const int32_t* hub_node,
                             const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs, int64_t n_slots,
                             const int32_t* seg_len, float* out, void* ws, int64_t nseg, int D, void* stream) {
  if (mode < 0 || mode > MRG_GCS_CCONV) return MRG_E_ENUM;
  if (nseg < 0 || D <= 0 || n_chunks < nseg || n_hubs < 0 || n_slots < 0) return MRG_E_SHAPE;
  if (nseg == 0) return MRG_OK;
  if (!out || !chunk_node || !chunk_start || !chunk_end || !chunk_slot || !seg_len) return MRG_E_NULLPTR;
  const bool needs_y = mode == MRG_GCS_SUB || mode == MRG_GCS_MUL || mode == MRG_GCS_CCORR || mode == MRG_GCS_CCONV;
  if (n_hubs > 0 && (!hub_node || !hub_first || !hub_count)) return MRG_E_NULLPTR;
  if (n_slots > 0 && !ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws_val = (float*)ws;
  RowGeom g = row_geom(D, aligned16(X) && aligned16(Y) && aligned16(out) && aligned16(ws));
  if (!g.ok) return MRG_E_SHAPE;
  (void)needs_y;
  // eight or sixteen outputs per lane (gcs_corr8_k; MRG_CORR8 = 0 / 8 / 16)
  const int corr8 = (corr8_on() && g.vec == 4 && D <= 256 && D >= 16) ? (corr8_on() == 16 ? 16 : (D % 8 == 0 ? 8 : 0)) : 0;
#define LAUNCH(KERN, V, L, K, M)                                                                                       \
  hipLaunchKernelGGL((KERN<V, L, K, M>), dim3(grid), dim3(MRG_BLOCK), 0, st, X, xi, Y, yi, scal, eid, chunk_node,       \
                     chunk_start, chunk_end, chunk_slot, n_chunks, out, ws_val, D)
#define CALL(V, L, K)                                                                                                  \
  do {                                                                                                                 \
    int grid = grid_for(n_chunks, MRG_BLOCK / L);                                                                      \
    switch (mode) {                                                                                                    \
      case MRG_GCS_SUB: LAUNCH(gcs_k, V, L, K, MRG_GCS_SUB); break;                                                    \
      case MRG_GCS_MUL: LAUNCH(gcs_k, V, L, K, MRG_GCS_MUL); break;                                                    \
      case MRG_GCS_COPY: LAUNCH(gcs_k, V, L, K, MRG_GCS_COPY); break;                                                  \
      case MRG_GCS_NEGS: LAUNCH(gcs_k, V, L, K, MRG_GCS_NEGS); break;                                                  \
      case MRG_GCS_CCORR:                                                                                              \
        if (corr8) { if (corr8 == 16) hipLaunchKernelGGL((gcs_corr8_k<MRG_GCS_CCORR, 16>), dim3(grid_for(n_chunks, MRG_BLOCK / 16)), dim3(MRG_BLOCK), 0, st, X, xi, Y, yi, scal, \
                                        eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, out, ws_val, D); \
          else hipLaunchKernelGGL((gcs_corr8_k<MRG_GCS_CCORR, 8>), dim3(grid_for(n_chunks, MRG_BLOCK / 32)), dim3(MRG_BLOCK), 0, st, X, xi, Y, yi, scal, \
                                        eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, out, ws_val, D); }  \
        else LAUNCH(gcs_corr_k, V, L, K, MRG_GCS_CCORR);                                                               \
        break;                                                                                                         \
      default:                                                                                                         \
        if (corr8) { if (corr8 == 16) hipLaunchKernelGGL((gcs_corr8_k<MRG_GCS_CCONV, 16>), dim3(grid_for(n_chunks, MRG_BLOCK / 16)), dim3(MRG_BLOCK), 0, st, X, xi, Y, yi, scal, \
                                        eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, out, ws_val, D); \
          else hipLaunchKernelGGL((gcs_corr8_k<MRG_GCS_CCONV, 8>), dim3(grid_for(n_chunks, MRG_BLOCK / 32)), dim3(MRG_BLOCK), 0, st, X, xi, Y, yi, scal, \
                                        eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, out, ws_val, D); }  \
        else LAUNCH(gcs_corr_k, V, L, K, MRG_GCS_CCONV);                                                               \
        break;                                                                                                         \
    }                                                                                                                  \
    if (n_hubs > 0) {                                                                                                  \
      int gh = n_hubs < 4096 ? (int)n_hubs : 4096;                                                                        \
      hipLaunchKernelGGL((seg_hub_k<V, L, K, false>), dim3(gh), dim3(MRG_BLOCK), 0, st, (const float*)nullptr, hub_node, \
                         hub_first, hub_count, n_hubs, seg_len, out, (int32_t*)nullptr, ws_val, (const int32_t*)nullptr, D, 0); \
    }                                                                                                                  \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
#undef LAUNCH
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// ======================================================================================
// Span formulation of the elementwise modes (SUB / MUL / COPY / NEGS): perfectly balanced.
// The elements are pre-sorted by segment and packed as {seg, xi, yi, scal} (16 bytes, one
// broadcast load per element).  Every LPR-lane group owns SPAN consecutive sorted elements,
// keeps 8 gathered rows in flight, and sums runs of equal segment id in registers.  A run that
// covers its whole segment is stored straight to out[seg]; only the first and the last run of
// a span can be partial -- they go to consecutive workspace slots that seg_hub_k adds up in
// list order (bitwise reproducible, no atomics).  Segments without elements are listed as hubs
// with zero partials, so every output row is written and no zero-fill pass is needed.
// ======================================================================================
namespace mrg {

// prefetch depth per mode (measured on the FB15k-237 shape, D = 200): two gathers per element (sub / mul) want 8
// elements in flight, one gather (copy / negs) 4 -- deeper costs occupancy, shallower exposes the gather latency
#ifndef MRG_SPAN_U1
#define MRG_SPAN_U1 4
#endif
#ifndef MRG_SPAN_UM
#define MRG_SPAN_UM 4
#endif
#ifndef MRG_SPAN_US
#define MRG_SPAN_US 8
#endif
template <int MODE> struct SPAN_U { static constexpr int value = MODE == MRG_GCS_SUB ? MRG_SPAN_US : (MODE == MRG_GCS_MUL ? MRG_SPAN_UM : MRG_SPAN_U1); };

template <int VEC, int LPR, int KMAX, int MODE, int UOVR = 0>
__global__ __launch_bounds__(MRG_BLOCK) void span_gcs_k(const float* __restrict__ X, const float* __restrict__ Y,
                                                        const int4* __restrict__ meta, const float* __restrict__ ext_scal,
                                                        int64_t E, int span, const int32_t* __restrict__ span_slot,
                                                        const int32_t* __restrict__ span_start, int64_t n_spans, float* __restrict__ out,
                                                        float* __restrict__ ws_val, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  // elements whose row gathers are in flight per lane group.  UOVR (sub mode on big inputs): 6 instead of 8 -- measured on the
  // C5 shape (10 M elements, 1 GB table, D = 256) 2.13 ms against 2.42 ms, while the cache-resident FB15k-237 shape prefers 8
  // (91.7 us against 97.5 us): tools/ns_sweep.py
  constexpr int U = UOVR > 0 ? UOVR : SPAN_U<MODE>::value;
  constexpr bool NY = NeedsY<MODE>::value;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t sp = (int64_t)blockIdx.x * RPB + rw; sp < n_spans; sp += (int64_t)gridDim.x * RPB) {
    // spans cut at segment ends (mrg_span_plan_build's span_start): only segments longer than a span leave partial runs
    const int64_t start = span_start ? span_start[sp] : sp * span;
    const int64_t end = span_start ? span_start[sp + 1] : (start + span < E ? start + span : E);
    if (start >= end) continue;                           // an empty span (the last cut moved to E)
    const int slot_first = span_slot[2 * sp], slot_last = span_slot[2 * sp + 1];
    Vec<VEC> acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = Vec<VEC>::fill(0.f);
    int cur_seg = meta[start].x;
    bool first_run = true;
    for (int64_t j = start; j < end; j += U) {
      int4 m[U];
      Vec<VEC> x[U][KMAX], y[U][KMAX];
#pragma unroll
      for (int q = 0; q < U; ++q) m[q] = meta[j + q < end ? j + q : end - 1];
#pragma unroll
      for (int q = 0; q < U; ++q) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (c < dv) {
            x[q][k] = Vec<VEC>::load(X + (int64_t)m[q].y * D + c * VEC);
            if (NY) y[q][k] = Vec<VEC>::load(Y + (int64_t)m[q].z * D + c * VEC);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (j + q < end) {
          if (m[q].x != cur_seg) {             // uniform within the lane group: a run ends
            float* dst = (first_run && slot_first >= 0) ? ws_val + (int64_t)slot_first * D : out + (int64_t)cur_seg * D;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
              int c = sl + k * LPR;
              if (c < dv) acc[k].store(dst + c * VEC);
              acc[k] = Vec<VEC>::fill(0.f);
            }
            cur_seg = m[q].x;
            first_run = false;
          }
          const float s = ext_scal ? ext_scal[m[q].w] : __int_as_float(m[q].w);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            if (sl + k * LPR < dv) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) {
                const float xv = x[q][k][i];
                float val;
                if (MODE == MRG_GCS_SUB) val = xv - y[q][k][i] * s;
                else if (MODE == MRG_GCS_MUL) val = xv * (y[q][k][i] * s);
                else if (MODE == MRG_GCS_COPY) val = xv * s;
                else val = -(xv * s);
                acc[k][i] += val;
              }
            }
          }
        }
      }
    }
    const int slot = first_run ? slot_first : slot_last;
    float* dst = slot >= 0 ? ws_val + (int64_t)slot * D : out + (int64_t)cur_seg * D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) acc[k].store(dst + c * VEC);
    }
  }
}

}  // namespace mrg

extern "C" int mrg_span_gcs(int mode, const float* X, const float* Y, const void* meta, const float* ext_scal, int64_t E, int span,
                            const int32_t* span_slot, const int32_t* span_start, int64_t n_spans, const int32_t* hub_seg, const int32_t* hub_first,
                            const int32_t* hub_count, int64_t n_hubs, int64_t n_slots, const int32_t* seg_len, float* out,
                            void* ws, int64_t nseg, int D, void* stream) {
  if (mode != MRG_GCS_SUB && mode != MRG_GCS_MUL && mode != MRG_GCS_COPY && mode != MRG_GCS_NEGS) return MRG_E_ENUM;
  if (nseg < 0 || D <= 0 || E < 0 || span < 1 || n_spans < 0 || n_hubs < 0 || n_slots < 0) return MRG_E_SHAPE;
  if (nseg == 0) return MRG_OK;
  if (!out || !seg_len) return MRG_E_NULLPTR;
  if (E > 0 && (!X || !meta || !span_slot)) return MRG_E_NULLPTR;
  if (E > 0 && (mode == MRG_GCS_SUB || mode == MRG_GCS_MUL) && !Y) return MRG_E_NULLPTR;
  if (n_hubs > 0 && (!hub_seg || !hub_first || !hub_count)) return MRG_E_NULLPTR;
  if (n_slots > 0 && !ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws_val = (float*)ws;
  RowGeom g = row_geom(D, aligned16(X) && aligned16(Y) && aligned16(out) && aligned16(ws));
  if (!g.ok) return MRG_E_SHAPE;
  const int4* m4 = (const int4*)meta;
#define LAUNCH(V, L, K, M)                                                                                              \
  hipLaunchKernelGGL((span_gcs_k<V, L, K, M>), dim3(grid), dim3(MRG_BLOCK), 0, st, X, Y, m4, ext_scal, E, span, span_slot, span_start, n_spans, out, ws_val, D)
#define CALL(V, L, K)                                                                                                  \
  do {                                                                                                                 \
    int grid = grid_for(n_spans, MRG_BLOCK / L);                                                                       \
    if (E > 0 && n_spans > 0) switch (mode) {                                                                          \
      case MRG_GCS_SUB:                                                                                                \
        if (E >= ((int64_t)1 << 22))                                                                                   \
          hipLaunchKernelGGL((span_gcs_k<V, L, K, MRG_GCS_SUB, 6>), dim3(grid), dim3(MRG_BLOCK), 0, st, X, Y, m4, ext_scal, E, span, span_slot, span_start, n_spans, out, ws_val, D); \
        else LAUNCH(V, L, K, MRG_GCS_SUB);                                                                             \
        break;                                                                                                         \
      case MRG_GCS_MUL: LAUNCH(V, L, K, MRG_GCS_MUL); break;                                                           \
      case MRG_GCS_COPY: LAUNCH(V, L, K, MRG_GCS_COPY); break;                                                         \
      default: LAUNCH(V, L, K, MRG_GCS_NEGS); break;                                                                   \
    }                                                                                                                  \
    if (n_hubs > 0) launch_hub_sum<V, L, K>(hub_wide(nseg, n_spans), hub_seg, hub_first, hub_count, n_hubs, seg_len, out, ws_val, D, st); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
#undef LAUNCH
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}


// ======================================================================================
// DistMult scores  score[t] = sum_c ent[s_t, c] * rel[r_t, c] * ent[o_t, c]
//   Network.calc_score, reference models/model_search_lp.py:169-176 (three [T, D] gathers, two
//   products and a row sum in the reference).  One lane group per triple; nothing of size [T, D]
//   is written.  Its backward is three mrg_span_gcs launches (mode MUL, ext_scal = dscore).
// ======================================================================================
namespace mrg {
template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void distmult_k(const float* __restrict__ ent, const float* __restrict__ rel,
                                                        const int32_t* __restrict__ si, const int32_t* __restrict__ ri,
                                                        const int32_t* __restrict__ oi, float* __restrict__ score, int64_t T, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t t = (int64_t)blockIdx.x * RPB + rw; t < T; t += (int64_t)gridDim.x * RPB) {
    const float* a = ent + (int64_t)si[t] * D;
    const float* b = rel + (int64_t)ri[t] * D;
    const float* c = ent + (int64_t)oi[t] * D;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int col = sl + k * LPR;
      if (col < dv) {
        Vec<VEC> x = Vec<VEC>::load(a + col * VEC), y = Vec<VEC>::load(b + col * VEC), z = Vec<VEC>::load(c + col * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += x[j] * y[j] * z[j];
      }
    }
    acc = group_sum<LPR>(acc);
    if (sl == 0) score[t] = acc;
  }
}
}  // namespace mrg

extern "C" int mrg_distmult_score(const float* ent, const float* rel, const int32_t* s_idx, const int32_t* r_idx,
                                  const int32_t* o_idx, float* score, int64_t T, int D, void* stream) {
  if (T < 0 || D <= 0) return MRG_E_SHAPE;
  if (T == 0) return MRG_OK;
  if (!ent || !rel || !s_idx || !r_idx || !o_idx || !score) return MRG_E_NULLPTR;
  RowGeom g = row_geom(D, aligned16(ent) && aligned16(rel));
  if (!g.ok) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define CALL(V, L, K) \
  hipLaunchKernelGGL((distmult_k<V, L, K>), dim3(grid_for(T, (MRG_BLOCK / L) * 4)), dim3(MRG_BLOCK), 0, st, ent, rel, s_idx, r_idx, o_idx, score, T, D)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
