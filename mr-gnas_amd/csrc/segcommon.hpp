// Shared pieces of the destination-segmented reducers (segreduce.hip, fused_gcs.hip):
// per-lane accumulators, row finalisation, and the ordered combine of hub partials.
#pragma once
#include "common.hpp"
#include <math.h>

namespace mrg {

template <int VEC, int LPR, int KMAX, bool IS_MAX>
struct Acc {
  Vec<VEC> val[KMAX];
  IVec<VEC> arg[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      val[k] = Vec<VEC>::fill(IS_MAX ? -INFINITY : 0.f);
      arg[k] = IVec<VEC>::fill(-1);
    }
  }
  // strict '>' keeps the first (lowest edge id) of equal maxima
  __device__ __forceinline__ void take(int k, const Vec<VEC>& x, int e) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (IS_MAX) {
        if (x[j] > val[k][j]) { val[k][j] = x[j]; arg[k][j] = e; }
      } else {
        val[k][j] += x[j];
      }
    }
  }
  __device__ __forceinline__ void take_partial(int k, const Vec<VEC>& x, const IVec<VEC>& a) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (IS_MAX) {
        if (x[j] > val[k][j]) { val[k][j] = x[j]; arg[k][j] = a[j]; }
      } else {
        val[k][j] += x[j];
      }
    }
  }
};

template <int VEC, int LPR, int KMAX, bool IS_MAX>
__device__ __forceinline__ void finalize_row(Acc<VEC, LPR, KMAX, IS_MAX>& acc, int v, int deg, bool is_mean,
                                             const float* __restrict__ self_rows, float* __restrict__ out,
                                             int32_t* __restrict__ arg, int D, int sl) {
  const int dv = D / VEC;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int c = sl + k * LPR;
    if (c < dv) {
      Vec<VEC> o = acc.val[k];
      if (IS_MAX) {
        if (deg == 0) o = Vec<VEC>::fill(0.f);
      } else if (is_mean) {
        float d = (float)(deg > 1 ? deg : 1);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = o[j] / d;
      }
      if (self_rows) {
        Vec<VEC> s = Vec<VEC>::load(self_rows + (int64_t)v * D + c * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += s[j];
      }
      o.store(out + (int64_t)v * D + c * VEC);
      if (IS_MAX && arg) acc.arg[k].store(arg + (int64_t)v * D + c * VEC);
    }
  }
}

// Combine of the partial results of a split (hub) list: one workgroup per hub.  The RPB lane
// groups take contiguous ranges of the hub's partial slots (8 loads in flight each) and their
// sums are combined in range order through LDS -- a fixed association, bitwise reproducible.
// THREADS = 1024 (16 lane groups at LPR = 64) is for plans with few, long hubs (segments = relations): the pass is
// bound by the latency of one workgroup walking its hub's partial rows, so more groups per hub shorten it.
template <int VEC, int LPR, int KMAX, bool IS_MAX, int THREADS = MRG_BLOCK>
__global__ __launch_bounds__(THREADS) void seg_hub_k(const float* __restrict__ self_rows, const int32_t* __restrict__ hub_node,
                                                       const int32_t* __restrict__ hub_first, const int32_t* __restrict__ hub_count,
                                                       int64_t n_hubs, const int32_t* __restrict__ in_degree,
                                                       float* __restrict__ out, int32_t* __restrict__ arg,
                                                       const float* __restrict__ ws_val, const int32_t* __restrict__ ws_arg,
                                                       int D, int is_mean) {
  constexpr int RPB = THREADS / LPR;
  constexpr int WIDTH = LPR * KMAX * VEC;
  constexpr int U = 8;
  __shared__ float sval[RPB * WIDTH];
  __shared__ int32_t sarg[IS_MAX ? RPB * WIDTH : 1];
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t h = blockIdx.x; h < n_hubs; h += gridDim.x) {
    const int v = hub_node[h];
    if (v < 0) continue;                                    // block-uniform: padding beyond the plan's real hubs (capacity-sized launch)
    const int s0 = hub_first[h], cnt = hub_count[h];
    const int per = (cnt + RPB - 1) / RPB;
    const int q0 = rw * per < cnt ? rw * per : cnt, q1 = q0 + per < cnt ? q0 + per : cnt;
    Acc<VEC, LPR, KMAX, IS_MAX> acc;
    acc.init();
    for (int q = q0; q < q1; q += U) {
      Vec<VEC> x[U][KMAX];
      IVec<VEC> a[U][KMAX];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (q + u < q1 && c < dv) {
            x[u][k] = Vec<VEC>::load(ws_val + (int64_t)(s0 + q + u) * D + c * VEC);
            a[u][k] = IS_MAX ? IVec<VEC>::load(ws_arg + (int64_t)(s0 + q + u) * D + c * VEC) : IVec<VEC>::fill(-1);
          }
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (q + u < q1) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (sl + k * LPR < dv) acc.take_partial(k, x[u][k], a[u][k]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sval[rw * WIDTH + c * VEC + j] = acc.val[k][j];
        if (IS_MAX) sarg[rw * WIDTH + c * VEC + j] = acc.arg[k][j];
      }
    }
    __syncthreads();
    if (rw == 0) {
      Acc<VEC, LPR, KMAX, IS_MAX> tot;
      tot.init();
      for (int g = 0; g < RPB; ++g) {
        if (g * per < cnt) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            int c = sl + k * LPR;
            Vec<VEC> xv;
            IVec<VEC> av = IVec<VEC>::fill(-1);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              xv[j] = sval[g * WIDTH + c * VEC + j];
              if (IS_MAX) av[j] = sarg[g * WIDTH + c * VEC + j];
            }
            tot.take_partial(k, xv, av);
          }
        }
      }
      finalize_row<VEC, LPR, KMAX, IS_MAX>(tot, v, in_degree[v], is_mean != 0, self_rows, out, arg, D, sl);
    }
  }
}

// few segments, many spans: hubs are few and long -> the wide workgroup.  Decided from the plan's exact sizes (never from
// hub / slot counts, which may still be capacities when the launch is issued) so that the association is fixed per plan.
inline bool hub_wide(int64_t nseg, int64_t n_spans) { return nseg <= 2048 && n_spans >= 4 * nseg; }

// the hub pass of a sum plan (span kernels): out rows of split / empty segments
template <int VEC, int LPR, int KMAX>
inline void launch_hub_sum(bool wide, const int32_t* hub_seg, const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs,
                           const int32_t* seg_len, float* out, const float* ws_val, int D, hipStream_t st) {
  const int gh = n_hubs < 4096 ? (int)n_hubs : 4096;
  if constexpr (KMAX == 1) {
    if (wide) {
      hipLaunchKernelGGL((seg_hub_k<VEC, LPR, KMAX, false, 1024>), dim3(gh), dim3(1024), 0, st, (const float*)nullptr, hub_seg, hub_first,
                         hub_count, n_hubs, seg_len, out, (int32_t*)nullptr, ws_val, (const int32_t*)nullptr, D, 0);
      return;
    }
  }
  hipLaunchKernelGGL((seg_hub_k<VEC, LPR, KMAX, false>), dim3(gh), dim3(MRG_BLOCK), 0, st, (const float*)nullptr, hub_seg, hub_first,
                     hub_count, n_hubs, seg_len, out, (int32_t*)nullptr, ws_val, (const int32_t*)nullptr, D, 0);
}

}  // namespace mrg
