// a4 / a5 / a6: destination-segmented max / sum / mean over in-edges (HBM-bound gather of
// whole rows through a CSR-by-destination), replacing DGL's gspmm(copy_e, max|sum|mean):
//   reference models/operations_lp.py:232-234 (a_max), :247-249 (a_mean), :261-263 (a_sum).
// One LPR-lane group walks one chunk of one destination's in-edge list; rows are read as
// float4 per lane (a whole 800-byte row per wave at D = 200).  No float atomics: partial
// results of split (hub) lists go to a workspace and are combined in list order, so the
// result is bitwise reproducible.
// Algorithmic bytes: fwd 4*D*(E + N) read + 4*D*N write (+ 4*E indices); bwd 4*D*N + 4*D*M.
#include "segcommon.hpp"

namespace mrg {

// HEADS (sum / mean only): msg is indexed by LIST POSITION and only the "head" rows hold data -- the run sums that the
// GEMM epilogue gemm_epilogue_segsum left at the first row of every run of one destination inside one half (rows 4h..4h+3
// of every group of 8) of a 32-row strip.  Position j of node v's list is a head iff the previous row of its half of the
// strip lies outside the strip or before the list (head_rowptr[v]): re-derived here, nothing is stored for it.
__device__ __forceinline__ bool seg_is_head(int j, int list_start) {
  const int q = j & 31;
  const int prev = (q & 3) ? j - 1 : (q >= 8 ? j - 5 : -1);
  return prev < list_start;                                // also true for prev == -1 (first group of the half in its strip)
}

template <int VEC, int LPR, int KMAX, bool IS_MAX, bool HEADS = false>
__global__ __launch_bounds__(MRG_BLOCK) void seg_chunk_k(const float* __restrict__ msg, const float* __restrict__ self_rows,
                                                         const int32_t* __restrict__ eid, const int32_t* __restrict__ chunk_node,
                                                         const int32_t* __restrict__ chunk_start, const int32_t* __restrict__ chunk_end,
                                                         const int32_t* __restrict__ chunk_slot, int64_t n_chunks,
                                                         const int32_t* __restrict__ in_degree, float* __restrict__ out,
                                                         int32_t* __restrict__ arg, float* __restrict__ ws_val,
                                                         int32_t* __restrict__ ws_arg, int D, int is_mean,
                                                         const int32_t* __restrict__ head_rowptr = nullptr) {
  constexpr int RPB = MRG_BLOCK / LPR;
  constexpr int U = 4;                                      // rows in flight per lane group
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t ch = (int64_t)blockIdx.x * RPB + rw; ch < n_chunks; ch += (int64_t)gridDim.x * RPB) {
    const int v = chunk_node[ch];
    if (v < 0) continue;                                    // padding beyond the plan's real chunks (capacity-sized launch, see plans.hip)
    const int j0 = chunk_start[ch], j1 = chunk_end[ch];
    const int slot = chunk_slot[ch];
    Acc<VEC, LPR, KMAX, IS_MAX> acc;
    acc.init();
    const int list_start = HEADS ? head_rowptr[v] : 0;
    for (int j = j0; j < j1; j += U) {
      int e[U];
      Vec<VEC> x[U][KMAX];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (HEADS) e[q] = (j + q < j1 && seg_is_head(j + q, list_start)) ? j + q : -1;
        else e[q] = (j + q < j1) ? eid[j + q] : -1;
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          int c = sl + k * LPR;
          if (e[q] >= 0 && c < dv) x[q][k] = Vec<VEC>::load(msg + (int64_t)e[q] * D + c * VEC);
        }
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (e[q] >= 0) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            if (sl + k * LPR < dv) acc.take(k, x[q][k], e[q]);
          }
        }
      }
    }
    if (slot < 0) {
      finalize_row<VEC, LPR, KMAX, IS_MAX>(acc, v, in_degree[v], is_mean != 0, self_rows, out, arg, D, sl);
    } else {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        int c = sl + k * LPR;
        if (c < dv) {
          acc.val[k].store(ws_val + (int64_t)slot * D + c * VEC);
          if (IS_MAX) acc.arg[k].store(ws_arg + (int64_t)slot * D + c * VEC);
        }
      }
    }
  }
}

// rows [0, E): gradient of the edge messages; rows [E, E + N): copy for the self-loop residual
template <int VEC, int LPR, int KMAX, int MODE>
__global__ __launch_bounds__(MRG_BLOCK) void seg_bwd_k(const float* __restrict__ gout, const int32_t* __restrict__ dst,
                                                       const int32_t* __restrict__ in_degree, const int32_t* __restrict__ arg,
                                                       float* __restrict__ gmsg, float* __restrict__ gself,
                                                       const float* __restrict__ relu_src, int64_t E, int64_t rows, int D,
                                                       const unsigned* __restrict__ relu_bits = nullptr,
                                                       const int32_t* __restrict__ order = nullptr) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  // order (round 5): the edge ids sorted by destination (the CSR-by-destination list of the graph's plan).  Walking the edges in THAT
  // order, consecutive lane groups and trips gather the SAME gout / arg row of the [N, D] tables: at the C5 shape (1 GB tables, far
  // beyond every cache) a walk in edge-id order read 2 KB of table per edge from HBM -- 20 GB per launch next to the 10 GB it writes
  // (5.05 ms = 6 TB/s of real traffic, "0.30" of the peak against its 12.3 GB of algorithmic bytes); in destination order the table
  // rows come from L1 / L2 after their first touch.  Same values, rows written in another order.
  for (int64_t pos = (int64_t)blockIdx.x * RPB + rw; pos < rows; pos += (int64_t)gridDim.x * RPB) {
    const bool edge = pos < E;
    const int64_t r = (edge && order) ? (int64_t)order[pos] : pos;
    const int64_t v = edge ? dst[r] : r - E;
    float inv = 1.0f;
    if (MODE == MRG_REDUCE_MEAN && edge) { int d = in_degree[v]; inv = (float)(d > 1 ? d : 1); }
    float* o = edge ? gmsg + r * D : gself + (r - E) * D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) {
        Vec<VEC> g = Vec<VEC>::load(gout + v * D + c * VEC);
        if (edge) {
          if (MODE == MRG_REDUCE_MEAN) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] = g[j] / inv;
          } else if (MODE == MRG_REDUCE_MAX) {
            IVec<VEC> a = IVec<VEC>::load(arg + v * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] = (a[j] == (int)r) ? g[j] : 0.f;
          }
          if (relu_src) {            // the messages were ReLU outputs: mask the gradient where they are 0
            Vec<VEC> y = Vec<VEC>::load(relu_src + r * D + c * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] = y[j] > 0.f ? g[j] : 0.f;
          } else if (relu_bits) {    // ... kept as one bit per element by the fused forward (word = 32 columns of row r)
            const int bld = (D + 31) >> 5;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const int col = c * VEC + j;
              g[j] = ((relu_bits[r * bld + (col >> 5)] >> (col & 31)) & 1u) ? g[j] : 0.f;
            }
          }
        }
        g.store(o + c * VEC);
      }
    }
  }
}

}  // namespace mrg

using namespace mrg;

extern "C" int64_t mrg_seg_reduce_workspace_bytes(int64_t n_slots, int D) {
  if (n_slots < 0 || D <= 0) return 0;
  return (n_slots + 1) * (int64_t)D * 8 + 64;   // float values + int32 arg per partial slot
}

static int seg_reduce_fwd_impl(int mode, const float* msg, const float* self_rows, const int32_t* eid,
                               const int32_t* chunk_node, const int32_t* chunk_start, const int32_t* chunk_end,
                               const int32_t* chunk_slot, int64_t n_chunks, const int32_t* hub_node,
                               const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs, int64_t n_slots,
                               const int32_t* in_degree, float* out, int32_t* arg, void* ws, int64_t N, int D,
                               void* stream, const int32_t* head_rowptr);

extern "C" int mrg_seg_reduce_fwd(int mode, const float* msg, const float* self_rows, const int32_t* eid,
                                  const int32_t* chunk_node, const int32_t* chunk_start, const int32_t* chunk_end,
                                  const int32_t* chunk_slot, int64_t n_chunks, const int32_t* hub_node,
                                  const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs, int64_t n_slots,
                                  const int32_t* in_degree, float* out, int32_t* arg, void* ws, int64_t N, int D,
                                  void* stream) {
  return seg_reduce_fwd_impl(mode, msg, self_rows, eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, hub_node, hub_first,
                             hub_count, n_hubs, n_slots, in_degree, out, arg, ws, N, D, stream, nullptr);
}

// Second level of the fused a_mean / a_sum (first level: mrg_linear_relu_segsum_fwd): part is indexed by list position
// (row j of the by-destination edge list) and holds run sums at the head rows only; rowptr [N + 1] is the list start of
// every node.  mode: MRG_REDUCE_SUM or MRG_REDUCE_MEAN.
extern "C" int mrg_seg_reduce_heads_fwd(int mode, const float* part, const float* self_rows, const int32_t* rowptr,
                                        const int32_t* chunk_node, const int32_t* chunk_start, const int32_t* chunk_end,
                                        const int32_t* chunk_slot, int64_t n_chunks, const int32_t* hub_node,
                                        const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs, int64_t n_slots,
                                        const int32_t* in_degree, float* out, void* ws, int64_t N, int D, void* stream) {
  if (mode != MRG_REDUCE_SUM && mode != MRG_REDUCE_MEAN) return MRG_E_ENUM;
  if (!rowptr) return MRG_E_NULLPTR;
  return seg_reduce_fwd_impl(mode, part, self_rows, nullptr, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, hub_node, hub_first,
                             hub_count, n_hubs, n_slots, in_degree, out, nullptr, ws, N, D, stream, rowptr);
}

static int seg_reduce_fwd_impl(int mode, const float* msg, const float* self_rows, const int32_t* eid,
                               const int32_t* chunk_node, const int32_t* chunk_start, const int32_t* chunk_end,
                               const int32_t* chunk_slot, int64_t n_chunks, const int32_t* hub_node,
                               const int32_t* hub_first, const int32_t* hub_count, int64_t n_hubs, int64_t n_slots,
                               const int32_t* in_degree, float* out, int32_t* arg, void* ws, int64_t N, int D,
                               void* stream, const int32_t* head_rowptr) {
  if (mode < 0 || mode > 2) return MRG_E_ENUM;
  if (N < 0 || D <= 0 || n_chunks < N || n_hubs < 0 || n_slots < 0) return MRG_E_SHAPE;
  if (N == 0) return MRG_OK;
  if (!out || !chunk_node || !chunk_start || !chunk_end || !chunk_slot || !in_degree) return MRG_E_NULLPTR;
  if (mode == MRG_REDUCE_MAX && !arg) return MRG_E_NULLPTR;
  if (n_hubs > 0 && (!hub_node || !hub_first || !hub_count)) return MRG_E_NULLPTR;
  if (n_slots > 0 && !ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws_val = (float*)ws;
  int32_t* ws_arg = ws ? (int32_t*)((float*)ws + (n_slots + 1) * (int64_t)D) : nullptr;
  // the arg half starts at a multiple of D floats: 16-B aligned iff D % 4 == 0 (as the vector path requires)
  RowGeom g = row_geom(D, aligned16(msg) && aligned16(self_rows) && aligned16(out) && aligned16(arg) && aligned16(ws));
  if (!g.ok) return MRG_E_SHAPE;
  const int is_mean = mode == MRG_REDUCE_MEAN;
#define CALL(V, L, K)                                                                                                  \
  do {                                                                                                                 \
    int grid = grid_for(n_chunks, MRG_BLOCK / L);                                                                      \
    if (mode == MRG_REDUCE_MAX)                                                                                        \
      hipLaunchKernelGGL((seg_chunk_k<V, L, K, true>), dim3(grid), dim3(MRG_BLOCK), 0, st, msg, self_rows, eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, in_degree, out, arg, ws_val, ws_arg, D, is_mean); \
    else if (head_rowptr)                                                                                              \
      hipLaunchKernelGGL((seg_chunk_k<V, L, K, false, true>), dim3(grid), dim3(MRG_BLOCK), 0, st, msg, self_rows, eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, in_degree, out, arg, ws_val, ws_arg, D, is_mean, head_rowptr); \
    else                                                                                                               \
      hipLaunchKernelGGL((seg_chunk_k<V, L, K, false>), dim3(grid), dim3(MRG_BLOCK), 0, st, msg, self_rows, eid, chunk_node, chunk_start, chunk_end, chunk_slot, n_chunks, in_degree, out, arg, ws_val, ws_arg, D, is_mean); \
    if (n_hubs > 0) {                                                                                                  \
      int gh = n_hubs < 4096 ? (int)n_hubs : 4096;                                                                        \
      if (mode == MRG_REDUCE_MAX)                                                                                      \
        hipLaunchKernelGGL((seg_hub_k<V, L, K, true>), dim3(gh), dim3(MRG_BLOCK), 0, st, self_rows, hub_node, hub_first, hub_count, n_hubs, in_degree, out, arg, ws_val, ws_arg, D, is_mean); \
      else                                                                                                             \
        hipLaunchKernelGGL((seg_hub_k<V, L, K, false>), dim3(gh), dim3(MRG_BLOCK), 0, st, self_rows, hub_node, hub_first, hub_count, n_hubs, in_degree, out, arg, ws_val, ws_arg, D, is_mean); \
    }                                                                                                                  \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

static int seg_reduce_bwd_impl(int mode, const float* gout, const int32_t* dst, const int32_t* in_degree, const int32_t* arg, float* gmsg,
                               float* gself, const float* relu_src, const unsigned* relu_bits, int64_t E, int64_t N, int D, void* stream,
                               const int32_t* order = nullptr);

// both forms with the edges walked in destination order: order [E] = the edge ids sorted by destination (NULL: edge-id order)
extern "C" int mrg_seg_reduce_bwd_ordered(int mode, const float* gout, const int32_t* dst, const int32_t* in_degree, const int32_t* arg,
                                          float* gmsg, float* gself, const float* relu_src, const unsigned* relu_bits, const int32_t* order,
                                          int64_t E, int64_t N, int D, void* stream) {
  if (relu_src && relu_bits) return MRG_E_SHAPE;
  return seg_reduce_bwd_impl(mode, gout, dst, in_degree, arg, gmsg, gself, relu_src, relu_bits, E, N, D, stream, order);
}

extern "C" int mrg_seg_reduce_bwd(int mode, const float* gout, const int32_t* dst, const int32_t* in_degree,
                                  const int32_t* arg, float* gmsg, float* gself, const float* relu_src, int64_t E, int64_t N,
                                  int D, void* stream) {
  return seg_reduce_bwd_impl(mode, gout, dst, in_degree, arg, gmsg, gself, relu_src, nullptr, E, N, D, stream);
}

// the same with the ReLU mask as bits (relu_bits [E, ceil(D / 32)], written by mrg_linear_relu_segsum_fwd)
extern "C" int mrg_seg_reduce_bwd_bits(int mode, const float* gout, const int32_t* dst, const int32_t* in_degree, float* gmsg, float* gself,
                                       const unsigned* relu_bits, int64_t E, int64_t N, int D, void* stream) {
  if (mode != MRG_REDUCE_SUM && mode != MRG_REDUCE_MEAN) return MRG_E_ENUM;
  if (E > 0 && !relu_bits) return MRG_E_NULLPTR;
  return seg_reduce_bwd_impl(mode, gout, dst, in_degree, nullptr, gmsg, gself, nullptr, relu_bits, E, N, D, stream);
}

static int seg_reduce_bwd_impl(int mode, const float* gout, const int32_t* dst, const int32_t* in_degree, const int32_t* arg, float* gmsg,
                               float* gself, const float* relu_src, const unsigned* relu_bits, int64_t E, int64_t N, int D, void* stream,
                               const int32_t* order) {
  if (mode < 0 || mode > 2) return MRG_E_ENUM;
  if (E < 0 || N < 0 || D <= 0) return MRG_E_SHAPE;
  if (!gout || (E > 0 && (!dst || !gmsg))) return MRG_E_NULLPTR;
  if (mode == MRG_REDUCE_MEAN && !in_degree) return MRG_E_NULLPTR;
  if (mode == MRG_REDUCE_MAX && !arg) return MRG_E_NULLPTR;
  const int64_t rows = E + (gself ? N : 0);
  if (rows == 0) return MRG_OK;
  hipStream_t st = (hipStream_t)stream;
  RowGeom g = row_geom(D, aligned16(gout) && aligned16(gmsg) && aligned16(gself) && aligned16(arg) && aligned16(relu_src));
  if (!g.ok) return MRG_E_SHAPE;
#define CALL(V, L, K)                                                                                                  \
  do {                                                                                                                 \
    int grid = grid_for(rows, (MRG_BLOCK / L) * 4);                                                                           \
    if (mode == MRG_REDUCE_SUM) hipLaunchKernelGGL((seg_bwd_k<V, L, K, 0>), dim3(grid), dim3(MRG_BLOCK), 0, st, gout, dst, in_degree, arg, gmsg, gself, relu_src, E, rows, D, relu_bits, order); \
    else if (mode == MRG_REDUCE_MEAN) hipLaunchKernelGGL((seg_bwd_k<V, L, K, 1>), dim3(grid), dim3(MRG_BLOCK), 0, st, gout, dst, in_degree, arg, gmsg, gself, relu_src, E, rows, D, relu_bits, order); \
    else hipLaunchKernelGGL((seg_bwd_k<V, L, K, 2>), dim3(grid), dim3(MRG_BLOCK), 0, st, gout, dst, in_degree, arg, gmsg, gself, relu_src, E, rows, D, relu_bits, order); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
