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

// ---- a_max backward: the input gradient WITHOUT a dense product (round 5) ------------------------------------------------------------
// a_max (reference models/operations_lp.py:230-235) is  h[v] = max over the in-edges e of v of ReLU(W x_e + b).  Its gradient w.r.t.
// the messages, gmsg[e][c] = g[v][c] if e is the arg-max of (v, c) and the maximum is positive, has ONE non-zero per (node, column):
// N * D of E * D entries (FB15k-237: 2.7 %).  The input gradient gx = gmsg W was a dense [E, D] x [D, K] product over that matrix
// (0.33 ms per launch after a 0.19 ms seg_bwd_k wrote gmsg).  Here ONE pass per edge row forms gmsg[e] from the gathered g / arg /
// max rows (written for the weight gradient, which stays dense) and adds the rows W[c, :] of the columns the edge won -- on average
// N * D / E of them (5.3) -- from a copy of W in LDS (D * K * 4 bytes <= 160 KB: D = K = 200 fills the CU's LDS exactly; one
// 1024-thread workgroup per CU).  Order of the sum per output element: the won columns c ascending within j = c % 4, j = 0..3 --
// deterministic; exact f32 FMAs (the dense product ran on the split core at f32-equivalent accuracy).
template <bool WLDS, int THREADS>
__global__ __launch_bounds__(THREADS) void segmax_bwd_gx_k(const float* __restrict__ gout, const float* __restrict__ mx,
                                                         const int32_t* __restrict__ dst, const int32_t* __restrict__ arg,
                                                         const float* __restrict__ W, float* __restrict__ gmsg, float* __restrict__ gx,
                                                         const int32_t* __restrict__ order, int64_t E, int D, int Kin) {
  extern __shared__ __align__(16) float wl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int WAVES = THREADS / 64;
  if constexpr (WLDS) {
    const int n4 = D * Kin / 4;
    for (int i = tid; i < n4; i += THREADS) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(W)[i];
    __syncthreads();
  }
  const float* __restrict__ wsrc = WLDS ? wl : W;          // lab (MRG_SEGMAX_BWD_LDS=0): the rows of W from L2, four waves per block, no LDS
  const int dv = D >> 2, kv = Kin >> 2;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  // A wave's rows are a chain edge id -> destination -> three gathered table rows -> stores: ~2 us of dependent latency per edge with
  // only sixteen waves on the CU (the LDS is full).  Software pipeline: the table rows of edge i + 1 and the destination of edge
  // i + 2 are in flight while edge i is processed (positions past the end are clamped to the last edge and never processed).
  auto edge_at = [&](int64_t p) -> int { const int64_t q = p < E ? p : E - 1; return order ? order[q] : (int)q; };
  const bool lv = lane < dv;
  auto load_row = [&](int64_t v, float4& g4, int4& a4, float4& x4) {
    g4 = make_float4(0.f, 0.f, 0.f, 0.f); a4 = make_int4(-1, -1, -1, -1); x4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (lv) {
      g4 = *reinterpret_cast<const float4*>(gout + v * D + lane * 4);
      a4 = *reinterpret_cast<const int4*>(arg + v * D + lane * 4);
      if (mx) x4 = *reinterpret_cast<const float4*>(mx + v * D + lane * 4);
    }
  };
  const int64_t first = (int64_t)blockIdx.x * WAVES + wave;
  if (first >= E) return;                                   // (after the only barrier)
  int r0 = __builtin_amdgcn_readfirstlane(edge_at(first));
  int r1 = __builtin_amdgcn_readfirstlane(edge_at(first + stride));
  int64_t v1 = __builtin_amdgcn_readfirstlane(dst[r1]);
  float4 g4, x4; int4 a4;
  load_row((int64_t)__builtin_amdgcn_readfirstlane(dst[r0]), g4, a4, x4);
  for (int64_t pos = first; pos < E; pos += stride) {
    const int r2 = __builtin_amdgcn_readfirstlane(edge_at(pos + 2 * stride));
    const int v2 = dst[r2];                                 // in flight until the bottom of the trip
    float4 gn, xn; int4 an;
    load_row(v1, gn, an, xn);                               // edge i + 1's rows: in flight during edge i's work
    const int r = r0;
    float m[4];
    m[0] = (a4.x == r && x4.x > 0.f) ? g4.x : 0.f;
    m[1] = (a4.y == r && x4.y > 0.f) ? g4.y : 0.f;
    m[2] = (a4.z == r && x4.z > 0.f) ? g4.z : 0.f;
    m[3] = (a4.w == r && x4.w > 0.f) ? g4.w : 0.f;
    if (lv && gmsg) *reinterpret_cast<float4*>(gmsg + (int64_t)r * D + lane * 4) = make_float4(m[0], m[1], m[2], m[3]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned long long live = __ballot(m[j] != 0.f);
      while (live) {                                        // wave-uniform: one trip per column this edge won
        const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(live));
        live &= live - 1;
        const float val = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m[j]), src));
        const int c = src * 4 + j;
        if (lane < kv) {
          const float4 w4 = *reinterpret_cast<const float4*>(wsrc + c * Kin + lane * 4);
          acc.x += val * w4.x; acc.y += val * w4.y; acc.z += val * w4.z; acc.w += val * w4.w;
        }
      }
    }
    if (lane < kv) *reinterpret_cast<float4*>(gx + (int64_t)r * Kin + lane * 4) = acc;
    g4 = gn; a4 = an; x4 = xn;
    r0 = r1; r1 = r2;
    v1 = __builtin_amdgcn_readfirstlane(v2);
  }
}

}  // namespace mrg

using namespace mrg;

// 1 when mrg_segmax_bwd_input takes the shape: W [D, Kin] fits the LDS of a CU, rows are whole float4s of at most one wave
extern "C" int mrg_segmax_bwd_input_ok(int D, int Kin) {
  return (D > 0 && Kin > 0 && D % 4 == 0 && Kin % 4 == 0 && D <= 256 && Kin <= 256 && (size_t)D * Kin * 4 <= 160 * 1024) ? 1 : 0;
}

// gmsg [E, D] (may be NULL) and gx [E, Kin] = gmsg W of a_max's backward in one pass (see segmax_bwd_gx_k).  gout / mx / arg: [N, D]
// (mx NULL: no ReLU mask); W [D, Kin] row-major (nn.Linear.weight of the aggregator); order: NULL or the edge ids by destination.
extern "C" int mrg_segmax_bwd_input(const float* gout, const float* mx, const int32_t* dst, const int32_t* arg, const float* W, float* gmsg,
                                    float* gx, const int32_t* order, int64_t E, int64_t N, int D, int Kin, void* stream) {
  if (E < 0 || N < 0 || E >= ((int64_t)1 << 31)) return MRG_E_SHAPE;
  if (!mrg_segmax_bwd_input_ok(D, Kin)) return MRG_E_SHAPE;
  if (E == 0) return MRG_OK;
  if (!gout || !dst || !arg || !W || !gx) return MRG_E_NULLPTR;
  if (!aligned16(gout) || !aligned16(arg) || !aligned16(W) || !aligned16(gx) || !aligned16(gmsg) || !aligned16(mx)) return MRG_E_SHAPE;
  static const int use_lds = [] { const char* e = getenv("MRG_SEGMAX_BWD_LDS"); return e ? atoi(e) : 1; }();   // lab switch
  if (use_lds) {
    const size_t lds = (size_t)D * Kin * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&segmax_bwd_gx_k<true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int64_t grid = (E + 15) / 16;
    if (grid > 256) grid = 256;                            // one 1024-thread workgroup (the whole LDS) per CU
    hipLaunchKernelGGL((segmax_bwd_gx_k<true, 1024>), dim3((unsigned)grid), dim3(1024), lds, (hipStream_t)stream, gout, mx, dst, arg, W, gmsg, gx, order, E, D, Kin);
  } else {
    int64_t grid = (E + 3) / 4;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((segmax_bwd_gx_k<false, 256>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, gout, mx, dst, arg, W, gmsg, gx, order, E, D, Kin);
  }
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

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
