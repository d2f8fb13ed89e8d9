// a1: compose ops and the gather that feeds them (HBM-streaming kernels).
//   reference models/operations_lp.py:71-98 (pre_mult_op / pre_sub_op / pre_add_op)
//   reference models/model_lp.py:126-131, models/model_search_lp.py:135-145 (the gather)
// Algorithmic bytes: fwd 12*D per row; bwd sub/add 12*D, mult 20*D per row.
#include "common.hpp"

namespace mrg {

template <int OP> __device__ __forceinline__ float compose1(float a, float b) {
  return OP == MRG_COMPOSE_MULT ? a * b : (OP == MRG_COMPOSE_SUB ? a - b : a + b);
}

// flat elementwise, n = rows*D / VEC vectors
template <int VEC, int OP>
__global__ __launch_bounds__(MRG_BLOCK) void compose_fwd_k(const float* __restrict__ s, const float* __restrict__ hr,
                                                           float* __restrict__ out, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * MRG_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * MRG_BLOCK + threadIdx.x; i < n; i += stride) {
    Vec<VEC> a = Vec<VEC>::load(s + i * VEC), b = Vec<VEC>::load(hr + i * VEC), o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = compose1<OP>(a[j], b[j]);
    o.store(out + i * VEC);
  }
}

// Two vectors per thread and trip, every load of the trip issued before the first store (the one-vector form loaded s only after it
// had stored gs: three dependent memory round trips per trip, 3.2 TB/s at the C5 size).
template <int VEC, int OP>
__global__ __launch_bounds__(MRG_BLOCK) void compose_bwd_k(const float* __restrict__ g, const float* __restrict__ s,
                                                           const float* __restrict__ hr, float* __restrict__ gs,
                                                           float* __restrict__ ghr, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * MRG_BLOCK;
  const bool mult = OP == MRG_COMPOSE_MULT;
  for (int64_t i0 = (int64_t)blockIdx.x * MRG_BLOCK + threadIdx.x; i0 < n; i0 += 2 * stride) {
    const int64_t i1 = i0 + stride;
    const bool two = i1 < n;
    Vec<VEC> g0 = Vec<VEC>::load(g + i0 * VEC), g1 = two ? Vec<VEC>::load(g + i1 * VEC) : g0;
    Vec<VEC> b0 = g0, b1 = g0, a0 = g0, a1 = g0;
    if (mult && gs) { b0 = Vec<VEC>::load(hr + i0 * VEC); if (two) b1 = Vec<VEC>::load(hr + i1 * VEC); }
    if (mult && ghr) { a0 = Vec<VEC>::load(s + i0 * VEC); if (two) a1 = Vec<VEC>::load(s + i1 * VEC); }
    if (gs) {
      Vec<VEC> o0 = g0, o1 = g1;
      if (mult) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { o0[j] = g0[j] * b0[j]; o1[j] = g1[j] * b1[j]; }
      }
      o0.store(gs + i0 * VEC);
      if (two) o1.store(gs + i1 * VEC);
    }
    if (ghr) {
      Vec<VEC> o0, o1;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        o0[j] = mult ? g0[j] * a0[j] : (OP == MRG_COMPOSE_SUB ? -g0[j] : g0[j]);
        o1[j] = mult ? g1[j] * a1[j] : (OP == MRG_COMPOSE_SUB ? -g1[j] : g1[j]);
      }
      o0.store(ghr + i0 * VEC);
      if (two) o1.store(ghr + i1 * VEC);
    }
  }
}

// one row per LPR-lane group; OP == -1: plain gather of ent rows
template <int VEC, int LPR, int KMAX, int OP>
__global__ __launch_bounds__(MRG_BLOCK) void gather_compose_k(const float* __restrict__ ent, const float* __restrict__ rel,
                                                              const int32_t* __restrict__ ei, const int32_t* __restrict__ ri,
                                                              float* __restrict__ out, int64_t rows, int D) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float* a = ent + (int64_t)ei[r] * D;
    const float* b = OP >= 0 ? rel + (int64_t)ri[r] * D : nullptr;
    float* o = out + r * D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int c = sl + k * LPR;
      if (c < dv) {
        Vec<VEC> x = Vec<VEC>::load(a + c * VEC);
        if (OP >= 0) {
          Vec<VEC> y = Vec<VEC>::load(b + c * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) x[j] = compose1<(OP < 0 ? 0 : OP)>(x[j], y[j]);
        }
        x.store(o + c * VEC);
      }
    }
  }
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_compose_fwd(int op, const float* s, const float* hr, float* out, int64_t rows, int D, void* stream) {
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (op < 0 || op > 2) return MRG_E_ENUM;
  int64_t n = rows * D;
  if (n == 0) return MRG_OK;
  if (!s || !hr || !out) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  bool v4 = (n % 4 == 0) && aligned16(s) && aligned16(hr) && aligned16(out);
  int64_t nv = v4 ? n / 4 : n;
  int grid = stream_grid_for(nv, MRG_BLOCK * 4);
#define GO(V, O) hipLaunchKernelGGL((compose_fwd_k<V, O>), dim3(grid), dim3(MRG_BLOCK), 0, st, s, hr, out, nv)
  if (v4) { if (op == 0) GO(4, 0); else if (op == 1) GO(4, 1); else GO(4, 2); }
  else    { if (op == 0) GO(1, 0); else if (op == 1) GO(1, 1); else GO(1, 2); }
#undef GO
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_compose_bwd(int op, const float* gout, const float* s, const float* hr, float* gs, float* ghr,
                               int64_t rows, int D, void* stream) {
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (op < 0 || op > 2) return MRG_E_ENUM;
  int64_t n = rows * D;
  if (n == 0 || (!gs && !ghr)) return MRG_OK;
  if (!gout) return MRG_E_NULLPTR;
  if (op == MRG_COMPOSE_MULT && ((gs && !hr) || (ghr && !s))) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  bool v4 = (n % 4 == 0) && aligned16(gout) && aligned16(s) && aligned16(hr) && aligned16(gs) && aligned16(ghr);
  int64_t nv = v4 ? n / 4 : n;
  int grid = stream_grid_for(nv, MRG_BLOCK * 4);
#define GO(V, O) hipLaunchKernelGGL((compose_bwd_k<V, O>), dim3(grid), dim3(MRG_BLOCK), 0, st, gout, s, hr, gs, ghr, nv)
  if (v4) { if (op == 0) GO(4, 0); else if (op == 1) GO(4, 1); else GO(4, 2); }
  else    { if (op == 0) GO(1, 0); else if (op == 1) GO(1, 1); else GO(1, 2); }
#undef GO
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_gather_compose_fwd(int op, const float* ent, const float* rel, const int32_t* ent_idx,
                                      const int32_t* rel_idx, float* out, int64_t rows, int D, void* stream) {
  if (op < -1 || op > 2) return MRG_E_ENUM;
  if (rows < 0 || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!ent || !ent_idx || !out) return MRG_E_NULLPTR;
  if (op >= 0 && (!rel || !rel_idx)) return MRG_E_NULLPTR;
  RowGeom g = row_geom(D, aligned16(ent) && aligned16(out) && (op < 0 || aligned16(rel)));
  if (!g.ok) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define CALL(V, L, K)                                                                                         \
  do {                                                                                                        \
    int grid = grid_for(rows, (MRG_BLOCK / L) * 4);                                                                 \
    if (op == -1) hipLaunchKernelGGL((gather_compose_k<V, L, K, -1>), dim3(grid), dim3(MRG_BLOCK), 0, st, ent, rel, ent_idx, rel_idx, out, rows, D); \
    else if (op == 0) hipLaunchKernelGGL((gather_compose_k<V, L, K, 0>), dim3(grid), dim3(MRG_BLOCK), 0, st, ent, rel, ent_idx, rel_idx, out, rows, D); \
    else if (op == 1) hipLaunchKernelGGL((gather_compose_k<V, L, K, 1>), dim3(grid), dim3(MRG_BLOCK), 0, st, ent, rel, ent_idx, rel_idx, out, rows, D); \
    else hipLaunchKernelGGL((gather_compose_k<V, L, K, 2>), dim3(grid), dim3(MRG_BLOCK), 0, st, ent, rel, ent_idx, rel_idx, out, rows, D); \
  } while (0)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
