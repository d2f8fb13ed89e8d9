// K-way sum of equally-shaped buffers: the gradient fan-in of a tensor read by several candidate
// operators (MixedOp hands one state to up to eleven operators, reference models/cell_lp.py:25-33;
// a cell hands h_in to every MixedOp, models/cell_lp.py:150-186).  Autograd's pairwise adds move
// 3 S bytes per extra consumer; one K-way pass moves (K + 1) S in total.  HBM-bound.
#include "common.hpp"

#define MRG_SUM_MAXK 8

namespace mrg {

struct SumPack { const float* p[MRG_SUM_MAXK]; };

template <int K, bool ACC>
__global__ __launch_bounds__(MRG_BLOCK) void sum_k(SumPack xs, float* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * MRG_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * MRG_BLOCK) {
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = reinterpret_cast<const float4*>(xs.p[k])[i];
    float4 a = ACC ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

template <bool ACC>
__global__ void sum_tail_k(SumPack xs, int K, float* __restrict__ out, int64_t first, int64_t n) {
  int64_t i = first + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = ACC ? out[i] : 0.f;
  for (int k = 0; k < K; ++k) a += xs.p[k][i];
  out[i] = a;
}

template <bool ACC>
static void launch_sum(const SumPack& xs, int K, float* out, int64_t n, bool vec, hipStream_t st) {
  int64_t n4 = vec ? n / 4 : 0;
  if (n4 > 0) {
    int grid = stream_grid_for(n4, MRG_BLOCK * 2);
    switch (K) {
#define CASE(KK) case KK: hipLaunchKernelGGL((sum_k<KK, ACC>), dim3(grid), dim3(MRG_BLOCK), 0, st, xs, out, n4); break;
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
  }
  int64_t first = n4 * 4;
  if (first < n) {
    int64_t rem = n - first;
    hipLaunchKernelGGL((sum_tail_k<ACC>), dim3((unsigned)((rem + 255) / 256)), dim3(256), 0, st, xs, K, out, first, n);
  }
}

// out[r] = sum_k xs[k][r] + (r < E ? gather_edge[dst[r]] : gather_self[r - E])      rows of D floats
// The gradient fan-in of a state one of whose readers is a_sum (reference models/operations_lp.py:252-264: h = sum over the in-edges
// of the message rows + the self row): that reader's gradient w.r.t. the [M, D] state is a GATHER of the [N, D] node gradient
// (edge row e receives row dst[e], self row n receives row n), so it is read here from the cache-resident [N, D] tensor instead of
// being written as an [M, D] tensor by a backward kernel and read back by the sum.
template <int VEC, int LPR, int KMAX>
__global__ __launch_bounds__(MRG_BLOCK) void sum_rows_gather_k(SumPack xs, int K, const float* __restrict__ gather_edge,
                                                               const float* __restrict__ gather_self, const int32_t* __restrict__ dst,
                                                               int64_t E, int64_t rows, int D, float* __restrict__ out) {
  constexpr int RPB = MRG_BLOCK / LPR;
  const int sl = threadIdx.x % LPR, rw = row_group_of_thread<LPR>();
  const int dv = D / VEC;
  for (int64_t r = (int64_t)blockIdx.x * RPB + rw; r < rows; r += (int64_t)gridDim.x * RPB) {
    const float* gb = r < E ? gather_edge + (int64_t)dst[r] * D : (gather_self ? gather_self + (r - E) * D : nullptr);
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
      const int c = sl + q * LPR;
      if (c < dv) {
        Vec<VEC> v[MRG_SUM_MAXK];
#pragma unroll
        for (int k = 0; k < MRG_SUM_MAXK; ++k) {
          v[k] = Vec<VEC>::fill(0.f);
          if (k < K) v[k] = Vec<VEC>::load(xs.p[k] + r * D + c * VEC);
        }
        Vec<VEC> a = gb ? Vec<VEC>::load(gb + c * VEC) : Vec<VEC>::fill(0.f);
#pragma unroll
        for (int k = 0; k < MRG_SUM_MAXK; ++k)
          if (k < K) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] += v[k][j];
          }
        a.store(out + r * D + c * VEC);
      }
    }
  }
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_sum_rows_gather(const float* const* xs_host, int K, const float* gather_edge, const float* gather_self, const int32_t* dst,
                                   int64_t E, int64_t rows, int D, float* out, void* stream) {
  if (K < 0 || K > MRG_SUM_MAXK || E < 0 || rows < E || D <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!out || (K > 0 && !xs_host) || (E > 0 && (!gather_edge || !dst))) return MRG_E_NULLPTR;
  SumPack xs{};
  bool al = aligned16(out) && aligned16(gather_edge) && aligned16(gather_self);
  for (int k = 0; k < K; ++k) {
    if (!xs_host[k]) return MRG_E_NULLPTR;
    xs.p[k] = xs_host[k];
    al = al && aligned16(xs_host[k]);
  }
  RowGeom g = row_geom(D, al);
  if (!g.ok) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define CALL(V, L, KM)                                                                                                  \
  hipLaunchKernelGGL((sum_rows_gather_k<V, L, KM>), dim3(grid_for(rows, (MRG_BLOCK / L) * 4)), dim3(MRG_BLOCK), 0, st, xs, K, gather_edge, \
                     gather_self, dst, E, rows, D, out)
  MRG_DISPATCH_GEOM(g, CALL);
#undef CALL
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// out[i] = (accumulate ? out[i] : 0) + sum_k xs_host[k][i],  i < n;  1 <= K <= 8 device buffers whose
// pointers are given in a HOST array.  Summation order is k = 0..K-1 (deterministic).
extern "C" int mrg_sum_buffers(const float* const* xs_host, int K, float* out, int64_t n, int accumulate, void* stream) {
  if (K < 1 || K > MRG_SUM_MAXK || n < 0) return MRG_E_SHAPE;
  if (n == 0) return MRG_OK;
  if (!xs_host || !out) return MRG_E_NULLPTR;
  SumPack xs{};
  bool vec = aligned16(out);
  for (int k = 0; k < K; ++k) {
    if (!xs_host[k]) return MRG_E_NULLPTR;
    xs.p[k] = xs_host[k];
    vec = vec && aligned16(xs_host[k]);
  }
  hipStream_t st = (hipStream_t)stream;
  if (accumulate) launch_sum<true>(xs, K, out, n, vec, st);
  else launch_sum<false>(xs, K, out, n, vec, st);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
