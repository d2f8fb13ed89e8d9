// f3: the [B, N] score function that is not a GEMM.
//   sf_TransE_op.forward   reference models/operations_lp.py:101-112
//       obj = sub + rel;  score[b, n] = sigmoid(gamma - sum_c |obj[b, c] - ent[n, c]|)      (torch.norm(p=1) / cdist)
// (sf_DisMult_op, :115-127, is sigmoid((sub * rel) ent^T): the compose kernel + the MFMA row GEMM with a sigmoid epilogue.)
// An L1 distance has no matrix form, so this is VALU work: B*N*D absolute differences (744 M at B = 256, FB15k-237,
// D = 200).  Tiles of 64 entity rows x 8 query rows are staged through LDS in 64-column slices (coalesced global reads,
// conflict-free LDS reads: the entity tile is padded to 65 floats per row, the query tile is read as a broadcast).
#include <hip/hip_runtime.h>
#include "common.hpp"

namespace mrg {

constexpr int TE_N = 64, TE_B = 8, TE_C = 64;

// dist[b, n] = sum_c |obj[b, c] - ent[n, c]|;  score = sigmoid(gamma - dist)
__global__ __launch_bounds__(256) void transe_fwd_k(const float* __restrict__ ent, const float* __restrict__ sub, const float* __restrict__ rel,
                                                    float gamma, float* __restrict__ score, int64_t B, int64_t N, int D) {
  __shared__ float se[TE_N][TE_C + 1];
  __shared__ float so[TE_B][TE_C];
  const int64_t n0 = (int64_t)blockIdx.x * TE_N, b0 = (int64_t)blockIdx.y * TE_B;
  const int tid = threadIdx.x;
  const int nl = tid & 63, bl = tid >> 6;               // thread owns (n0 + nl, b0 + bl) and (n0 + nl, b0 + bl + 4)
  float acc0 = 0.f, acc1 = 0.f;
  for (int c0 = 0; c0 < D; c0 += TE_C) {
    for (int i = tid; i < TE_N * TE_C; i += 256) {
      const int r = i >> 6, c = i & 63;
      se[r][c] = (n0 + r < N && c0 + c < D) ? ent[(n0 + r) * D + c0 + c] : 0.f;
    }
    for (int i = tid; i < TE_B * TE_C; i += 256) {
      const int r = i >> 6, c = i & 63;
      so[r][c] = (b0 + r < B && c0 + c < D) ? sub[(b0 + r) * D + c0 + c] + rel[(b0 + r) * D + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < TE_C; ++c) {
      const float e = se[nl][c];
      acc0 += fabsf(so[bl][c] - e);
      acc1 += fabsf(so[bl + 4][c] - e);
    }
    __syncthreads();
  }
  if (n0 + nl < N) {
    if (b0 + bl < B) score[(b0 + bl) * N + n0 + nl] = sigmoidf_fast(gamma - acc0);
    if (b0 + bl + 4 < B) score[(b0 + bl + 4) * N + n0 + nl] = sigmoidf_fast(gamma - acc1);
  }
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

// dd[b, n] = d loss / d dist[b, n] = -g[b, n] * s (1 - s)
// gobj[b, c] = sum_n dd[b, n] * sign(obj[b, c] - ent[n, c])            (returned for both sub and rel)
// block: 8 query rows x 64 columns; 4 n-lanes per column; every n is visited by exactly one lane, lanes combined in order
__global__ __launch_bounds__(256) void transe_bwd_obj_k(const float* __restrict__ ent, const float* __restrict__ sub, const float* __restrict__ rel,
                                                        const float* __restrict__ g, const float* __restrict__ score, float* __restrict__ gobj,
                                                        int64_t B, int64_t N, int D) {
  __shared__ float sdd[TE_B][256];
  __shared__ float red[4][TE_B][TE_C];
  const int64_t b0 = (int64_t)blockIdx.y * TE_B;
  const int c = blockIdx.x * TE_C + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float o[TE_B], acc[TE_B];
#pragma unroll
  for (int b = 0; b < TE_B; ++b) {
    o[b] = (b0 + b < B && c < D) ? sub[(b0 + b) * D + c] + rel[(b0 + b) * D + c] : 0.f;
    acc[b] = 0.f;
  }
  for (int64_t n0 = 0; n0 < N; n0 += 256) {
    for (int i = threadIdx.x; i < TE_B * 256; i += 256) {
      const int b = i >> 8, n = i & 255;
      float v = 0.f;
      if (b0 + b < B && n0 + n < N) { const float s = score[(b0 + b) * N + n0 + n]; v = -g[(b0 + b) * N + n0 + n] * s * (1.f - s); }
      sdd[b][n] = v;
    }
    __syncthreads();
    for (int n = q; n < 256 && n0 + n < N; n += 4) {
      const float e = c < D ? ent[(n0 + n) * D + c] : 0.f;
#pragma unroll
      for (int b = 0; b < TE_B; ++b) acc[b] += sdd[b][n] * sgn(o[b] - e);
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < TE_B; ++b) red[q][b][threadIdx.x & 63] = acc[b];
  __syncthreads();
  if (q == 0 && c < D) {
#pragma unroll
    for (int b = 0; b < TE_B; ++b)
      if (b0 + b < B) gobj[(b0 + b) * D + c] = ((red[0][b][threadIdx.x] + red[1][b][threadIdx.x]) + red[2][b][threadIdx.x]) + red[3][b][threadIdx.x];
  }
}

// gent[n, c] = -sum_b dd[b, n] * sign(obj[b, c] - ent[n, c]);   block: 4 entity rows x 64 columns, loop over b in order
__global__ __launch_bounds__(256) void transe_bwd_ent_k(const float* __restrict__ ent, const float* __restrict__ sub, const float* __restrict__ rel,
                                                        const float* __restrict__ g, const float* __restrict__ score, float* __restrict__ gent,
                                                        int64_t B, int64_t N, int D) {
  const int64_t n = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.x * TE_C + (threadIdx.x & 63);
  if (n >= N || c >= D) return;
  const float e = ent[n * D + c];
  float acc = 0.f;
  for (int64_t b = 0; b < B; ++b) {
    const float s = score[b * N + n];
    const float dd = -g[b * N + n] * s * (1.f - s);
    acc -= dd * sgn(sub[b * D + c] + rel[b * D + c] - e);
  }
  gent[n * D + c] = acc;
}


// ---- the [B, N] scorer's output gradient, activation folded in, TRANSPOSED ------------------------------------------------------
// gT[n][b] = g[b][n] * act'(y[b][n])   (act = sigmoid: y (1 - y); ReLU: [y > 0]; none: 1)
// Both gradients of a wide, short Linear (functional._Linear.backward: the DistMult [B, N] scorer, N = all entities) stream along the
// N entity rows and read the score gradient as [N, B] rows.  torch formed it as mul, rsub, mul and a strided copy: four launches, eleven
// [B, N] passes (3.6 ms at B = 256, N = 1 M); here one: g and y read once in 64 x 64 tiles (256-byte row pieces), transposed through
// LDS (pitch 65: conflict free both ways), written once.
constexpr int GT_TILE = 64;
__global__ __launch_bounds__(256) void act_grad_transpose_k(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ gT,
                                                            int64_t B, int64_t N, int act) {
  __shared__ float tile[GT_TILE][GT_TILE + 1];
  const int64_t n0 = (int64_t)blockIdx.x * GT_TILE, b0 = (int64_t)blockIdx.y * GT_TILE;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;             // 4 rows of 64 per pass
#pragma unroll 4
  for (int r = ty; r < GT_TILE; r += 4) {
    const int64_t b = b0 + r, n = n0 + tx;
    float v = 0.f;
    if (b < B && n < N) {
      v = g[b * N + n];
      if (act != MRG_ACT_NONE) {
        const float s = y[b * N + n];
        v = act == MRG_ACT_SIGMOID ? v * s * (1.f - s) : (s > 0.f ? v : 0.f);
      }
    }
    tile[r][tx] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < GT_TILE; r += 4) {
    const int64_t n = n0 + r, b = b0 + tx;
    if (n < N && b < B) gT[n * B + b] = tile[tx][r];
  }
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_transe_score_fwd(const float* ent, const float* sub, const float* rel, float gamma, float* score, int64_t B, int64_t N, int D,
                                    void* stream) {
  if (B < 0 || N < 0 || D <= 0) return MRG_E_SHAPE;
  if (B == 0 || N == 0) return MRG_OK;
  if (!ent || !sub || !rel || !score) return MRG_E_NULLPTR;
  dim3 grid((unsigned)((N + TE_N - 1) / TE_N), (unsigned)((B + TE_B - 1) / TE_B));
  hipLaunchKernelGGL(transe_fwd_k, grid, dim3(256), 0, (hipStream_t)stream, ent, sub, rel, gamma, score, B, N, D);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_transe_score_bwd(const float* ent, const float* sub, const float* rel, const float* gscore, const float* score, float* gent,
                                    float* gobj, int64_t B, int64_t N, int D, void* stream) {
  if (B < 0 || N < 0 || D <= 0) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0 || N == 0) {                         // an empty batch / entity table: the gradients that do exist are zero
    if (gent && N > 0 && hipMemsetAsync(gent, 0, (size_t)N * D * 4, st) != hipSuccess) return MRG_E_WORKSPACE;
    if (gobj && B > 0 && hipMemsetAsync(gobj, 0, (size_t)B * D * 4, st) != hipSuccess) return MRG_E_WORKSPACE;
    return MRG_OK;
  }
  if (!ent || !sub || !rel || !gscore || !score) return MRG_E_NULLPTR;
  if (gobj) {
    dim3 grid((unsigned)((D + TE_C - 1) / TE_C), (unsigned)((B + TE_B - 1) / TE_B));
    hipLaunchKernelGGL(transe_bwd_obj_k, grid, dim3(256), 0, st, ent, sub, rel, gscore, score, gobj, B, N, D);
  }
  if (gent) {
    dim3 grid((unsigned)((D + TE_C - 1) / TE_C), (unsigned)((N + 3) / 4));
    hipLaunchKernelGGL(transe_bwd_ent_k, grid, dim3(256), 0, st, ent, sub, rel, gscore, score, gent, B, N, D);
  }
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_act_grad_transpose(const float* g, const float* y, float* gT, int64_t B, int64_t N, int act, void* stream) {
  if (B < 0 || N < 0) return MRG_E_SHAPE;
  if (act != MRG_ACT_NONE && act != MRG_ACT_RELU && act != MRG_ACT_SIGMOID) return MRG_E_ENUM;
  if (B == 0 || N == 0) return MRG_OK;
  if (!g || !gT || (act != MRG_ACT_NONE && !y)) return MRG_E_NULLPTR;
  const int64_t gx = (N + GT_TILE - 1) / GT_TILE, gy = (B + GT_TILE - 1) / GT_TILE;
  if (gx > 0x7fffffffLL || gy > 65535) return MRG_E_SHAPE;
  hipLaunchKernelGGL(act_grad_transpose_k, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, g, y, gT, B, N, act);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
