// Dense linears on edge / node rows with exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//   nn.Linear inside a_max_op / a_mean_op: reference models/operations_lp.py:228,231,243,246
//   W_O / W_I / W_S / W_R of CompGraphConv:  reference models/compgcn.py:36-41,77-78,100,103
// These are tall-skinny GEMMs (rows ~ 5e5, K and Nout ~ 2e2): 2*rows*K*Nout flop against
// 4*rows*(K+Nout) bytes is ~100 flop/B at D = 200, above the f32-MFMA ridge (157 TF/s over
// 8 TB/s ~ 20 flop/B), so the roofline here is the matrix pipe, not HBM.
//
// Tiling (wave = 64 lanes): a workgroup of 4 waves owns 128 rows x (NT*32) columns; wave w
// owns rows [32w, 32w+32) and all NT column tiles, so X is read from HBM exactly once.
// LDS tiles are k-contiguous with a 4-float pad (stride 36): the ds_read_b128 fragment reads
// (4 consecutive k per lane -> 4 MFMAs) and the staging ds_write_b128 are bank-conflict-free.
#include "common.hpp"

namespace mrg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LBM = 128;   // rows per workgroup
constexpr int LBK = 32;    // k per LDS tile
constexpr int LLD = LBK + 4;

__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ base, int64_t row, int64_t nrows, int k, int K, int64_t ld,
                                            bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows) {
    const float* p = base + row * ld + k;
    if (vec_ok && k + 3 < K) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

// C[rows, N] = act(A[rows, K] * B^T + bias),  B given as [N, K] (BT = false, nn.Linear weight)
// or as [K, N] (BT = true: the same weight used for the input gradient).
template <int NT, bool BT>
__global__ __launch_bounds__(MRG_BLOCK) void linear_k(const float* __restrict__ A, const float* __restrict__ B,
                                                      const float* __restrict__ bias, float* __restrict__ C, int64_t rows,
                                                      int K, int N, int act, int vecA, int vecB) {
  __shared__ float As[LBM * LLD];
  __shared__ float Bs[NT * 32 * LLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * LBM;
  const int col0 = blockIdx.y * (NT * 32);
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += LBK) {
    // ---- stage A: 128 x 32 floats = 1024 float4, 4 per thread
#pragma unroll
    for (int i = 0; i < (LBM * LBK / 4) / MRG_BLOCK; ++i) {
      int idx = tid + i * MRG_BLOCK;
      int r = idx >> 3, k4 = idx & 7;
      float4 v = ld4_guard(A, row0 + r, rows, k0 + k4 * 4, K, K, vecA != 0);
      *reinterpret_cast<float4*>(&As[r * LLD + k4 * 4]) = v;
    }
    // ---- stage B
    if (!BT) {
#pragma unroll
      for (int i = 0; i < (NT * 32 * LBK / 4 + MRG_BLOCK - 1) / MRG_BLOCK; ++i) {
        int idx = tid + i * MRG_BLOCK;
        if (idx < NT * 32 * LBK / 4) {
          int j = idx >> 3, k4 = idx & 7;
          float4 v = ld4_guard(B, col0 + j, N, k0 + k4 * 4, K, K, vecB != 0);
          *reinterpret_cast<float4*>(&Bs[j * LLD + k4 * 4]) = v;
        }
      }
    } else {
      // B[k][n] row-major: lanes run along n (coalesced), scatter into the k-contiguous tile
      for (int e = tid; e < NT * 32 * LBK; e += MRG_BLOCK) {
        int j = e % (NT * 32), kk = e / (NT * 32);
        float v = 0.f;
        if (k0 + kk < K && col0 + j < N) v = B[(int64_t)(k0 + kk) * N + col0 + j];
        Bs[j * LLD + kk] = v;
      }
    }
    __syncthreads();
    const int kt = K - k0 < LBK ? K - k0 : LBK;
    const int nt8 = (kt + 7) >> 3;
    for (int t = 0; t < nt8; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(&As[(wave * 32 + li) * LLD + t * 8 + lh * 4]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float4 b = *reinterpret_cast<const float4*>(&Bs[(n * 32 + li) * LLD + t * 8 + lh * 4]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 32 + li;
    if (col < N) {
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < rows) {
          float v = acc[n][r] + bv;
          if (act == MRG_ACT_RELU) v = v > 0.f ? v : 0.f;
          C[row * N + col] = v;
        }
      }
    }
  }
}

// Weight gradient: partial[g][n][c] = sum over the block's rows r of gY[r][n] * X'[r][c],
// X' = [X | 1] (the extra column of ones yields the bias gradient for free).
constexpr int WBR = 32;    // rows per LDS tile

template <int TPW>
__global__ __launch_bounds__(MRG_BLOCK) void linear_wgrad_k(const float* __restrict__ gY, const float* __restrict__ X,
                                                            float* __restrict__ ws, int64_t rows, int K, int Nout, int TM,
                                                            int TN, int TNB, int64_t rows_per_block, int vecG, int vecX) {
  extern __shared__ float smem[];
  const int tn0 = blockIdx.y * TNB;                       // first X' column tile of this workgroup
  const int tnb = TN - tn0 < TNB ? TN - tn0 : TNB;        // column tiles it owns
  const int ldg = TM * 32, ldx = TNB * 32;
  float* Gs = smem;                 // [WBR][ldg]
  float* Xs = smem + WBR * ldg;     // [WBR][ldx]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ntiles = TM * tnb;
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += WBR) {
    // stage gY tile (zero beyond Nout / beyond r_end)
    for (int idx = tid; idx < WBR * (ldg / 4); idx += MRG_BLOCK) {
      int r = idx / (ldg / 4), c4 = idx % (ldg / 4);
      float4 v = ld4_guard(gY, r0 + r, r_end, c4 * 4, Nout, Nout, vecG != 0);
      *reinterpret_cast<float4*>(&Gs[r * ldg + c4 * 4]) = v;
    }
    // stage the X' tile: columns [tn0*32, tn0*32 + ldx) of [X | 1]
    for (int idx = tid; idx < WBR * (ldx / 4); idx += MRG_BLOCK) {
      int r = idx / (ldx / 4), c4 = idx % (ldx / 4);
      int c = tn0 * 32 + c4 * 4;
      float4 v = ld4_guard(X, r0 + r, r_end, c, K, K, vecX != 0);
      if (r0 + r < r_end) {
        if (c == K) v.x = 1.f; else if (c + 1 == K) v.y = 1.f; else if (c + 2 == K) v.z = 1.f; else if (c + 3 == K) v.w = 1.f;
      }
      *reinterpret_cast<float4*>(&Xs[r * ldx + c4 * 4]) = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int t = 0; t < WBR / 2; ++t) {
      const float* grow = Gs + (2 * t + lh) * ldg + li;
      const float* xrow = Xs + (2 * t + lh) * ldx + li;
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int id = wave + 4 * i;
        if (id < ntiles) {
          const int m = id / tnb, n = id - m * tnb;
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[m * 32], xrow[n * 32], acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  const int ldw = TN * 32;
  float* out = ws + (int64_t)blockIdx.x * ldg * ldw;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int id = wave + 4 * i;
    if (id < ntiles) {
      const int m = id / tnb, n = id - m * tnb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(int64_t)row * ldw + (tn0 + n) * 32 + li] = acc[i][r];
      }
    }
  }
}

// gW[n][c] = sum_g ws[g][n][c] (c < K);  gbias[n] = sum_g ws[g][n][K]   -- fixed order:
// thread row ty sums the partial tiles g = ty, ty+16, ..., the 16 row sums are added in order.
__global__ void linear_wgrad_reduce_k(const float* __restrict__ ws, float* __restrict__ gW, float* __restrict__ gbias,
                                      int G, int K, int Nout, int ldg, int ldx) {
  __shared__ float part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int n = blockIdx.y;
  float acc = 0.f;
  if (c <= K)
    for (int g = ty; g < G; g += 16) acc += ws[((int64_t)g * ldg + n) * ldx + c];
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c <= K) {
    float tot = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += part[i][tx];
    if (c < K) gW[(int64_t)n * K + c] = tot;
    else if (gbias) gbias[n] = tot;
  }
}

static int pick_nt(int ncols) {
  int t = (ncols + 31) / 32;
  if (t <= 1) return 1;
  if (t <= 2) return 2;
  if (t <= 4) return 4;
  if (t <= 7) return 7;
  return 8;
}

template <bool BT>
static int launch_linear(const float* A, const float* B, const float* bias, float* C, int64_t rows, int K, int N, int act,
                         hipStream_t st) {
  const int nt = pick_nt(N);
  dim3 grid((unsigned)((rows + LBM - 1) / LBM), (unsigned)((N + nt * 32 - 1) / (nt * 32)));
  const int vecA = (K % 4 == 0) && aligned16(A);
  const int vecB = (K % 4 == 0) && aligned16(B);
#define GO(NTV) hipLaunchKernelGGL((linear_k<NTV, BT>), grid, dim3(MRG_BLOCK), 0, st, A, B, bias, C, rows, K, N, act, vecA, vecB)
  switch (nt) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 7: GO(7); break;
    default: GO(8); break;
  }
#undef GO
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

struct WgradPlan {
  int TM, TN, TNB, tpw, G;
  int64_t rows_per_block;
  size_t lds;
  bool ok;
};

static WgradPlan wgrad_plan(int64_t rows, int K, int Nout) {
  WgradPlan p{};
  p.TM = (Nout + 31) / 32;
  p.TN = (K + 1 + 31) / 32;
  // at most 52 accumulator tiles (13 per wave) per workgroup; wider outputs split X' columns over grid.y
  p.TNB = p.TM * p.TN <= 52 ? p.TN : (52 / p.TM > 0 ? 52 / p.TM : 0);
  int per_wave = p.TNB > 0 ? (p.TM * p.TNB + 3) / 4 : 99;
  const int opts[] = {1, 2, 4, 7, 13};
  p.tpw = 0;
  for (int o : opts) if (per_wave <= o) { p.tpw = o; break; }
  p.lds = (size_t)WBR * (p.TM + p.TNB) * 32 * sizeof(float);
  p.ok = p.tpw > 0 && p.lds <= 160 * 1024;
  int64_t tiles = (rows + WBR - 1) / WBR;
  int64_t G = tiles < 512 ? tiles : 512;
  if (G < 1) G = 1;
  int64_t tpb = (tiles + G - 1) / G;
  if (tpb < 1) tpb = 1;
  p.rows_per_block = tpb * WBR;
  p.G = (int)((rows + p.rows_per_block - 1) / p.rows_per_block);
  if (p.G < 1) p.G = 1;
  return p;
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_linear_fwd(const float* X, const float* W, const float* bias, float* Y, int64_t rows, int K, int Nout,
                              int act, void* stream) {
  if (rows < 0 || K <= 0 || Nout <= 0) return MRG_E_SHAPE;
  if (act != MRG_ACT_NONE && act != MRG_ACT_RELU) return MRG_E_ENUM;
  if (rows == 0) return MRG_OK;
  if (!X || !W || !Y) return MRG_E_NULLPTR;
  return launch_linear<false>(X, W, bias, Y, rows, K, Nout, act, (hipStream_t)stream);
}

extern "C" int mrg_linear_bwd_input(const float* gY, const float* W, float* gX, int64_t rows, int K, int Nout, void* stream) {
  if (rows < 0 || K <= 0 || Nout <= 0) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!gY || !W || !gX) return MRG_E_NULLPTR;
  // gX[rows, K] = gY[rows, Nout] * W[Nout, K]: reduction over Nout, W read as [k = Nout][n = K]
  return launch_linear<true>(gY, W, nullptr, gX, rows, Nout, K, MRG_ACT_NONE, (hipStream_t)stream);
}

extern "C" int64_t mrg_linear_bwd_weight_workspace_bytes(int64_t rows, int K, int Nout) {
  if (rows < 0 || K <= 0 || Nout <= 0) return 0;
  WgradPlan p = wgrad_plan(rows, K, Nout);
  return (int64_t)p.G * p.TM * 32 * p.TN * 32 * sizeof(float);
}

extern "C" int mrg_linear_bwd_weight(const float* gY, const float* X, float* gW, float* gbias, void* ws, int64_t rows, int K,
                                     int Nout, void* stream) {
  if (rows < 0 || K <= 0 || Nout <= 0) return MRG_E_SHAPE;
  if (!gW) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {                       // no rows: the gradients are exactly zero
    hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)Nout * K, st);
    if (e == hipSuccess && gbias) e = hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)Nout, st);
    return (int)e;
  }
  if (!gY || !X) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  WgradPlan p = wgrad_plan(rows, K, Nout);
  if (!p.ok) return MRG_E_SHAPE;
  const int vecG = (Nout % 4 == 0) && aligned16(gY);
  const int vecX = (K % 4 == 0) && aligned16(X);
#define GO(T)                                                                                                          \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
    hipLaunchKernelGGL((linear_wgrad_k<T>), dim3(p.G, (p.TN + p.TNB - 1) / p.TNB), dim3(MRG_BLOCK), p.lds, st, gY, X, (float*)ws, rows, K, Nout, p.TM, p.TN, p.TNB, p.rows_per_block, vecG, vecX); \
  } while (0)
  switch (p.tpw) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 7: GO(7); break;
    default: GO(13); break;
  }
#undef GO
  MRG_LAUNCH_CHECK();
  hipLaunchKernelGGL(linear_wgrad_reduce_k, dim3((K + 1 + 63) / 64, Nout), dim3(1024), 0, st, (const float*)ws, gW, gbias,
                     p.G, K, Nout, p.TM * 32, p.TN * 32);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
