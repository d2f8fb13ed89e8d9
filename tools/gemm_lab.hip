// Kernel lab: timing-only variants of the LDS row GEMM inner loop (results are NOT checked).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mr-gnas_amd/csrc tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include "gemm.hpp"
#include <cstdio>
#include <vector>
using namespace mrg;

// FLAGS: 1 = no barrier, 2 = no stash (LDS writes), 4 = no fetch, 8 = no epilogue, 16 = no LDS fragment reads,
//        32 = stagger by hardware wave slot parity, 64 = stagger by blockIdx parity, 128 = stagger by (blockIdx/256)&1
template <int NT, int FLAGS>
__global__ __launch_bounds__(MRG_BLOCK, 2) void lab_k(GemmArgs a) {
  constexpr int MT = 1, GBK = 16, GBM = 128, GLD = GBK + 4, F4R = GBK / 4;
  constexpr int NA = GBM * F4R / MRG_BLOCK, NBT = NT * 32 * F4R, NB = (NBT + MRG_BLOCK - 1) / MRG_BLOCK;
  extern __shared__ __align__(16) float smem[];
  constexpr int A_TILE = GBM * GLD, B_TILE = NT * 32 * GLD, STAGE = A_TILE + B_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * GBM;
  const int K = a.K1;
  const int nkt = (K + GBK - 1) / GBK;
  f32x16 acc[NT];
  for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float4 pa[NA], pb[NB];
  const float* arow[NA]; const float* brow[NB];
  for (int i = 0; i < NA; ++i) { int64_t row = row0 + (tid + i * 256) / F4R; arow[i] = a.A1 + (row < a.rows ? row : a.rows - 1) * a.K1; }
  for (int i = 0; i < NB; ++i) { int f = tid + i * 256; f = f < NBT ? f : NBT - 1; int col = f / F4R; brow[i] = a.B + (int64_t)(col < a.N ? col : a.N - 1) * a.ldb; }
  auto fetch = [&](int k0) {
    for (int i = 0; i < NA; ++i) { int k = k0 + ((tid + i * 256) % F4R) * 4; pa[i] = *reinterpret_cast<const float4*>(arow[i] + (k + 4 <= K ? k : K - 4)); }
    for (int i = 0; i < NB; ++i) { int f = tid + i * 256; f = f < NBT ? f : NBT - 1; int k = k0 + (f % F4R) * 4; pb[i] = *reinterpret_cast<const float4*>(brow[i] + (k + 4 <= K ? k : K - 4)); }
  };
  auto stash = [&](int buf) {
    for (int i = 0; i < NA; ++i) { int f = tid + i * 256; *reinterpret_cast<float4*>(&smem[buf * STAGE + (f / F4R) * GLD + (f % F4R) * 4]) = pa[i]; }
    for (int i = 0; i < NB; ++i) { int f = tid + i * 256; if (f < NBT) *reinterpret_cast<float4*>(&smem[buf * STAGE + A_TILE + (f / F4R) * GLD + (f % F4R) * 4]) = pb[i]; }
  };
  if (FLAGS & 32) { unsigned hw = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 4 /* HW_ID[3:0] wave_id */); if (hw & 1) for (int i = 0; i < 14; ++i) __builtin_amdgcn_s_sleep(2); }
  if ((FLAGS & 64) && (blockIdx.x & 1)) for (int i = 0; i < 14; ++i) __builtin_amdgcn_s_sleep(2);
  if ((FLAGS & 128) && ((blockIdx.x >> 8) & 1)) for (int i = 0; i < 14; ++i) __builtin_amdgcn_s_sleep(2);
  fetch(0); stash(0); __syncthreads();
  int cur = 0;
  float4 af = make_float4(1.f, 2.f, 3.f, 4.f), bfix = make_float4(.5f, .25f, .125f, 1.f);
  for (int kt = 0; kt < nkt; ++kt) {
    if (!(FLAGS & 4) && kt + 1 < nkt) fetch((kt + 1) * GBK);
    const float* At = smem + cur * STAGE + (wave * 32 + li) * GLD + lh * 4;
    const float* Bt = smem + cur * STAGE + A_TILE + li * GLD + lh * 4;
#pragma unroll
    for (int t = 0; t < GBK / 8; ++t) {
      float4 a4 = (FLAGS & 16) ? af : *reinterpret_cast<const float4*>(At + t * 8);
      float4 b4[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b4[n] = (FLAGS & 16) ? bfix : *reinterpret_cast<const float4*>(Bt + n * 32 * GLD + t * 8);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[n].x, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[n].y, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[n].z, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[n].w, acc[n], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      if (!(FLAGS & 2)) stash(cur ^ 1);
      if (!(FLAGS & 1)) __syncthreads();
      cur ^= 1;
    }
  }
  if (FLAGS & 8) {
    float s = 0.f;
    for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.678f) a.C[0] = s;
    return;
  }
  for (int n = 0; n < NT; ++n) {
    const int col = n * 32 + li;
    if (col < a.N)
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.rows) a.C[row * a.ldc + col] = acc[n][r];
      }
  }
}

// LDS-DMA variant: tiles go global -> LDS directly (global_load_lds_dwordx4), unpadded rows of 16 floats.
template <int NT, int FLAGS>
__global__ __launch_bounds__(MRG_BLOCK, 2) void dma_k(GemmArgs a) {
  constexpr int GBK = 16, GBM = 128, GLD = GBK, F4R = GBK / 4;
  constexpr int NA = GBM * F4R / MRG_BLOCK, NBT = NT * 32 * F4R, NB = (NBT + MRG_BLOCK - 1) / MRG_BLOCK;
  extern __shared__ __align__(16) float smem[];
  constexpr int A_TILE = GBM * GLD, B_TILE = NT * 32 * GLD, STAGE = A_TILE + B_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * GBM;
  const int K = a.K1;
  const int nkt = (K + GBK - 1) / GBK;
  f32x16 acc[NT];
  for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const float* arow[NA]; const float* brow[NB];
  for (int i = 0; i < NA; ++i) { int64_t row = row0 + (tid + i * 256) / F4R; arow[i] = a.A1 + (row < a.rows ? row : a.rows - 1) * a.K1; }
  for (int i = 0; i < NB; ++i) { int f = tid + i * 256; f = f < NBT ? f : NBT - 1; int col = f / F4R; brow[i] = a.B + (int64_t)(col < a.N ? col : a.N - 1) * a.ldb; }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  auto fetch = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int f = tid + i * 256; int k = k0 + (f % F4R) * 4;
      // LDS destination of a wave-instruction = wave-uniform base + lane*16: rows are unpadded, so f*4 floats is exactly that
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(arow[i] + (k + 4 <= K ? k : K - 4)), (lds_ptr_t)(smem + buf * STAGE + (f - lane) * 4), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int f = tid + i * 256; int fc = f < NBT ? f : NBT - 1; int k = k0 + (fc % F4R) * 4;
      if (f - lane < NBT) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(brow[i] + (k + 4 <= K ? k : K - 4)), (lds_ptr_t)(smem + buf * STAGE + A_TILE + (f - lane) * 4), 16, 0, 0);
    }
  };
  fetch(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (!(FLAGS & 4) && kt + 1 < nkt) fetch(cur ^ 1, (kt + 1) * GBK);
    const float* At = smem + cur * STAGE + (wave * 32 + li) * GLD + lh * 4;
    const float* Bt = smem + cur * STAGE + A_TILE + li * GLD + lh * 4;
#pragma unroll
    for (int t = 0; t < GBK / 8; ++t) {
      float4 a4 = *reinterpret_cast<const float4*>(At + t * 8);
      float4 b4[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b4[n] = *reinterpret_cast<const float4*>(Bt + n * 32 * GLD + t * 8);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[n].x, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[n].y, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[n].z, acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[n].w, acc[n], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  if (FLAGS & 8) {
    float s = 0.f;
    for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 12345.678f) a.C[0] = s;
    return;
  }
  for (int n = 0; n < NT; ++n) {
    const int col = n * 32 + li;
    if (col < a.N)
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < a.rows) a.C[row * a.ldc + col] = acc[n][r];
      }
  }
}

template <int FLAGS> void run_dma(GemmArgs a, const char* tag) {
  constexpr int NT = 7;
  size_t lds = (size_t)2 * (128 + NT * 32) * 16 * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_k<NT, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((unsigned)((a.rows + 127) / 128));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((dma_k<NT, FLAGS>), grid, dim3(256), lds, 0, a); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
  }
  double flop = 2.0 * a.rows * a.K1 * 224;
  printf("%-44s %.3f ms  executed %.1f TF/s (useful %.1f)\n", tag, best, flop / best / 1e9, 2.0 * a.rows * a.K1 * a.N / best / 1e9);
}

template <int FLAGS> void run(GemmArgs a, const char* tag) {
  constexpr int NT = 7;
  size_t lds = (size_t)2 * (128 + NT * 32) * 20 * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_k<NT, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((unsigned)((a.rows + 127) / 128));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((lab_k<NT, FLAGS>), grid, dim3(256), lds, 0, a); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
  }
  double flop = 2.0 * a.rows * a.K1 * 224;
  printf("%-44s %.3f ms  executed %.1f TF/s (useful %.1f)\n", tag, best, flop / best / 1e9, 2.0 * a.rows * a.K1 * a.N / best / 1e9);
}

template <typename KernelT> void run_lib(KernelT kern, GemmArgs a, size_t lds, int gbm, const char* tag) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((unsigned)((a.rows + gbm - 1) / gbm));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 1 && ms < best) best = ms;
  }
  printf("%-44s %.3f ms  useful %.1f TF/s\n", tag, best, 2.0 * a.rows * (a.K1 + a.K2) * a.N / best / 1e9);
}

int main() {
  int64_t rows = 544230; int K = 200, N = 200;
  float *A, *B, *C;
  hipMalloc(&A, rows * K * 4); hipMalloc(&B, N * K * 4); hipMalloc(&C, rows * N * 4);
  std::vector<float> h(rows * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(A, h.data(), rows * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), N * K * 4, hipMemcpyHostToDevice);
  GemmArgs a{}; a.A1 = A; a.A2 = A; a.K1 = K; a.K2 = 0; a.B = B; a.ldb = K; a.C = C; a.ldc = N; a.N = N; a.rows = rows;
  { GemmArgs b = a; b.bias = B; b.act = 1;
    run_lib(rowgemm_k<7, 1, 16, EPI_BIAS_ACT, true>, b, gemm_lds_bytes(7, 1, 16), 128, "LIB rowgemm_k<7,1,16> (register staging)");
    run_lib(rowgemm_dma_k<7, EPI_BIAS_ACT, false>, b, (size_t)2 * (128 + 224) * 16 * 4, 128, "LIB rowgemm_dma_k<7>");
    GemmArgs c = a;
    run_lib(rowgemm_dma_k<7, EPI_BIAS_ACT, false>, c, (size_t)2 * (128 + 224) * 16 * 4, 128, "LIB rowgemm_dma_k<7> no bias/act");
    c.rows = 544256;   /* multiple of 128: allocate enough */
    }
  run<0>(a, "full");
  run<8>(a, "no epilogue");
  run<8 | 4>(a, "no epilogue, no fetch");
  run<8 | 4 | 2>(a, "no epilogue, no fetch, no stash");
  run<8 | 4 | 2 | 1>(a, "no epilogue, no fetch, no stash, no barrier");
  run<8 | 4 | 2 | 1 | 16>(a, "MFMA only (no LDS reads either)");
  run<8 | 1>(a, "no epilogue, no barrier (fetch+stash kept)");
  run<1>(a, "full but no barrier");
  run<32>(a, "full, stagger by wave slot");
  run<64>(a, "full, stagger by blockIdx&1");
  run<128>(a, "full, stagger by (blockIdx>>8)&1");
  run_dma<0>(a, "LDS-DMA full");
  run_dma<8>(a, "LDS-DMA no epilogue");
  run_dma<8 | 4>(a, "LDS-DMA no epilogue, no fetch");
  run<8 | 32>(a, "no epilogue, stagger by wave slot");
  run<8 | 64>(a, "no epilogue, stagger by blockIdx&1");
  return 0;
}
