// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 (operands in registers) for the issue patterns
// the 3-way split GEMM can choose from: CHAIN dependent MFMAs per accumulator in a row, NVALU independent
// VALU instructions after every MFMA, ILV accumulators interleaved round-robin.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int CHAIN, int NVALU, int ILV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(b0 + threadIdx.x * 1e-3f - j); }
  unsigned d0 = threadIdx.x, d1 = threadIdx.x * 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i += ILV)
#pragma unroll
      for (int c = 0; c < CHAIN; ++c)
#pragma unroll
        for (int v = 0; v < ILV; ++v) {
          acc[i + v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i + v], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NVALU; ++q) {
            if (q & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d0) : "v"(d1));
            else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d1) : "v"(d0));
          }
        }
  }
  float s = d0 + d1;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// distinct A / B operand registers per MFMA, as in the GEMM (3 A planes x 2 row tiles, 3 B planes per column tile)
template <int NT>
__global__ __launch_bounds__(256) void kd(float* out, int iters, float a0, float b0) {
  f32x16 acc[2][NT];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
  bf16x8 A[2][3], B[NT][3];
  for (int m = 0; m < 2; ++m) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) A[m][p][j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j + p + m);
  for (int n = 0; n < NT; ++n) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) B[n][p][j] = (__bf16)(b0 + threadIdx.x * 1e-3f - j + p + n);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#define T(AP, BP) _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m][AP], B[n][BP], acc[m][n], 0, 0, 0)
      T(1, 1); T(2, 0); T(0, 2); T(1, 0); T(0, 1); T(0, 0);
#undef T
    }
  }
  float s = 0;
  for (int m = 0; m < 2; ++m) for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) s += acc[m][i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT> void rund(int blocks, int iters, const char* tag) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((kd<NT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * NT * 12;
    if (rep == 2) printf("%-12s distinct operands NT=%d: %7.1f TFLOP/s  (%.1f clk/MFMA/SIMD @2.4GHz)\n", tag, NT, blocks * 4.0 * n * 32768.0 / ms / 1e9,
                         ms * 1e-3 * 2.4e9 / ((double)((blocks + 255) / 256) * n));
  }
  hipFree(out);
}
template <int NACC, int CHAIN, int NVALU, int ILV> void run(int blocks, int iters, const char* tag) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, CHAIN, NVALU, ILV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * NACC * CHAIN * 32768.0;
    if (rep == 2) printf("%-12s nacc=%2d chain=%d valu/mfma=%d interleave=%d: %7.1f TFLOP/s  (%.1f clk/MFMA/SIMD @2.4GHz)\n", tag, NACC, CHAIN, NVALU, ILV,
                         flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)((blocks + 255) / 256) * iters * NACC * CHAIN));
  }
  hipFree(out);
}
int main() {
  run<14, 1, 0, 1>(256, 4000, "1 wave/SIMD");
  run<14, 6, 0, 1>(256, 1000, "1 wave/SIMD");
  run<14, 6, 2, 1>(256, 1000, "1 wave/SIMD");
  run<14, 6, 4, 1>(256, 1000, "1 wave/SIMD");
  run<14, 6, 6, 1>(256, 1000, "1 wave/SIMD");
  run<14, 6, 0, 2>(256, 1000, "1 wave/SIMD");
  run<14, 6, 4, 2>(256, 1000, "1 wave/SIMD");
  run<14, 6, 0, 7>(256, 1000, "1 wave/SIMD");
  rund<7>(256, 2000, "1 wave/SIMD");
  rund<7>(2183, 25, "short WGs");
  run<14, 6, 0, 2>(2183, 25, "short WGs");
  run<14, 6, 0, 2>(2048, 25, "short WGs");
  run<14, 6, 0, 2>(256, 25 * 8, "same work");
  run<14, 6, 0, 2>(256, 20000, "long");
  run<7, 6, 4, 1>(512, 1000, "2 waves/SIMD");
  run<7, 6, 0, 1>(512, 1000, "2 waves/SIMD");
  return 0;
}
