// Lab: does a co-resident wave that issues global stores / loads / VALU slow the MFMA stream of the wave sharing its SIMD?
// 512-thread workgroups, one per CU: waves 0..3 run a chain of v_mfma_f32_32x32x16_bf16 (7 accumulators, the row GEMM's order),
// waves 4..7 play a partner role.  Prints cycles per MFMA of the compute waves (s_memtime) per role.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_partner_lab.hip -o tools/labbin/mfma_partner
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>   // 0 partner exits, 1 dword stores to scattered rows (the GEMM epilogue's pattern), 2 dwordx4 loads, 3 VALU loop, 4 partner is a second MFMA wave, 5 float4 row-order stores
__global__ __launch_bounds__(512) void k(float* out, const float* in, unsigned long long* stamp, int iters, int64_t rows_total) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave < 4 || ROLE == 4) {
    f32x16 acc[7];
    for (int i = 0; i < 7; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 A[3], B[7][3];
    for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) A[p][j] = (__bf16)(1.0f + lane * 1e-3f + j + p);
    for (int n = 0; n < 7; ++n) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) B[n][p][j] = (__bf16)(0.5f + lane * 1e-3f - j + p + n);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int n0 = 2 * pp, n1 = 2 * pp + 1;
#define T(AP, BP) acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[AP], B[n0][BP], acc[n0], 0, 0, 0); if (n1 < 7) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[AP], B[n1 < 7 ? n1 : 0][BP], acc[n1 < 7 ? n1 : 0], 0, 0, 0)
        T(1, 1); T(2, 0); T(0, 2); T(1, 0); T(0, 1); T(0, 0);
#undef T
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 7; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if (lane == 0) stamp[blockIdx.x * 8 + wave] = t1 - t0;
    return;
  }
  if (ROLE == 0) return;
  const int w = wave - 4, li = lane & 31, lh = lane >> 5;
  if (ROLE == 1 || ROLE == 5) {
    // strips of 32 rows x 224 columns of a [rows_total, 200] tensor, as the epilogue walks them
    float v = (float)lane;
    for (int it = 0; it < iters / 13 + 1; ++it) {
      const int64_t strip = ((int64_t)it * gridDim.x * 4 + blockIdx.x * 4 + w) % (rows_total / 32);
      float* base = out + strip * 32 * 200;
      if (ROLE == 1) {
#pragma unroll
        for (int n = 0; n < 7; ++n) {
          if (n * 32 + li >= 200) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2) + 4 * lh) * 200 + n * 32 + li] = v + r;
        }
      } else {
        float4* b4 = reinterpret_cast<float4*>(base);
#pragma unroll
        for (int i = 0; i < 25; ++i) b4[i * 64 + lane] = make_float4(v, v + 1, v + 2, v + i);
      }
    }
    return;
  }
  if (ROLE == 2) {
    float4 s = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters / 13 + 1; ++it) {
      const int64_t strip = ((int64_t)it * gridDim.x * 4 + blockIdx.x * 4 + w) % (rows_total / 32);
      const float4* b4 = reinterpret_cast<const float4*>(in + strip * 32 * 200);
#pragma unroll
      for (int i = 0; i < 25; ++i) { float4 x = b4[i * 64 + lane]; s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w; }
    }
    if (s.x == 123.456f) out[1] = s.x + s.y + s.z + s.w;
    return;
  }
  if (ROLE == 3) {
    unsigned d0 = lane, d1 = lane * 3;
    for (int it = 0; it < iters * 40; ++it) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(d0) : "v"(d1));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d1) : "v"(d0));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(d0) : "v"(d1));
      asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d1) : "v"(d0));
    }
    if (d0 + d1 == 12345u) out[2] = 1.f;
  }
}

template <int ROLE> void run(const char* name, float* out, const float* in, unsigned long long* st, int iters, int64_t rows) {
  hipMemset(st, 0, 256 * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROLE>), dim3(256), dim3(512), 0, 0, out, in, st, iters, rows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROLE>), dim3(256), dim3(512), 0, 0, out, in, st, iters, rows);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < (ROLE == 4 ? 8 : 4); ++w) if (h[b * 8 + w]) v.push_back((double)h[b * 8 + w] / ((double)iters * 42));
  std::sort(v.begin(), v.end());
  printf("%-44s %8.3f ms   cycles per MFMA of a compute wave: median %.1f  p10 %.1f  p90 %.1f\n", name, ms, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int64_t rows = 558771;
  float *out, *in; unsigned long long* st;
  hipMalloc(&out, rows * 200 * 4 + 4096); hipMalloc(&in, rows * 200 * 4 + 4096); hipMalloc(&st, 256 * 8 * 8);
  hipMemset(in, 0, rows * 200 * 4);
  run<0>("partner exits", out, in, st, iters, rows);
  run<3>("partner: VALU loop", out, in, st, iters, rows);
  run<2>("partner: float4 loads (row order)", out, in, st, iters, rows);
  run<1>("partner: dword stores, epilogue pattern", out, in, st, iters, rows);
  run<5>("partner: float4 stores, row order", out, in, st, iters, rows);
  run<4>("partner: a second MFMA wave", out, in, st, iters, rows);
  run<0>("partner exits (again)", out, in, st, iters, rows);
  return 0;
}
