#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o, const float* in) {
  unsigned a0 = __builtin_bit_cast(unsigned, in[threadIdx.x]), b0 = __builtin_bit_cast(unsigned, in[threadIdx.x + 64]);
  unsigned a1 = __builtin_bit_cast(unsigned, in[threadIdx.x + 128]), b1 = __builtin_bit_cast(unsigned, in[threadIdx.x + 192]);
  auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  o[threadIdx.x] = __builtin_bit_cast(float, s0[0]);
  o[threadIdx.x + 64] = __builtin_bit_cast(float, s0[1]);
  o[threadIdx.x + 128] = __builtin_bit_cast(float, s1[0]);
  o[threadIdx.x + 192] = __builtin_bit_cast(float, s1[1]);
}
__global__ void k2(float* o, const float* in) {
  unsigned a0 = __builtin_bit_cast(unsigned, in[threadIdx.x]), b0 = __builtin_bit_cast(unsigned, in[threadIdx.x + 64]);
  unsigned r0, r1;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "=v"(r0), "=v"(r1) : "0"(a0), "1"(b0));
  o[threadIdx.x] = __builtin_bit_cast(float, r0);
  o[threadIdx.x + 64] = __builtin_bit_cast(float, r1);
}
