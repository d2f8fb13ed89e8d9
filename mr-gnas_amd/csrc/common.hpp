// Shared device helpers for the gfx950 kernels of libmrgnas_hip.so.
// Wavefront = 64 lanes on CDNA4; every cross-lane idiom below is written for 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrgnas.h"

#define MRG_WAVE 64
#define MRG_BLOCK 256            // 4 waves per workgroup, one per SIMD
#ifndef MRG_MAX_GRID
#define MRG_MAX_GRID 2048        // 256 CUs x 8 blocks: streaming kernels grid-stride past this
#endif

#define MRG_LAUNCH_CHECK()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

namespace mrg {

// ---- vector-of-VEC floats ---------------------------------------------------
template <int VEC> struct Vec;
template <> struct Vec<4> {
  float4 v;
  __device__ __forceinline__ static Vec load(const float* p) { Vec r; r.v = *reinterpret_cast<const float4*>(p); return r; }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ static Vec fill(float x) { Vec r; r.v = make_float4(x, x, x, x); return r; }
  __device__ __forceinline__ float& operator[](int i) { return (&v.x)[i]; }
  __device__ __forceinline__ float operator[](int i) const { return (&v.x)[i]; }
};
template <> struct Vec<1> {
  float v;
  __device__ __forceinline__ static Vec load(const float* p) { Vec r; r.v = *p; return r; }
  __device__ __forceinline__ void store(float* p) const { *p = v; }
  __device__ __forceinline__ static Vec fill(float x) { Vec r; r.v = x; return r; }
  __device__ __forceinline__ float& operator[](int) { return v; }
  __device__ __forceinline__ float operator[](int) const { return v; }
};

template <int VEC> struct IVec;
template <> struct IVec<4> {
  int4 v;
  __device__ __forceinline__ static IVec load(const int32_t* p) { IVec r; r.v = *reinterpret_cast<const int4*>(p); return r; }
  __device__ __forceinline__ void store(int32_t* p) const { *reinterpret_cast<int4*>(p) = v; }
  __device__ __forceinline__ static IVec fill(int x) { IVec r; r.v = make_int4(x, x, x, x); return r; }
  __device__ __forceinline__ int& operator[](int i) { return (&v.x)[i]; }
  __device__ __forceinline__ int operator[](int i) const { return (&v.x)[i]; }
};
template <> struct IVec<1> {
  int v;
  __device__ __forceinline__ static IVec load(const int32_t* p) { IVec r; r.v = *p; return r; }
  __device__ __forceinline__ void store(int32_t* p) const { *p = v; }
  __device__ __forceinline__ static IVec fill(int x) { IVec r; r.v = x; return r; }
  __device__ __forceinline__ int& operator[](int) { return v; }
  __device__ __forceinline__ int operator[](int) const { return v; }
};


// Which of the block's row groups a thread belongs to.  With LPR == 64 a row group is a wave: the index is wave-uniform, and saying
// so (readfirstlane) turns every `base + r * D` of the kernel into scalar arithmetic and every row access into the
// SGPR-base + 32-bit lane offset form -- no 64-bit address registers and adds per tensor stream.
template <int LPR>
__device__ __forceinline__ int row_group_of_thread() {
  const int rw = (int)threadIdx.x / LPR;
  if constexpr (LPR == 64) return __builtin_amdgcn_readfirstlane(rw);
  else return rw;
}

// Sum over the 64 lanes of a wave, in every lane, without the LDS crossbar: four DPP steps inside each 16-lane row (quad
// permutes, half-row mirror, row mirror -- VALU latency, no ds_bpermute round trip), then the four row sums through scalar
// registers.  The xor butterfly over 64 lanes is a chain of six dependent ds_bpermute (~100 cycles each), which a kernel with two or
// three waves per SIMD cannot hide.
__device__ __forceinline__ float wave_sum_dpp(float x) {
  float t = x;
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, true));   // row_half_mirror
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xF, 0xF, true));   // row_mirror
  const int ti = __builtin_bit_cast(int, t);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sum over the LPR consecutive lanes that share one row (LPR = 16, 32 or 64), in every lane of the group.  A whole wave (LPR = 64)
// takes the DPP form above; 16 / 32 lanes use the xor butterfly.
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
  if constexpr (LPR == 64) {
    return wave_sum_dpp(x);
  } else {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, MRG_WAVE);
    return x;
  }
}

__device__ __forceinline__ float sigmoidf_fast(float z) { return 1.0f / (1.0f + __expf(-z)); }

// ---- row-tile geometry ---------------------------------------------------------
// A row of D floats is covered by LPR lanes x KMAX steps x VEC floats.
struct RowGeom {
  int vec;    // 4 (D % 4 == 0 and 16-B aligned pointers) or 1
  int lpr;    // lanes per row: 16 / 32 / 64
  int kmax;   // 1 / 2 / 4
  bool ok;
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline RowGeom row_geom(int D, bool all_aligned) {
  RowGeom g{};
  g.vec = (D % 4 == 0 && all_aligned) ? 4 : 1;
  int dv = D / g.vec;
  g.lpr = dv <= 16 ? 16 : (dv <= 32 ? 32 : 64);
  int k = (dv + g.lpr - 1) / g.lpr;
  g.kmax = k <= 1 ? 1 : (k <= 2 ? 2 : 4);
  g.ok = D > 0 && k <= 4;
  return g;
}

// Dispatch a kernel template<VEC, LPR, KMAX> on a RowGeom.
#define MRG_DISPATCH_GEOM(G, CALL)                                             \
  do {                                                                         \
    if ((G).vec == 4) {                                                        \
      if ((G).lpr == 16)      { MRG_DISPATCH_K(4, 16, (G).kmax, CALL); }       \
      else if ((G).lpr == 32) { MRG_DISPATCH_K(4, 32, (G).kmax, CALL); }       \
      else                    { MRG_DISPATCH_K(4, 64, (G).kmax, CALL); }       \
    } else {                                                                   \
      if ((G).lpr == 16)      { MRG_DISPATCH_K(1, 16, (G).kmax, CALL); }       \
      else if ((G).lpr == 32) { MRG_DISPATCH_K(1, 32, (G).kmax, CALL); }       \
      else                    { MRG_DISPATCH_K(1, 64, (G).kmax, CALL); }       \
    }                                                                          \
  } while (0)
// KMAX > 1 only ever occurs with LPR == 64.
#define MRG_DISPATCH_K(V, L, K, CALL)                                          \
  do {                                                                         \
    if ((K) == 1) { CALL(V, L, 1); }                                           \
    else if ((L) == 64 && (K) == 2) { CALL(V, 64, 2); }                        \
    else if ((L) == 64) { CALL(V, 64, 4); }                                    \
  } while (0)

// Deterministic second stage of the two-stage reductions: out[t] = sum_b ws[(b0 + b) * ld + t],
// b < nb.  A 64 x 16 thread block owns 64 consecutive t; thread row ty sums the partials
// b = ty, ty + 16, ... (coalesced along t; four interleaved accumulators), then the 16 row sums are added in order.
// thread row ty's share of column t: partial rows b = ty, ty + 16, ... of ws[(b0 + b) * ld + t], four loads in flight
// (the pass is latency-bound: a thread walks up to 64 partial rows); fixed association
template <typename T> struct ordered_acc { typedef T type; };
template <> struct ordered_acc<float> { typedef double type; };
template <typename T>
__device__ __forceinline__ typename ordered_acc<T>::type ordered_partial(const T* __restrict__ ws, int b0, int nb, int ld, int t, int ty) {
  typedef typename ordered_acc<T>::type A;                  // float partials are added in double (round 5: the float32 control)
  A a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int b = ty;
  for (; b + 48 < nb; b += 64) {
    const T v0 = ws[(int64_t)(b0 + b) * ld + t], v1 = ws[(int64_t)(b0 + b + 16) * ld + t];
    const T v2 = ws[(int64_t)(b0 + b + 32) * ld + t], v3 = ws[(int64_t)(b0 + b + 48) * ld + t];
    a0 += (A)v0; a1 += (A)v1; a2 += (A)v2; a3 += (A)v3;
  }
  for (; b < nb; b += 16) a0 += (A)ws[(int64_t)(b0 + b) * ld + t];
  return (a0 + a1) + (a2 + a3);
}

// blockIdx.y selects one of up to four row ranges [bs.x[y], bs.x[y + 1]) of ws, reduced into out + y * out_stride
struct ReduceRanges { int x[5]; };
template <typename T>
__global__ void ordered_reduce_k(const T* __restrict__ ws, T* __restrict__ out, ReduceRanges bs, int out_stride, int ld, int len) {
  typedef typename ordered_acc<T>::type A;
  __shared__ A part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  const int b0 = bs.x[blockIdx.y], nb = bs.x[blockIdx.y + 1] - b0;
  out += (int64_t)blockIdx.y * out_stride;
  const A acc = t < len ? ordered_partial<T>(ws, b0, nb, ld, t, ty) : A(0);
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && t < len) {
    A tot = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += part[i][tx];
    out[t] = (T)tot;
  }
}

template <typename T>
inline void launch_ordered_reduce(const T* ws, T* out, int b0, int nb, int ld, int len, hipStream_t st) {
  ReduceRanges bs{{b0, b0 + nb, 0, 0, 0}};
  hipLaunchKernelGGL((ordered_reduce_k<T>), dim3((len + 63) / 64, 1), dim3(1024), 0, st, ws, out, bs, 0, ld, len);
}
// n_ranges (<= 4) consecutive row ranges bounds[0..n_ranges] in one launch: out + r * out_stride receives range r
template <typename T>
inline void launch_ordered_reduce_ranges(const T* ws, T* out, const int* bounds, int n_ranges, int out_stride, int ld, int len, hipStream_t st) {
  ReduceRanges bs{};
  for (int i = 0; i <= n_ranges; ++i) bs.x[i] = bounds[i];
  hipLaunchKernelGGL((ordered_reduce_k<T>), dim3((len + 63) / 64, n_ranges), dim3(1024), 0, st, ws, out, bs, out_stride, ld, len);
}

// Grid of the FLAT HBM-streaming kernels (compose, K-way sums, the MixedOp combine / statistics / gradient-reduction passes:
// every lane walks consecutive float4 with a grid stride and nothing depends on a gathered index).
// Measured on MI355X (tools/stream_lab.hip, 2 reads + 1 write over [558 771, 200] floats): 512 blocks of 256 threads, one
// float4 per lane in flight, stream at 5.9 TB/s; 1024 blocks 5.6, 2048 blocks 5.3, and MORE loads in flight per lane are
// slower still (U = 4: 4.7-5.1) -- fewer concurrent address windows per tensor keep more of each DRAM page's bursts together.
// In the supernet step (profiles/r3_stream_grid.txt): mrg_sum_buffers 4.90 -> 4.50 ms, mrg_mix_fwd 3.72 -> 3.47, statistics
// 3.10 -> 2.95.  The row-per-wave kernels (gates, gathers, reducers' backward, mrg_mix_bwd_apply with its up to 14 address
// streams) are latency-bound per wave and LOSE with fewer blocks (gate_fwd 1.4 -> 3.1 ms at 512): they keep grid_for.
inline int& stream_blocks() { static int b = 512; return b; }
inline int stream_grid_for(int64_t work_items, int items_per_block) {
  int64_t b = (work_items + items_per_block - 1) / items_per_block;
  if (b < 1) b = 1;
  if (b > stream_blocks()) b = stream_blocks();
  return (int)b;
}

inline int grid_for(int64_t work_items, int items_per_block) {
  int64_t b = (work_items + items_per_block - 1) / items_per_block;
  if (b < 1) b = 1;
  if (b > MRG_MAX_GRID) b = MRG_MAX_GRID;
  return (int)b;
}

}  // namespace mrg
