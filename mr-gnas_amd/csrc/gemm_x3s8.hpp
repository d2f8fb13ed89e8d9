// The LDS-weight split-core row GEMM (gemm_x3s.hpp) for outputs of EIGHT column tiles (225 .. 256 columns: D = 256, BASELINE config 5)
// as ONE column block.
//
// gemm_x3s.hpp runs 256 columns as two blocks of four tiles (seven is the widest block whose weight ring of three slabs stays below
// the 64 KiB an LDS-DMA can address), which reads the activation operand twice -- at the C5 shape that operand is 10.2 GB per launch,
// beyond every cache, and the launch moves 30.7 GB at 2.85 TB/s with the matrix pipe busy 0.13 (profiles/r4_sq_counters_c5.txt).
// Here a workgroup owns 128 rows x 8 tiles: 128 accumulator registers per lane, a weight ring of TWO 24 KB slabs (the DMA of slab
// s + 1 is issued at the top of slab s into the buffer slab s - 1 was read from, and has that slab's MFMAs to land), activations two
// slabs ahead in a register ring of two.  Same fragments, same six terms in the same order per accumulator as every other split-core
// kernel: bit-identical results (tests/test_ops_gpu.py::test_split_core_kernels_are_bit_exact_with_each_other, D = 256 cases).
// Plain launches only (no direction groups, single source): the input gradient / forward of the aggregators' Linear and the fused
// a_max / a_mean at D = 256; the gate epilogue does not fit the register file at eight tiles (55 spills) and keeps the two-block form.
//
// In-order vector-memory history of a wave:  A(0) B(0) A(1) | B(1) A(2) | B(2) A(3) | ...   (slab s issues B(s+1), A(s+2) at its top)
//   top of slab s, before the splits of A(s+1):   younger than A(s+1) = B(s+1) A(s+2)          -> vmcnt(NBW + 2)
//   end of slab s, before the barrier:            younger than B(s+1) = A(s+2)                 -> vmcnt(2)
#pragma once
#include "gemm_x3s.hpp"

namespace mrg {

template <int NT, int EPI>
__global__ __launch_bounds__(256, 2) void rowgemm_x3s8_k(GemmArgs a, const char* __restrict__ Bp, int ntile) {
  constexpr int GBM = 128;
  constexpr int NCH = NT * 3;                   // 1 KB chunks of one pre-split B slab
  constexpr int BSLAB = NCH * 1024;
  constexpr int NBW = (NCH + 3) / 4;            // DMA instructions per wave and slab (the last wave repeats the last chunk: same bytes, same place)
  constexpr int NP = (NT + 1) / 2;
  extern __shared__ __align__(16) char smem_b8[];    // [2][BSLAB] = 48 KB at eight tiles
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * GBM;
  const int64_t roww = row0 + wave * 32;
  const int K = a.K1;
  const int nslab = (K + 15) >> 4;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  int64_t rc = roww + li < a.rows ? roww + li : a.rows - 1;
  if (rc < 0) rc = 0;
  if (a.row_index) rc = a.row_index[rc];                 // gathered rows (fused a_max / a_mean: edges in destination order)
  const float* ar1 = a.A1 + rc * a.K1;
  v4f xr[2][2];                                          // raw fragments: a ring of two slabs
  auto load_a = [&](int slab, v4f (&x)[2]) {
    const int sl = slab < nslab ? slab : nslab - 1;      // beyond the end: re-read the last slab (an asynchronous fill is never conditional)
    const int k = sl * 16 + lh * 8;
    const float* p0 = ar1 + (k + 4 <= K ? k : K - 4);    // beyond K: any finite values, the weight's rows there are zero
    const float* p1 = ar1 + (k + 8 <= K ? k + 4 : K - 4);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[0]) : "v"(p0));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[1]) : "v"(p1));
  };
  auto fetch_b = [&](int slab, int buf) {
    const char* src = Bp + (int64_t)slab * ntile * 3072;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      int c = wave * NBW + i;
      c = c < NCH ? c : NCH - 1;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + c * 1024 + lane * 16), (lds_ptr_t)(smem_b8 + buf * BSLAB + c * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem_b8 + (unsigned)lane * 16u;
  u32x4 bq2[2][2][3];                                    // [double buffer][tile of the pair][plane]
  auto read_b = [&](int n, int buf, u32x4 (&q)[3]) {
    const unsigned ad = lds0 + (unsigned)(buf * BSLAB + n * 3072);
    asm volatile("ds_read_b128 %0, %1" : "=v"(q[0]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q[1]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(q[2]) : "v"(ad));
  };
  u32x4 ch, cm, cl, nh, nm, nl;
  auto split_pair_of = [&](const v4f (&x)[2], int q, u32x4& H, u32x4& M, u32x4& L) {     // q = 0..3: floats 2q, 2q + 1 of the 8
    const v4f& v = x[q >> 1];
    unsigned h, m, l;
    if (q & 1) split_pair(v.z, v.w, h, m, l); else split_pair(v.x, v.y, h, m, l);
    H[q] = h; M[q] = m; L[q] = l;
  };

  // ---- prologue:  A(0) B(0) A(1)
  load_a(0, xr[0]);
  fetch_b(0, 0);
  load_a(1, xr[1]);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");           // A(0) and this wave's share of B(0) have landed (younger: A(1))
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair_of(xr[0], q, ch, cm, cl);
  __builtin_amdgcn_s_barrier();                              // everybody's share of B(0) is in LDS

  auto slab = [&](auto r_c, int s) {
    constexpr int R = decltype(r_c)::value;
    const bool has_next = s + 1 < nslab;
    if (has_next) fetch_b(s + 1, R ^ 1);                     // into the buffer slab s - 1 was read from: every wave is past that barrier
    load_a(s + 2, xr[R]);                                    // into the raw registers slab s - 1 split from
    read_b(0, R, bq2[0][0]);
    if (NT > 1) read_b(1, R, bq2[0][1]);
    if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 2) : "memory");    // A(s+1): younger = B(s+1) A(s+2)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const int n0 = 2 * pp, n1 = 2 * pp + 1 < NT ? 2 * pp + 1 : n0;
      const bool two = 2 * pp + 1 < NT;                      // an odd NT ends on a single tile
      if (n0 + 2 < NT) read_b(n0 + 2, R, bq2[(pp + 1) & 1][0]);
      if (n0 + 3 < NT) read_b(n0 + 3, R, bq2[(pp + 1) & 1][1]);
      // the reads just issued may still be in flight
      if (n0 + 3 < NT) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      else if (n0 + 2 < NT) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (has_next && pp < 4) split_pair_of(xr[R ^ 1], pp, nh, nm, nl);   // one float pair of A(s+1) in the shadow of each tile pair
      const bf16x8 Ah = __builtin_bit_cast(bf16x8, ch), Am = __builtin_bit_cast(bf16x8, cm), Al = __builtin_bit_cast(bf16x8, cl);
      const bf16x8 Bh0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][0]), Bm0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][1]),
                   Bl0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][2]);
      const bf16x8 Bh1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][0]), Bm1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][1]),
                   Bl1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][2]);
      // small terms first, the leading term last (the order of every split-core kernel)
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm1, acc[n1], 0, 0, 0);
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh1, acc[n1], 0, 0, 0);
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl1, acc[n1], 0, 0, 0);
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh1, acc[n1], 0, 0, 0);
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm1, acc[n1], 0, 0, 0);
      acc[n0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh0, acc[n0], 0, 0, 0);
      if (two) acc[n1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh1, acc[n1], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < (two ? 12 : 6); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // ... then up to three VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next) {
#pragma unroll
      for (int q = NP; q < 4; ++q) split_pair_of(xr[R ^ 1], q, nh, nm, nl);   // fewer than four tile pairs: the rest of A(s+1) here
      ch = nh; cm = nm; cl = nl;
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         // this wave's share of B(s+1) is in LDS (younger: A(s+2))
      __builtin_amdgcn_s_barrier();                            // ... and everybody's; all reads of this slab's buffer are done
    }
  };
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab(std::integral_constant<int, 0>{}, s);
    slab(std::integral_constant<int, 1>{}, s + 1);
  }
  if (s < nslab) slab(std::integral_constant<int, 0>{}, s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tail's unused A fills: their registers must stay until they land
  asm volatile("" :: "v"(xr[0][0]), "v"(xr[0][1]), "v"(xr[1][0]), "v"(xr[1][1]));

  if constexpr (EPI == EPI_SEGMAX) gemm_epilogue_segmax<NT>(a, acc, roww, 0, li, lh);
  else if constexpr (EPI == EPI_SEGSUM) gemm_epilogue_segsum<NT>(a, acc, roww, 0, li, lh);
  else gemm_epilogue<NT, EPI>(a, acc, roww, 0, li, lh, row0 + GBM <= a.rows);
}

// 1 (default): plain launches whose output is eight column tiles wide run on rowgemm_x3s8_k; 0: two four-tile column blocks (round 3)
inline int& gemm_wide8() { static int m = 1; return m; }

template <int EPI>
inline bool x3s8_eligible(const GemmArgs& a) {
  if (EPI == EPI_GATE) return false;                       // 55 register spills at eight tiles: keeps the two-block form
  if (!(gemm_wide8() && a.grp.n == 0 && (a.K2 == 0 || !a.A2) && a.K1 >= 32 && x3s_eligible(a))) return false;
  return (a.N > 224 && a.N <= 256) || (gemm_wide8() == 2 && a.N > 192 && a.N <= 224);   // 2 (lab): seven-tile outputs on the ring of two as well
}

// Bp: the split of B prepared by launch_bsplit(..., nt = gemm_pick_nt(a.N) = 4, ...): [slab][8 tiles][plane][lane], the layout of one
// eight-tile block as well
template <int EPI>
inline int launch_rowgemm_x3s8(GemmArgs a, const void* Bp, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  a.A2 = a.A1; a.K2 = 0;
  const int ntile = x3_tiles(a.N, gemm_pick_nt(a.N));      // 8 (7: the lab's seven-tile form)
  dim3 grid((unsigned)((a.rows + 127) / 128), 1);
  const size_t lds = (size_t)2 * ntile * 3 * 1024;
  if (ntile == 8) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3s8_k<8, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((rowgemm_x3s8_k<8, EPI>), grid, dim3(256), lds, st, a, (const char*)Bp, ntile);
  } else if (ntile == 7) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3s8_k<7, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((rowgemm_x3s8_k<7, EPI>), grid, dim3(256), lds, st, a, (const char*)Bp, ntile);
  } else {
    return MRG_E_SHAPE;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

}  // namespace mrg
