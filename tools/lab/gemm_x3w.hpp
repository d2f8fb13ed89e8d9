// LAB ONLY since round 4 (it ran only behind mrg_gemm_set_mode(4); build with -I mr-gnas_amd/csrc -I tools/lab).
// Split-bf16 row GEMM, two waves per SIMD.
//
// What round 2 measured on rowgemm_x3_k (gemm_x3.hpp; tools/gemm_x3_lab.hip, profiles/r2_rowgemm_rounds.txt): with 224
// accumulators a wave needs the whole register file of its SIMD, so a CU runs ONE workgroup whose four waves go through
// prologue (A / B latency) -> k-loop -> epilogue (stores) one after the other, and all CUs do so in lock step:
// 1 063 workgroups on 256 CUs are 4.15 rounds that cost 5, and a round is 48 us of which the MFMAs are 17.6.  Starting the
// rounds out of phase changes nothing (+-2 %): the phases of ONE CU have to overlap.
//
// Here a wave owns 64 rows x 4 (or 3) column tiles: 128 (96) accumulators, <= 256 registers in all, so two workgroups
// share a CU and while one wave waits for its first slabs or drains its stores the other wave of the SIMD multiplies.
// The seven column tiles of a row block are two workgroups (tiles 0-3 and 4-6) that are dispatched to the same XCD one
// after the other, so the second reads the A rows from that XCD's L2.  Everything else is the proven wave-autonomous
// scheme of gemm_x3.hpp (no barrier; A by LDS-DMA into a private 4-slot ring three slabs ahead; pre-split B fragments
// straight from L2 into the registers the previous slab released; counted s_waitcnt) minus its one-slab-ahead
// software pipelining of the A split, which the second wave makes unnecessary and the register budget forbids.
#pragma once
#include "gemm_x3.hpp"

// s_sleep(127) repetitions (~3.8 us each) the second workgroup of a CU waits in the first round; 0 = off
#ifndef MRG_X3W_MAP
#define MRG_X3W_MAP 0
#endif
#ifndef MRG_X3W_STAGGER
#define MRG_X3W_STAGGER 4
#endif

namespace mrg {

// one wave: rows [roww, roww + 64) x column tiles [tile0, tile0 + NT)
template <int NT, int EPI, bool DUAL>
__device__ __forceinline__ void x3w_wave_tile(const GemmArgs& a, const char* __restrict__ Bp, int ntile, int64_t roww, int tile0,
                                              float* ring, int lane, bool full, int64_t trace_slot) {
  constexpr int MT = 2, WROWS = 64;
  constexpr int SLOT_CH = WROWS * 4;          // 16-byte chunks per ring slot
  constexpr int NA = SLOT_CH / 64;            // DMA instructions per slab (4)
  constexpr int NBL = 3 * NT;                 // B loads per slab
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int li = lane & 31, lh = lane >> 5;
  const int K = a.K1 + a.K2;
  const int nslab = (K + 15) >> 4;

  MRG_X3_STAMP(trace_slot, 0);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // DMA sources: ring chunk f = lane + 64 i holds (row f/4, 4-float column c = (f%4) ^ ((f/16)&3)) of the slab
  const float* arow1[NA]; const float* arow2[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = lane + 64 * i;
    const int64_t row = roww + (f >> 2);
    const int64_t rc = row < a.rows ? row : a.rows - 1;
    arow1[i] = a.A1 + rc * a.K1;
    arow2[i] = DUAL ? a.A2 + rc * a.K2 : nullptr;
  }
  const int acol = 4 * ((lane & 3) ^ ((lane >> 4) & 3));           // the same for every i: 64 i shifts f/16 by a multiple of 4
  auto fetch_a = [&](int slab) {
    const int k = slab * 16 + acol;
    float* dst = ring + (slab % X3_SLOTS) * (SLOT_CH * 4);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float* p;
      if (DUAL) {
        const bool first = k < a.K1;
        const int kk = first ? k : k - a.K1, ld = first ? a.K1 : a.K2;
        p = (first ? arow1[i] : arow2[i]) + (kk + 4 <= ld ? kk : ld - 4);
      } else {
        p = arow1[i] + (k + 4 <= K ? k : K - 4);
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(dst + 64 * i * 4), 16, 0, 0);
    }
  };
  const unsigned lds_ring = (unsigned)(size_t)(lds_ptr_t)ring;
  unsigned a_off[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = 32 * m + li, sw = (r >> 2) & 3;
    a_off[m][0] = (unsigned)((r * 4 + ((2 * lh) ^ sw)) * 16);
    a_off[m][1] = (unsigned)((r * 4 + ((2 * lh + 1) ^ sw)) * 16);
  }
  v4f x[MT][2];
  auto read_a = [&](int slab) {
    const unsigned base = lds_ring + (slab % X3_SLOTS) * (SLOT_CH * 16);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][0]) : "v"(base + a_off[m][0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][1]) : "v"(base + a_off[m][1]));
    }
  };
  // asynchronous register fills, first read behind the matching counted s_waitcnt (see gemm_x3.hpp)
  u32x4 bq[NT][3];
  const unsigned voff = (unsigned)lane * 16u;
  const char* bcol = Bp + (int64_t)tile0 * 3072;
  auto load_b = [&](int n, int slab) {
    const char* sb = bcol + ((int64_t)slab * ntile + n) * 3072;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[n][0]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(bq[n][1]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(bq[n][2]) : "v"(voff), "s"(sb));
  };
  u32x4 ch[MT], cm[MT], cl[MT];

  // ---- prologue: A slabs 0..2 and B slab 0 in flight (the host guarantees nslab >= 4)
#pragma unroll
  for (int s = 0; s < 3; ++s) fetch_a(s);
#pragma unroll
  for (int n = 0; n < NT; ++n) load_b(n, 0);

  // One k-slab; MODE fixes every wait count at compile time:
  //   0 steady state (s + 3 < nslab)   1: s == nslab-3   2: s == nslab-2   3: s == nslab-1
  // In-order vector-memory history when slab s starts: ... A(s) | A(s+1) | A(s+2) | B(s) tiles 0..NT-1   (each A = NA, B(s) = NBL)
  auto slab = [&](auto mode_c, int s) {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool has_next = MODE != 3;
    constexpr bool do_dma = MODE == 0;
    // A(s) has landed once at most { A(s+1), A(s+2), B(s) } are outstanding (the A slabs that exist)
    if (MODE <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NA + NBL) : "memory");
    else if (MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBL) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBL) : "memory");
    read_a(s);
    if (do_dma && !((MRG_X3_DBG & 4) && s > 0)) fetch_a(s + 3);                       // slot (s+3)%4 held slab s-1, whose fragments were consumed a slab ago
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if ((MRG_X3_DBG & 512) && s == 0) MRG_X3_STAMP(trace_slot, 1);
    if (!((MRG_X3_DBG & 64) && s > 0))
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f& v = x[m][q >> 1];
        unsigned h, mm, l;
        if (q & 1) split_pair(v.z, v.w, h, mm, l);
        else split_pair(v.x, v.y, h, mm, l);
        ch[m][q] = h; cm[m][q] = mm; cl[m][q] = l;
      }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      // B(s) tile n: younger = rest of B(s), this slab's A DMA, the B(s+1) tiles issued so far
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1) + NA) : "memory");
      else if (MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1 - n)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, bq[n][0]), Bm = __builtin_bit_cast(bf16x8, bq[n][1]),
                   Bl = __builtin_bit_cast(bf16x8, bq[n][2]);
#define MRG_X3W_TERM(AP, BP)                                                                            \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                          \
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AP[m]), BP, acc[m][n], 0, 0, 0)
      MRG_X3W_TERM(cm, Bm);
      MRG_X3W_TERM(cl, Bh);
      MRG_X3W_TERM(ch, Bl);
      MRG_X3W_TERM(cm, Bh);
      MRG_X3W_TERM(ch, Bm);
      MRG_X3W_TERM(ch, Bh);
#undef MRG_X3W_TERM
      __builtin_amdgcn_sched_barrier(0);
      if (has_next && !((MRG_X3_DBG & 32) && s > 0)) load_b(n, s + 1);
    }
  };
  for (int s = 0; s + 3 < nslab; ++s) slab(std::integral_constant<int, 0>{}, s);
  slab(std::integral_constant<int, 1>{}, nslab - 3);
  slab(std::integral_constant<int, 2>{}, nslab - 2);
  slab(std::integral_constant<int, 3>{}, nslab - 1);
  MRG_X3_STAMP(trace_slot, 2);
  if ((MRG_X3_DBG & 1) && acc[0][0][0] != 123.456f) return;
#pragma unroll
  for (int m = 0; m < MT; ++m) gemm_epilogue<NT, EPI>(a, acc[m], roww + m * 32, tile0 * 32, li, lh, full);
  if (MRG_X3_DBG & 512) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MRG_X3_STAMP(trace_slot, 3); }
}

// grid.x = 2 * (row blocks rounded up to 8): id -> (xcd = id % 8, j = id / 8): column half j & 1 of row block (j >> 1) * 8 + xcd,
// so that the two halves of a row block run back to back on ONE XCD.  grid.y = blocks of seven column tiles.
template <int EPI, bool DUAL>
__global__ __launch_bounds__(X3_THREADS, 2) void rowgemm_x3w_k(GemmArgs a, const char* __restrict__ Bp, int ntile, int row_blocks) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int id = blockIdx.x, j = id >> 3;
#if MRG_X3W_MAP == 1
  // workgroups j and j + 32 of an XCD share a CU (breadth-first placement): give them the two halves of one row block
  const int rb = ((j >> 6) * 32 + (j & 31)) * 8 + (id & 7);
  const int half = (j >> 5) & 1;
#else
  const int rb = (j >> 1) * 8 + (id & 7);
  const int half = j & 1;
#endif
  if (rb >= row_blocks) return;
  const int tile0 = blockIdx.y * 7 + half * 4;
  if (tile0 * 32 >= a.N) return;                       // N <= 128 within this block of seven: the second half has no columns
  const int64_t row0 = (int64_t)rb * 256;
  const int64_t roww = row0 + wave * 64;
  if (roww >= a.rows) return;                          // wave-uniform; no barrier anywhere
  // The two workgroups of a CU must not move in lock step (both loading, both multiplying, both storing): of the
  // workgroups of the first round the one that sits in the upper half of the CU's LDS starts late.
  if (MRG_X3W_STAGGER > 0 && id < 512) {
    const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6);          // HW_REG_LDS_ALLOC[7:0]: LDS_BASE
    if (lds_base != 0)
      for (int i = 0; i < MRG_X3W_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
  float* ring = smem + wave * (X3_SLOTS * 64 * 4 * 4);
  const bool full = row0 + 256 <= a.rows;
  [[maybe_unused]] const int64_t trace_slot = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
  if (half == 0) x3w_wave_tile<4, EPI, DUAL>(a, Bp, ntile, roww, tile0, ring, lane, full, trace_slot);
  else x3w_wave_tile<3, EPI, DUAL>(a, Bp, ntile, roww, tile0, ring, lane, full, trace_slot);
}

// operands the kernel is built for: B split for blocks of seven tiles (gemm_pick_nt == 7), enough rows to fill the chip twice
inline bool x3w_eligible(const GemmArgs& a) { return x3_eligible(a) && gemm_pick_nt(a.N) == 7 && a.rows >= 128 * 512; }

template <int EPI>
inline int launch_rowgemm_x3w(GemmArgs a, const void* Bp, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int ntile = x3_tiles(a.N, 7);
  const int row_blocks = (int)((a.rows + 255) / 256);
  const int rb_round = MRG_X3W_MAP == 1 ? 256 : 8;
  dim3 grid((unsigned)(2 * ((row_blocks + rb_round - 1) / rb_round) * rb_round), (unsigned)(ntile / 7));
  const size_t lds = (size_t)(X3_THREADS / 64) * X3_SLOTS * 64 * 64;                 // 64 KB: two workgroups per CU
  static bool attr_done[2] = {};
#define MRG_GOW(DV)                                                                                                   \
  do {                                                                                                                \
    if (!attr_done[DV ? 1 : 0]) {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3w_k<EPI, DV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_done[DV ? 1 : 0] = true;                                                                                   \
    }                                                                                                                 \
    hipLaunchKernelGGL((rowgemm_x3w_k<EPI, DV>), grid, dim3(X3_THREADS), lds, st, a, (const char*)Bp, ntile, row_blocks); \
  } while (0)
  if (a.K2 > 0) MRG_GOW(true); else MRG_GOW(false);
#undef MRG_GOW
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

}  // namespace mrg
