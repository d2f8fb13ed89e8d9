// Split-bf16 row GEMM with the weight operand staged through LDS: 128-row workgroups, two per CU.
//
// What rounds 2 and 3 measured on the wave-autonomous kernels (gemm_x3.hpp, gemm_x3w.hpp; profiles/r2_rowgemm_rounds.txt,
// profiles/r3_rowgemm_epilogue.txt): a wave that keeps the pre-split B fragments of a whole k-slab in registers (84 of them for
// seven column tiles) next to 224 accumulators owns its SIMD's register file, so a CU runs one workgroup whose prologue,
// k-loop and store tail never overlap (MFMA busy 36 % of the wave cycles); and every wave streams the whole weight
// (21 KB per slab) from L2 by itself -- 1.1 MB per 256-row block against 0.2 MB of activations.
//
// Here the four waves of a workgroup SHARE the weight slab: it travels L2 -> LDS once per workgroup by LDS-DMA (double
// buffered, one barrier per slab) and the MFMA operands are read from there (ds_read_b128, conflict free: the pre-split
// layout is already fragment order), so a wave needs only the fragments of the tile it is multiplying.  A wave owns 32 rows
// x NT column tiles (NT * 16 accumulators); the activations are read straight into registers in fragment order (two
// global_load_dwordx4 per slab and lane, two slabs ahead) and split in the shadow of the previous slab's MFMAs.  That is
// <= 256 registers and 3 * NT * 3 KB of LDS per workgroup (the weight slabs two ahead, the activations three ahead -- a
// wave's vector-memory operations complete in issue order, so the two depths are coupled): two workgroups share a CU, and
// while one waits at its barrier, for its first slabs or for its stores, the other multiplies.
// LDS-DMA addresses its destination through M0[15:0]: everything it writes lies below 64 KiB of the workgroup's LDS
// (3 x 21 KB here).
#pragma once
#include "gemm_x3.hpp"

// lab switches (tools/x3s_dbg_lab.hip, timing only -- wrong results): 1 no epilogue, 2 no A loads after the prologue,
// 4 no B DMA after the prologue, 8 no barriers, 16 phase stamps, 64 no split arithmetic, 128 no fragment reads after the first slab,
// 256 nothing but the epilogue, 512 the tile's stores paced through the k-loop (use with 1).  (A "no splits" switch left asynchronous register fills unconsumed and
// faulted: every asm load's registers must be read after its wait -- see gemm_x3.hpp.)
#ifndef MRG_X3S_DBG
#define MRG_X3S_DBG 0
#endif
// lab: s_sleep(127) repetitions (~3.5 us each) the SECOND workgroup of every CU waits in the first round (blocks 256..511 share
// their CUs with blocks 0..255 under round-robin dispatch): desynchronises the two workgroups of a CU.  0 = off.
#ifndef MRG_X3S_STAGGER
#define MRG_X3S_STAGGER 0
#endif
// 1 (round 4 lab; bit-identical, measured equal): the weight fragments of a slab's FIRST column-tile pair are read during the previous slab's LAST pair -- the
// barrier that publishes slab s + 1 stands in front of slab s's last pair instead of behind it -- so that the LDS round trip and
// the barrier's skew no longer open every slab (round 4; 0 = round 3's schedule).  Needs an even number of tile pairs per slab.
#ifndef MRG_X3S_PIPE
#define MRG_X3S_PIPE 0
#endif
// 1: the slabs with s + 4 < nslab run a branch-free instance of the slab body with a running activation pointer (round 4 lab;
// bit-identical, measured equal: 0.407-0.414 vs 0.410-0.414 ms at rows 558 771 -- profiles/r4_rowgemm_phases.txt).  0 (default) = round 3's form.
#ifndef MRG_X3S_STEADY
#define MRG_X3S_STEADY 0
#endif

namespace mrg {

// lab switch 16: per-wave phase stamps (shader clock, s_memtime) -> mrg_x3s_trace[wave slot * 24 + i]:
// 0 start, 1 first slab, 2 + s = end of slab s (s < 16), 18 k-loop done, 19 stores issued, 20 stores landed, 21 HW_ID | XCC_ID << 32,
// 22 / 23 the 100 MHz clock at the end / start
#if MRG_X3S_DBG & 16
__device__ unsigned long long* mrg_x3s_trace;
#define MRG_X3S_STAMP(i) do { if (lane == 0) mrg_x3s_trace[trace_slot * 24 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MRG_X3S_STAMP(i) do { } while (0)
#endif

// TR: the MFMA operands change places (weight fragment first, activation fragment second), which transposes the accumulator
// tile -- a lane then holds ONE row's columns 8g + 4 lh + {0..3} in registers 4g..4g+3 -- so that the epilogue loads and stores
// 16 bytes per lane (gemm_epilogue_tr below): 4 * NT store instructions per strip instead of 16 * NT.  Same products, same
// order of accumulation: bit-identical results.
#ifndef MRG_X3S_WPS4
#define MRG_X3S_WPS4 2      // lab: waves per SIMD the NT <= 4 instances are compiled for (3 = at most 168 registers)
#endif
template <int NT, int EPI, bool DUAL, bool TR>
__global__ __launch_bounds__(256, (NT <= 4 ? MRG_X3S_WPS4 : 2)) void rowgemm_x3s_k(GemmArgs a, const char* __restrict__ Bp, int ntile) {
  constexpr int GBM = 128;
  constexpr int NCH = NT * 3;                   // 1 KB chunks (64 lanes x 16 B) of one pre-split B slab of this column block
  constexpr int BSLAB = NCH * 1024;
  constexpr int NBW = (NCH + 3) / 4;            // DMA instructions per wave and slab
  extern __shared__ __align__(16) char smem_b[];     // [3][BSLAB]: 63 KB for seven column tiles, below the DMA's 64 KiB
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int64_t row0 = (int64_t)blockIdx.x * GBM;
  int sg = 0;
  if (a.grp.n > 0) sg = ((int)blockIdx.x >= a.grp.tile0[1] ? 1 : 0) + ((int)blockIdx.x >= a.grp.tile0[2] ? 1 : 0);
  sg = __builtin_amdgcn_readfirstlane(sg);
  const char* __restrict__ Bq = Bp + (int64_t)sg * a.grp.bp_stride;
  if (a.grp.n > 0) {                                     // grouped launch, as in rowgemm_x3_k (constant indices only)
#define MRG_PICK(F) (sg == 0 ? a.grp.F[0] : (sg == 1 ? a.grp.F[1] : a.grp.F[2]))
    row0 = MRG_PICK(lo) + (int64_t)((int)blockIdx.x - MRG_PICK(tile0)) * GBM;
    a.rows = MRG_PICK(hi);
    a.bias = MRG_PICK(bias);
    a.scale = MRG_PICK(scale);
    if (!MRG_PICK(use_rowscale)) a.rowscale = nullptr;
#undef MRG_PICK
  }
  const int64_t roww = row0 + wave * 32;
  const int col0 = blockIdx.y * (NT * 32);
  const int K = a.K1 + a.K2;
  const int nslab = (K + 15) >> 4;
  if (MRG_X3S_STAGGER > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = 0; i < MRG_X3S_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }

  [[maybe_unused]] const int64_t trace_slot = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
  MRG_X3S_STAMP(0);
#if MRG_X3S_DBG & 16
  if (lane == 0) mrg_x3s_trace[trace_slot * 24 + 23] = __builtin_amdgcn_s_memrealtime();
#endif
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

#if MRG_X3S_DBG & 256
  // lab: nothing but the epilogue (what the store pattern alone costs at this grid and residency)
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = (float)(lane + n + r);
  if constexpr (EPI == EPI_SEGMAX) gemm_epilogue_segmax<NT>(a, acc, roww, col0, li, lh);
  else if constexpr (EPI == EPI_SEGSUM) gemm_epilogue_segsum<NT>(a, acc, roww, col0, li, lh);
  else if constexpr (TR) gemm_epilogue_tr<NT, EPI>(a, acc, roww, col0, li, lh);
  else gemm_epilogue<NT, EPI>(a, acc, roww, col0, li, lh, row0 + GBM <= a.rows);
  return;
#endif
#if MRG_X3S_DBG & 512
  float* paced_c = (roww + 32 <= a.rows && col0 + NT * 32 <= a.ldc + 31) ? a.C + (roww + 4 * lh) * a.ldc + (col0 + li < a.N ? col0 + li : a.N - 1) : nullptr;
#endif
  // ---- A: this lane's fragment of a slab = row li, k = slab * 16 + lh * 8 + {0..3, 4..7}: two 16-byte loads
  int64_t rc = roww + li < a.rows ? roww + li : a.rows - 1;
  if (rc < 0) rc = 0;
  if (a.row_index) rc = a.row_index[rc];                 // gathered rows (EPI_SEGMAX / EPI_SEGSUM: edges in destination order)
  const float* ar1 = a.A1 + rc * a.K1;
  const float* ar2 = a.A2 + rc * a.K2;
  auto a_ptr = [&](int k) -> const float* {
    if (DUAL) {
      const bool first = k < a.K1;
      const int kk = first ? k : k - a.K1, ld = first ? a.K1 : a.K2;
      return (first ? ar1 : ar2) + (kk + 4 <= ld ? kk : ld - 4);
    }
    return ar1 + (k + 4 <= K ? k : K - 4);               // beyond K: any finite values, the weight's rows there are zero
  };
  // asynchronous register fills, first read behind the matching counted s_waitcnt (see gemm_x3.hpp)
  v4f xr[3][2];                                          // raw fragments: a ring of three slabs
  auto load_a = [&](int slab, v4f (&x)[2]) {
    const int sl = slab < nslab ? slab : nslab - 1;      // beyond the end: re-read the last slab (an asynchronous fill is never conditional)
    const int k = sl * 16 + lh * 8;
    const float* p0 = a_ptr(k);
    const float* p1 = a_ptr(k + 4);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[0]) : "v"(p0));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[1]) : "v"(p1));
  };
  // steady state (slab + 3 < nslab - 1, single source): no clamp is needed, and the address is a running pointer that advances by one
  // slab per call -- two VALU instructions instead of thirteen
  const float* pa = ar1 + 3 * 16 + lh * 8;
  auto load_a_run = [&](v4f (&x)[2]) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[0]) : "v"(pa));
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(x[1]) : "v"(pa));
    pa += 16;
  };
  // ---- B: the slab's NCH chunks, NBW per wave (the last wave repeats the last chunk: same bytes to the same place)
  const char* bcol = Bq + (int64_t)blockIdx.y * NT * 3072;
  auto fetch_b = [&](int slab, int buf) {
    const char* src = bcol + (int64_t)slab * ntile * 3072;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      int c = wave * NBW + i;
      c = c < NCH ? c : NCH - 1;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + c * 1024 + lane * 16), (lds_ptr_t)(smem_b + buf * BSLAB + c * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem_b + (unsigned)lane * 16u;
  u32x4 bq2[2][2][3];                                    // [double buffer][tile of the pair][plane]
  bool lab_reads = true;                                 // lab switch 128: no fragment reads after the first slab
  auto read_b = [&](int n, int buf, u32x4 (&q)[3]) {
    if ((MRG_X3S_DBG & 128) && !lab_reads) return;
    const unsigned ad = lds0 + (unsigned)(buf * BSLAB + n * 3072);
    asm volatile("ds_read_b128 %0, %1" : "=v"(q[0]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q[1]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(q[2]) : "v"(ad));
  };
  u32x4 ch, cm, cl, nh, nm, nl;
  auto split_pair_of = [&](const v4f (&x)[2], int q, u32x4& H, u32x4& M, u32x4& L) {     // q = 0..3: floats 2q, 2q + 1 of the 8
    const v4f& v = x[q >> 1];
    unsigned h, m, l;
    if (MRG_X3S_DBG & 64) { h = __builtin_bit_cast(unsigned, (q & 1) ? v.z : v.x); m = __builtin_bit_cast(unsigned, (q & 1) ? v.w : v.y); l = h; }   // lab: no split
    else if (q & 1) split_pair(v.z, v.w, h, m, l); else split_pair(v.x, v.y, h, m, l);
    H[q] = h; M[q] = m; L[q] = l;
  };
  auto nb_issued = [&](int j) { return (j >= -2 && j + 2 < nslab) ? NBW : 0; };   // B DMAs issued at the top of slab j (j < 0: prologue)
  // lab switch 512: stores a wave issues inside slab j (they count in vmcnt like the loads); 0 in the product
  auto paced_in = [&](int j) -> int {
#if MRG_X3S_DBG & 512
    if (paced_c == nullptr || j < 0 || j >= nslab) return 0;
    const int np = (NT + 1) / 2, per = (16 * NT + nslab * np - 1) / (nslab * np);
    int n = 0;
    for (int pp = 0; pp < np; ++pp)
      for (int q = 0; q < 3; ++q) n += (q < per && (j * np + pp) * per + q < 16 * NT) ? 1 : 0;
    return n;
#else
    (void)j;
    return 0;
#endif
  };
  constexpr int NP = (NT + 1) / 2;                          // column-tile pairs per slab
  constexpr bool PIPE = MRG_X3S_PIPE && (NP % 2 == 0);      // the pair buffers alternate across slabs: their parity must not depend on s
  auto nbw = [&](int j) { return (j >= 1 && j < nslab) ? NBW : 0; };              // PIPE: DMAs of B(j) that stand BEHIND older A loads (B(0) leads)

  if constexpr (PIPE) {
    // issue order of a wave:  A(0) B(0) A(1) A(2) B(1) | A(3) B(2) | A(4) B(3) | ...   (slab s issues A(s+3) at its top and
    // B(s+2) behind the barrier in front of its last pair)
    load_a(0, xr[0]);
    fetch_b(0, 0);
    load_a(1, xr[1]);
    load_a(2, xr[2]);
    if (nslab > 1) fetch_b(1, 1);
    wait_vmcnt(4 + nbw(1));                                  // A(0) and this wave's share of B(0) have landed
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair_of(xr[0], q, ch, cm, cl);
    __builtin_amdgcn_s_barrier();                            // everybody's share of B(0) is in LDS
    read_b(0, 0, bq2[0][0]);                                 // slab 0's first pair
    if (NT > 1) read_b(1, 0, bq2[0][1]);
  } else {
  // ---- prologue, in the steady state's issue order: A(0) | B(0) A(1) | B(1) A(2)
  load_a(0, xr[0]);
  fetch_b(0, 0);
  load_a(1, xr[1]);
  if (nslab > 1) fetch_b(1, 1);
  load_a(2, xr[2]);
  wait_vmcnt(2 + nb_issued(-1) + 2);                         // A(0) and this wave's share of B(0) have landed
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair_of(xr[0], q, ch, cm, cl);
  __builtin_amdgcn_s_barrier();                              // everybody's share of B(0) is in LDS
  }
  MRG_X3S_STAMP(1);

  // One k-slab; R = s % 3 at compile time (ring positions of the raw fragments and of the B buffers).
  // In-order vector-memory history of a wave at the top of slab s:  ... A(s+1) | B(s+1) A(s+2)      (B(s) in LDS: barrier)
  //   top: issue B(s+2) into buffer (s+2) % 3 -- read last during slab s-1, every wave is past that barrier -- and A(s+3) into
  //        the raw registers slab s-1 split from;
  //   the splits need A(s+1): younger = B(s+1) A(s+2) B(s+2) A(s+3);
  //   end: B(s+1) must be in LDS before the barrier: younger = A(s+2) B(s+2) A(s+3).
#define MRG_MM(AF, BF, C) (TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF, AF, C, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, BF, C, 0, 0, 0))
  // ST (steady state, s + 4 < nslab): every condition and every wait count below is a compile-time constant -- the scalar branches of
  // the run-time form cost a wave that has its SIMD's matrix pipe to itself about one MFMA slot each, ~450 of a slab's 1 800 cycles
  // with nothing but MFMAs left in the loop (profiles/r4_rowgemm_phases.txt)
  auto slab = [&](auto r_c, auto st_c, int s) {
    constexpr int R = decltype(r_c)::value;
    constexpr bool ST = decltype(st_c)::value;
    const bool has_next = ST ? true : (s + 1 < nslab);
    if (!PIPE && (ST || s + 2 < nslab) && !(MRG_X3S_DBG & 4)) fetch_b(s + 2, (R + 2) % 3);
    if (!(MRG_X3S_DBG & 2)) {
      if constexpr (ST && !DUAL) load_a_run(xr[R]);
      else load_a(s + 3, xr[R]);
    }
    // Column tiles in PAIRS: the twelve MFMAs of a pair alternate between its two accumulators (a dependent MFMA issued back to
    // back waits for its predecessor's result; with another accumulator's MFMA in between the pipe stays busy) and the VALU
    // instructions of the A split are spread between them (sched_group_barrier: 1 MFMA, then up to 3 VALU) instead of
    // standing in front of the MFMAs.  Each accumulator still receives its six terms in the same order: bit-identical results.
    constexpr int SPP = (4 + NP - 1) / NP;                    // split pairs handled in the shadow of one tile pair
    if constexpr (PIPE) {
      // A(s+1) is split during this slab; younger than it: B(s) [unless s == 0: B(0) leads the prologue] A(s+2) B(s+1) A(s+3)
      if (has_next && !(MRG_X3S_DBG & 6)) wait_vmcnt(nbw(s) + 2 + nbw(s + 1) + 2);
      if (has_next && (MRG_X3S_DBG & 6)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    read_b(0, R, bq2[0][0]);
    if (NT > 1) read_b(1, R, bq2[0][1]);
    if constexpr (ST && !(MRG_X3S_DBG & 6)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBW + 4) : "memory");
    else if (has_next && !(MRG_X3S_DBG & 6)) wait_vmcnt(nb_issued(s - 1) + 2 + nb_issued(s) + 2 + paced_in(s - 2) + paced_in(s - 1));
    if (has_next && (MRG_X3S_DBG & 6)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      constexpr int dummy = 0; (void)dummy;
      const int n0 = 2 * pp, n1 = 2 * pp + 1;
      if (PIPE && pp + 1 == NP) {
        // in front of the last pair: publish slab s + 1, start the DMA of slab s + 2 into the buffer slab s - 1 was read from
        // (every wave is past that slab), and read slab s + 1's first pair
        if (has_next) {
          if (MRG_X3S_DBG & 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // this wave's share of B(s+1): only A(s+3) is younger
          if (!(MRG_X3S_DBG & 8)) __builtin_amdgcn_s_barrier();
          if (s + 2 < nslab && !(MRG_X3S_DBG & 4)) fetch_b(s + 2, (R + 2) % 3);
          read_b(0, (R + 1) % 3, bq2[(pp + 1) & 1][0]);
          if (NT > 1) {
            read_b(1, (R + 1) % 3, bq2[(pp + 1) & 1][1]);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          }
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      } else if (pp + 1 < NP) {
        read_b(n0 + 2, R, bq2[(pp + 1) & 1][0]);
        if (n1 + 2 < NT) {
          read_b(n1 + 2, R, bq2[(pp + 1) & 1][1]);
          asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // the reads just issued may still be in flight
        } else {
          asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
#pragma unroll
        for (int q = pp * SPP; q < (pp + 1) * SPP && q < 4; ++q) split_pair_of(xr[(R + 1) % 3], q, nh, nm, nl);
      }
      const bf16x8 Ah = __builtin_bit_cast(bf16x8, ch), Am = __builtin_bit_cast(bf16x8, cm), Al = __builtin_bit_cast(bf16x8, cl);
      const bf16x8 Bh0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][0]), Bm0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][1]),
                   Bl0 = __builtin_bit_cast(bf16x8, bq2[pp & 1][0][2]);
      if (n1 < NT) {
        const bf16x8 Bh1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][0]), Bm1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][1]),
                     Bl1 = __builtin_bit_cast(bf16x8, bq2[pp & 1][1][2]);
        // small terms first, the leading term last (same order per accumulator as rowgemm_x3_k)
        // (lab switch 1024, timing only: the three 2^-16 terms are left out -- what a two-plane operand format would issue)
        if (!(MRG_X3S_DBG & 1024)) {
          acc[n0] = MRG_MM(Am, Bm0, acc[n0]);
          acc[n1] = MRG_MM(Am, Bm1, acc[n1]);
          acc[n0] = MRG_MM(Al, Bh0, acc[n0]);
          acc[n1] = MRG_MM(Al, Bh1, acc[n1]);
          acc[n0] = MRG_MM(Ah, Bl0, acc[n0]);
          acc[n1] = MRG_MM(Ah, Bl1, acc[n1]);
        }
        acc[n0] = MRG_MM(Am, Bh0, acc[n0]);
        acc[n1] = MRG_MM(Am, Bh1, acc[n1]);
        acc[n0] = MRG_MM(Ah, Bm0, acc[n0]);
        acc[n1] = MRG_MM(Ah, Bm1, acc[n1]);
        acc[n0] = MRG_MM(Ah, Bh0, acc[n0]);
        acc[n1] = MRG_MM(Ah, Bh1, acc[n1]);
      } else {
        if (!(MRG_X3S_DBG & 1024)) {
          acc[n0] = MRG_MM(Am, Bm0, acc[n0]);
          acc[n0] = MRG_MM(Al, Bh0, acc[n0]);
          acc[n0] = MRG_MM(Ah, Bl0, acc[n0]);
        }
        acc[n0] = MRG_MM(Am, Bh0, acc[n0]);
        acc[n0] = MRG_MM(Ah, Bm0, acc[n0]);
        acc[n0] = MRG_MM(Ah, Bh0, acc[n0]);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // ... then up to three VALU
      }
      __builtin_amdgcn_sched_barrier(0);
#if MRG_X3S_DBG & 512
      // lab (timing only, wrong results): the tile's 16 * NT dword stores PACED through the k-loop -- two or three behind every tile
      // pair, every (tile, register) address of the strip written once per tile -- instead of the burst behind it (switch 1 drops that)
      if (paced_c != nullptr) {
        const int slot = s * NP + pp;                       // 0 .. nslab * NP - 1
        const int per = (16 * NT + nslab * NP - 1) / (nslab * NP);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int idx = slot * per + j;                   // which (tile, register) of the strip: run-time, only the address depends on it
          if (j < per && idx < 16 * NT) {
            const int tn = idx >> 4, tr = idx & 15;
            paced_c[(int64_t)((tr & 3) + 8 * (tr >> 2)) * a.ldc + tn * 32] = acc[(pp * 2 + (j & 1)) % NT][(R * 4 + j) & 15];
          }
        }
      }
#endif
    }
    if (NT < 4 && has_next) {
#pragma unroll
      for (int q = NT; q < 4; ++q) split_pair_of(xr[(R + 1) % 3], q, nh, nm, nl);
    }
    if (has_next) {
      ch = nh; cm = nm; cl = nl;
      if constexpr (!PIPE) {
      if constexpr (ST && !(MRG_X3S_DBG & 6)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 4) : "memory");
      else if (!(MRG_X3S_DBG & 6)) wait_vmcnt(2 + nb_issued(s) + 2 + paced_in(s - 1) + paced_in(s));     // this wave's share of B(s+1) is in LDS
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(MRG_X3S_DBG & 8)) __builtin_amdgcn_s_barrier();   // ... and everybody's; all reads of this slab's buffer are done
      }
    }
#if MRG_X3S_DBG & 16
    if (s < 16) MRG_X3S_STAMP(2 + s);
#endif
    if (MRG_X3S_DBG & 128) lab_reads = false;
  };
#undef MRG_MM
  int s = 0;
  constexpr bool STEADY = MRG_X3S_STEADY && !PIPE;
  if constexpr (STEADY) {
    for (; s + 6 < nslab; s += 3) {             // all three slabs satisfy s' + 4 < nslab
      slab(std::integral_constant<int, 0>{}, std::true_type{}, s);
      slab(std::integral_constant<int, 1>{}, std::true_type{}, s + 1);
      slab(std::integral_constant<int, 2>{}, std::true_type{}, s + 2);
    }
  }
  for (; s + 2 < nslab; s += 3) {
    slab(std::integral_constant<int, 0>{}, std::false_type{}, s);
    slab(std::integral_constant<int, 1>{}, std::false_type{}, s + 1);
    slab(std::integral_constant<int, 2>{}, std::false_type{}, s + 2);
  }
  if (s < nslab) { slab(std::integral_constant<int, 0>{}, std::false_type{}, s); ++s; }
  if (s < nslab) { slab(std::integral_constant<int, 1>{}, std::false_type{}, s); ++s; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tail's unused A fills: their registers must stay until they land
  asm volatile("" :: "v"(xr[0][0]), "v"(xr[0][1]), "v"(xr[1][0]), "v"(xr[1][1]), "v"(xr[2][0]), "v"(xr[2][1]));

  MRG_X3S_STAMP(18);
  if ((MRG_X3S_DBG & 1) && acc[0][0] != 123.456f) return;
  if constexpr (EPI == EPI_SEGMAX) gemm_epilogue_segmax<NT>(a, acc, roww, col0, li, lh);
  else if constexpr (EPI == EPI_SEGSUM) gemm_epilogue_segsum<NT>(a, acc, roww, col0, li, lh);
  else if constexpr (TR) gemm_epilogue_tr<NT, EPI>(a, acc, roww, col0, li, lh);
  else gemm_epilogue<NT, EPI>(a, acc, roww, col0, li, lh, row0 + GBM <= a.rows);
#if MRG_X3S_DBG & 16
  MRG_X3S_STAMP(19);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MRG_X3S_STAMP(20);
  if (lane == 0) {
    mrg_x3s_trace[trace_slot * 24 + 21] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))            // HW_REG_HW_ID
                                        | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);   // HW_REG_XCC_ID
    mrg_x3s_trace[trace_slot * 24 + 22] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

inline bool x3s_eligible(const GemmArgs& a) { return x3_eligible(a) && a.rows > 0; }

// Bp: the split of B prepared by launch_bsplit(..., nt = gemm_pick_nt(a.N), ...)
template <int EPI>
inline int launch_rowgemm_x3s(GemmArgs a, const void* Bp, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int nt = gemm_pick_nt(a.N);
  const int ntile = x3_tiles(a.N, nt);
  const int gbm = 128;
  if (a.grp.n > 0) {
    a.grp.tile0[0] = 0;
    for (int i = 0; i < 3; ++i) {
      const int64_t r = i < a.grp.n && a.grp.hi[i] > a.grp.lo[i] ? a.grp.hi[i] - a.grp.lo[i] : 0;
      a.grp.tile0[i + 1] = a.grp.tile0[i] + (int)((r + gbm - 1) / gbm);
    }
    if (a.grp.tile0[3] == 0) return MRG_OK;
  }
  dim3 grid((unsigned)(a.grp.n > 0 ? a.grp.tile0[3] : (a.rows + gbm - 1) / gbm), (unsigned)(ntile / nt));
  size_t lds = (size_t)3 * nt * 3 * 1024;
#if MRG_X3S_DBG
  if (getenv("MRG_X3S_LDS_EXTRA")) lds += (size_t)atoi(getenv("MRG_X3S_LDS_EXTRA"));     // lab: one workgroup per CU
#endif
  bool tr = false;
  if constexpr (EPI != EPI_SEGMAX && EPI != EPI_SEGSUM) tr = gemm_epi_mode() == 2 && gemm_epilogue_tr_ok<EPI>(a);
#define MRG_GOS3(NTV, DV, TV)                                                                                         \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3s_k<NTV, EPI, DV, TV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((rowgemm_x3s_k<NTV, EPI, DV, TV>), grid, dim3(256), lds, st, a, (const char*)Bp, ntile);       \
  } while (0)
#define MRG_GOS2(NTV, DV)                                                                                             \
  do {                                                                                                                \
    if constexpr (EPI == EPI_SEGMAX || EPI == EPI_SEGSUM) MRG_GOS3(NTV, DV, false);                                   \
    else { if (tr) MRG_GOS3(NTV, DV, true); else MRG_GOS3(NTV, DV, false); }                                          \
  } while (0)
#define MRG_GOS(NTV) do { if (a.K2 > 0) MRG_GOS2(NTV, true); else MRG_GOS2(NTV, false); } while (0)
  switch (nt) {
    case 1: MRG_GOS(1); break;
    case 2: MRG_GOS(2); break;
    case 4: MRG_GOS(4); break;
    default: MRG_GOS(7); break;
  }
#undef MRG_GOS
#undef MRG_GOS2
#undef MRG_GOS3
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

}  // namespace mrg
