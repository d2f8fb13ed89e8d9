// LAB ONLY since round 4 (it lost to gemm_x3s.hpp and ran only behind mrg_gemm_set_mode(3); build with
// -I mr-gnas_amd/csrc -I tools/lab, see tools/gemm_x3_lab.hip).
// Persistent form of the split-core row GEMM (gemm_x3.hpp) with transposed accumulators.
//
// What the lab measured on the round-1 kernel (tools/gemm_x3_lab.hip, rows 272 115, N = 200): K = 200 ran 0.236 ms of
// which 0.090 ms were the epilogue (0.146 ms with the stores compiled out); K = 400: 0.366 / 0.307 ms.  Removing the A
// split, the A DMA, the B loads or the fragment reads from the loop changed nothing: the k-loop already runs at the
// matrix pipe's pace, but a wave (one per SIMD, 224 accumulator registers) executed prologue -> k-loop -> epilogue
// strictly in sequence and then retired: HBM latency of the first slabs and 224 dword stores per tile (16 KB in flight
// at most: vmcnt counts 64 instructions) were fully exposed, 4-5 times per CU.
//
// This kernel removes both exposures:
//   * accumulators are TRANSPOSED: the weight fragment is the MFMA's A operand and the activation fragment its B
//     operand (both fragment layouts are the same: lane = index, 8 consecutive k), so D[i = column][j = row]: a lane
//     owns ONE output row and 4 consecutive columns per register quad -> the epilogue is 4 float4 loads / stores per
//     32x32 tile instead of 16 dword ones (56 instead of 224 store instructions per wave tile at N = 200: they all fit
//     in flight), and the gate / accumulate inputs are read as float4 as well;
//   * the wave is PERSISTENT: it walks its wave tiles (32*MT rows) with ONE continuous DMA / B-load pipeline -- the A slabs of
//     the next tile's first three k-steps are fetched during the current tile's last three and the B fragments of its
//     k-step 0 during the last one, so after the epilogue the next tile's MFMAs start at once.
// vmcnt bookkeeping stays the round-1 kernel's (waits count the younger operations, fixed at compile time); behind each
// epilogue the wave waits for vmcnt(0) once -- the look-ahead loads are older than the stores and have long landed, so only
// the tail of the stores is exposed.  Two rules this kernel learnt the hard way (both faulted on the GPU):
//   * an LDS-DMA addresses its destination through M0[15:0]: everything it writes must lie below 64 KiB of LDS;
//   * a register that an asynchronous asm load writes must stay allocated until the load has returned: no load whose
//     result is never consumed (the register is reused and overwritten later), no conditional asm definition (the join
//     copies the not-yet-arrived register).
#pragma once
#include "gemm_x3.hpp"
#include "gemm_x3w.hpp"
#include "gemm_x3s.hpp"

namespace mrg {

typedef float v4f_t __attribute__((ext_vector_type(4)));

template <int EPI> struct X3pEpiOps {      // lower bound of vector-memory instructions one FULL column tile of one row tile issues
  static constexpr int value = 4;          // 4 float4 stores (the gate's S loads / aux stores, the accumulate loads only add to it)
};

template <int NT, int MT, int EPI, bool DUAL, int ACT>       // ACT: MRG_ACT_* of EPI_BIAS_ACT (one kernel per activation: a run-time
                                                             // choice between three inlined epilogues made hipcc shuffle the 224
                                                             // accumulators between register ranges and spill ~110 registers)
__global__ __launch_bounds__(X3_THREADS, 1) void rowgemm_x3p_k(GemmArgs a, const char* __restrict__ Bp, int ntile, int total_wtiles) {
  constexpr int WROWS = 32 * MT;
  constexpr int SLOT_CH = WROWS * 4;          // 16-byte chunks per ring slot
  constexpr int NA = SLOT_CH / 64;            // DMA instructions per slab
  constexpr int NBL = 3 * NT;                 // B loads per slab
  constexpr int NPAIR = MT * 4;
  constexpr int PP = NT > 1 ? (NPAIR + NT - 2) / (NT - 1) : NPAIR;
  constexpr int RING_F = X3_SLOTS * SLOT_CH * 4;                 // floats per wave ring
  extern __shared__ __align__(16) float smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wstride = gridDim.x * (X3_THREADS / 64);
  int t = blockIdx.x * (X3_THREADS / 64) + wave;
  if (t >= total_wtiles) return;                                // waves share nothing: no barrier anywhere
  const int col0 = blockIdx.y * (NT * 32);
  const int K = a.K1 + a.K2;
  const int nslab = (K + 15) >> 4;

  // LDS: one ring of A slabs per wave, 64 KiB per workgroup at MT = 2 -- exactly the round-1 kernel's footprint (an LDS-DMA
  // addresses its destination through M0[15:0]; a first version that also kept a bias copy in LDS needed 69 KiB and faulted)
  float* ring = smem + wave * RING_F;

  f32x16 acc[MT][NT];
  auto init_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // DMA sources: ring chunk f = lane + 64 i holds (row f/4 = lane/4 + 16 i, 4-float column c = (f%4) ^ ((f/16)&3)) of the slab;
  // the column does not depend on i (64 i is a multiple of 16 * 4), the row is kept as one base per tile
  const int acol = 4 * ((lane & 3) ^ ((lane >> 4) & 3));
  const int last_row = (int)(a.rows - 1);
  auto fetch_a = [&](int row_base, int slab, int pos) {
    const int k = slab * 16 + acol;
    float* dst = ring + (pos & (X3_SLOTS - 1)) * (SLOT_CH * 4);
    const float* src;
    int ld, kc;
    if (DUAL) {
      const bool first = k < a.K1;
      const int kk = first ? k : k - a.K1;
      ld = first ? a.K1 : a.K2;
      src = first ? a.A1 : a.A2;
      kc = kk + 4 <= ld ? kk : ld - 4;
    } else {
      ld = K; src = a.A1; kc = k + 4 <= K ? k : K - 4;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = row_base + 16 * i;
      const float* p = src + (int64_t)(r < last_row ? r : last_row) * ld + kc;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(dst + 64 * i * 4), 16, 0, MRG_A_CPOL);
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
  v4f_t x[MT][2];
  auto read_a = [&](int pos) {
    const unsigned base = lds_ring + (pos & (X3_SLOTS - 1)) * (SLOT_CH * 16);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][0]) : "v"(base + a_off[m][0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][1]) : "v"(base + a_off[m][1]));
    }
  };
  u32x4 bq[NT][3];
  const unsigned voff = (unsigned)lane * 16u;
  const char* bcol = Bp + (int64_t)blockIdx.y * NT * 3072;
  auto load_b = [&](int n, int slab) {
    const char* sb = bcol + ((int64_t)slab * ntile + n) * 3072;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[n][0]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(bq[n][1]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(bq[n][2]) : "v"(voff), "s"(sb));
  };
  u32x4 ch[MT], cm[MT], cl[MT], nh[MT], nm[MT], nl[MT];
  auto split_one = [&](int j, u32x4 (&H)[MT], u32x4 (&M)[MT], u32x4 (&L)[MT]) {
    const int m = j >> 2, q = j & 3;
    const v4f_t& v = x[m][q >> 1];
    unsigned h, mm, l;
    if (q & 1) split_pair(v.z, v.w, h, mm, l);
    else split_pair(v.x, v.y, h, mm, l);
    H[m][q] = h; M[m][q] = mm; L[m][q] = l;
  };

  int tn = t + wstride;
  int rb_cur = t * WROWS + (lane >> 2);                                  // this lane's first DMA row of the current / next tile
  int rb_nxt = (tn < total_wtiles ? tn : t) * WROWS + (lane >> 2);
  int pos = 0;                                       // global k-step counter of this wave: ring slot = pos % 4
  // ---- prologue of the wave: A slabs 0..2 of the first tile and B of k-step 0 in flight; slab 0 split
#pragma unroll
  for (int s = 0; s < 3; ++s) fetch_a(rb_cur, s, s);           // the host guarantees nslab >= 5
#pragma unroll
  for (int n = 0; n < NT; ++n) load_b(n, 0);
  wait_vmcnt(NBL);
  read_a(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NPAIR; ++j) split_one(j, ch, cm, cl);

  // One k-step; every k-step of every tile runs this same code: ONE instantiation (with the round-1 kernel's four compile-time
  // variants inside a tile loop hipcc spilled ~90 registers and shuffled the 224 accumulators between register ranges).  The
  // pipeline never drains inside the wave's run: the A DMA always fetches three k-steps ahead -- into the next tile's first
  // slabs near a tile end (the wave's last tile re-reads itself; never consumed) -- and the B loads always fetch the next
  // k-step's fragments, wrapping to slab 0 at a tile end: B does not depend on the tile, so they ARE the next tile's
  // k-step-0 fragments.  No load is conditional and every loaded register is consumed (see the header).
  auto slab = [&](int s) {
    const int s1 = s + 1 < nslab ? s + 1 : 0;
    // A(pos+1) was issued two k-steps ago; younger: B(pos-1) [NBL], A(pos+2) [NA], B(pos) [NBL]
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBL + NA) : "memory");
    read_a(pos + 1);
    {
      const bool nxt = s + 3 >= nslab;
      fetch_a(nxt ? rb_nxt : rb_cur, nxt ? s + 3 - nslab : s + 3, pos + 3);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      // B(pos) tile n: younger = rest of B(pos), this k-step's A DMA, the B(pos+1) tiles issued so far
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1) + NA) : "memory");
      constexpr int N0 = NT > 1 ? 1 : 0;
      if (n == N0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (n >= N0) {
#pragma unroll
        for (int j = (n - N0) * PP; j < (n - N0 + 1) * PP && j < NPAIR; ++j) split_one(j, nh, nm, nl);
      }
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, bq[n][0]), Bm = __builtin_bit_cast(bf16x8, bq[n][1]),
                   Bl = __builtin_bit_cast(bf16x8, bq[n][2]);
      // transposed product: the weight fragment is the A operand, the activation fragment the B operand
#define MRG_X3P_TERM(AP, BP)                                                                            \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                          \
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BP, __builtin_bit_cast(bf16x8, AP[m]), acc[m][n], 0, 0, 0)
      MRG_X3P_TERM(cm, Bm);
      MRG_X3P_TERM(cl, Bh);
      MRG_X3P_TERM(ch, Bl);
      MRG_X3P_TERM(cm, Bh);
      MRG_X3P_TERM(ch, Bm);
      MRG_X3P_TERM(ch, Bh);
#undef MRG_X3P_TERM
      if (n == NT - 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { ch[m] = nh[m]; cm[m] = nm[m]; cl[m] = nl[m]; }
      }
#pragma unroll
      for (int i = 0; i < 6 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MRG_X3_VPM, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_b(n, s1);
    }
    ++pos;
  };

  // epilogue of one wave tile: lane (li, lh) owns row `li` of every row tile and the columns 8q + 4 lh .. + 3 of every column tile
  auto epilogue = [&](int tile) {
    constexpr int act = ACT;
    const int64_t roww = (int64_t)tile * WROWS;
    const bool full = roww + WROWS <= a.rows;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int64_t row = roww + m * 32 + li;
      const bool rv = row < a.rows;
      const int64_t rcl = rv ? row : a.rows - 1;
      float* crow = a.C + rcl * a.ldc + col0 + 4 * lh;
      const float* srow = nullptr;
      float* xrow = nullptr;
      float cs = 1.f;
      if (EPI == EPI_GATE) { srow = a.S + rcl * a.ld_s + col0 + 4 * lh; xrow = a.aux ? a.aux + rcl * a.N + col0 + 4 * lh : nullptr; }
      if (EPI == EPI_ACCUM) srow = a.Cin + rcl * a.ld_cin + col0 + 4 * lh;
      if (EPI == EPI_GATE || EPI == EPI_SCALE) cs = a.scale * (a.rowscale ? a.rowscale[rcl] : 1.0f);
      const float* brow = a.bias ? a.bias + col0 + 4 * lh : nullptr;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = n * 32 + 8 * q;
          if (col0 + off < a.N) {                                  // wave-uniform (N % 8 == 0): both halves of the chunk pair exist
            v4f_t in = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_GATE || EPI == EPI_ACCUM) in = *reinterpret_cast<const v4f_t*>(srow + off);
            if (EPI != EPI_ACCUM && brow) bv = *reinterpret_cast<const v4f_t*>(brow + off);
            v4f_t v, g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float xacc = acc[m][n][4 * q + j] + bv[j];
              if (EPI == EPI_BIAS_ACT) v[j] = act == MRG_ACT_RELU ? (xacc > 0.f ? xacc : 0.f) : (act == MRG_ACT_SIGMOID ? sigmoidf_fast(xacc) : xacc);
              else if (EPI == EPI_GATE) { g[j] = sigmoidf_fast(xacc); v[j] = g[j] * in[j] * cs; }
              else if (EPI == EPI_SCALE) v[j] = xacc * cs;
              else v[j] = xacc + in[j];
            }
            if (full || rv) {
              *reinterpret_cast<v4f_t*>(crow + off) = v;
              if (EPI == EPI_GATE && xrow) *reinterpret_cast<v4f_t*>(xrow + off) = g;
            }
          }
        }
      }
    }
  };

  for (;;) {
    init_acc();                                                  // the accumulators never cross an iteration of this loop
    for (int s = 0; s < nslab; ++s) slab(s);
    if (!((MRG_X3_DBG & 1) && acc[0][0][0] != 123.456f)) epilogue(t);        // lab switch 1: no epilogue
    // Everything issued so far retires here: the stores, and the look-ahead loads that were issued before them (long landed).
    // The k-loop's waits count younger operations; with the queue empty they are trivially right for the next tile's first steps.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tn >= total_wtiles) break;
    t = tn;
    tn = t + wstride;
    rb_cur = rb_nxt;
    rb_nxt = (tn < total_wtiles ? tn : t) * WROWS + (lane >> 2);
  }
  // the B fragments fetched by the wave's last k-step are never multiplied: hold their registers until here (vmcnt(0) above)
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" ::"v"(bq[n][p]));
}

// what the transposed, vectorised epilogue and the per-tile look-ahead need on top of x3_eligible
inline bool x3p_eligible(const GemmArgs& a) {
  const int K = a.K1 + a.K2;
  bool ok = (a.N % 8 == 0) && (a.ldc % 4 == 0) && aligned16(a.C) && K >= 80 && a.rows * (int64_t)(K > a.N ? K : a.N) < (int64_t)1 << 40;
  if (a.bias) ok = ok && aligned16(a.bias);
  if (a.S) ok = ok && (a.ld_s % 4 == 0) && aligned16(a.S);
  if (a.aux) ok = ok && aligned16(a.aux);
  if (a.Cin) ok = ok && (a.ld_cin % 4 == 0) && aligned16(a.Cin);
  return ok && a.rows < ((int64_t)1 << 31);
}

template <int EPI>
inline int launch_rowgemm_x3p(GemmArgs a, const void* Bp, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int nt = gemm_pick_nt(a.N);
  const int ntile = x3_tiles(a.N, nt);
  const int mt = a.rows > 128 * 512 ? 2 : 1;
  const int wrows = 32 * mt;
  const int total = (int)((a.rows + wrows - 1) / wrows);
  const int ny = ntile / nt;
  int gx = (total + 3) / 4;
  const int cap = 256 / ny > 0 ? 256 / ny : 1;                    // one workgroup per CU
  if (gx > cap) gx = cap;
  dim3 grid((unsigned)gx, (unsigned)ny);
  const size_t lds = (size_t)(X3_THREADS / 64) * X3_SLOTS * wrows * 64;
  const int actv = EPI == EPI_BIAS_ACT ? a.act : 0;
  static bool attr_done[4][2][2][3] = {};                        // the LDS attribute is set once per kernel instance, not per launch
#define MRG_GOP3(NTV, MTV, DV, SLOT, AV)                                                                              \
  do {                                                                                                                \
    if (!attr_done[SLOT][MTV - 1][DV ? 1 : 0][AV]) {                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3p_k<NTV, MTV, EPI, DV, AV>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
      attr_done[SLOT][MTV - 1][DV ? 1 : 0][AV] = true;                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((rowgemm_x3p_k<NTV, MTV, EPI, DV, AV>), grid, dim3(X3_THREADS), lds, st, a, (const char*)Bp, ntile, total); \
  } while (0)
#define MRG_GOP2(NTV, MTV, DV, SLOT)                                                                                  \
  do {                                                                                                                \
    if (EPI == EPI_BIAS_ACT && actv == MRG_ACT_RELU) MRG_GOP3(NTV, MTV, DV, SLOT, (EPI == EPI_BIAS_ACT ? MRG_ACT_RELU : 0)); \
    else if (EPI == EPI_BIAS_ACT && actv == MRG_ACT_SIGMOID) MRG_GOP3(NTV, MTV, DV, SLOT, (EPI == EPI_BIAS_ACT ? MRG_ACT_SIGMOID : 0)); \
    else MRG_GOP3(NTV, MTV, DV, SLOT, 0);                                                                             \
  } while (0)
#define MRG_GOP(NTV, SLOT)                                                                                            \
  do {                                                                                                                \
    if (mt == 2) { if (a.K2 > 0) MRG_GOP2(NTV, 2, true, SLOT); else MRG_GOP2(NTV, 2, false, SLOT); }                  \
    else { if (a.K2 > 0) MRG_GOP2(NTV, 1, true, SLOT); else MRG_GOP2(NTV, 1, false, SLOT); }                          \
  } while (0)
  switch (nt) {
    case 1: MRG_GOP(1, 0); break;
    case 2: MRG_GOP(2, 1); break;
    case 4: MRG_GOP(4, 2); break;
    default: MRG_GOP(7, 3); break;
  }
#undef MRG_GOP
#undef MRG_GOP2
#undef MRG_GOP3
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

}  // namespace mrg
