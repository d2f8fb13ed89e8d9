// Split-bf16 row GEMM on 16 x 16 x 32 tiles: 64-row workgroups, THREE per CU (round 5).
//
// What round 4's phase stamps said about rowgemm_x3s_k (profiles/r4_rowgemm_phases.txt): a wave that owns a 32-row strip of
// seven 32 x 32 tiles carries 112 accumulator registers, so a SIMD holds two waves, and a wave's life is prologue -> k-loop ->
// store tail IN SEQUENCE (7 500 / 48 200 / 18 800 cycles for 17 472 cycles of MFMA issue): two waves per SIMD are not enough to
// keep the matrix pipe busy through each other's load and store phases (busy 0.42-0.47).  The structural candidate the stamps
// named -- an output tile small enough for three waves per SIMD -- is this kernel:
//
//   * a wave owns 16 rows x 14 column tiles of 16 (v_mfma_f32_16x16x32_bf16: 4 accumulator registers per tile, 56 in all; the
//     same flop per cycle as the 32 x 32 x 16 form), a workgroup of four waves 64 rows; <= 168 registers -> three workgroups per
//     CU, whose prologues, k-loops and store tails interleave on every SIMD;
//   * k advances in slabs of 32; the pre-split weight of a slab travels L2 -> LDS by LDS-DMA as two half-slabs of seven column
//     tiles (21 KB each) into a ring of two buffers (42 KB per workgroup: three workgroups fit the CU's 160 KB); one barrier per
//     half-slab; the MFMA operands are read back with ds_read_b128 (fragment order: conflict free);
//   * activations go straight into registers in fragment order (lane (row r, k-group g) reads the 8 consecutive floats
//     k = 32 s + 8 g ..: two global_load_dwordx4 per slab), two slabs ahead, and are split into their bf16 planes in the shadow
//     of the second half-slab's MFMAs;
//   * the epilogue turns PAIRS of 16 x 16 tiles into the 32 x 32 store shape with v_permlane16_swap_b32 (one VALU instruction per
//     register pair): a store instruction then writes 2 rows x 128 bytes, exactly what gemm_epilogue's stores write -- four rows x
//     64 bytes per instruction (the 16 x 16 accumulator as it stands) would double the memory pipeline's line transactions.
//
// Same six cross terms per product and the same order of the terms per accumulator as the other split-core kernels; the sums over
// k are formed 32 at a time by the matrix instruction instead of 16, so results agree with them to rounding, not bit for bit
// (tests/test_ops_gpu.py pins the error against float64 next to the exact-f32 core: <= 1.5 x).
//
// The weight split for this kernel has its own layout (launch_bsplitq): [slab32][half][tile 0..6][plane][lane] x 16 bytes with
// lane (column c = lane % 16, k-group g = lane / 16) holding B[n = (half * 7 + tile) * 16 + c][k = 32 slab + 8 g + 0..7].
#pragma once
#include "gemm_x3.hpp"

namespace mrg {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// lab / rollback switch: 0 = every launch stays on rowgemm_x3s_k (mrg_gemm_set_q)
inline int& gemm_q() { static int m = 1; return m; }

constexpr int X3Q_HT = 7;                       // 16-column tiles per half-slab
constexpr int X3Q_NT = 2 * X3Q_HT;              // 14 tiles = 224 columns
constexpr int X3Q_CHUNK = X3Q_HT * 3 * 1024;    // bytes of one half-slab of the pre-split weight
constexpr int X3Q_ROWS = 64;                    // rows per workgroup

inline int x3q_slabs(int K) { return (K + 31) / 32; }
inline size_t x3q_bsplit_bytes(int K) { return (size_t)x3q_slabs(K) * 2 * X3Q_CHUNK; }

struct BSplitQ { const float* B[3]; u32x4* out[3]; const float* B2[3]; int ksplit; };
static __global__ void bsplitq_k(BSplitQ p, int64_t sn, int64_t sk, int N, int K, int nslab) {
  const float* __restrict__ B = p.B[blockIdx.y];
  const float* __restrict__ B2 = p.B2[blockIdx.y];
  const int ksplit = (p.ksplit > 0 && B2) ? p.ksplit : K;
  u32x4* __restrict__ Bp = p.out[blockIdx.y];
  if (!B) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nslab * X3Q_NT * 64) return;
  const int lane = idx & 63, tile = (idx >> 6) % X3Q_NT, slab = (idx >> 6) / X3Q_NT;
  const int n = tile * 16 + (lane & 15), k0 = slab * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    v[j] = (n < N && k < K) ? (k < ksplit ? B[n * sn + k * sk] : B2[n * sn + (k - ksplit) * sk]) : 0.f;
  }
  u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned a, b, c;
    split_pair(v[2 * j], v[2 * j + 1], a, b, c);
    h[j] = a; m[j] = b; l[j] = c;
  }
  // [slab][half][tile in half][plane][lane]
  const int half = tile / X3Q_HT, t = tile - half * X3Q_HT;
  u32x4* o = Bp + ((int64_t)((slab * 2 + half) * X3Q_HT + t) * 3) * 64 + lane;
  o[0] = h; o[64] = m; o[128] = l;
}

inline void launch_bsplitq3(const float* const* B, int64_t sn, int64_t sk, int N, int K, void* const* out, int count, hipStream_t st,
                            const float* const* B2 = nullptr, int ksplit = 0) {
  const int nslab = x3q_slabs(K);
  const int total = nslab * X3Q_NT * 64;
  BSplitQ p{};
  for (int i = 0; i < 3; ++i) {
    p.B[i] = i < count ? B[i] : nullptr;
    p.out[i] = i < count ? (u32x4*)out[i] : nullptr;
    p.B2[i] = (B2 && i < count) ? B2[i] : nullptr;
  }
  p.ksplit = B2 ? ksplit : 0;
  hipLaunchKernelGGL(bsplitq_k, dim3((total + 255) / 256, count), dim3(256), 0, st, p, sn, sk, N, K, nslab);
}

// ---- epilogue on PAIRS of 16 x 16 tiles ------------------------------------------------------------------------------------------
// C/D map of a 16 x 16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.  v_permlane16_swap_b32 a, b exchanges a's odd 16-lane
// rows with b's even ones, so for tiles 2j (a) and 2j + 1 (b) and register r:
//   a' : lanes  0-15 row r      of tile 2j | lanes 16-31 row r      of tile 2j+1 | lanes 32-47 row 8+r  of 2j | lanes 48-63 row 8+r  of 2j+1
//   b' : lanes  0-15 row 4+r    of tile 2j | lanes 16-31 row 4+r    of tile 2j+1 | lanes 32-47 row 12+r of 2j | lanes 48-63 row 12+r of 2j+1
// i.e. with li = lane & 31, lh = lane >> 5 a lane holds column 32 j + li of rows i + 8 lh, i = r (a') or 4 + r (b'): one store
// instruction writes two 128-byte row pieces.  Arithmetic per element is gemm_epilogue's (bias add, activation / gate / scale /
// accumulate in the same order).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_q(const GemmArgs& a, f32x4v (&acc)[X3Q_NT], int64_t rowbase, int li, int lh, bool full) {
  if (rowbase >= a.rows) return;                           // wave-uniform: the whole 16-row strip is out of range
  const int last = (int)((a.rows - 1 - rowbase) < 15 ? (a.rows - 1 - rowbase) : 15);
  const int colw = li < a.N ? li : a.N - 1;
  float* crow[8];
  const float* srow[8];
  float* xrow[8];
  float cs[8];
  bool rok[8];
  const bool cstore = EPI != EPI_GATE || a.C != nullptr;   // EPI_GATE with C == NULL: only the gate (aux) is stored
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i + 8 * lh;
    const int rc = row < last ? row : last;
    rok[i] = row <= last;
    crow[i] = a.C + (rowbase + rc) * a.ldc + colw;
    if (EPI == EPI_GATE) srow[i] = a.S + (rowbase + rc) * a.ld_s + colw;
    if (EPI == EPI_ACCUM) srow[i] = a.Cin + (rowbase + rc) * a.ld_cin + colw;
    if (EPI == EPI_GATE) xrow[i] = a.aux ? a.aux + (rowbase + rc) * a.N + colw : nullptr;
    cs[i] = 1.0f;
    if (EPI == EPI_GATE || EPI == EPI_SCALE) cs[i] = a.scale * (a.rowscale ? a.rowscale[rowbase + rc] : 1.0f);
  }
  const int lc = li & 15;
#pragma unroll
  for (int j = 0; j < X3Q_HT; ++j) {
    // The bias is added BEFORE the lanes exchange (same sum per element: the exchange only moves data), so that the first reader of
    // an accumulator is an instruction the compiler sees -- it inserts the MFMA-result wait states itself; the exchange is inline
    // asm (the builtin __builtin_amdgcn_permlane16_swap of ROCm 7.2 returns its FIRST result twice and merges calls with different
    // operands: tools/lab/permlane16_swap_probe.hip) with the two wait states a VALU write of a swap operand needs inside the string.
    const int ca = (2 * j) * 16 + lc, cb = (2 * j + 1) * 16 + lc;
    const float bva = a.bias ? a.bias[ca < a.N ? ca : a.N - 1] : 0.f;
    const float bvb = a.bias ? a.bias[cb < a.N ? cb : a.N - 1] : 0.f;
    float acc8[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float xa = acc[2 * j][r] + bva, xb = acc[2 * j + 1][r] + bvb;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
      acc8[r] = xa;
      acc8[4 + r] = xb;
    }
    const int col = j * 32 + li;
    const bool cok = col < a.N;                             // columns of this pair that exist; a lane beyond them re-reads its column of pair 0 (never stored)
    const int off = cok ? j * 32 : 0;
    float in[8], v[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) in[i] = 0.f;
    if ((EPI == EPI_GATE && cstore) || EPI == EPI_ACCUM) {
#pragma unroll
      for (int i = 0; i < 8; ++i) in[i] = srow[i][off];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = acc8[i];
      if (EPI == EPI_BIAS_ACT) {
        v[i] = (a.act == MRG_ACT_RELU) ? (x > 0.f ? x : 0.f) : (a.act == MRG_ACT_SIGMOID ? sigmoidf_fast(x) : x);
      } else if (EPI == EPI_GATE) {
        g[i] = sigmoidf_fast(x);
        v[i] = g[i] * in[i] * cs[i];
      } else if (EPI == EPI_SCALE) {
        v[i] = x * cs[i];
      } else {
        v[i] = x + in[i];
      }
    }
    if (full) {
      if (cok) {
        if (cstore) {
#pragma unroll
          for (int i = 0; i < 8; ++i) crow[i][j * 32] = v[i];
        }
        if (EPI == EPI_GATE && a.aux) {
#pragma unroll
          for (int i = 0; i < 8; ++i) xrow[i][j * 32] = g[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (cok && rok[i]) {
          if (cstore) crow[i][j * 32] = v[i];
          if (EPI == EPI_GATE && a.aux) xrow[i][j * 32] = g[i];
        }
      }
    }
  }
}

#ifndef MRG_X3Q_WPS
#define MRG_X3Q_WPS 3        // lab: waves per SIMD the kernel is compiled for (3: <= 168 registers)
#endif
// lab switches (timing only -- wrong results; tools/lab/kernel_lab.sh libs NAME=-DMRG_X3Q_DBG=V:linear.hip,dense.hip builds one library per value into tools/labso/):
// 1 no epilogue, 2 no A loads after the prologue, 4 no weight DMA after the prologue, 8 no barriers after the prologue,
// 16 no fragment reads after the first tile, 32 no split arithmetic
#ifndef MRG_X3Q_DBG
#define MRG_X3Q_DBG 0
#endif

template <int EPI, bool DUAL>
__global__ __launch_bounds__(256, MRG_X3Q_WPS) void rowgemm_x3q_k(GemmArgs a, const char* __restrict__ Bp) {
  constexpr int NCH = X3Q_HT * 3;               // 1 KB pieces (64 lanes x 16 B) of a half-slab
  constexpr int NBW = (NCH + 3) / 4;            // DMA instructions per wave and half-slab
  extern __shared__ __align__(16) char smem_q[];          // [2][X3Q_CHUNK]
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lg = lane >> 4;              // MFMA operand coordinates: row (column for B) and k-group
  int64_t row0 = (int64_t)blockIdx.x * X3Q_ROWS;
  int sg = 0;
  if (a.grp.n > 0) sg = ((int)blockIdx.x >= a.grp.tile0[1] ? 1 : 0) + ((int)blockIdx.x >= a.grp.tile0[2] ? 1 : 0);
  sg = __builtin_amdgcn_readfirstlane(sg);
  const char* __restrict__ Bq = Bp + (int64_t)sg * a.grp.bp_stride;
  if (a.grp.n > 0) {                                     // grouped launch, as in rowgemm_x3_k (constant indices only)
#define MRG_PICK(F) (sg == 0 ? a.grp.F[0] : (sg == 1 ? a.grp.F[1] : a.grp.F[2]))
    row0 = MRG_PICK(lo) + (int64_t)((int)blockIdx.x - MRG_PICK(tile0)) * X3Q_ROWS;
    a.rows = MRG_PICK(hi);
    a.bias = MRG_PICK(bias);
    a.scale = MRG_PICK(scale);
    if (!MRG_PICK(use_rowscale)) a.rowscale = nullptr;
#undef MRG_PICK
  }
  const int64_t roww = row0 + wave * 16;
  const int K = a.K1 + a.K2;
  const int nslab = (K + 31) >> 5;

  f32x4v acc[X3Q_NT];
#pragma unroll
  for (int n = 0; n < X3Q_NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;

  // ---- A: this lane's fragment of slab s = row lr, k = 32 s + 8 lg + {0..3, 4..7}: two 16-byte loads
  int64_t rc = roww + lr < a.rows ? roww + lr : a.rows - 1;
  if (rc < 0) rc = 0;
  if (a.row_index) rc = a.row_index[rc];                 // gathered rows
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
  v4f xr[2][2];                                          // raw fragments: a ring of two slabs
  auto load_a = [&](int slab, v4f (&x)[2]) {
    const int sl = slab < nslab ? slab : nslab - 1;      // beyond the end: re-read the last slab (an asynchronous fill is never conditional)
    const int k = sl * 32 + lg * 8;
    const float* p0 = a_ptr(k);
    const float* p1 = a_ptr(k + 4);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[0]) : "v"(p0));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x[1]) : "v"(p1));
  };
  // ---- B: half-slab c = 2 s + half lives in buffer `half`; NBW pieces per wave (the last wave repeats the last piece)
  auto fetch_b = [&](int c) {
    const char* src = Bq + (int64_t)c * X3Q_CHUNK;
    const int buf = c & 1;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      int pc = wave * NBW + i;
      pc = pc < NCH ? pc : NCH - 1;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + pc * 1024 + lane * 16), (lds_ptr_t)(smem_q + buf * X3Q_CHUNK + pc * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem_q + (unsigned)lane * 16u;
  u32x4 bq[2][3];                                        // [ping-pong][plane]: the fragments of one column tile
  bool lab_first = true;
  auto read_b = [&](int t, int buf, u32x4 (&q)[3]) {
    if ((MRG_X3Q_DBG & 16) && !lab_first) return;
    lab_first = false;
    const unsigned ad = lds0 + (unsigned)(buf * X3Q_CHUNK + t * 3072);
    asm volatile("ds_read_b128 %0, %1" : "=v"(q[0]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q[1]) : "v"(ad));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(q[2]) : "v"(ad));
  };
  u32x4 ch, cm, cl, nh, nm, nl;
  auto split_pair_of = [&](const v4f (&x)[2], int q, u32x4& H, u32x4& M, u32x4& L) {     // q = 0..3: floats 2q, 2q + 1 of the 8
    const v4f& v = x[q >> 1];
    unsigned h, m, l;
    if (MRG_X3Q_DBG & 32) { h = __builtin_bit_cast(unsigned, (q & 1) ? v.z : v.x); m = __builtin_bit_cast(unsigned, (q & 1) ? v.w : v.y); l = h; }
    else if (q & 1) split_pair(v.z, v.w, h, m, l); else split_pair(v.x, v.y, h, m, l);
    H[q] = h; M[q] = m; L[q] = l;
  };

  // ---- prologue.  Issue order of a wave:  A(0) B(0) B(1) A(1) | s = 0, half 1: B(2) A(2) | s = 1, half 0: B(3) | half 1: B(4) A(3) | ...
  load_a(0, xr[0]);
  fetch_b(0);
  fetch_b(1);
  load_a(1, xr[1]);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 2) : "memory");       // A(0) and this wave's share of B(0) have landed
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair_of(xr[0], q, ch, cm, cl);
  __builtin_amdgcn_s_barrier();                                          // everybody's share of B(0) is in LDS

  // One half-slab: seven column tiles, six MFMAs each on the tile's own accumulator (a single accumulation chain of this
  // instruction needs no interleaving with other accumulators: MI355X_MICROARCH.md, cycle constants), the next tile's fragments
  // read meanwhile.  SPLIT: the four float pairs of the NEXT slab's activations are split in the shadow of tiles 0..3.
  auto half_slab = [&](auto half_c, auto split_c, const v4f (&xn)[2]) {
    constexpr int HF = decltype(half_c)::value;
    constexpr bool SPLIT = decltype(split_c)::value;
    read_b(0, HF, bq[0]);
#pragma unroll
    for (int t = 0; t < X3Q_HT; ++t) {
      if (t + 1 < X3Q_HT) {
        read_b(t + 1, HF, bq[(t + 1) & 1]);
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");       // tile t's fragments are there; the three just issued may be in flight
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (SPLIT && t < 4) split_pair_of(xn, t, nh, nm, nl);
      const bf16x8 Ah = __builtin_bit_cast(bf16x8, ch), Am = __builtin_bit_cast(bf16x8, cm), Al = __builtin_bit_cast(bf16x8, cl);
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, bq[t & 1][0]), Bm = __builtin_bit_cast(bf16x8, bq[t & 1][1]), Bl = __builtin_bit_cast(bf16x8, bq[t & 1][2]);
      f32x4v c = acc[HF * X3Q_HT + t];
      // small terms first, the leading term last (the order of the other split-core kernels)
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, c, 0, 0, 0);
      acc[HF * X3Q_HT + t] = c;
      if (SPLIT) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // ... then up to two VALU (an MFMA holds the vector issue for 8 of its 16 cycles)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // One k-slab of 32; R = s % 2 at compile time (ring position of the raw fragments).
  //   half 0 (buffer 0): [s >= 1: DMA of half-slab 2 s + 1 into buffer 1 -- read last during slab s - 1's half 1, every wave is
  //           past that barrier]; at its end this wave's share of half-slab 2 s + 1 must be in LDS: it is the youngest -> vmcnt(0)
  //           (which also lands A(s + 1), issued half a slab earlier);
  //   half 1 (buffer 1): DMA of half-slab 2 s + 2 into buffer 0, then A(s + 2) into the raw registers slab s - 1 split from;
  //           A(s + 1) is split in the shadow of its first four tiles; at its end only A(s + 2) is younger than the DMA -> vmcnt(2).
  auto slab = [&](auto r_c, int s) {
    constexpr int R = decltype(r_c)::value;
    const bool has_next = s + 1 < nslab;
    if (s > 0 && !(MRG_X3Q_DBG & 4)) fetch_b(2 * s + 1);
    half_slab(std::integral_constant<int, 0>{}, std::false_type{}, xr[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(MRG_X3Q_DBG & 8)) __builtin_amdgcn_s_barrier();
    if (has_next && !(MRG_X3Q_DBG & 4)) fetch_b(2 * s + 2);
    if (!(MRG_X3Q_DBG & 2)) load_a(s + 2, xr[R]);
    half_slab(std::integral_constant<int, 1>{}, std::true_type{}, xr[R ^ 1]);
    ch = nh; cm = nm; cl = nl;
    if (has_next) {
      if (MRG_X3Q_DBG & 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (!(MRG_X3Q_DBG & 8)) __builtin_amdgcn_s_barrier();
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

  const int li = lane & 31, lh = lane >> 5;
  if ((MRG_X3Q_DBG & 1) && acc[0][0] != 123.456f) return;
  gemm_epilogue_q<EPI>(a, acc, roww, li, lh, row0 + X3Q_ROWS <= a.rows);
}

inline bool x3q_eligible(const GemmArgs& a) { return x3_eligible(a) && a.rows > 0 && a.N <= X3Q_NT * 16 && !a.row_seg; }

// Bp: the split of B prepared by launch_bsplitq3
template <int EPI>
inline int launch_rowgemm_x3q(GemmArgs a, const void* Bp, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int gbm = X3Q_ROWS;
  if (a.grp.n > 0) {
    a.grp.tile0[0] = 0;
    for (int i = 0; i < 3; ++i) {
      const int64_t r = i < a.grp.n && a.grp.hi[i] > a.grp.lo[i] ? a.grp.hi[i] - a.grp.lo[i] : 0;
      a.grp.tile0[i + 1] = a.grp.tile0[i] + (int)((r + gbm - 1) / gbm);
    }
    if (a.grp.tile0[3] == 0) return MRG_OK;
  }
  dim3 grid((unsigned)(a.grp.n > 0 ? a.grp.tile0[3] : (a.rows + gbm - 1) / gbm));
  const size_t lds = (size_t)2 * X3Q_CHUNK;
  if constexpr (EPI == EPI_SEGMAX || EPI == EPI_SEGSUM) {
    return MRG_E_SHAPE;                                   // the fused aggregators' epilogues stay on rowgemm_x3s_k (x3q_shape says so)
  } else {
#define MRG_GOQ(DV)                                                                                                   \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3q_k<EPI, DV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((rowgemm_x3q_k<EPI, DV>), grid, dim3(256), lds, st, a, (const char*)Bp);                       \
  } while (0)
    if (a.K2 > 0) MRG_GOQ(true); else MRG_GOQ(false);
#undef MRG_GOQ
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MRG_OK : (int)e;
  }
}

}  // namespace mrg
