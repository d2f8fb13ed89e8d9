// Tall-skinny GEMM core on the bf16 matrix pipe with f32-equivalent accuracy ("3-way split").
//
//   C[rows, N] = epilogue( [A1 | A2][rows, K1+K2] * B[N, K1+K2]^T )          (same contract as gemm.hpp)
//
// Every f32 operand is written as the exact-to-2^-26 sum of three bf16 numbers,
//   x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)       (round-to-nearest-even each)
// and a product is evaluated as the six leading cross terms, accumulated in f32 by the MFMA:
//   a*b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm        dropped: am*bl + al*bm + al*bl  (<= 2^-25 |a b|)
// so the result carries an error below one f32 ulp of each product -- the same class as the f32 pipe
// (tests pin |err| against a float64 product next to the exact-f32 kernel).  Why: v_mfma_f32_32x32x2_f32
// retires 64 flop/cycle/SIMD (157 TF/s chip peak), v_mfma_f32_32x32x16_bf16 1024; six bf16 MFMAs replace
// eight f32 MFMAs per 16 k-columns at 1/16 of the cycles each: 2.7x less matrix-pipe time for the same
// answer.  The price is VALU work for the splits, which is why
//   * B (a weight matrix, a few hundred KB) is split ONCE per call by bsplit3_k into fragment order
//     ([k-slab][column tile][plane][lane][8 bf16], one ds_read_b128 per fragment, conflict free), and
//   * A (activations, read once from HBM as f32 by LDS-DMA) is split in registers right after its
//     fragment read: 44 VALU instructions per 16 k-columns per wave against 6*NT MFMAs.
// Geometry as gemm.hpp: 256 threads own 128 rows x NT*32 columns, wave w owns rows [32w, 32w+32); two
// workgroups per CU; k walked in slabs of 16 with double-buffered LDS-DMA.  The DMA writes LDS linearly
// (wave base + lane*16 B), so the A tile is swizzled by choosing WHICH global chunk each lane fetches:
// chunk (row, c) lives at 16-byte slot row*4 + (c ^ ((row >> 2) & 3)), which makes the fragment reads
// (8 consecutive k per lane = two ds_read_b128) bank-conflict free.
#pragma once
#include <type_traits>
#include "gemm.hpp"

// lab switches (tools/gemm_x3_lab.hip): 1 no epilogue stores, 2 no A split, 4 no DMA after the first slab, 8 no barrier,
// 32 no B loads after the first slab, 64 no split at all, 128 no fragment reads, 256 staggered start (desynchronised rounds)
#ifndef MRG_X3_DBG
#define MRG_X3_DBG 0
#endif
// cache policy of the streamed A operand's DMA (lab: 0 default, 2 = nt)
#ifndef MRG_A_CPOL
#define MRG_A_CPOL 0
#endif
// VALU instructions the scheduler may place after each MFMA of a tile (lab sweep: 2 / 3 / 4 / 6)
// lab switch 256: s_sleep(127) repetitions (64 * 127 cycles each) the odd workgroups of the first round wait
#ifndef MRG_X3_STAGGER
#define MRG_X3_STAGGER 6
#endif
#ifndef MRG_X3_VPM
#define MRG_X3_VPM 3
#endif

namespace mrg {

// lab switch 512: per-wave phase timestamps (100 MHz wall clock) -> mrg_x3_trace[wave slot * 4 + {start, loop, epilogue, end}]
#if MRG_X3_DBG & 512
__device__ unsigned long long* mrg_x3_trace;
#define MRG_X3_STAMP(slot, i) do { if (lane == 0) mrg_x3_trace[(int64_t)(slot) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define MRG_X3_STAMP(slot, i) do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two floats -> three packed bf16 pairs (element 0 in the low half).
// The residual subtractions are spelled as single v_sub_f32: hipcc would SLP-pack the pair into v_pk_add_f32,
// which costs ~13 cycles of matrix-pipe time each when issued beside MFMAs (MI355X_MICROARCH.md, "price of one
// filler"), against ~0 for a plain 4-cycle VALU instruction.
__device__ __forceinline__ float sub1(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  f32x2 v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 r = {sub1(x0, __builtin_bit_cast(float, h << 16)), sub1(x1, __builtin_bit_cast(float, h & 0xffff0000u))};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 r2 = {sub1(r.x, __builtin_bit_cast(float, m << 16)), sub1(r.y, __builtin_bit_cast(float, m & 0xffff0000u))};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

inline int x3_tiles(int N, int nt) { return ((N + nt * 32 - 1) / (nt * 32)) * nt; }     // column tiles, padded to blocks of nt
inline int x3_slabs(int K) { return (K + 15) / 16; }
inline size_t x3_bsplit_bytes(int N, int K, int nt) { return (size_t)x3_slabs(K) * x3_tiles(N, nt) * 3 * 64 * 16; }

// B(n, k) = B[n * sn + k * sk]  ->  Bp[slab][tile][plane][lane] (16 B = 8 bf16: n = tile*32 + lane%32,
// k = slab*16 + (lane/32)*8 + j).  Rows >= N and columns >= K are zero, which is also what makes the
// clamped out-of-range A chunks harmless.  sk != 1 presents W^T without a transpose pass.
// Up to three weights of the same shape in one launch (blockIdx.y): the direction segments of a dense filter.
// Optional second source along k (the input gradient of two candidates in one product, [dz_a | dz_b] [W_a ; W_b]):
// columns k >= ksplit come from B2 at k - ksplit (ksplit = 0: single source).
struct BSplit3 { const float* B[3]; u32x4* out[3]; const float* B2[3]; int ksplit; };
static __global__ void bsplit3_k(BSplit3 p, int64_t sn, int64_t sk, int N, int K, int ntile, int nslab) {
  const float* __restrict__ B = p.B[blockIdx.y];
  const float* __restrict__ B2 = p.B2[blockIdx.y];
  const int ksplit = (p.ksplit > 0 && B2) ? p.ksplit : K;
  u32x4* __restrict__ Bp = p.out[blockIdx.y];
  if (!B) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nslab * ntile * 64) return;
  const int lane = idx & 63, tile = (idx >> 6) % ntile, slab = (idx >> 6) / ntile;
  const int n = tile * 32 + (lane & 31), k0 = slab * 16 + (lane >> 5) * 8;
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
  u32x4* o = Bp + ((int64_t)(slab * ntile + tile) * 3) * 64 + lane;
  o[0] = h; o[64] = m; o[128] = l;
}

inline void launch_bsplit3(const float* const* B, int64_t sn, int64_t sk, int N, int K, int nt, void* const* out, hipStream_t st,
                           const float* const* B2 = nullptr, int ksplit = 0) {
  const int ntile = x3_tiles(N, nt), nslab = x3_slabs(K);
  const int total = nslab * ntile * 64;
  BSplit3 p{};
  for (int i = 0; i < 3; ++i) { p.B[i] = B[i]; p.out[i] = (u32x4*)out[i]; p.B2[i] = B2 ? B2[i] : nullptr; }
  p.ksplit = B2 ? ksplit : 0;
  hipLaunchKernelGGL(bsplit3_k, dim3((total + 255) / 256, 3), dim3(256), 0, st, p, sn, sk, N, K, ntile, nslab);
}

inline void launch_bsplit(const float* B, int64_t sn, int64_t sk, int N, int K, int nt, void* Bp, hipStream_t st) {
  const int ntile = x3_tiles(N, nt), nslab = x3_slabs(K);
  const int total = nslab * ntile * 64;
  BSplit3 p{};
  p.B[0] = B; p.out[0] = (u32x4*)Bp;
  hipLaunchKernelGGL(bsplit3_k, dim3((total + 255) / 256, 1), dim3(256), 0, st, p, sn, sk, N, K, ntile, nslab);
}

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n <= 63
__device__ __forceinline__ void wait_vmcnt(int n) {
#define MRG_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
#define MRG_VM8(B) MRG_VM(B + 0) MRG_VM(B + 1) MRG_VM(B + 2) MRG_VM(B + 3) MRG_VM(B + 4) MRG_VM(B + 5) MRG_VM(B + 6) MRG_VM(B + 7)
  switch (n) {
    MRG_VM(0) MRG_VM(1) MRG_VM(2) MRG_VM(3) MRG_VM(4) MRG_VM(5) MRG_VM(6) MRG_VM(7)
    MRG_VM(8) MRG_VM(9) MRG_VM(10) MRG_VM(11) MRG_VM(12) MRG_VM(13) MRG_VM(14) MRG_VM(15)
    MRG_VM(16) MRG_VM(17) MRG_VM(18) MRG_VM(19) MRG_VM(20) MRG_VM(21) MRG_VM(22) MRG_VM(23)
    MRG_VM(24) MRG_VM(25) MRG_VM(26) MRG_VM(27) MRG_VM(28) MRG_VM(29) MRG_VM(30) MRG_VM(31)
    MRG_VM(32) MRG_VM(33) MRG_VM(34) MRG_VM(35) MRG_VM(36) MRG_VM(37) MRG_VM(38) MRG_VM(39)
    MRG_VM(40) MRG_VM(41) MRG_VM(42) MRG_VM(43) MRG_VM(44) MRG_VM(45) MRG_VM(46) MRG_VM(47)
    MRG_VM(48) MRG_VM(49) MRG_VM(50) MRG_VM(51) MRG_VM(52) MRG_VM(53) MRG_VM(54) MRG_VM(55)
    MRG_VM(56) MRG_VM(57) MRG_VM(58) MRG_VM(59) MRG_VM(60) MRG_VM(61) MRG_VM(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef MRG_VM8
#undef MRG_VM
}

// mrg_gemm_set_epilogue: 1 = row-order 16-byte stores through LDS where the operands allow, 0 (default) = accumulator-order stores.
// Round 3 measured the store tail NOT to be bound by the number of store instructions: with 4.5x fewer (25 instead of 112 per
// strip) the plain epilogue is 9 % slower alone (0.249 vs 0.228 ms at rows 272 115, K = N = 200; the LDS round trip is pure
// overhead), the accumulate epilogue 5-8 % faster (its input is read in row order too), the gate epilogue equal; in the supernet
// step the row GEMM entry points lose 0.9 ms / step in total (profiles/r3_rowgemm_epilogue.txt).  Kept as a tested option.
inline int& gemm_epi_lds() { static int m = 0; return m; }
// Round 4: 2 = TRANSPOSED accumulators in the LDS-weight kernel (gemm_x3s.hpp, gemm_epilogue_tr): the MFMA operands change
// places, a lane owns one row's 4-column chunks, and every epilogue load / store is 16 bytes per lane with no LDS round trip (28
// store instructions per 32-row strip instead of 112).  Bit-identical, and SLOWER (rows 558 771, K = N = 200: plain 0.426 vs
// 0.393 ms, accumulate 0.598 vs 0.463, gate 0.717 vs 0.611): a store instruction that writes 32 rows x 32 bytes costs the memory
// pipeline more than one that writes 2 rows x 128 bytes, and the epilogue is not bound by its instruction count -- the same
// 112 stores alone, at the kernel's grid, move 4.0 TB/s (profiles/r4_rowgemm_phases.txt).  Kept as a tested comparison point.
inline int& gemm_epi_mode() { static int m = 0; return m; }

constexpr int X3_THREADS = 256;     // 4 waves, one per SIMD
constexpr int X3_SLOTS = 4;         // per-wave LDS ring of A slabs

// Wave-autonomous kernel: a wave owns MT*32 rows x NT*32 columns and shares NOTHING with the other
// waves of its workgroup -- no barrier anywhere.
//   A: the wave's own [MT*32 rows x 16 k] f32 slab travels HBM -> LDS by DMA into a private 4-slot ring,
//      three slabs ahead; it is read back as fragments (2 ds_read_b128 per row tile) one slab ahead and
//      split into bf16 planes in the shadow of the current slab's MFMAs.
//   B: the pre-split fragments are read straight from global memory (L1/L2 resident: all waves of the chip
//      read the same few hundred KB) into the registers the MFMAs of the previous slab just released.
// With one wave per SIMD (MT = 2: 224 accumulator + ~200 other registers) every latency is covered
// inside the wave's own instruction stream: the in-order vmcnt counter is waited on with the exact number
// of younger operations (3 per column tile, 2*MT per A slab), never drained.
template <int NT, int MT, int EPI, bool DUAL, bool LDSEPI>
__global__ __launch_bounds__(X3_THREADS, 1) void rowgemm_x3_k(GemmArgs a, const char* __restrict__ Bp, int ntile) {   // a and Bp are re-pointed by a grouped launch
  constexpr int WROWS = 32 * MT, GBM = WROWS * (X3_THREADS / 64);
  constexpr int SLOT_CH = WROWS * 4;          // 16-byte chunks per ring slot
  constexpr int NA = SLOT_CH / 64;            // DMA instructions per slab
  constexpr int NBL = 3 * NT;                 // B loads per slab
  constexpr int NPAIR = MT * 4;               // float pairs to split per slab and lane
  constexpr int PP = NT > 1 ? (NPAIR + NT - 2) / (NT - 1) : NPAIR;   // pairs split in the shadow of one column tile
  extern __shared__ __align__(16) float smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int64_t row0 = (int64_t)blockIdx.x * GBM;
  // grouped launch: this workgroup's row range and its weight.  The range index and the weight pointer are formed by
  // unconditional scalar arithmetic (bp_stride = 0 in a plain launch): the B loads address through an SGPR pair.
  int sg = 0;
  if (a.grp.n > 0) sg = ((int)blockIdx.x >= a.grp.tile0[1] ? 1 : 0) + ((int)blockIdx.x >= a.grp.tile0[2] ? 1 : 0);
  sg = __builtin_amdgcn_readfirstlane(sg);
  const char* __restrict__ Bq = Bp + (int64_t)sg * a.grp.bp_stride;
  if (a.grp.n > 0) {                                     // constant indices only: a dynamic one would move the argument block to scratch
#define MRG_PICK(F) (sg == 0 ? a.grp.F[0] : (sg == 1 ? a.grp.F[1] : a.grp.F[2]))
    row0 = MRG_PICK(lo) + (int64_t)((int)blockIdx.x - MRG_PICK(tile0)) * GBM;
    a.rows = MRG_PICK(hi);
    a.bias = MRG_PICK(bias);
    a.scale = MRG_PICK(scale);
    if (!MRG_PICK(use_rowscale)) a.rowscale = nullptr;
#undef MRG_PICK
  }
  const int64_t roww = row0 + wave * WROWS;
  const int col0 = blockIdx.y * (NT * 32);
  const int K = a.K1 + a.K2;
  const int nslab = (K + 15) >> 4;

  if ((MRG_X3_DBG & 256) && blockIdx.x < 256 && (blockIdx.x & 1)) {     // lab: first-round workgroups of every other CU start late
    for (int i = 0; i < MRG_X3_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
  [[maybe_unused]] const int64_t trace_slot = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
  MRG_X3_STAMP(trace_slot, 0);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // DMA sources: ring chunk f = lane + 64 i holds (row f/4, 4-float column c = (f%4) ^ ((f/16)&3)) of the slab
  const float* arow1[NA]; const float* arow2[NA]; int acol[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int f = lane + 64 * i;
    const int64_t row = roww + (f >> 2);
    int64_t rc = row < a.rows ? row : a.rows - 1;
    if (a.row_index) rc = a.row_index[rc];                 // gathered rows (EPI_SEGMAX: edges in destination order)
    arow1[i] = a.A1 + rc * a.K1;
    arow2[i] = a.A2 + rc * a.K2;
    acol[i] = 4 * ((f & 3) ^ ((f >> 4) & 3));
  }
  float* ring = smem + wave * a.wave_lds_floats;        // wave-private region: the A ring, later the epilogue's strip
  auto fetch_a = [&](int slab) {
    const int k0 = slab * 16;
    float* dst = ring + (slab % X3_SLOTS) * (SLOT_CH * 4);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = k0 + acol[i];
      const float* p;
      if (DUAL) {
        const bool first = k < a.K1;
        const int kk = first ? k : k - a.K1, ld = first ? a.K1 : a.K2;
        p = (first ? arow1[i] : arow2[i]) + (kk + 4 <= ld ? kk : ld - 4);
      } else {
        p = arow1[i] + (k + 4 <= K ? k : K - 4);
      }
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
  v4f x[MT][2];                                   // raw fragments of the NEXT slab
  auto read_a = [&](int slab) {
    const unsigned base = lds_ring + (slab % X3_SLOTS) * (SLOT_CH * 16);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][0]) : "v"(base + a_off[m][0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(x[m][1]) : "v"(base + a_off[m][1]));
    }
  };

  // Asynchronous register fills: the asm statements below return at once, the data arrives later and is first
  // read behind the matching s_waitcnt.  This relies on hipcc keeping each u32x4 in the 4-register tuple the asm
  // wrote (it is exactly the MFMA operand tuple, so there is nothing to copy); a copy scheduled between the load and
  // the wait would read stale registers -- the float64 comparisons of tests/test_ops_gpu.py would catch that.
  u32x4 bq[NT][3];
  const unsigned voff = (unsigned)lane * 16u;
  const char* bcol = Bq + (int64_t)blockIdx.y * NT * 3072;
  auto load_b = [&](int n, int slab) {
    const char* sb = bcol + ((int64_t)slab * ntile + n) * 3072;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[n][0]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(bq[n][1]) : "v"(voff), "s"(sb));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(bq[n][2]) : "v"(voff), "s"(sb));
  };

  u32x4 ch[MT], cm[MT], cl[MT];                   // split planes of the CURRENT slab
  u32x4 nh[MT], nm[MT], nl[MT];                   // ... being produced for the next one
  auto split_one = [&](int j, u32x4 (&H)[MT], u32x4 (&M)[MT], u32x4 (&L)[MT]) {   // pair j of 4*MT
    const int m = j >> 2, q = j & 3;
    const v4f& v = x[m][q >> 1];
    unsigned h, mm, l;
    if (MRG_X3_DBG & 2) { h = __builtin_bit_cast(unsigned, v.x); mm = h; l = h; }
    else if (q & 1) split_pair(v.z, v.w, h, mm, l);
    else split_pair(v.x, v.y, h, mm, l);
    H[m][q] = h; M[m][q] = mm; L[m][q] = l;
  };

  // ---- prologue: A slabs 0..2 and B slab 0 in flight; slab 0 split
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < nslab) fetch_a(s);
#pragma unroll
  for (int n = 0; n < NT; ++n) load_b(n, 0);
  wait_vmcnt(NBL);                                // everything older than the B loads has landed
  read_a(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  MRG_X3_STAMP(trace_slot, 1);
#pragma unroll
  for (int j = 0; j < NPAIR; ++j) split_one(j, ch, cm, cl);

  // One k-slab.  MODE fixes the number of younger vector-memory operations at every wait at compile time
  // (a run-time count costs scalar branches the single wave of a SIMD cannot hide):
  //   0 steady state (s + 3 < nslab)   1: s == nslab-3   2: s == nslab-2   3: s == nslab-1   -1: run-time counts
  auto slab = [&](auto mode_c, int s) {
    constexpr int MODE = decltype(mode_c)::value;
    const bool has_next = MODE < 0 ? (s + 1 < nslab) : (MODE != 3);
    const bool do_dma = MODE < 0 ? (s + 3 < nslab) : (MODE == 0);
    const int dma = do_dma ? NA : 0;
    if (has_next) {
      // A(s+1) was issued two slabs ago; younger: B(s-1) [NBL], A(s+2) [NA if any], B(s) [NBL]
      if (MODE < 0) { if (s >= 2) wait_vmcnt(2 * NBL + ((s + 2 < nslab) ? NA : 0)); }
      else if (MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBL) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBL + NA) : "memory");
      if (!(MRG_X3_DBG & 128)) read_a(s + 1);
    }
    if (do_dma && !((MRG_X3_DBG & 4) && s > 0)) fetch_a(s + 3);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      // B(s) tile n: younger = rest of B(s), this slab's A DMA, the B(s+1) tiles issued so far
      if (MODE < 0) wait_vmcnt(3 * (NT - 1 - n) + dma + (has_next ? 3 * n : 0));
      else if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1) + NA) : "memory");
      else if (MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1 - n)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NT - 1)) : "memory");
      constexpr int N0 = NT > 1 ? 1 : 0;              // the splits start one tile late: the LDS reads issued at the top have landed by then
      if (n == N0 && has_next) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // raw fragments of the next slab
      __builtin_amdgcn_sched_barrier(0);
      if (has_next && n >= N0 && !(MRG_X3_DBG & 64)) {                      // VALU work for the shadow of this tile's MFMAs
#pragma unroll
        for (int j = (n - N0) * PP; j < (n - N0 + 1) * PP && j < NPAIR; ++j) split_one(j, nh, nm, nl);
      }
      const bf16x8 Bh = __builtin_bit_cast(bf16x8, bq[n][0]), Bm = __builtin_bit_cast(bf16x8, bq[n][1]),
                   Bl = __builtin_bit_cast(bf16x8, bq[n][2]);
      // Row tiles interleaved (a dependent MFMA issued back to back costs ~6 extra cycles, measured with
      // tools/mfma_bf16_peak.hip); small terms first, the leading term last.
#define MRG_X3_TERM(AP, BP)                                                                             \
  _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                          \
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AP[m]), BP, acc[m][n], 0, 0, 0)
      MRG_X3_TERM(cm, Bm);
      MRG_X3_TERM(cl, Bh);
      MRG_X3_TERM(ch, Bl);
      MRG_X3_TERM(cm, Bh);
      MRG_X3_TERM(ch, Bm);
      MRG_X3_TERM(ch, Bh);
#undef MRG_X3_TERM
      if (has_next && n == NT - 1 && !(MRG_X3_DBG & 64)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { ch[m] = nh[m]; cm[m] = nm[m]; cl[m] = nl[m]; }
      }
      // spread the VALU instructions of this tile over the MFMA issue slots: 1 MFMA, then up to 3 VALU
#pragma unroll
      for (int i = 0; i < 6 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MRG_X3_VPM, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next && !((MRG_X3_DBG & 32) && s > 0)) load_b(n, s + 1);
    }
  };
  // the host guarantees nslab >= 4 (x3_eligible)
  for (int s = 0; s + 3 < nslab; ++s) slab(std::integral_constant<int, 0>{}, s);
  slab(std::integral_constant<int, 1>{}, nslab - 3);
  slab(std::integral_constant<int, 2>{}, nslab - 2);
  slab(std::integral_constant<int, 3>{}, nslab - 1);
  MRG_X3_STAMP(trace_slot, 2);
  if ((MRG_X3_DBG & 1) && acc[0][0][0] != 123.456f) return;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if constexpr (EPI == EPI_SEGMAX) gemm_epilogue_segmax<NT>(a, acc[m], roww + m * 32, col0, li, lh);
    else if constexpr (EPI == EPI_SEGSUM) gemm_epilogue_segsum<NT>(a, acc[m], roww + m * 32, col0, li, lh);
    else if constexpr (LDSEPI) gemm_epilogue_lds<NT, EPI>(a, acc[m], roww + m * 32, col0, lane, ring);
    else gemm_epilogue<NT, EPI>(a, acc[m], roww + m * 32, col0, li, lh, row0 + GBM <= a.rows);
  }
  if (MRG_X3_DBG & 512) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MRG_X3_STAMP(trace_slot, 3); }
}

inline bool x3_eligible(const GemmArgs& a) {
  return (a.K1 % 4 == 0) && (a.K2 % 4 == 0) && a.K1 >= 4 && (a.K1 + a.K2) > 48 && (a.K2 == 0 || a.K2 >= 4) && aligned16(a.A1) && aligned16(a.A2);
}

// Bp: the split of B prepared by launch_bsplit(..., nt = gemm_pick_nt(a.N), ...)
template <int EPI>
inline int launch_rowgemm_x3(GemmArgs a, const void* Bp, hipStream_t st, int nt_force = 0) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int nt = nt_force > 0 ? nt_force : gemm_pick_nt(a.N);    // nt_force: narrower column blocks for few rows (gemm_dispatch.hpp: x3n_shape)
  const int ntile = x3_tiles(a.N, nt);
  const int mt = a.rows > 128 * 512 ? 2 : 1;            // short operands: more, smaller workgroups
  const int gbm = 32 * mt * (X3_THREADS / 64);
  if (a.grp.n > 0) {                                    // grouped: blocks of range s follow those of range s - 1
    a.grp.tile0[0] = 0;
    for (int i = 0; i < 3; ++i) {
      const int64_t r = i < a.grp.n && a.grp.hi[i] > a.grp.lo[i] ? a.grp.hi[i] - a.grp.lo[i] : 0;
      a.grp.tile0[i + 1] = a.grp.tile0[i] + (int)((r + gbm - 1) / gbm);
    }
    if (a.grp.tile0[3] == 0) return MRG_OK;
  }
  dim3 grid((unsigned)(a.grp.n > 0 ? a.grp.tile0[3] : (a.rows + gbm - 1) / gbm), (unsigned)(ntile / nt));
  const size_t ring_floats = (size_t)X3_SLOTS * 32 * mt * 16;
  a.epi_lds = (gemm_epi_lds() && gemm_epilogue_lds_ok<EPI>(a)) ? 1 : 0;
  a.wave_lds_floats = (int)((a.epi_lds && gemm_stage_floats(nt) > ring_floats) ? gemm_stage_floats(nt) : ring_floats);
  const size_t lds = (size_t)(X3_THREADS / 64) * a.wave_lds_floats * sizeof(float);
#define MRG_GOX3(NTV, MTV, DV, LV)                                                                                    \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_x3_k<NTV, MTV, EPI, DV, LV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((rowgemm_x3_k<NTV, MTV, EPI, DV, LV>), grid, dim3(X3_THREADS), lds, st, a, (const char*)Bp, ntile); \
  } while (0)
#define MRG_GOX2(NTV, MTV, DV)                                                                                        \
  do {                                                                                                                \
    if constexpr (EPI == EPI_SEGMAX || EPI == EPI_SEGSUM) MRG_GOX3(NTV, MTV, DV, false);                              \
    else { if (a.epi_lds) MRG_GOX3(NTV, MTV, DV, true); else MRG_GOX3(NTV, MTV, DV, false); }                         \
  } while (0)
#define MRG_GOX(NTV)                                                                                                  \
  do {                                                                                                                \
    if (mt == 2) { if (a.K2 > 0) MRG_GOX2(NTV, 2, true); else MRG_GOX2(NTV, 2, false); }                              \
    else { if (a.K2 > 0) MRG_GOX2(NTV, 1, true); else MRG_GOX2(NTV, 1, false); }                                      \
  } while (0)
  switch (nt) {
    case 1: MRG_GOX(1); break;
    case 2: MRG_GOX(2); break;
    case 4: MRG_GOX(4); break;
    default: MRG_GOX(7); break;
  }
#undef MRG_GOX
#undef MRG_GOX2
#undef MRG_GOX3
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

}  // namespace mrg
