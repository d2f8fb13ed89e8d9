// Software-pipelined tall-skinny GEMM core on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   C[rows, N] = epilogue( [A1 | A2][rows, K1+K2] * B[N, K1+K2]^T )
//
// rows ~ 5e5, N and K ~ 2e2..4e2: 2*rows*K*N flop over 4*rows*(K+N) bytes is ~100 flop/B, far above
// the f32-MFMA ridge (157 TF/s / 8 TB/s ~ 20 flop/B), so the matrix pipe is the roofline.
//
// Geometry (wave = 64 lanes): a 256-thread workgroup owns 128 rows x NT*32 columns; wave w owns
// rows [32w, 32w+32) and all NT column tiles of 32x32 (NT*16 accumulator registers), so A is read
// from HBM exactly once.  Registers are held to <= 256 per lane so that TWO workgroups share a CU:
// one workgroup's prologue / epilogue overlaps the other's MFMA phase.
// K is walked in tiles of 16: while the 8*NT MFMAs (64 cycles each) of tile t run, the global
// loads of tile t+1 are already in flight into registers (issued branch-free from clamped
// addresses; masking happens when they are written to the other LDS buffer after the MFMAs),
// one barrier per tile.  LDS tiles are k-contiguous with a 4-float pad (row stride 20 floats):
// the ds_read_b128 fragment reads are bank-conflict free; one ds_read_b128 (4 consecutive k)
// feeds 4 MFMAs; the fragments of step t+1 are read while the MFMAs of step t issue.
// Measured (MI355X, rows 544k, K = N = 200): 66-68 TF/s (rocBLAS sgemm via torch: 38.6 TF/s);
// with 13 barriers per workgroup and lock-stepped partners the matrix pipe is ~55 % busy.
// "Dual source": the reduction dimension may be the concatenation of two row-major tensors
// (the reference's torch.cat([s, s_in], dim=1) is never materialised).
#pragma once
#include "common.hpp"

namespace mrg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM_MAX = 256;      // rows per workgroup = 128 * MT (MT row tiles of 32 per wave)

enum { EPI_BIAS_ACT = 0, EPI_GATE = 1, EPI_SCALE = 2, EPI_ACCUM = 3, EPI_SEGMAX = 4, EPI_SEGSUM = 5 };

// Up to three row ranges of ONE launch of the split-core row GEMM that share every tensor but differ in the weight matrix
// (the in / out / self direction segments of a dense filter: one launch instead of three).  Workgroup blocks
// [tile0[s], tile0[s+1]) own range s = rows [lo[s], hi[s]) (absolute row numbers of the shared tensors).
struct GemmGroups {
  int n;                              // 0: plain GEMM over rows [0, rows)
  int tile0[4];
  int64_t lo[3], hi[3];
  int64_t bp_stride;                  // range s reads the pre-split weight at Bp + s * bp_stride (launch_bsplit3)
  const float* bias[3];
  float scale[3];
  int use_rowscale[3];
};

struct GemmArgs {
  const float* A1; const float* A2;   // [rows][K1], [rows][K2] row-major; A2 may be NULL (K2 = 0)
  int K1, K2;
  const float* B; int ldb;            // [N][ldb] row-major, columns [0, K1+K2) are used
  const float* bias;                  // [N] or NULL
  float* C; int ldc; int N;           // output [rows][ldc], N valid columns
  int64_t rows;
  int act;                            // EPI_BIAS_ACT: MRG_ACT_*
  const float* S; int ld_s;           // EPI_GATE: C = sigmoid(acc + bias) * S[row][col] * c_row
  const float* rowscale; float scale; // c_row = scale * (rowscale ? rowscale[row] : 1)     (EPI_GATE, EPI_SCALE)
  float* aux;                         // EPI_GATE: optional store of sigmoid(acc + bias), [rows][N]
  const float* Cin; int ld_cin;       // EPI_ACCUM: C = acc + Cin
  // split core only (gemm_x3.hpp): output row r reads A row row_index[r] (NULL: r itself)
  const int32_t* row_index;
  // EPI_SEGMAX (no C): seg_out[row_seg[row_index[r]] * N + col] = max over rows r of pack(ReLU(acc + bias), r), see gemm_epilogue_segmax
  const int32_t* row_seg;
  unsigned long long* seg_out;
  // EPI_SEGSUM (no C): run sums of ReLU(acc + bias) at seg_part[head position * N + col], ReLU bit mask at relu_bits, see gemm_epilogue_segsum
  float* seg_part;
  unsigned* relu_bits; int bits_ld;
  // split core, one-wave kernel: stores in row order through a wave-private LDS strip (gemm_epilogue_lds); set by the launcher
  int epi_lds; int wave_lds_floats;
  GemmGroups grp;                     // split core, one-wave kernel only
};

// ---- a_max: ReLU + destination-segmented max as the GEMM's epilogue (reference models/operations_lp.py:230-234) ----------
// The GEMM walks the edges in destination order (row_index = the by-destination edge list), so the 32 rows of an accumulator
// strip are a few runs of equal destination.  A lane reduces its 16 rows run by run in registers and publishes one
// 64-bit atomic max per (run, column): key = float bits of the ReLU output (>= +0, so unsigned order = float order) in the
// high word, 0xFFFFFFFF - sorted row position in the low word.  max() is exact, so the result does not depend on the
// order of the atomics; among equal values the smallest position (= lowest edge id: the list of a destination is in edge
// order) wins, like DGL's first-index argmax; key 0 = "no in-edge".  The [E, D] ReLU output is never written.
// ---- a_mean / a_sum over ReLU(Linear): the ordered first level of a destination-segmented SUM as the GEMM's epilogue ---------
// Same walk as gemm_epilogue_segmax (edges in destination order).  A lane adds the ReLU outputs of each run of equal
// destination among ITS 16 rows of a strip (ascending rows: a fixed order) and stores the run's sum at the position of the
// run's first row ("head"): seg_part[head * N + col].  Only head rows are ever written or read; which rows are heads follows
// from the destination list alone (mrg_seg_reduce_heads_fwd re-derives it), so the second level adds a destination's run
// sums in ascending head order -- deterministic, no float atomics.  relu_bits[e * bits_ld + tile] keeps y > 0 per column
// (bit = column within the tile) of ORIGINAL edge row e for the backward: the [E, D] message tensor is never written.
template <int NT>
__device__ __forceinline__ void gemm_epilogue_segsum(const GemmArgs& a, f32x16 (&acc)[NT], int64_t rowbase, int col0, int li, int lh) {
  if (rowbase >= a.rows) return;
  int seg[16], eidv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const bool ok = row < a.rows;
    const int64_t rc = ok ? row : a.rows - 1;
    const int e = a.row_index ? a.row_index[rc] : (int)rc;
    eidv[r] = ok ? e : -1;
    seg[r] = ok ? a.row_seg[e] : -1;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 32 + li;
    const bool cok = col < a.N;
    const float bv = (a.bias && cok) ? a.bias[col] : 0.f;
    int cur = -1;
    int64_t head = 0;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = acc[n][r] + bv;
      const bool pos = x > 0.f;
      const unsigned long long ball = __ballot(pos);          // lanes 0-31: this register's row of half 0, lanes 32-63: of half 1
      // a padding tile (gemm_pick_nt rounds 3 tiles up to 4, 5-6 up to 7) has no word in the row of bits_ld = ceil(N/32) words
      if (li == 0 && eidv[r] >= 0 && a.relu_bits && (col0 >> 5) + n < a.bits_ld)
        a.relu_bits[(int64_t)eidv[r] * a.bits_ld + (col0 >> 5) + n] = lh ? (unsigned)(ball >> 32) : (unsigned)ball;
      if (seg[r] != cur) {                                   // uniform over the 32 lanes of a half wave
        if (cur >= 0 && cok) a.seg_part[head * a.N + col] = sum;
        cur = seg[r];
        head = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sum = 0.f;
      }
      sum += pos ? x : 0.f;
    }
    if (cur >= 0 && cok) a.seg_part[head * a.N + col] = sum;
  }
}

template <int NT>
__device__ __forceinline__ void gemm_epilogue_segmax(const GemmArgs& a, f32x16 (&acc)[NT], int64_t rowbase, int col0, int li, int lh) {
  if (rowbase >= a.rows) return;
  int seg[16];
  unsigned low[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const bool ok = row < a.rows;
    const int64_t rc = ok ? row : a.rows - 1;
    const int e = a.row_index ? a.row_index[rc] : (int)rc;
    const int d = a.row_seg[e];
    seg[r] = ok ? d : -1;
    low[r] = 0xFFFFFFFFu - (unsigned)rc;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 32 + li;
    const bool cok = col < a.N;
    const float bv = (a.bias && cok) ? a.bias[col] : 0.f;
    int cur = -1;
    unsigned long long best = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (seg[r] != cur) {                                   // uniform over the 32 lanes of a half wave
        if (cur >= 0 && cok) atomicMax(a.seg_out + (int64_t)cur * a.N + col, best);
        cur = seg[r];
        best = 0;
      }
      const float x = acc[n][r] + bv;
      const float y = x > 0.f ? x : 0.f;                     // ReLU; NaN -> 0 like the unfused epilogue
      const unsigned long long key = ((unsigned long long)__float_as_uint(y) << 32) | low[r];
      best = key > best ? key : best;
    }
    if (cur >= 0 && cok) atomicMax(a.seg_out + (int64_t)cur * a.N + col, best);
  }
}

// Branch-free guarded loads, split in two halves so the data is not needed until the tile is
// written to LDS (after the MFMAs of the previous tile):
//   gemm_raw4   issues the load from a clamped, always valid address -- no branch, no use of the data
//               (hipcc turns a load inside a data-dependent branch into branch + load + vmcnt(0));
//   gemm_mask4  zeroes the lanes that were out of range, at stash time.
//   VEC4: K % 4 == 0 and 16-byte aligned rows -> one global_load_dwordx4 per call.
template <bool VEC4>
__device__ __forceinline__ float4 gemm_raw4(const float* __restrict__ base, int64_t row, int64_t nrows, int k, int K, int ld) {
  const int64_t rc = row < nrows ? row : nrows - 1;
  const float* p = base + rc * ld;
  if (VEC4) {
    const int kc = k + 4 <= K ? k : (K >= 4 ? K - 4 : 0);
    return *reinterpret_cast<const float4*>(p + kc);
  }
  const int km = K - 1;
  return make_float4(p[k < K ? k : km], p[k + 1 < K ? k + 1 : km], p[k + 2 < K ? k + 2 : km], p[k + 3 < K ? k + 3 : km]);
}

template <bool VEC4>
__device__ __forceinline__ float4 gemm_mask4(float4 v, int64_t row, int64_t nrows, int k, int K) {
  const bool rv = row < nrows;
  if (VEC4) {
    const bool ok = rv && k + 4 <= K;
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
  } else {
    v.x = (rv && k < K) ? v.x : 0.f; v.y = (rv && k + 1 < K) ? v.y : 0.f;
    v.z = (rv && k + 2 < K) ? v.z : 0.f; v.w = (rv && k + 3 < K) ? v.w : 0.f;
  }
  return v;
}

// the concatenated operand [A1 | A2] (K1 % 4 == 0 whenever K2 > 0; the host passes A2 = A1 when
// there is no second source): which tensor / local column a global column k maps to
struct ASel { const float* base; int ld, kk; };
__device__ __forceinline__ ASel gemm_sel_a(const GemmArgs& a, int k) {
  const bool first = k < a.K1 || a.K2 == 0;
  ASel s;
  s.base = first ? a.A1 : a.A2;
  s.ld = first ? a.K1 : a.K2;
  s.kk = first ? k : k - a.K1;
  return s;
}

// raw / mask of a float4 of the concatenated operand [A1 | A2] at global column k.  The scalar
// path selects the source per element (K1 need not be a multiple of 4 there).
template <bool VEC4>
__device__ __forceinline__ float4 gemm_raw_a(const GemmArgs& a, int64_t row, int k) {
  if (VEC4) {
    const ASel sa = gemm_sel_a(a, k);
    return gemm_raw4<true>(sa.base, row, a.rows, sa.kk, sa.ld, sa.ld);
  }
  const int64_t rc = row < a.rows ? row : a.rows - 1;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ASel sa = gemm_sel_a(a, k + j);
    const int kc = sa.kk < sa.ld ? sa.kk : sa.ld - 1;
    v[j] = sa.base[rc * sa.ld + kc];
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}

template <bool VEC4>
__device__ __forceinline__ float4 gemm_mask_a(float4 v, const GemmArgs& a, int64_t row, int k) {
  if (VEC4) {
    const ASel sa = gemm_sel_a(a, k);
    return gemm_mask4<true>(v, row, a.rows, sa.kk, sa.ld);
  }
  const int K = a.K1 + a.K2;
  const bool rv = row < a.rows;
  v.x = (rv && k < K) ? v.x : 0.f; v.y = (rv && k + 1 < K) ? v.y : 0.f;
  v.z = (rv && k + 2 < K) ? v.z : 0.f; v.w = (rv && k + 3 < K) ? v.w : 0.f;
  return v;
}

// Epilogue of one wave: NT tiles of 32x32 at rows [rowbase, rowbase+32), columns [col0, col0 + NT*32).
// C/D map of a tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
// Every load (bias, row scale, gate multiplicand, accumulate input) is issued unconditionally from a
// clamped address BEFORE the stores, and a full tile stores without per-element branches: a store inside
// a data-dependent branch makes hipcc wait vmcnt(0) per store, which serialises the 16*NT stores of a lane.
#ifndef MRG_C_NT
#define MRG_C_NT 0
#endif
template <int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[NT], int64_t rowbase, int col0, int li, int lh,
                                              bool full) {
  if (rowbase >= a.rows) return;                           // wave-uniform: the whole 32-row strip is out of range
  // One pointer per accumulator row (clamped to the last valid row) and per array, set up once; a column tile is then
  // a constant 128-byte offset from it, i.e. an immediate of the load / store.  (Per-tile 64-bit address arithmetic
  // made hipcc spill hundreds of addresses of the 224-accumulator kernels to scratch memory.)
  const int last = (int)((a.rows - 1 - rowbase) < 31 ? (a.rows - 1 - rowbase) : 31);
  const int colw = col0 + li < a.N ? col0 + li : a.N - 1;   // this lane's column in tile 0 (clamped: only lanes of a partial tile)
  float* crow[16];
  const float* srow[16];
  float* xrow[16];
  float cs[16];
  bool rok[16];
  // EPI_GATE with C == NULL: only the gate (aux) is stored -- the gated output is recomputed from gate, S and the row scale
  // by its consumer (the MixedOp epilogue's "virtual" candidate, mixedop.hip) instead of being written and re-read four times
  const bool cstore = EPI != EPI_GATE || a.C != nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int rc = row < last ? row : last;
    rok[r] = row <= last;
    crow[r] = a.C + (rowbase + rc) * a.ldc + colw;
    if (EPI == EPI_GATE) srow[r] = a.S + (rowbase + rc) * a.ld_s + colw;
    if (EPI == EPI_ACCUM) srow[r] = a.Cin + (rowbase + rc) * a.ld_cin + colw;
    if (EPI == EPI_GATE) xrow[r] = a.aux ? a.aux + (rowbase + rc) * a.N + colw : nullptr;
    cs[r] = 1.0f;
    if (EPI == EPI_GATE || EPI == EPI_SCALE) cs[r] = a.scale * (a.rowscale ? a.rowscale[rowbase + rc] : 1.0f);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 32 + li;
    // columns of this tile that exist; a lane beyond them re-reads its tile-0 column (never stored)
    const bool cok = col < a.N;
    const int off = cok ? n * 32 : 0;
    const float bv = a.bias ? a.bias[cok ? col : colw] : 0.f;
    float in[16], v[16], g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) in[r] = 0.f;
    if ((EPI == EPI_GATE && cstore) || EPI == EPI_ACCUM) {     // (gate only: the multiplicand is not read at all)
#pragma unroll
      for (int r = 0; r < 16; ++r) in[r] = srow[r][off];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = acc[n][r] + bv;
      if (EPI == EPI_BIAS_ACT) {
        v[r] = (a.act == MRG_ACT_RELU) ? (x > 0.f ? x : 0.f) : (a.act == MRG_ACT_SIGMOID ? sigmoidf_fast(x) : x);
      } else if (EPI == EPI_GATE) {
        g[r] = sigmoidf_fast(x);
        v[r] = g[r] * in[r] * cs[r];
      } else if (EPI == EPI_SCALE) {
        v[r] = x * cs[r];
      } else {
        v[r] = x + in[r];
      }
    }
    if (full) {
      if (cok) {
        if (cstore) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#if MRG_C_NT
            __builtin_nontemporal_store(v[r], crow[r] + n * 32);         // lab: C is written once and re-read by a later kernel
#else
            crow[r][n * 32] = v[r];
#endif
          }
        }
        if (EPI == EPI_GATE && a.aux) {
#pragma unroll
          for (int r = 0; r < 16; ++r) xrow[r][n * 32] = g[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (cok && rok[r]) {
          if (cstore) crow[r][n * 32] = v[r];
          if (EPI == EPI_GATE && a.aux) xrow[r][n * 32] = g[r];
        }
      }
    }
  }
}

// ---- the same epilogues on a TRANSPOSED accumulator tile (rowgemm_x3s_k<..., TR = true>) ---------------------------------------
// With the MFMA operands exchanged, lane (li, lh) holds row rowbase + li and, in registers 4g .. 4g+3 of tile n, the columns
// col0 + 32 n + 8 g + 4 lh + {0..3}: sixteen consecutive bytes of C.  Bias, gate multiplicand and accumulate input are read as
// float4 and the results leave as global_store_dwordx4: 4 * NT store (and load) instructions per strip instead of 16 * NT, one
// address computation per row instead of per (row, array).  Arithmetic and its order per element are gemm_epilogue's.
// Requires gemm_epilogue_tr_ok (N % 4 == 0: a 4-column chunk is valid or invalid as a whole; 16-byte aligned rows).
template <int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue_tr(const GemmArgs& a, f32x16 (&acc)[NT], int64_t rowbase, int col0, int li, int lh) {
  if (rowbase >= a.rows) return;                           // wave-uniform: the whole 32-row strip is out of range
  const int64_t row = rowbase + li;
  const bool rok = row < a.rows;
  const int64_t rc = rok ? row : a.rows - 1;
  const int cb = col0 + 4 * lh;                            // this lane's first column of (tile 0, group 0)
  const bool cstore = EPI != EPI_GATE || a.C != nullptr;   // EPI_GATE with C == NULL: only the gate is stored (see gemm_epilogue)
  float* crow = cstore ? a.C + rc * a.ldc + cb : nullptr;
  const float* srow = nullptr;
  if (EPI == EPI_GATE && cstore) srow = a.S + rc * a.ld_s + cb;
  if (EPI == EPI_ACCUM) srow = a.Cin + rc * a.ld_cin + cb;
  float* xrow = (EPI == EPI_GATE && a.aux) ? a.aux + rc * a.N + cb : nullptr;
  float cs = 1.0f;
  if (EPI == EPI_GATE || EPI == EPI_SCALE) cs = a.scale * (a.rowscale ? a.rowscale[rc] : 1.0f);
  const float* brow = a.bias ? a.bias + cb : nullptr;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int off = n * 32 + g * 8;
      if (cb + off >= a.N) continue;                       // beyond the last column (whole chunk)
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), in = make_float4(0.f, 0.f, 0.f, 0.f);
      if (brow) bv = *reinterpret_cast<const float4*>(brow + off);
      if (srow) in = *reinterpret_cast<const float4*>(srow + off);
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w}, ii[4] = {in.x, in.y, in.z, in.w};
      float v[4], gt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = acc[n][4 * g + j] + bb[j];
        if (EPI == EPI_BIAS_ACT) {
          v[j] = (a.act == MRG_ACT_RELU) ? (x > 0.f ? x : 0.f) : (a.act == MRG_ACT_SIGMOID ? sigmoidf_fast(x) : x);
        } else if (EPI == EPI_GATE) {
          gt[j] = sigmoidf_fast(x);
          v[j] = gt[j] * ii[j] * cs;
        } else if (EPI == EPI_SCALE) {
          v[j] = x * cs;
        } else {
          v[j] = x + ii[j];
        }
      }
      if (rok) {
        if (cstore) *reinterpret_cast<float4*>(crow + off) = make_float4(v[0], v[1], v[2], v[3]);
        if (EPI == EPI_GATE && xrow) *reinterpret_cast<float4*>(xrow + off) = make_float4(gt[0], gt[1], gt[2], gt[3]);
      }
    }
  }
}

// host side: may this launch run on transposed accumulators?
template <int EPI>
inline bool gemm_epilogue_tr_ok(const GemmArgs& a) {
  if (EPI != EPI_BIAS_ACT && EPI != EPI_GATE && EPI != EPI_SCALE && EPI != EPI_ACCUM) return false;
  if (a.N % 4 != 0) return false;
  const bool gate_only = EPI == EPI_GATE && !a.C && a.aux;
  if (!gate_only && (!a.C || a.ldc % 4 != 0 || !aligned16(a.C))) return false;
  if (EPI == EPI_GATE && ((a.C && (a.ld_s % 4 != 0 || !aligned16(a.S))) || (a.aux && !aligned16(a.aux)))) return false;
  if (EPI == EPI_ACCUM && (a.ld_cin % 4 != 0 || !aligned16(a.Cin))) return false;
  if (a.bias && !aligned16(a.bias)) return false;
  for (int i = 0; i < a.grp.n && i < 3; ++i)
    if (a.grp.bias[i] && !aligned16(a.grp.bias[i])) return false;
  return true;
}

// ---- the same epilogues with the stores (and the [rows, N] inputs) in ROW order, through a wave-private LDS strip -----------
// gemm_epilogue above stores an accumulator register as it stands: one global_store_dword per (row quad, column tile) =
// two 128-byte pieces of two rows, 16 * NT store instructions per 32-row strip.  The row GEMM's store tail is bound by the
// NUMBER of store instructions, not by their bytes (lab: the one-wave kernel at rows 272 115, K = N = 200 spends 0.09 of its
// 0.237 ms there; MI355X_MICROARCH.md "attention epilogue store tail", cdna_hip_programming.md T21).  Here the strip is
// written to LDS in the accumulator layout (ds_write_b32: 32 consecutive floats per half wave, conflict free) and read back as
// float4 in row-major order, so one global_store_dwordx4 writes 1 KB of consecutive addresses: W / 8 store instructions per
// strip (25 instead of 112 at W = 200), and the gate multiplicand / accumulate input are read the same way.
// Arithmetic and its order per element are those of gemm_epilogue (bit-identical results): the bias add, the activation /
// sigmoid and the row scale of EPI_SCALE happen before the LDS round trip, the products with row-major inputs after it.
// stage: wave-private LDS, 32 * NT * 32 + 32 floats.  Requires N % 4 == 0 and 16-byte aligned rows of C / S / aux / Cin.
constexpr int GEMM_STAGE_PAD = 32;       // the strip's 32 row scales (EPI_GATE)
inline size_t gemm_stage_floats(int nt) { return (size_t)32 * nt * 32 + GEMM_STAGE_PAD; }

template <int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& a, f32x16 (&acc)[NT], int64_t rowbase, int col0, int lane,
                                                  float* __restrict__ stage) {
  if (rowbase >= a.rows) return;                            // wave-uniform
  const int li = lane & 31, lh = lane >> 5;
  const int W = (a.N - col0) < NT * 32 ? (a.N - col0) : NT * 32;                  // columns of this block: wave-uniform, % 4 == 0
  const int w4 = W >> 2;
  const int nrows = (int)((a.rows - rowbase) < 32 ? (a.rows - rowbase) : 32);
  float* cs_lds = stage + 32 * NT * 32;
  // ---- accumulator layout: bias, activation, (EPI_SCALE) row scale; then the strip goes to LDS as [32][W]
  float cs[16];
  if (EPI == EPI_GATE || EPI == EPI_SCALE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int rc = row < nrows ? row : nrows - 1;
      cs[r] = a.scale * (a.rowscale ? a.rowscale[rowbase + rc] : 1.0f);
      if (EPI == EPI_GATE && li == 0) cs_lds[row] = cs[r];
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n * 32 + li;
    const bool cok = col < W;
    const float bv = a.bias ? a.bias[col0 + (cok ? col : 0)] : 0.f;
    float* sp = stage + (4 * lh) * W + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = acc[n][r] + bv;
      float v;
      if (EPI == EPI_BIAS_ACT) v = (a.act == MRG_ACT_RELU) ? (x > 0.f ? x : 0.f) : (a.act == MRG_ACT_SIGMOID ? sigmoidf_fast(x) : x);
      else if (EPI == EPI_GATE) v = sigmoidf_fast(x);
      else if (EPI == EPI_SCALE) v = x * cs[r];
      else v = x;
      if (cok) sp[((r & 3) + 8 * (r >> 2)) * W] = v;
    }
  }
  // (same wave: the LDS executes a wave's operations in order, no barrier)
  // ---- row-major order: float4 q of the strip = row q / w4, columns 4 (q % w4) ...
  const int total = nrows * w4;
  const int sr = 64 / w4, sc = 64 - sr * w4;                 // what 64 float4 further means in (row, float4 column)
  int r = lane / w4, c = lane - r * w4;
  float* __restrict__ Cb = a.C + rowbase * a.ldc + col0;
  const float* __restrict__ Sb = nullptr;
  float* __restrict__ Xb = nullptr;
  int ld_in = 0;
  if (EPI == EPI_GATE) { Sb = a.S + rowbase * a.ld_s + col0; ld_in = a.ld_s; Xb = a.aux ? a.aux + rowbase * a.N + col0 : nullptr; }
  if (EPI == EPI_ACCUM) { Sb = a.Cin + rowbase * a.ld_cin + col0; ld_in = a.ld_cin; }
  constexpr int U = 5;                                       // float4 per lane in flight
  for (int q0 = lane; q0 - lane < total; q0 += 64 * U) {
    float4 v[U], in[U];
    float csr[U];
    int ro[U], co[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 64 * u;
      ok[u] = q < total;
      const int rr = ok[u] ? r : 0, cc = ok[u] ? c : 0;       // clamped: loads are unconditional, only the stores are guarded
      ro[u] = rr; co[u] = cc * 4;
      v[u] = *reinterpret_cast<const float4*>(stage + rr * W + cc * 4);
      if (EPI == EPI_GATE || EPI == EPI_ACCUM) in[u] = *reinterpret_cast<const float4*>(Sb + (int64_t)rr * ld_in + cc * 4);
      if (EPI == EPI_GATE) csr[u] = cs_lds[rr];
      c += sc; r += sr;
      if (c >= w4) { c -= w4; ++r; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 o = v[u];
      if (EPI == EPI_GATE) {
        o.x = v[u].x * in[u].x * csr[u]; o.y = v[u].y * in[u].y * csr[u]; o.z = v[u].z * in[u].z * csr[u]; o.w = v[u].w * in[u].w * csr[u];
      } else if (EPI == EPI_ACCUM) {
        o.x = v[u].x + in[u].x; o.y = v[u].y + in[u].y; o.z = v[u].z + in[u].z; o.w = v[u].w + in[u].w;
      }
      if (ok[u]) {
        *reinterpret_cast<float4*>(Cb + (int64_t)ro[u] * a.ldc + co[u]) = o;
        if (EPI == EPI_GATE && Xb) *reinterpret_cast<float4*>(Xb + (int64_t)ro[u] * a.N + co[u]) = v[u];
      }
    }
  }
}

// host side: may this launch use gemm_epilogue_lds?
template <int EPI>
inline bool gemm_epilogue_lds_ok(const GemmArgs& a) {
  if (EPI != EPI_BIAS_ACT && EPI != EPI_GATE && EPI != EPI_SCALE && EPI != EPI_ACCUM) return false;
  if (a.N % 4 != 0 || a.ldc % 4 != 0 || !a.C || !aligned16(a.C)) return false;
  if (EPI == EPI_GATE && (a.ld_s % 4 != 0 || !aligned16(a.S) || (a.aux && !aligned16(a.aux)))) return false;
  if (EPI == EPI_ACCUM && (a.ld_cin % 4 != 0 || !aligned16(a.Cin))) return false;
  return true;
}

template <int NT, int MT, int GBK, int EPI, bool VEC4>
__global__ __launch_bounds__(MRG_BLOCK, (MT == 1 ? 2 : 1)) void rowgemm_k(GemmArgs a) {
  constexpr int GBM = 128 * MT;
  constexpr int GLD = GBK + 4;                 // padded LDS row stride (floats)
  constexpr int F4R = GBK / 4;                 // float4 per tile row
  constexpr int NA = GBM * F4R / MRG_BLOCK;    // A float4 per thread per tile
  constexpr int NBT = NT * 32 * F4R;           // B float4 per tile
  constexpr int NB = (NBT + MRG_BLOCK - 1) / MRG_BLOCK;
  extern __shared__ __align__(16) float smem[];
  constexpr int A_TILE = GBM * GLD, B_TILE = NT * 32 * GLD, STAGE = A_TILE + B_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * GBM;
  const int col0 = blockIdx.y * (NT * 32);
  const int K = a.K1 + a.K2;
  const int nkt = (K + GBK - 1) / GBK;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  float4 pa[NA];
  float4 pb[NB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int f = tid + i * MRG_BLOCK;
      pa[i] = gemm_raw_a<VEC4>(a, row0 + f / F4R, k0 + (f % F4R) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int f = tid + i * MRG_BLOCK;
      f = f < NBT ? f : NBT - 1;
      pb[i] = gemm_raw4<VEC4>(a.B, col0 + f / F4R, a.N, k0 + (f % F4R) * 4, K, a.ldb);
    }
  };
  auto stash = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int f = tid + i * MRG_BLOCK;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + (f / F4R) * GLD + (f % F4R) * 4]) =
          gemm_mask_a<VEC4>(pa[i], a, row0 + f / F4R, k0 + (f % F4R) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int f = tid + i * MRG_BLOCK;
      if (f < NBT)
        *reinterpret_cast<float4*>(&smem[buf * STAGE + A_TILE + (f / F4R) * GLD + (f % F4R) * 4]) =
            gemm_mask4<VEC4>(pb[i], col0 + f / F4R, a.N, k0 + (f % F4R) * 4, K);
    }
  };

  fetch(0);
  stash(0, 0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int k0 = kt * GBK;
    if (kt + 1 < nkt) fetch(k0 + GBK);                 // in flight during the MFMAs below
    const int kv = K - k0 < GBK ? K - k0 : GBK;
    const int nt8 = (kv + 7) >> 3;
    const float* At = smem + cur * STAGE + (wave * 32 * MT + li) * GLD + lh * 4;
    const float* Bt = smem + cur * STAGE + A_TILE + li * GLD + lh * 4;
    // fragment reads of step t+1 are issued before the MFMAs of step t (register double buffer)
    float4 af[2][MT], bf[2][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const float4*>(At + m * 32 * GLD);
#pragma unroll
    for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const float4*>(Bt + n * 32 * GLD);
#pragma unroll
    for (int t = 0; t < GBK / 8; ++t) {
      if (t < nt8) {
        const int c = t & 1, nx = c ^ 1;
        if (t + 1 < GBK / 8) {               // harmless over-read of zero-filled columns when t + 1 >= nt8
#pragma unroll
          for (int m = 0; m < MT; ++m) af[nx][m] = *reinterpret_cast<const float4*>(At + m * 32 * GLD + (t + 1) * 8);
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[nx][n] = *reinterpret_cast<const float4*>(Bt + n * 32 * GLD + (t + 1) * 8);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][m].x, bf[c][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][m].y, bf[c][n].y, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][m].z, bf[c][n].z, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][m].w, bf[c][n].w, acc[m][n], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) {
      stash(cur ^ 1, k0 + GBK);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue
#pragma unroll
  for (int m = 0; m < MT; ++m)
    gemm_epilogue<NT, EPI>(a, acc[m], row0 + wave * 32 * MT + m * 32, col0, li, lh, row0 + GBM <= a.rows);
}

// ---- LDS-DMA variant (the fast path) ---------------------------------------------------------
// Same tiling as rowgemm_k, but the A / B tiles travel global -> LDS with global_load_lds_dwordx4
// (no VGPR staging, no ds_write, no vmcnt -> ds_write dependency in the instruction stream).
// Measured on MI355X (rows 544k, K = N = 200): staging through registers cost 0.35 ms on top of
// 0.38 ms of MFMA time; with LDS-DMA the same kernel runs 0.55 ms instead of 0.86 ms.
// The DMA destination is wave-uniform base + lane*16, so LDS rows are unpadded (64 B); the
// resulting 4-way bank conflict of the ds_read_b128 fragment reads is invisible next to the
// 64-cycle f32 MFMAs.  The DMA cannot zero-fill: out-of-range rows / columns are read from
// clamped addresses (their results are never stored) and K must be a multiple of 8 so that an
// 8-deep MFMA step never contains columns beyond K.
template <int NT, int EPI, bool DUAL>
__global__ __launch_bounds__(MRG_BLOCK, 2) void rowgemm_dma_k(GemmArgs a) {
  constexpr int GBK = 16, GBM = 128, GLD = GBK, F4R = GBK / 4;
  constexpr int NA = GBM * F4R / MRG_BLOCK, NBT = NT * 32 * F4R, NB = (NBT + MRG_BLOCK - 1) / MRG_BLOCK;
  extern __shared__ __align__(16) float smem[];
  constexpr int A_TILE = GBM * GLD, B_TILE = NT * 32 * GLD, STAGE = A_TILE + B_TILE;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * GBM;
  const int col0 = blockIdx.y * (NT * 32);
  const int K = a.K1 + a.K2;
  const int nkt = (K + GBK - 1) / GBK;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const float* arow1[NA]; const float* arow2[NA]; const float* brow[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t row = row0 + (tid + i * MRG_BLOCK) / F4R;
    const int64_t rc = row < a.rows ? row : a.rows - 1;
    arow1[i] = a.A1 + rc * a.K1;
    arow2[i] = a.A2 + rc * a.K2;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int f = tid + i * MRG_BLOCK;
    f = f < NBT ? f : NBT - 1;
    const int col = col0 + f / F4R;
    brow[i] = a.B + (int64_t)(col < a.N ? col : a.N - 1) * a.ldb;
  }
  auto fetch = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + i * MRG_BLOCK;
      const int k = k0 + (f % F4R) * 4;
      const float* p;
      if (DUAL) {
        const bool first = k < a.K1;
        const int kk = first ? k : k - a.K1, ld = first ? a.K1 : a.K2;
        p = (first ? arow1[i] : arow2[i]) + (kk + 4 <= ld ? kk : ld - 4);
      } else {
        p = arow1[i] + (k + 4 <= K ? k : K - 4);
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(smem + buf * STAGE + (f - lane) * 4), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + i * MRG_BLOCK;
      const int fc = f < NBT ? f : NBT - 1;
      const int k = k0 + (fc % F4R) * 4;
      if (f - lane < NBT)                      // wave-uniform
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(brow[i] + (k + 4 <= K ? k : K - 4)),
                                         (lds_ptr_t)(smem + buf * STAGE + A_TILE + (f - lane) * 4), 16, 0, 0);
    }
  };

  // Fragment reads are issued through inline asm: hipcc otherwise puts s_waitcnt vmcnt(0) in front of the
  // first ds_read that follows a global_load_lds (it assumes the DMA may alias it), which would drain the
  // DMA of the NEXT tile before this tile's MFMAs start.  Hazards are handled by hand: the reads only touch
  // the buffer whose DMA was waited for (vmcnt(0)) before the barrier that ended the previous iteration.
  typedef float v4f __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned a_off = (unsigned)(((wave * 32 + li) * GLD + lh * 4) * sizeof(float));
  const unsigned b_off = (unsigned)((A_TILE + li * GLD + lh * 4) * sizeof(float));
  auto step = [&](unsigned abase, unsigned bbase, int t) {     // 8 k-columns: 1 + NT fragment reads, 4*NT MFMAs
    v4f a4, b4[NT];
    asm volatile("ds_read_b128 %0, %1" : "=v"(a4) : "v"(abase + t * 32));
#pragma unroll
    for (int n = 0; n < NT; ++n) asm volatile("ds_read_b128 %0, %1" : "=v"(b4[n]) : "v"(bbase + n * (32 * GLD * 4) + t * 32));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);                         // keep the MFMAs below the wait (they do not touch memory)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[n].x, acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[n].y, acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[n].z, acc[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[n].w, acc[n], 0, 0, 0);
  };
  const int nfull = K / GBK;                                // tiles with both 8-column steps valid
  fetch(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int kt = 0; kt < nfull; ++kt) {                      // DMA of tile kt+1 runs under the MFMAs of tile kt
    if (kt + 1 < nkt) fetch(cur ^ 1, (kt + 1) * GBK);
    const unsigned abase = lds0 + cur * (STAGE * 4) + a_off, bbase = lds0 + cur * (STAGE * 4) + b_off;
    step(abase, bbase, 0);
    step(abase, bbase, 1);
    if (kt + 1 < nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA has landed ...
      __builtin_amdgcn_s_barrier();                        // ... and so has everyone else's; all reads of `cur` are done
      cur ^= 1;
    }
  }
  if (nfull < nkt) {                                        // K % 16 == 8: one last half tile
    step(lds0 + cur * (STAGE * 4) + a_off, lds0 + cur * (STAGE * 4) + b_off, 0);
  }

  gemm_epilogue<NT, EPI>(a, acc, row0 + wave * 32, col0, li, lh, row0 + GBM <= a.rows);
}

#ifndef MRG_FORCE_NT4
#define MRG_FORCE_NT4 0     // lab: column blocks of four tiles whatever the width (N = 200: 2 blocks, 8 tiles, A read twice)
#endif
inline int gemm_pick_nt(int ncols) {
  int t = (ncols + 31) / 32;
  if (MRG_FORCE_NT4 && t > 4) return 4;
  if (t <= 1) return 1;
  if (t <= 2) return 2;
  if (t <= 4) return 4;
  if (t <= 7) return 7;
  // wider outputs use several column blocks: the block width that pads the fewest all-zero tiles
  // (256 columns: 2 x 4 tiles, not 7 + 1 padded to 14), the wider one on ties (A is re-read per column block)
  int best = 7, waste = ((t + 6) / 7) * 7 - t;
  if (((t + 3) / 4) * 4 - t < waste) best = 4;
  return best;
}

inline size_t gemm_lds_bytes(int nt, int mt, int bk) { return (size_t)2 * (128 * mt + nt * 32) * (bk + 4) * sizeof(float); }

template <int EPI>
inline int launch_rowgemm(GemmArgs a, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const bool vec = (a.K1 % 4 == 0) && (a.K2 % 4 == 0) && (a.ldb % 4 == 0) && ((a.K1 + a.K2) % 4 == 0) && aligned16(a.A1) &&
                   aligned16(a.A2) && aligned16(a.B) && a.K1 >= 4 && (a.K2 == 0 || a.K2 >= 4);
  const int nt = gemm_pick_nt(a.N);
  if (vec && (a.K1 % 8 == 0) && (a.K2 % 8 == 0)) {            // LDS-DMA fast path
    dim3 gridd((unsigned)((a.rows + 127) / 128), (unsigned)((a.N + nt * 32 - 1) / (nt * 32)));
    const size_t ldsd = (size_t)2 * (128 + nt * 32) * 16 * sizeof(float);
#define MRG_GOD(NTV)                                                                                                  \
  do {                                                                                                                \
    if (a.K2 > 0) {                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_dma_k<NTV, EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd); \
      hipLaunchKernelGGL((rowgemm_dma_k<NTV, EPI, true>), gridd, dim3(MRG_BLOCK), ldsd, st, a);                        \
    } else {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_dma_k<NTV, EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd); \
      hipLaunchKernelGGL((rowgemm_dma_k<NTV, EPI, false>), gridd, dim3(MRG_BLOCK), ldsd, st, a);                       \
    }                                                                                                                 \
  } while (0)
    switch (nt) {
      case 1: MRG_GOD(1); break;
      case 2: MRG_GOD(2); break;
      case 4: MRG_GOD(4); break;
      default: MRG_GOD(7); break;
    }
#undef MRG_GOD
    hipError_t ed = hipGetLastError();
    return ed == hipSuccess ? MRG_OK : (int)ed;
  }
  const int mt = 1;
  const int gbm = 128 * mt;
  dim3 grid((unsigned)((a.rows + gbm - 1) / gbm), (unsigned)((a.N + nt * 32 - 1) / (nt * 32)));
  const size_t lds = gemm_lds_bytes(nt, mt, mt == 2 ? 32 : 16);
#define MRG_GO1(NTV, MTV, BKV, VV)                                                                                         \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_k<NTV, MTV, BKV, EPI, VV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((rowgemm_k<NTV, MTV, BKV, EPI, VV>), grid, dim3(MRG_BLOCK), lds, st, a);                             \
  } while (0)
#define MRG_GO(NTV)                                                                                                   \
  do {                                                                                                                \
    if (vec) MRG_GO1(NTV, 1, 16, true);                                                                               \
    else MRG_GO1(NTV, 1, 16, false);                                                                                  \
  } while (0)
  switch (nt) {
    case 1: MRG_GO(1); break;
    case 2: MRG_GO(2); break;
    case 4: MRG_GO(4); break;
    default: MRG_GO(7); break;
  }
#undef MRG_GO
#undef MRG_GO1
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MRG_OK : (int)e;
}

// Bt[c][r] = B[r][c]  (small weight matrices; used to present W^T row-major to the core)
static __global__ void transpose_k(const float* __restrict__ B, float* __restrict__ Bt, int rows, int cols, int ldb) {
  __shared__ float tile[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (r0 + i < rows && c < cols) tile[i][threadIdx.x] = B[(int64_t)(r0 + i) * ldb + c];
  __syncthreads();
  int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (c0 + i < cols && r < rows) Bt[(int64_t)(c0 + i) * rows + r] = tile[threadIdx.x][i];
}

inline void launch_transpose(const float* B, float* Bt, int rows, int cols, int ldb, hipStream_t st) {
  hipLaunchKernelGGL(transpose_k, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st, B, Bt, rows, cols, ldb);
}

}  // namespace mrg
