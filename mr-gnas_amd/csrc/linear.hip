// Dense linears on edge / node rows with exact-f32 MFMA (core: gemm.hpp).
//   nn.Linear inside a_max_op / a_mean_op: reference models/operations_lp.py:228,231,243,246
//   W_O / W_I / W_S / W_R of CompGraphConv:  reference models/compgcn.py:36-41,77-78,100,103
// forward / input-gradient use the pipelined row GEMM; the weight gradient is a split-over-rows
// GEMM (each workgroup reduces a chunk of rows into register-resident output tiles, partial
// tiles are combined in a fixed order), also software-pipelined, and optionally dual-source.
#include "gemm_dispatch.hpp"

namespace mrg {

// Weight gradient: partial[g][n][c] = sum over the workgroup's rows r of gY[r][n] * X'[r][c],
// X' = [X1 | X2 | 1] (the appended column of ones yields the bias gradient for free).
constexpr int WBR = 16;    // rows per LDS tile

struct WgradArgs {
  const float* gY; int Nout; int ldg;        // [rows][ldg], Nout valid columns (a column block of a wider gradient)
  const float* X1; const float* X2; int K1, K2;
  float* ws;
  int64_t rows, rows_per_block;
  int TM, TN, TNB;
  // grouped launch (wgrad_x3_k / wgrad_reduce3_k, gridDim.z = 3): row range, plan and partial-tile workspace of range z
  int ngrp;
  int64_t g_lo[3], g_hi[3], g_rpb[3], g_ws_off[3];      // g_ws_off in floats
  int g_G[3];
};

// X' = [X1 | X2 | 1 | 0...]: which tensor / local column a global column c maps to
struct XSel { const float* base; int ld, kk; };
__device__ __forceinline__ XSel wgrad_sel_x(const WgradArgs& a, int c) {
  const bool first = c < a.K1 || a.K2 == 0;
  XSel s;
  s.base = first ? a.X1 : a.X2;
  s.ld = first ? a.K1 : a.K2;
  s.kk = first ? c : c - a.K1;
  return s;
}

template <int TPW, int NPF, bool VEC4>      // TPW accumulator tiles per wave, NPF prefetch float4 per thread
__global__ __launch_bounds__(MRG_BLOCK, (TPW <= 7 ? 2 : 1)) void wgrad_k(WgradArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int tn0 = (int)(((int64_t)blockIdx.y * a.TN) / gridDim.y);                 // balanced split of the column tiles
  const int tnb = (int)(((int64_t)(blockIdx.y + 1) * a.TN) / gridDim.y) - tn0;
  const int ldg = a.TM * 32, ldx = tnb * 32, stage = WBR * (ldg + a.TNB * 32);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ntiles = a.TM * tnb;
  const int g4 = ldg / 4, x4 = ldx / 4, nf4 = WBR * (g4 + x4);      // float4 per staged tile
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r_end = r_begin + a.rows_per_block;
  if (r_end > a.rows) r_end = a.rows;

  float4 pf[NPF];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      int f = tid + i * MRG_BLOCK;
      // every lane issues ONE load from a clamped address; which operand it is, is a select
      const bool isg = f < WBR * g4;
      const int f2 = isg ? f : (f < nf4 ? f - WBR * g4 : 0);
      const int per = isg ? g4 : x4;
      const int r = f2 / per, c4 = f2 - r * per;
      if (VEC4) {
        const XSel sx = wgrad_sel_x(a, tn0 * 32 + c4 * 4);
        const float* base = isg ? a.gY : sx.base;
        const int ld = isg ? a.Nout : sx.ld;
        const int kk = isg ? c4 * 4 : sx.kk;
        pf[i] = gemm_raw4<true>(base, r0 + r, r_end, kk, ld, ld);
      } else {                                  // scalar path: per-element source selection
        const int64_t rc = r0 + r < r_end ? r0 + r : r_end - 1;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const XSel sx = wgrad_sel_x(a, tn0 * 32 + c4 * 4 + j);
          const float* base = isg ? a.gY : sx.base;
          const int ld = isg ? a.Nout : sx.ld;
          const int kk = isg ? c4 * 4 + j : sx.kk;
          v[j] = base[rc * ld + (kk < ld ? kk : ld - 1)];
        }
        pf[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto stash = [&](int buf, int64_t r0) {
    const int K = a.K1 + a.K2;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      int f = tid + i * MRG_BLOCK;
      if (f < nf4) {
        const bool isg = f < WBR * g4;
        const int f2 = isg ? f : f - WBR * g4;
        const int per = isg ? g4 : x4;
        const int r = f2 / per, c4 = f2 - r * per;
        const int c = tn0 * 32 + c4 * 4;
        const XSel sx = wgrad_sel_x(a, c);
        float4 v;
        if (VEC4) {
          v = gemm_mask4<true>(pf[i], r0 + r, r_end, isg ? c4 * 4 : sx.kk, isg ? a.Nout : sx.ld);
        } else {                                // columns are valid up to Nout (gY) or K (X'); the ones column is set below
          v = gemm_mask4<false>(pf[i], r0 + r, r_end, isg ? c4 * 4 : c, isg ? a.Nout : K);
        }
        if (!isg && r0 + r < r_end) {            // the appended column of ones
          if (c == K) v.x = 1.0f;
          if (!VEC4) { if (c + 1 == K) v.y = 1.0f; if (c + 2 == K) v.z = 1.0f; if (c + 3 == K) v.w = 1.0f; }
        }
        *reinterpret_cast<float4*>(&smem[buf * stage + f * 4]) = v;                    // Gs then Xs, both dense row-major
      }
    }
  };

  if (r_begin < r_end) {
    fetch(r_begin);
    stash(0, r_begin);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += WBR) {
    const bool more = r0 + WBR < r_end;
    if (more) fetch(r0 + WBR);
    const float* Gs = smem + cur * stage;
    const float* Xs = Gs + WBR * ldg;
#pragma unroll 1
    for (int t = 0; t < WBR / 2; ++t) {
      const float* grow = Gs + (2 * t + lh) * ldg + li;
      const float* xrow = Xs + (2 * t + lh) * ldx + li;
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int id = wave + 4 * i;
        if (id < ntiles) {
          const int m = id / tnb, n = id - m * tnb;
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[m * 32], xrow[n * 32], acc[i], 0, 0, 0);
        }
      }
    }
    if (more) {
      stash(cur ^ 1, r0 + WBR);
      __syncthreads();
      cur ^= 1;
    }
  }
  const int ldw = a.TN * 32;
  float* out = a.ws + (int64_t)blockIdx.x * ldg * ldw;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int id = wave + 4 * i;
    if (id < ntiles) {
      const int m = id / tnb, n = id - m * tnb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(int64_t)row * ldw + (tn0 + n) * 32 + li] = acc[i][r];
      }
    }
  }
}

// ---- LDS-DMA variant of the weight gradient (vector path) -------------------------------------
// Tiles go global -> LDS with global_load_lds_dwordx4 (no register staging, no ds_write); the DMA
// cannot synthesise values, so rows beyond the chunk and padding columns are sourced from a block
// of zeros and the appended bias column from a {1,0,0,0} constant.  MFMA operands are read with
// inline-asm ds_read_b32 so that hipcc does not drain the in-flight DMA before every LDS read.
static __device__ float mrg_zeros16[4] = {0.f, 0.f, 0.f, 0.f};
static __device__ float mrg_ones16[4] = {1.f, 0.f, 0.f, 0.f};

template <int TPW, int NPF>
__global__ __launch_bounds__(MRG_BLOCK, (TPW <= 7 ? 2 : 1)) void wgrad_dma_k(WgradArgs a) {
  extern __shared__ __align__(16) float smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tn0 = (int)(((int64_t)blockIdx.y * a.TN) / gridDim.y);
  const int tnb = (int)(((int64_t)(blockIdx.y + 1) * a.TN) / gridDim.y) - tn0;
  const int ldg = a.TM * 32, ldx = tnb * 32, stage = WBR * (ldg + a.TNB * 32);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int g4 = ldg / 4, x4 = ldx / 4, nf4 = WBR * (g4 + x4);
  const int K = a.K1 + a.K2;
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r_end = r_begin + a.rows_per_block;
  if (r_end > a.rows) r_end = a.rows;

  // loop-invariant DMA metadata of this thread's float4 slots
  const float* src[NPF]; int64_t stride[NPF]; int rr[NPF]; int kind[NPF];     // kind: 0 data, 1 ones, 2 zeros
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int f = tid + i * MRG_BLOCK;
    const bool isg = f < WBR * g4;
    const int f2 = isg ? f : (f < nf4 ? f - WBR * g4 : 0);
    const int per = isg ? g4 : x4;
    const int r = f2 / per, c = (f2 - r * per) * 4 + (isg ? 0 : tn0 * 32);
    rr[i] = r;
    if (isg) {
      kind[i] = c < a.Nout ? 0 : 2;
      src[i] = a.gY + (r_begin + r) * a.ldg + (c < a.Nout ? c : 0);
      stride[i] = (int64_t)WBR * a.ldg;
    } else {
      const XSel sx = wgrad_sel_x(a, c < K ? c : 0);
      kind[i] = c < K ? 0 : (c == K ? 1 : 2);
      src[i] = sx.base + (r_begin + r) * sx.ld + sx.kk;
      stride[i] = (int64_t)WBR * sx.ld;
    }
  }
  auto fetch = [&](int buf, int64_t r0, int64_t tile) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int f = tid + i * MRG_BLOCK;
      if (f - lane < nf4) {                                 // wave-uniform: this wave-instruction has work
        const bool rv = r0 + rr[i] < r_end;
        const float* p = (rv && kind[i] == 0) ? src[i] + tile * stride[i] : ((rv && kind[i] == 1) ? mrg_ones16 : mrg_zeros16);
        if (f < nf4)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(smem + buf * stage + (f - lane) * 4), 16, 0, 0);
      }
    }
  };
  // per-tile LDS byte offsets of this lane's operands
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  // wave w owns column tile n = w of the block (tnb <= 4) and every row tile m = i: one X' fragment read
  // feeds TM MFMAs (LDS delivers 64 B/clk per CU; two fragment reads per 64-cycle MFMA on 8 waves saturate it)
  unsigned goff[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) goff[i] = (unsigned)((lh * ldg + (i < a.TM ? i : 0) * 32 + li) * 4);
  const unsigned xoff = (unsigned)((WBR * ldg + lh * ldx + (wave < tnb ? wave : 0) * 32 + li) * 4);
  const bool active = wave < tnb;
  // 3-slot LDS ring: the DMA runs two 16-row tiles (~3 us of MFMA time) ahead; vmcnt is waited on with the
  // number of younger DMA instructions of this wave, never drained
  int per_tile = 0;
#pragma unroll
  for (int i = 0; i < NPF; ++i) per_tile += (wave * 64 + i * MRG_BLOCK < nf4) ? 1 : 0;
  if (r_begin < r_end) fetch(0, r_begin, 0);
  if (r_begin + WBR < r_end) fetch(1, r_begin + WBR, 1);
  int cur = 0;
  int64_t tile = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += WBR, ++tile) {
    wait_vmcnt(r0 + WBR < r_end ? per_tile : 0);           // tile `tile` has landed (this wave's part) ...
    __builtin_amdgcn_s_barrier();                          // ... everyone's part; all reads of the previous tile are done
    if (r0 + 2 * WBR < r_end) fetch(cur >= 1 ? cur - 1 : 2, r0 + 2 * WBR, tile + 2);   // slot (tile + 2) % 3
    const unsigned base = lds0 + cur * stage * 4;
    // fragment reads of k-step t+1 are in flight while the MFMAs of k-step t issue
    float gv[2][TPW], xv[2];
    auto frag = [&](int b, int t) {
      asm volatile("ds_read_b32 %0, %1" : "=v"(xv[b]) : "v"(base + xoff + t * 2 * ldx * 4));
#pragma unroll
      for (int i = 0; i < TPW; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(gv[b][i]) : "v"(base + goff[i] + t * 2 * ldg * 4));
    };
    frag(0, 0);
#pragma unroll
    for (int t = 0; t < WBR / 2; ++t) {
      const int c = t & 1;
      if (t + 1 < WBR / 2) {
        frag(c ^ 1, t + 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TPW + 1) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (active) {
#pragma unroll
        for (int i = 0; i < TPW; ++i)
          if (i < a.TM) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[c][i], xv[c], acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  const int ldw = a.TN * 32;
  float* out = a.ws + (int64_t)blockIdx.x * ldg * ldw;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (active && i < a.TM) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(int64_t)row * ldw + (tn0 + wave) * 32 + li] = acc[i][r];
      }
    }
  }
}

// ---- split-bf16 weight gradient (see gemm_x3.hpp for the arithmetic) ------------------------------
// Both operands are activations here, so both are split in registers.  512 threads = 8 waves (two per SIMD:
// the splits of one wave run under the MFMAs of the other); the workgroup owns all TM <= 7 row tiles of gW
// and KT = 8 (NG = 2) or 16 (NG = 1, TM <= 4) column tiles; wave (g, p) owns row tiles [4g, 4g+4) x column
// tiles {2p, 2p+1}: 6 fragments are split (264 VALU instructions) for 48 MFMAs per 16 rows.
// LDS: 4-slot ring (the DMA runs three tiles ahead) of [16 rows][A: 57 chunks | B: KT*8+1 chunks] (16-byte chunks; the odd pitch makes the
// transposed fragment reads -- 8 ds_read_b32, rows 8h..8h+7 of one column per lane -- conflict free).
constexpr int WX_THREADS = 512;
constexpr int WX_APITCH = 57 * 4;            // floats per A row in LDS (224 columns + one pad chunk)

template <int NG>
__global__ __launch_bounds__(WX_THREADS, 1) void wgrad_x3_k(WgradArgs a) {
  constexpr int KP = 8 / NG, KT = 2 * KP;
  constexpr int SLOTS = NG == 2 ? 4 : 3;              // LDS ring; the DMA runs SLOTS-1 tiles ahead
  constexpr int BPITCH = (KT * 8 + 1) * 4;
  constexpr int ACH = WBR * 57, BCH = WBR * (KT * 8 + 1), STAGE_CH = ACH + BCH;
  constexpr int NPF = (STAGE_CH + WX_THREADS - 1) / WX_THREADS;
  extern __shared__ __align__(16) float smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tn0 = (int)(((int64_t)blockIdx.y * a.TN) / gridDim.y);
  const int tnb = (int)(((int64_t)(blockIdx.y + 1) * a.TN) / gridDim.y) - tn0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wg = wave / KP, wp = wave % KP;
  const int m0 = wg * 4;
  const int an = a.TM - m0 < 4 ? (a.TM - m0 > 0 ? a.TM - m0 : 0) : 4;      // row tiles of this wave
  const int kn = tnb - 2 * wp < 2 ? (tnb - 2 * wp > 0 ? tnb - 2 * wp : 0) : 2;   // column tiles of this wave
  const int K = a.K1 + a.K2;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r_end = r_begin + a.rows_per_block;
  if (r_end > a.rows) r_end = a.rows;
  int64_t ws_off = 0;
  if (a.ngrp > 0) {                                       // one row range per blockIdx.z (constant indices: no scratch copy of the arguments)
    const int z = blockIdx.z;
#define MRG_PICKZ(F) (z == 0 ? a.F[0] : (z == 1 ? a.F[1] : a.F[2]))
    if ((int)blockIdx.x >= MRG_PICKZ(g_G)) return;          // workgroup-uniform, before any barrier
    const int64_t rpb = MRG_PICKZ(g_rpb), hi = MRG_PICKZ(g_hi);
    r_begin = MRG_PICKZ(g_lo) + (int64_t)blockIdx.x * rpb;
    r_end = r_begin + rpb < hi ? r_begin + rpb : hi;
    ws_off = MRG_PICKZ(g_ws_off);
#undef MRG_PICKZ
  }

  // loop-invariant DMA metadata of this thread's chunks
  const float* src[NPF]; int64_t stride[NPF]; int rr[NPF]; int kind[NPF];     // kind: 0 data, 1 ones, 2 zeros
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int f = tid + i * WX_THREADS;
    const bool isg = f < ACH;
    const int f2 = isg ? f : (f < STAGE_CH ? f - ACH : 0);
    const int per = isg ? 57 : KT * 8 + 1;
    const int r = f2 / per, c = (f2 - r * per) * 4;
    rr[i] = r;
    if (isg) {
      kind[i] = c < a.Nout ? 0 : 2;
      src[i] = a.gY + (r_begin + r) * a.ldg + (c < a.Nout ? c : 0);
      stride[i] = (int64_t)WBR * a.ldg;
    } else {
      const int cg = tn0 * 32 + c;
      const bool incol = c < tnb * 32;
      const XSel sx = wgrad_sel_x(a, (incol && cg < K) ? cg : 0);
      kind[i] = (incol && cg < K) ? 0 : ((incol && cg == K) ? 1 : 2);
      src[i] = sx.base + (r_begin + r) * sx.ld + sx.kk;
      stride[i] = (int64_t)WBR * sx.ld;
    }
  }
  int per_tile = 0;
#pragma unroll
  for (int i = 0; i < NPF; ++i) per_tile += (wave * 64 + i * WX_THREADS < STAGE_CH) ? 1 : 0;
  auto fetch = [&](int buf, int64_t r0, int64_t tile) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int f = tid + i * WX_THREADS;
      if (f - lane < STAGE_CH) {                            // wave-uniform
        const bool rv = r0 + rr[i] < r_end;
        const float* p = (rv && kind[i] == 0) ? src[i] + tile * stride[i] : ((rv && kind[i] == 1) ? mrg_ones16 : mrg_zeros16);
        if (f < STAGE_CH)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(smem + (buf * STAGE_CH + (f - lane)) * 4), 16, 0, 0);
      }
    }
  };

  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned a_off = (unsigned)((8 * lh * WX_APITCH + m0 * 32 + li) * 4);
  const unsigned b_off = (unsigned)((ACH * 4 + 8 * lh * BPITCH + 2 * wp * 32 + li) * 4);
  // one fragment: rows 8h..8h+7 of one column, split into three bf16 planes
  auto frag = [&](unsigned addr, auto pitch_c, u32x4& H, u32x4& M, u32x4& L) {
    constexpr int PB = decltype(pitch_c)::value * 4;
    float v[8];
    asm volatile("ds_read_b32 %0, %1" : "=v"(v[0]) : "v"(addr));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[1]) : "v"(addr), "n"(PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[2]) : "v"(addr), "n"(2 * PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[3]) : "v"(addr), "n"(3 * PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[4]) : "v"(addr), "n"(4 * PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[5]) : "v"(addr), "n"(5 * PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[6]) : "v"(addr), "n"(6 * PB));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[7]) : "v"(addr), "n"(7 * PB));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h, m, l;
      split_pair(v[2 * j], v[2 * j + 1], h, m, l);
      H[j] = h; M[j] = m; L[j] = l;
    }
  };

  if (r_begin < r_end) fetch(0, r_begin, 0);
#pragma unroll
  for (int t = 1; t < SLOTS - 1; ++t)
    if (r_begin + t * WBR < r_end) fetch(t, r_begin + t * WBR, t);
  int cur = 0;
  int64_t tile = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += WBR, ++tile) {
    {                                                      // younger DMA: the tiles already issued behind this one
      int64_t left = (r_end - r0 + WBR - 1) / WBR - 1;
      wait_vmcnt((int)(left < SLOTS - 2 ? left : SLOTS - 2) * per_tile);
    }
    __builtin_amdgcn_s_barrier();
    if (r0 + (SLOTS - 1) * WBR < r_end)                    // into the slot read during the previous tile
      fetch(cur == 0 ? SLOTS - 1 : cur - 1, r0 + (SLOTS - 1) * WBR, tile + SLOTS - 1);
    const unsigned base = lds0 + cur * (STAGE_CH * 16);
    if (an > 0 && kn > 0) {
      u32x4 bh[2], bm[2], bl[2];
      frag(base + b_off, std::integral_constant<int, BPITCH>{}, bh[0], bm[0], bl[0]);
      if (kn > 1) frag(base + b_off + 128, std::integral_constant<int, BPITCH>{}, bh[1], bm[1], bl[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < an) {
          u32x4 ah, am, al;
          frag(base + a_off + i * 128, std::integral_constant<int, WX_APITCH>{}, ah, am, al);
          const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
          const bf16x8 Bh0 = __builtin_bit_cast(bf16x8, bh[0]), Bm0 = __builtin_bit_cast(bf16x8, bm[0]), Bl0 = __builtin_bit_cast(bf16x8, bl[0]);
          if (kn > 1) {                                    // two accumulators interleaved: no back-to-back dependent MFMAs
            const bf16x8 Bh1 = __builtin_bit_cast(bf16x8, bh[1]), Bm1 = __builtin_bit_cast(bf16x8, bm[1]), Bl1 = __builtin_bit_cast(bf16x8, bl[1]);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh1, acc[i][1], 0, 0, 0);
          } else {
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh0, acc[i][0], 0, 0, 0);
          }
        }
      }
    }
    cur = cur == SLOTS - 1 ? 0 : cur + 1;
  }
  const int ldw = a.TN * 32;
  float* out = a.ws + ws_off + (int64_t)blockIdx.x * (a.TM * 32) * ldw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (i < an && j < kn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (m0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          out[(int64_t)row * ldw + (tn0 + 2 * wp + j) * 32 + li] = acc[i][j][r];
        }
      }
}

// ---- the same weight gradient with every fragment split ONCE per workgroup ------------------------------------------------
// wgrad_x3_k stages raw f32 tiles in LDS and every wave splits the fragments it multiplies: a gY fragment is split by the four
// waves that share its row tiles, an X fragment by two -- 264 VALU instructions per wave and 16-row tile beside 48 MFMAs, and
// eight ds_read_b32 per fragment.  Here fragment f of a tile is produced by ONE wave (f % 8): eight coalesced global_load_dword
// straight into the MFMA operand layout (lane = column f*32 + lane%32, rows 8*(lane/32) .. +7 -- 128 contiguous bytes per row
// and half wave), one split, three ds_write_b128 (bf16 planes, lane-contiguous: conflict free); the consumers read three
// ds_read_b128 per fragment.  Per workgroup and tile: 15 splits instead of 48 (TM = 7, KT = 8).  Two LDS buffers, one barrier per
// tile; the loads of tile t + 2 are issued before the barrier of tile t and consumed (split) during the MFMAs of tile t + 1.
// Same operands, same products, same accumulation order per output element as wgrad_x3_k: bit-identical partial tiles.
// lab (tools/wgrad_trace_lab.hip, -DMRG_WGRAD_TRACE=1): shader-clock stamps of workgroup (0, 0, 0), per wave and 16-row tile:
// mrg_wgrad_trace[(wave * 64 + tile) * 6 + i]: 0 top of the tile, 1 MFMAs issued, 2 fragments split and written, 3 next loads issued,
// 4 past the barrier, 5 the 100 MHz clock at the top
#ifndef MRG_WGRAD_TRACE
#define MRG_WGRAD_TRACE 0
#endif
#if MRG_WGRAD_TRACE
__device__ unsigned long long* mrg_wgrad_trace;
#define MRG_WG_STAMP(i) do { if (traced && lane == 0 && t < 64) mrg_wgrad_trace[(wave * 64 + (int)t) * 6 + (i)] = \
    (i) == 5 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MRG_WG_STAMP(i) do { } while (0)
#endif

template <int NG>
__global__ __launch_bounds__(WX_THREADS, 1) void wgrad_x3v_k(WgradArgs a) {
  constexpr int KP = 8 / NG, KT = 2 * KP;
  constexpr int ASLOTS = NG == 2 ? 8 : 4;                 // row-tile slots (TM <= 7 / <= 4)
  constexpr int NFRAG = ASLOTS + KT;
  constexpr int FPW = (NFRAG + 7) / 8;                    // fragments a wave produces per tile
  extern __shared__ __align__(16) float smem[];
    const int tn0 = (int)(((int64_t)blockIdx.y * a.TN) / gridDim.y);
  const int tnb = (int)(((int64_t)(blockIdx.y + 1) * a.TN) / gridDim.y) - tn0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wg = wave / KP, wp = wave % KP;
  const int m0 = wg * 4;
  const int an = a.TM - m0 < 4 ? (a.TM - m0 > 0 ? a.TM - m0 : 0) : 4;      // row tiles of this wave
  const int kn = tnb - 2 * wp < 2 ? (tnb - 2 * wp > 0 ? tnb - 2 * wp : 0) : 2;   // column tiles of this wave
  const int K = a.K1 + a.K2;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r_end = r_begin + a.rows_per_block;
  if (r_end > a.rows) r_end = a.rows;
  int64_t ws_off = 0;
  if (a.ngrp > 0) {
    const int z = blockIdx.z;
#define MRG_PICKZ(F) (z == 0 ? a.F[0] : (z == 1 ? a.F[1] : a.F[2]))
    if ((int)blockIdx.x >= MRG_PICKZ(g_G)) return;          // workgroup-uniform, before any barrier
    const int64_t rpb = MRG_PICKZ(g_rpb), hi = MRG_PICKZ(g_hi);
    r_begin = MRG_PICKZ(g_lo) + (int64_t)blockIdx.x * rpb;
    r_end = r_begin + rpb < hi ? r_begin + rpb : hi;
    ws_off = MRG_PICKZ(g_ws_off);
#undef MRG_PICKZ
  }

  // the fragments this wave produces: per lane one column, eight rows; constants (the bias column of ones, padding, fragments
  // that do not exist) come from a 4-float device array with row stride 0: the loads are the same straight-line code for every
  // wave and tile (a conditional load would make the compiler wait for it before the loop's back edge)
  const float* src[FPW]; int64_t ld[FPW]; bool live[FPW];
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int f = wave + 8 * i;
    live[i] = f < NFRAG && (f < ASLOTS ? f < a.TM : f - ASLOTS < tnb);
    const float* base = mrg_zeros16; int64_t l = 0;
    if (live[i]) {
      if (f < ASLOTS) {
        const int cg = f * 32 + li;
        if (cg < a.Nout) { base = a.gY + (r_begin + 8 * lh) * a.ldg + cg; l = a.ldg; }
      } else {
        const int cg = (tn0 + f - ASLOTS) * 32 + li;
        if (cg < K) {
          const XSel sx = wgrad_sel_x(a, cg);
          base = sx.base + (r_begin + 8 * lh) * sx.ld + sx.kk; l = sx.ld;
        } else if (cg == K) {
          base = mrg_ones16;
        }
      }
    }
    src[i] = base; ld[i] = l;
  }
  constexpr unsigned BUF_BYTES = FPW * 8 * 3 * 1024;

  // Full 16-row tiles run through the pipelined loop with plain loads; a ragged last tile (nv < 16 rows) is one extra, unpipelined
  // trip: its k positions >= nv are loaded from the rows 16 ABOVE their own (inside the tensor: the host guarantees r_end >= 16) and
  // zeroed at the split -- in-bounds loads without masks, and exactly the operand wgrad_x3_k builds (valid rows first, zeros after):
  // bit-identical with it for any row count.
  const int64_t nfull = (r_end - r_begin) / WBR;
  const int nv_tail = (int)((r_end - r_begin) - nfull * WBR);
  const int kbase = 8 * lh;
  float raw[FPW][8];
  auto fetch = [&](int64_t t) {                            // a FULL tile (t clamped by the caller)
    const int64_t toff = t * WBR;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const float* p = src[i] + toff * ld[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[i][j] = p[j * ld[i]];
    }
  };
  auto produce = [&](int buf, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      if (live[i]) {                                        // wave-uniform
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (!TAIL || kbase + j < nv_tail) ? raw[i][j] : 0.f;
        u32x4 H, M, L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned h, m, l;
          split_pair(x[2 * j], x[2 * j + 1], h, m, l);
          H[j] = h; M[j] = m; L[j] = l;
        }
        const int f = wave + 8 * i;
        u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + buf * BUF_BYTES + f * 3072) + lane;
        dst[0] = H; dst[64] = M; dst[128] = L;
      }
    }
  };
  auto rdfrag = [&](int buf, int f, bf16x8& H, bf16x8& M, bf16x8& L) {
    const u32x4* p = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(smem) + buf * BUF_BYTES + f * 3072) + lane;
    H = __builtin_bit_cast(bf16x8, p[0]); M = __builtin_bit_cast(bf16x8, p[64]); L = __builtin_bit_cast(bf16x8, p[128]);
  };
  auto consume = [&](int cur) {
    if (an > 0 && kn > 0) {
      bf16x8 Bh0, Bm0, Bl0, Bh1, Bm1, Bl1;
      rdfrag(cur, ASLOTS + 2 * wp, Bh0, Bm0, Bl0);
      if (kn > 1) rdfrag(cur, ASLOTS + 2 * wp + 1, Bh1, Bm1, Bl1);
      else { Bh1 = Bh0; Bm1 = Bm0; Bl1 = Bl0; }
      bf16x8 An[3];                                        // the NEXT row tile's planes: read while this one's MFMAs run
      rdfrag(cur, m0, An[0], An[1], An[2]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < an) {
          const bf16x8 Ah = An[0], Am = An[1], Al = An[2];
          if (i + 1 < an) rdfrag(cur, m0 + i + 1, An[0], An[1], An[2]);
          if (kn > 1) {                                    // two accumulators interleaved: no back-to-back dependent MFMAs
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm1, acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh1, acc[i][1], 0, 0, 0);
          } else {
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm0, acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh0, acc[i][0], 0, 0, 0);
          }
        }
      }
    }
  };

  if (nfull > 0) {
    fetch(0);
    produce(0, std::false_type{});
    fetch(nfull > 1 ? 1 : 0);
    __syncthreads();
    int cur = 0;
#if MRG_WGRAD_TRACE
    const bool traced = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#endif
    for (int64_t t = 0; t < nfull; ++t) {
      MRG_WG_STAMP(5); MRG_WG_STAMP(0);
      consume(cur);
      MRG_WG_STAMP(1);
      // tile t + 1 (its loads were issued one tile ago) is split under this tile's MFMAs; then the loads of tile t + 2 (clamped:
      // past the end the last full tile is simply loaded again and never used)
      if (t + 1 < nfull) produce(cur ^ 1, std::false_type{});
      MRG_WG_STAMP(2);
      fetch(t + 2 < nfull ? t + 2 : nfull - 1);
      MRG_WG_STAMP(3);
      __syncthreads();
      MRG_WG_STAMP(4);
      cur ^= 1;
    }
  }
  if (nv_tail > 0) {                                       // every wave is past the loop's last barrier: both buffers are free
    const int64_t toff = nfull * WBR;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const float* p = src[i] + toff * ld[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[i][j] = p[(int64_t)(kbase + j >= nv_tail ? j - WBR : j) * ld[i]];
    }
    produce(0, std::true_type{});
    __syncthreads();
    consume(0);
  }
  const int ldw = a.TN * 32;
  float* out = a.ws + ws_off + (int64_t)blockIdx.x * (a.TM * 32) * ldw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (i < an && j < kn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (m0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          out[(int64_t)row * ldw + (tn0 + 2 * wp + j) * 32 + li] = acc[i][j][r];
        }
      }
}

static int g_wgrad_variant = 1;           // 1: fragments split once per workgroup (wgrad_x3v_k), 0: per consuming wave (wgrad_x3_k)
inline int wgrad_variant() { return g_wgrad_variant; }

template <int NG>
static void launch_wgrad_x3(dim3 grid, const WgradArgs& a, hipStream_t st) {
  const int kt = 16 / NG;
  bool v1 = wgrad_variant() == 1 && a.rows >= WBR;         // (its ragged last tile is the 16 rows ENDING at the range's end)
  for (int i = 0; i < a.ngrp; ++i) v1 = v1 && !(a.g_hi[i] > a.g_lo[i] && a.g_hi[i] < WBR);
  if (v1) {
    const size_t lds = (size_t)2 * ((((NG == 2 ? 8 : 4) + kt) + 7) / 8 * 8) * 3 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3v_k<NG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((wgrad_x3v_k<NG>), grid, dim3(WX_THREADS), lds, st, a);
  } else {
    const size_t ldsx = (size_t)(NG == 2 ? 4 : 3) * WBR * (57 + kt * 8 + 1) * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_x3_k<NG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx);
    hipLaunchKernelGGL((wgrad_x3_k<NG>), grid, dim3(WX_THREADS), ldsx, st, a);
  }
}

// gW[n][c] = sum_g ws[g][n][c] (c < K);  gbias[n] = sum_g ws[g][n][K]   -- fixed order:
// thread row ty sums the partial tiles g = ty, ty+16, ..., the 16 row sums are added in order.
// Up to three row ranges in one launch (blockIdx.z).
struct WgradReduce3 { const float* ws[3]; float* gW[3]; float* gbias[3]; int G[3]; };
__global__ void wgrad_reduce3_k(WgradReduce3 p, int K, int Nout, int ldg, int ldx) {
  __shared__ float part[16][64];
  const int z = blockIdx.z;
  const float* __restrict__ ws = z == 0 ? p.ws[0] : (z == 1 ? p.ws[1] : p.ws[2]);
  float* __restrict__ gW = z == 0 ? p.gW[0] : (z == 1 ? p.gW[1] : p.gW[2]);
  float* __restrict__ gbias = z == 0 ? p.gbias[0] : (z == 1 ? p.gbias[1] : p.gbias[2]);
  const int G = z == 0 ? p.G[0] : (z == 1 ? p.G[1] : p.G[2]);
  if (!gW) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int n = blockIdx.y;
  float acc = 0.f;
  if (c <= K)
    for (int g = ty; g < G; g += 16) acc += ws[((int64_t)g * ldg + n) * ldx + c];
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c <= K) {
    float tot = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) tot += part[i][tx];
    if (c < K) gW[(int64_t)n * K + c] = tot;
    else if (gbias) gbias[n] = tot;
  }
}

struct WgradPlan {
  int TM, TN, TNB, tpw, npf, G;
  int64_t rows_per_block;
  size_t lds;
  bool ok;
};

// How many row ranges share one weight-gradient launch (the three direction segments of a dense filter: mrg_linear_bwd_weight3), so
// that the ranges TOGETHER, not each of them, get about one workgroup per CU: with 15 000-row ranges (the 30 000-edge search step) three
// ranges of 59 row blocks x 2 column blocks were 354 workgroups = 1.4 rounds of the chip, 46 us where one round of 23-tile blocks
// takes 33.  A lab path that launches the ranges one by one says so with mrg_wgrad_set_share, to keep the same partial sums.
static int g_wgrad_share = 1;
static int wgrad_share_max_blocks() { static const int v = [] { const char* e = getenv("MRG_WGRAD_SHARE_MAX"); return e ? atoi(e) : 400; }(); return v; }   // lab; measured: 15 000-row ranges -13 %, 87 000-row ranges (WN18RR) -10 %, 272 000-row ranges (1 063 blocks) +20 %

static WgradPlan wgrad_plan(int64_t rows, int K, int Nout, bool dma = false, int share = 0) {
  WgradPlan p{};
  p.TM = (Nout + 31) / 32;
  p.TN = (K + 1 + 31) / 32;
  const int opts[] = {1, 2, 4, 7, 13};
  p.tpw = 0;
  if (dma && p.TM <= 7) {
    // DMA kernel: wave w owns column tile w of the block (<= 4 per block) and all TM row tiles
    p.TNB = p.TN < 4 ? p.TN : 4;
    for (int o : opts) if (p.TM <= o) { p.tpw = o; break; }
  } else {
    // at most 28 accumulator tiles (7 per wave, 112 registers) per workgroup so that two workgroups
    // share a CU; wider outputs split the X' column tiles over grid.y (gY is then re-read per split)
    p.TNB = p.TM * p.TN <= 28 ? p.TN : (28 / p.TM > 0 ? 28 / p.TM : 0);
    int per_wave = p.TNB > 0 ? (p.TM * p.TNB + 3) / 4 : 99;
    for (int o : opts) if (per_wave <= o) { p.tpw = o; break; }
  }
  int nf4 = WBR * (p.TM + p.TNB) * 8;                    // float4 per staged tile
  p.npf = (nf4 + MRG_BLOCK - 1) / MRG_BLOCK;
  p.lds = (size_t)(dma ? 3 : 2) * WBR * (p.TM + p.TNB) * 32 * sizeof(float);
  p.ok = p.tpw > 0 && p.lds <= 160 * 1024 && p.npf <= 16;
  int64_t tiles = (rows + WBR - 1) / WBR;
  // >= 16 row tiles per workgroup: every extra workgroup costs a TM*32 x TN*32 partial tile (372 KB at 200 x 400)
  // that the ordered reduction has to read back
  // ... and about one workgroup per CU: 256 / (column blocks of the split-core kernel) row blocks (the figure depends on
  // the shape only, never on the kernel chosen: the workspace query and the launch must agree)
  const int ny = (p.TN + (p.TM <= 4 ? 16 : 8) - 1) / (p.TM <= 4 ? 16 : 8);
  // (only up to 400 row blocks of 16 tiles per range: measured -13 % at 15 000-row ranges,
  //  -10 % at 87 000-row ranges, but +20 % at 272 000-row ranges, where three rounds of 128 shorter workgroups per range beat one round of long ones)
  const int sh = (share > 0 ? share : g_wgrad_share);
  const bool shared = sh > 1 && (tiles + 15) / 16 <= wgrad_share_max_blocks();
  const int64_t gshare = 256 / (ny * sh);
  static const int gmul = [] { const char* e = getenv("MRG_WGRAD_GMUL"); return e ? atoi(e) : 1; }();      // lab: rounds of workgroups per range
  const int64_t gmax = !shared ? ((256 / ny > 32 ? 256 / ny : 32) * gmul) : (gshare > 16 ? gshare : 16);
  int64_t G = (tiles + 15) / 16 < gmax ? (tiles + 15) / 16 : gmax;
  // few rows (a sampled step graph, a rank's node chunk): a workgroup walks its 16-row tiles one barrier at a time (~2 us each), so
  // one or two workgroups of 16 tiles are a 30 us latency chain on an idle chip -- up to eight workgroups of >= 4 tiles instead
  // (the extra partial tiles are a few MB for the ordered reduction)
  const int64_t gsmall = (tiles + 3) / 4 < 8 ? (tiles + 3) / 4 : 8;
  if (G < gsmall) G = gsmall;
  if (G < 1) G = 1;
  int64_t tpb = (tiles + G - 1) / G;
  if (tpb < 1) tpb = 1;
  p.rows_per_block = tpb * WBR;
  p.G = (int)((rows + p.rows_per_block - 1) / p.rows_per_block);
  if (p.G < 1) p.G = 1;
  return p;
}

int64_t wgrad_workspace_bytes(int64_t rows, int K, int Nout) {
  WgradPlan p = wgrad_plan(rows, K, Nout);
  return (int64_t)p.G * p.TM * 32 * p.TN * 32 * sizeof(float);
}

// gW[Nout][K1+K2] = gY^T [X1 | X2], gbias = column sums of gY
static int launch_wgrad_one(const float* gY, int ldg, const float* X1, const float* X2, int K1, int K2, float* gW, float* gbias, void* ws,
                           int64_t rows, int Nout, hipStream_t st);

int launch_wgrad(const float* gY, const float* X1, const float* X2, int K1, int K2, float* gW, float* gbias, void* ws,
                 int64_t rows, int Nout, hipStream_t st) {
  // more than 7 row tiles of gW (Nout > 224, e.g. D = 256): balanced column blocks of gY, each a launch of the
  // <= 7-tile kernels (X is re-read per block); same workspace, stream ordered
  const bool splittable = Nout > 224 && rows > 0 && (Nout % 4 == 0) && (K1 % 4 == 0) && (K2 % 4 == 0) && aligned16(gY) && aligned16(X1) &&
                          (!X2 || K2 == 0 || aligned16(X2)) && K1 >= 4 && (K2 == 0 || K2 >= 4);
  if (splittable) {
    const int nblk = (Nout + 223) / 224;
    const int cw = (((Nout + nblk - 1) / nblk) + 31) / 32 * 32;
    for (int n0 = 0; n0 < Nout; n0 += cw) {
      const int nc = Nout - n0 < cw ? Nout - n0 : cw;
      int rc = launch_wgrad_one(gY + n0, Nout, X1, X2, K1, K2, gW + (int64_t)n0 * (K1 + K2), gbias ? gbias + n0 : nullptr, ws, rows, nc, st);
      if (rc != MRG_OK) return rc;
    }
    return MRG_OK;
  }
  return launch_wgrad_one(gY, Nout, X1, X2, K1, K2, gW, gbias, ws, rows, Nout, st);
}

static int launch_wgrad_one(const float* gY, int ldg, const float* X1, const float* X2, int K1, int K2, float* gW, float* gbias, void* ws,
                           int64_t rows, int Nout, hipStream_t st) {
  const int K = K1 + K2;
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)Nout * K, st);
    if (e == hipSuccess && gbias) e = hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)Nout, st);
    return (int)e;
  }
  const bool vec0 = (Nout % 4 == 0) && (K1 % 4 == 0) && (K2 % 4 == 0) && aligned16(gY) && aligned16(X1) && (!X2 || K2 == 0 || aligned16(X2)) &&
                    Nout >= 4 && K1 >= 4 && (K2 == 0 || K2 >= 4) && Nout <= 224;
  WgradPlan p = wgrad_plan(rows, K, Nout, vec0);
  if (!p.ok) return MRG_E_SHAPE;
  WgradArgs a{};
  a.gY = gY; a.Nout = Nout; a.ldg = ldg; a.X1 = X1; a.X2 = X2; a.K1 = K1; a.K2 = K2; a.ws = (float*)ws;
  a.rows = rows; a.rows_per_block = p.rows_per_block; a.TM = p.TM; a.TN = p.TN; a.TNB = p.TNB;
  if (!X2 || K2 == 0) { a.X2 = X1; a.K2 = 0; }
  const bool vec = vec0;
  if (vec && gemm_mode() != 1 && p.TM <= 7) {        // split-bf16 core
    const int ng = p.TM <= 4 ? 1 : 2, kt = 16 / ng;
    dim3 gridx(p.G, (p.TN + kt - 1) / kt);
    if (ng == 1) launch_wgrad_x3<1>(gridx, a, st);
    else launch_wgrad_x3<2>(gridx, a, st);
    MRG_LAUNCH_CHECK();
    WgradReduce3 red{};
    red.ws[0] = (const float*)ws; red.gW[0] = gW; red.gbias[0] = gbias; red.G[0] = p.G;
    hipLaunchKernelGGL(wgrad_reduce3_k, dim3((K + 1 + 63) / 64, Nout, 1), dim3(1024), 0, st, red, K, Nout, p.TM * 32, p.TN * 32);
    MRG_LAUNCH_CHECK();
    return MRG_OK;
  }
  dim3 grid(p.G, (p.TN + p.TNB - 1) / p.TNB);      // y-blocks own ~TN/grid.y column tiles each (<= TNB)
#define GO(T, F)                                                                                                       \
  do {                                                                                                                 \
    if (vec) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_k<T, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
      hipLaunchKernelGGL((wgrad_dma_k<T, F>), grid, dim3(MRG_BLOCK), p.lds, st, a);                                    \
    } else {                                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k<T, F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
      hipLaunchKernelGGL((wgrad_k<T, F, false>), grid, dim3(MRG_BLOCK), p.lds, st, a);                                 \
    }                                                                                                                  \
  } while (0)
#define GOF(T)                                                                                                         \
  do {                                                                                                                 \
    if (p.npf <= 2) GO(T, 2); else if (p.npf <= 4) GO(T, 4); else if (p.npf <= 8) GO(T, 8); else GO(T, 16);            \
  } while (0)
  switch (p.tpw) {
    case 1: GOF(1); break;
    case 2: GOF(2); break;
    case 4: GOF(4); break;
    case 7: GOF(7); break;
    default: GOF(13); break;
  }
#undef GOF
#undef GO
  MRG_LAUNCH_CHECK();
  WgradReduce3 red1{};
  red1.ws[0] = (const float*)ws; red1.gW[0] = gW; red1.gbias[0] = gbias; red1.G[0] = p.G;
  hipLaunchKernelGGL(wgrad_reduce3_k, dim3((K + 1 + 63) / 64, Nout, 1), dim3(1024), 0, st, red1, K, Nout, p.TM * 32, p.TN * 32);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// a_max, second half: unpack the 64-bit keys of gemm_epilogue_segmax.  out = max (0 without in-edge) + self row;
// arg = the winning edge id (-1 without in-edge); mx = the max itself (the backward's ReLU mask: mx > 0).
__global__ void segmax_finalize_k(const unsigned long long* __restrict__ keys, const float* __restrict__ self_rows,
                                  const int32_t* __restrict__ eid, float* __restrict__ out, int32_t* __restrict__ arg,
                                  float* __restrict__ mx, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    const unsigned lo = (unsigned)k;
    const float v = __uint_as_float((unsigned)(k >> 32));
    out[i] = self_rows ? v + self_rows[i] : v;
    if (arg) arg[i] = lo == 0u ? -1 : eid[0xFFFFFFFFu - lo];
    if (mx) mx[i] = v;
  }
}

}  // namespace mrg

using namespace mrg;

// ---- a_max as two launches: split-core GEMM with the ReLU + segmented-max epilogue, then the unpack pass ---------------
static bool segmax_shape_ok(int K, int Nout) { return K > 48 && K % 4 == 0 && Nout > 0; }

extern "C" int64_t mrg_linear_relu_segmax_workspace_bytes(int64_t N, int K, int Nout) {
  if (N < 0 || !segmax_shape_ok(K, Nout) || gemm_mode() == 1) return 0;   // 0: the split core cannot take the shape or is switched off (use the unfused entry points)
  return ((N * (int64_t)Nout * 8 + 255) / 256) * 256 + (int64_t)bsplit_bytes_any(Nout, K) + 256;
}

extern "C" int mrg_linear_relu_segmax_fwd(const float* X, const float* W, const float* bias, const int32_t* eid, const int32_t* dst,
                                          const float* self_rows, float* out, int32_t* arg, float* mx, void* ws, int64_t E, int64_t N,
                                          int K, int Nout, void* stream) {
  if (E < 0 || N < 0 || K <= 0 || Nout <= 0 || E >= ((int64_t)1 << 32) - 1) return MRG_E_SHAPE;
  if (!segmax_shape_ok(K, Nout)) return MRG_E_SHAPE;
  if (N == 0) return MRG_OK;
  if (!out || !W) return MRG_E_NULLPTR;
  if (E > 0 && (!X || !eid || !dst)) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  if (!aligned16(X) || !aligned16(ws)) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)ws;
  const size_t key_bytes = (size_t)((N * (int64_t)Nout * 8 + 255) / 256) * 256;
  void* bsplit = (char*)ws + key_bytes;
  if (hipMemsetAsync(keys, 0, (size_t)N * Nout * 8, st) != hipSuccess) return MRG_E_WORKSPACE;
  if (E > 0) {
    GemmArgs a{};
    a.A1 = X; a.K1 = K; a.B = W; a.bias = bias; a.N = Nout; a.rows = E; a.act = MRG_ACT_RELU;
    a.row_index = eid; a.row_seg = dst; a.seg_out = keys;
    if (!x3_eligible(a)) return MRG_E_SHAPE;
    launch_bsplit_any(EPI_SEGMAX, E, W, K, 1, Nout, K, bsplit, st);
    const int rc = launch_rowgemm_x3_mode<EPI_SEGMAX>(a, bsplit, st);
    if (rc != MRG_OK) return rc;
  }
  const int64_t total = N * (int64_t)Nout;
  hipLaunchKernelGGL(segmax_finalize_k, dim3(stream_grid_for(total, 256 * 4)), dim3(256), 0, st, keys, self_rows, eid, out, arg, mx, total);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

// ---- a_mean (a_sum over ReLU(Linear)) first level: the split-core GEMM over the edges in destination order with the
// run-sum epilogue (gemm_epilogue_segsum); second level: mrg_seg_reduce_heads_fwd.  part: [E, Nout] floats (only head rows
// are written), relu_bits: [E, ceil(Nout / 32)] words.  ws: mrg_gemm_workspace_bytes(K, Nout).
extern "C" int mrg_linear_relu_segsum_fwd(const float* X, const float* W, const float* bias, const int32_t* eid, const int32_t* dst,
                                          float* part, unsigned* relu_bits, void* ws, int64_t E, int K, int Nout, void* stream) {
  if (E < 0 || K <= 0 || Nout <= 0 || E >= ((int64_t)1 << 31)) return MRG_E_SHAPE;
  if (!segmax_shape_ok(K, Nout)) return MRG_E_SHAPE;
  if (E == 0) return MRG_OK;
  if (!X || !W || !eid || !dst || !part) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  if (!aligned16(X) || !aligned16(ws)) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.A1 = X; a.K1 = K; a.B = W; a.bias = bias; a.N = Nout; a.rows = E; a.act = MRG_ACT_RELU;
  a.row_index = eid; a.row_seg = dst; a.seg_part = part; a.relu_bits = relu_bits; a.bits_ld = (Nout + 31) / 32;
  if (!x3_eligible(a)) return MRG_E_SHAPE;
  launch_bsplit_any(EPI_SEGSUM, E, W, K, 1, Nout, K, ws, st);
  return launch_rowgemm_x3_mode<EPI_SEGSUM>(a, ws, st);
}

extern "C" int64_t mrg_gemm_workspace_bytes(int K, int Nout) {
  if (K <= 0 || Nout <= 0) return 0;
  return (int64_t)gemm_workspace_bytes(K, Nout);
}

extern "C" int mrg_gemm_set_mode(int mode) {
  if (mode < 0 || mode > 2) return MRG_E_ENUM;
  gemm_mode() = mode;
  return MRG_OK;
}

extern "C" int mrg_wgrad_set_variant(int variant) {
  if (variant != 0 && variant != 1) return MRG_E_ENUM;
  g_wgrad_variant = variant;
  return MRG_OK;
}

extern "C" int mrg_gemm_set_wide8(int on) {
  if (on < 0 || on > 2) return MRG_E_ENUM;          // 2 (lab): seven-tile plain launches on the ring-of-two kernel as well
  gemm_wide8() = on;
  return MRG_OK;
}

extern "C" int mrg_wgrad_set_share(int n) {
  if (n < 1 || n > 3) return MRG_E_ENUM;
  g_wgrad_share = n;
  return MRG_OK;
}

extern "C" int mrg_gemm_set_small(int on) {
  if (on < 0) return MRG_E_ENUM;
  gemm_small() = on == 1 ? X3N_MAX_ROWS : (int64_t)on;      // 0 off, 1 the default bound, > 1 (lab) that many rows
  return MRG_OK;
}

extern "C" int mrg_gemm_set_q(int on) {
  if (on < 0 || on > 2) return MRG_E_ENUM;          // 2 (lab, tests): every eligible K, not only K > 224
  gemm_q() = on;
  return MRG_OK;
}

extern "C" int mrg_gemm_set_epilogue(int mode) {
  if (mode < 0 || mode > 2) return MRG_E_ENUM;
  gemm_epi_lds() = mode == 1 ? 1 : 0;
  gemm_epi_mode() = mode;
  return MRG_OK;
}

extern "C" int mrg_linear_fwd(const float* X, const float* W, const float* bias, float* Y, void* ws, int64_t rows, int K, int Nout,
                              int act, void* stream) {
  if (rows < 0 || K <= 0 || Nout <= 0) return MRG_E_SHAPE;
  if (act != MRG_ACT_NONE && act != MRG_ACT_RELU && act != MRG_ACT_SIGMOID) return MRG_E_ENUM;
  if (rows == 0) return MRG_OK;
  if (!X || !W || !Y) return MRG_E_NULLPTR;
  GemmArgs a{};
  a.A1 = X; a.K1 = K; a.B = W; a.bias = bias; a.C = Y; a.ldc = Nout; a.N = Nout; a.rows = rows; a.act = act;
  return launch_gemm<EPI_BIAS_ACT>(a, K, 1, ws, (hipStream_t)stream);
}

extern "C" int64_t mrg_linear_bwd_input_workspace_bytes(int K, int Nout) {
  if (K <= 0 || Nout <= 0) return 0;
  return (int64_t)gemm_workspace_bytes(Nout, K);
}

// gX[rows, K] (+)= gY[rows, Nout] * W[:, 0:K]   where W is [Nout][ldw] row-major (ldw >= K: a column block
// of a wider weight, e.g. one half of an nn.Linear(2D, D)); accumulate != 0 adds into the existing gX.
// The core sees B(n = k_in, k = n_out) = W[n_out * ldw + k_in]: a strided view, no transpose pass on the split path.
extern "C" int mrg_linear_bwd_input(const float* gY, const float* W, float* gX, void* ws, int64_t rows, int K, int Nout,
                                    int ldw, int accumulate, void* stream) {
  if (rows < 0 || K <= 0 || Nout <= 0 || ldw < K) return MRG_E_SHAPE;
  if (rows == 0) return MRG_OK;
  if (!gY || !W || !gX) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.A1 = gY; a.K1 = Nout; a.B = W; a.C = gX; a.ldc = K; a.N = K; a.rows = rows; a.act = MRG_ACT_NONE;
  if (accumulate) {
    a.Cin = gX; a.ld_cin = K;
    return launch_gemm<EPI_ACCUM>(a, 1, ldw, ws, st);
  }
  return launch_gemm<EPI_BIAS_ACT>(a, 1, ldw, ws, st);
}

// ---- the three direction segments of a dense filter in one launch each (split core only) -----------------------------------
// gX rows [0, b0) (+)= gY W[0][:, 0:K], rows [b0, b1) with W[1], rows [b1, M) with W[2]; W: HOST array of three device
// pointers to [Nout][ldw] weights (a column block when offset by the caller).
static size_t bwd_input3_each(int K, int Nout) { return (size_t)(((int64_t)bsplit_bytes_any(K, Nout) + 255) / 256 * 256); }

extern "C" int64_t mrg_linear_bwd_input3_workspace_bytes(int K, int Nout) {
  if (K <= 0 || Nout <= 48 || Nout % 4 != 0 || gemm_mode() == 1) return 0;      // 0: the split core cannot take the shape / is switched off
  return 3 * (int64_t)bwd_input3_each(K, Nout);
}

extern "C" int mrg_linear_bwd_input3(const float* gY, const float* const* W_host, float* gX, void* ws, int64_t b0, int64_t b1, int64_t M,
                                     int K, int Nout, int ldw, int accumulate, void* stream) {
  if (K <= 0 || Nout <= 0 || ldw < K || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (K <= 0 || Nout <= 48 || Nout % 4 != 0) return MRG_E_SHAPE;
  if (M == 0) return MRG_OK;
  if (!gY || !W_host || !gX) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.A1 = gY; a.K1 = Nout; a.C = gX; a.ldc = K; a.N = K; a.rows = M; a.act = MRG_ACT_NONE;
  if (accumulate) { a.Cin = gX; a.ld_cin = K; }
  if (!x3_eligible(a)) return MRG_E_SHAPE;
  const int64_t lo[3] = {0, b0, b1}, hi[3] = {b0, b1, M};
  const size_t each = bwd_input3_each(K, Nout);
  const float* Bs[3]; void* outs[3];
  a.grp.n = 3;
  a.grp.bp_stride = (int64_t)each;
  for (int i = 0; i < 3; ++i) {
    const bool live = hi[i] > lo[i];
    if (live && !W_host[i]) return MRG_E_NULLPTR;
    Bs[i] = live ? W_host[i] : nullptr;
    outs[i] = (char*)ws + i * each;
    a.grp.lo[i] = lo[i]; a.grp.hi[i] = live ? hi[i] : lo[i];
    a.grp.scale[i] = 1.0f;
  }
  launch_bsplit3_any(accumulate ? EPI_ACCUM : EPI_BIAS_ACT, M, Bs, 1, ldw, K, Nout, outs, st);         // B(n = k_in, k = n_out) = W[n_out * ldw + k_in]
  MRG_LAUNCH_CHECK();
  if (accumulate) return launch_rowgemm_x3_mode<EPI_ACCUM>(a, outs[0], st);
  return launch_rowgemm_x3_mode<EPI_BIAS_ACT>(a, outs[0], st);
}

// gX rows of the three ranges (+)= gY1 W1[s][:, 0:K] + gY2 W2[s][:, 0:K] as ONE product over the concatenated reduction
// dimension, [gY1 | gY2] [W1 ; W2]: the input gradient of two candidates that read the same rows (f_dense_comp and f_comp of
// one MixedOp, reference models/cell_lp.py:95-113, models/operations_lp.py:266-288,356-390) written once instead of two
// gradients that a fan-in pass adds.
static size_t bwd_input3_pair_each(int K, int Nout) { return (size_t)(((int64_t)bsplit_bytes_any(K, 2 * Nout) + 255) / 256 * 256); }

extern "C" int64_t mrg_linear_bwd_input3_pair_workspace_bytes(int K, int Nout) {
  if (K <= 0 || Nout <= 24 || Nout % 4 != 0 || gemm_mode() == 1) return 0;
  return 3 * (int64_t)bwd_input3_pair_each(K, Nout);
}

extern "C" int mrg_linear_bwd_input3_pair(const float* gY1, const float* gY2, const float* const* W1_host, const float* const* W2_host, float* gX,
                                          void* ws, int64_t b0, int64_t b1, int64_t M, int K, int Nout, int ldw, int accumulate, void* stream) {
  if (K <= 0 || Nout <= 24 || Nout % 4 != 0 || ldw < K || M < 0 || b0 < 0 || b1 < b0 || M < b1) return MRG_E_SHAPE;
  if (M == 0) return MRG_OK;
  if (!gY1 || !gY2 || !W1_host || !W2_host || !gX) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.A1 = gY1; a.K1 = Nout; a.A2 = gY2; a.K2 = Nout; a.C = gX; a.ldc = K; a.N = K; a.rows = M; a.act = MRG_ACT_NONE;
  if (accumulate) { a.Cin = gX; a.ld_cin = K; }
  if (!x3_eligible(a)) return MRG_E_SHAPE;
  const int64_t lo[3] = {0, b0, b1}, hi[3] = {b0, b1, M};
  const size_t each = bwd_input3_pair_each(K, Nout);
  const float* Bs[3]; const float* Bs2[3]; void* outs[3];
  a.grp.n = 3;
  a.grp.bp_stride = (int64_t)each;
  for (int i = 0; i < 3; ++i) {
    const bool live = hi[i] > lo[i];
    if (live && (!W1_host[i] || !W2_host[i])) return MRG_E_NULLPTR;
    Bs[i] = live ? W1_host[i] : nullptr;
    Bs2[i] = live ? W2_host[i] : nullptr;
    outs[i] = (char*)ws + i * each;
    a.grp.lo[i] = lo[i]; a.grp.hi[i] = live ? hi[i] : lo[i];
    a.grp.scale[i] = 1.0f;
  }
  // B(n = k_in, k) = k < Nout ? W1[k * ldw + k_in] : W2[(k - Nout) * ldw + k_in]
  launch_bsplit3_any(accumulate ? EPI_ACCUM : EPI_BIAS_ACT, M, Bs, 1, ldw, K, 2 * Nout, outs, st, Bs2, Nout);
  MRG_LAUNCH_CHECK();
  if (accumulate) return launch_rowgemm_x3_mode<EPI_ACCUM>(a, outs[0], st);
  return launch_rowgemm_x3_mode<EPI_BIAS_ACT>(a, outs[0], st);
}

// gW[s][Nout][K1+K2] = gY[lo_s:hi_s]^T [X1 | X2][lo_s:hi_s], gbias[s] = column sums, for the three row ranges in one launch
static bool wgrad3_ok(int K1, int K2, int Nout) {
  return Nout >= 4 && Nout <= 224 && Nout % 4 == 0 && K1 >= 4 && K1 % 4 == 0 && K2 >= 0 && K2 % 4 == 0;
}

extern "C" int64_t mrg_linear_bwd_weight3_workspace_bytes(int64_t b0, int64_t b1, int64_t M, int K1, int K2, int Nout) {
  if (M < 0 || b0 < 0 || b1 < b0 || M < b1 || !wgrad3_ok(K1, K2, Nout) || gemm_mode() == 1) return 0;
  const int64_t rows[3] = {b0, b1 - b0, M - b1};
  int64_t total = 0;
  for (int i = 0; i < 3; ++i) total += (wgrad_workspace_bytes(rows[i], K1 + K2, Nout) + 255) / 256 * 256;
  return total;
}

extern "C" int mrg_linear_bwd_weight3(const float* gY, const float* X1, const float* X2, float* const* gW_host, float* const* gb_host, void* ws,
                                      int64_t b0, int64_t b1, int64_t M, int K1, int K2, int Nout, void* stream) {
  if (M < 0 || b0 < 0 || b1 < b0 || M < b1 || !wgrad3_ok(K1, K2, Nout)) return MRG_E_SHAPE;
  if (!gW_host) return MRG_E_NULLPTR;
  if (M > 0 && (!gY || !X1 || (K2 > 0 && !X2))) return MRG_E_NULLPTR;
  if (!ws) return MRG_E_WORKSPACE;
  if (!aligned16(gY) || !aligned16(X1) || (K2 > 0 && !aligned16(X2))) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int K = K1 + K2;
  const int64_t lo[3] = {0, b0, b1}, hi[3] = {b0, b1, M};
  WgradArgs a{};
  a.gY = gY; a.Nout = Nout; a.ldg = Nout; a.X1 = X1; a.X2 = K2 > 0 ? X2 : X1; a.K1 = K1; a.K2 = K2; a.ws = (float*)ws; a.rows = M;
  a.ngrp = 3;
  WgradReduce3 red{};
  int maxG = 0;
  int64_t off = 0;
  WgradPlan p0 = wgrad_plan(1, K, Nout, true);
  if (!p0.ok || p0.TM > 7) return MRG_E_SHAPE;
  a.TM = p0.TM; a.TN = p0.TN; a.TNB = p0.TNB;
  int nlive = 0;
  for (int i = 0; i < 3; ++i) nlive += (gW_host[i] != nullptr && hi[i] > lo[i]) ? 1 : 0;
  if (nlive < 1) nlive = 1;
  for (int i = 0; i < 3; ++i) {
    const int64_t rows = hi[i] - lo[i];
    WgradPlan p = wgrad_plan(rows, K, Nout, true, nlive);
    const bool live = gW_host[i] != nullptr;
    a.g_lo[i] = lo[i]; a.g_hi[i] = hi[i]; a.g_rpb[i] = p.rows_per_block; a.g_G[i] = live ? p.G : 0; a.g_ws_off[i] = off / (int64_t)sizeof(float);
    red.ws[i] = (const float*)((const char*)ws + off); red.gW[i] = gW_host[i]; red.gbias[i] = gb_host ? gb_host[i] : nullptr; red.G[i] = p.G;
    off += (wgrad_workspace_bytes(rows, K, Nout) + 255) / 256 * 256;
    if (live && p.G > maxG) maxG = p.G;
  }
  if (maxG == 0) return MRG_OK;
  const int ng = a.TM <= 4 ? 1 : 2, kt = 16 / ng;
  dim3 gridx(maxG, (a.TN + kt - 1) / kt, 3);
  if (ng == 1) launch_wgrad_x3<1>(gridx, a, st);
  else launch_wgrad_x3<2>(gridx, a, st);
  MRG_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_reduce3_k, dim3((K + 1 + 63) / 64, Nout, 3), dim3(1024), 0, st, red, K, Nout, a.TM * 32, a.TN * 32);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int64_t mrg_linear_bwd_weight_workspace_bytes(int64_t rows, int K, int Nout) {
  if (rows < 0 || K <= 0 || Nout <= 0) return 0;
  return wgrad_workspace_bytes(rows, K, Nout);
}

// gW[Nout][K1+K2] = gY^T [X1 | X2] (X2 NULL / K2 = 0: single source), gbias[Nout] = column sums of gY (NULL ok)
extern "C" int mrg_linear_bwd_weight(const float* gY, const float* X1, const float* X2, float* gW, float* gbias, void* ws,
                                     int64_t rows, int K1, int K2, int Nout, void* stream) {
  if (rows < 0 || K1 <= 0 || K2 < 0 || Nout <= 0) return MRG_E_SHAPE;
  if (!gW) return MRG_E_NULLPTR;
  if (rows > 0 && (!gY || !X1 || (K2 > 0 && !X2))) return MRG_E_NULLPTR;
  if (rows > 0 && !ws) return MRG_E_WORKSPACE;
  return launch_wgrad(gY, X1, K2 > 0 ? X2 : nullptr, K1, K2, gW, gbias, ws, rows, Nout, (hipStream_t)stream);
}
