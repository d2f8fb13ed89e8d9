// Dispatch between the two matrix cores of the tall-skinny GEMM (gemm.hpp: exact f32; gemm_x3.hpp / gemm_x3s.hpp: split bf16).
// Rounds 2-3 also carried a persistent transposed-accumulator kernel (mode 3) and a two-waves-per-SIMD kernel (mode 4) as tested
// comparison points; both lost to gemm_x3s.hpp and live in tools/lab/ since round 4 (tools/gemm_x3_lab.hip still times them).
#pragma once
#include "gemm_x3.hpp"
#include "gemm_x3s.hpp"
#include "gemm_x3s8.hpp"
#include "gemm_x3q.hpp"

namespace mrg {

// ---- dispatch between the two cores ---------------------------------------------------------------
// mode 0 (default): split-bf16 core whenever the operands qualify and a workspace was given;
// mode 1: exact-f32 core only (v_mfma_f32_32x32x2_f32) -- the comparison point of the tests and of bench.py.
// mode 2: the same arithmetic on the wave-autonomous one-wave-per-SIMD kernel of gemm_x3.hpp (rounds 1-2's default).
// Since round 3 the default split-core kernel is gemm_x3s.hpp (weight slabs shared through LDS, 128-row workgroups, two per
// CU): bit-identical results; alone 0.191 vs 0.22-0.23 ms at rows 272 115, K = N = 200 and equal at 558 771 rows, but inside the
// supernet step every row-GEMM entry point gains 9-22 % (input gradients 10.3 -> 8.4 ms, fused a_max / a_mean 4.1 -> 3.2,
// dense-filter forward 6.5 -> 5.9; 68.2 -> 65.5 ms / step, profiles/r3_rowgemm_lds_weight.txt).
inline int& gemm_mode() { static int m = 0; return m; }

// Which launches take the 16 x 16 x 32 kernel of gemm_x3q.hpp (round 5): a pure function of the epilogue kind and of the product's
// shape, so that the code that prepares the weight split and the code that launches the GEMM agree without talking.  Its column
// block is 14 tiles of 16 (129..224 columns: narrower products keep the four-tile blocks of rowgemm_x3s_k, wider ones its eight-tile
// sibling); the fused aggregators' epilogues (EPI_SEGMAX / EPI_SEGSUM: run structure tied to the 32-row strip) stay on rowgemm_x3s_k.
// Measured (profiles/r5_rowgemm_q.txt, rows 558 771, interleaved rounds in one process): at K = 400 (the dense filters' forward on
// two operands, the paired input gradient) 0.552 vs 0.584 ms (-5 %), gate-only forward -7..-12 %; at K = 200 equal (0.986-1.017): the
// reduction dimension has to be long enough for the smaller tile's doubled weight traffic (L2 -> LDS 2.6 GB instead of 1.2 GB per
// launch) to be paid back, so gemm_q() == 1 (default) takes only K > X3Q_MIN_K; 2 (lab) takes every K > 48.
constexpr int X3Q_MIN_K = 224;
inline bool x3q_shape(int epi, int N, int K) {
  return gemm_mode() == 0 && gemm_q() != 0 && epi != EPI_SEGMAX && epi != EPI_SEGSUM && N > 128 && N <= X3Q_NT * 16
         && K > (gemm_q() == 2 ? 48 : X3Q_MIN_K);
}

// FEW ROWS (round 5: the sampled search step of graph_batch_size 300, a rank's node chunk).  A 128-row workgroup of rowgemm_x3s_k
// walks all seven column tiles of its rows: 546 matrix instructions per wave one behind the other, ~20 us however few rows there
// are, on an otherwise idle chip.  Up to X3N_MAX_ROWS rows the product runs on the wave-autonomous kernel of gemm_x3.hpp with column
// blocks of TWO tiles (grid.y = 4 for 200 columns): 3.5 x more waves, each with a 3.5 x shorter chain.  Same operands, same order of
// the k-sum per output element: bit-identical with rowgemm_x3s_k (tests/test_ops_gpu.py::test_few_rows_row_gemm_is_bit_exact), so a
// grouped launch and its per-range launches may fall on different sides of the bound.  Not for the shapes of rowgemm_x3q_k (another
// summation order: those keep one kernel for every row count).  mrg_gemm_set_small(0) switches it off.
constexpr int64_t X3N_MAX_ROWS = 16384;     // measured crossover (linear 200 x 200, one MI355X): 13 vs 24 us at 4 096 rows, 19 vs 27 at 16 384, 31 vs 30 at 32 768
constexpr int X3N_NT = 2;
inline int64_t& gemm_small() { static int64_t m = X3N_MAX_ROWS; return m; }     // the row bound; 0 = off (mrg_gemm_set_small)
inline bool x3n_shape(int epi, int64_t rows, int N, int K) {
  return gemm_mode() == 0 && rows > 0 && rows <= gemm_small() && gemm_pick_nt(N) > X3N_NT && N <= 224 && !x3q_shape(epi, N, K);
}

// bytes of ONE pre-split weight in whichever layout a launch of this shape may use
inline size_t bsplit_bytes_any(int N, int K) {
  const size_t a = x3_bsplit_bytes(N, K, gemm_pick_nt(N));
  const size_t q = (N > 128 && N <= X3Q_NT * 16) ? x3q_bsplit_bytes(K) : 0;
  const size_t n = (gemm_pick_nt(N) > X3N_NT && N <= 224) ? x3_bsplit_bytes(N, K, X3N_NT) : 0;
  const size_t m = a > q ? a : q;
  return m > n ? m : n;
}

// the weight split of a launch of `rows` rows with epilogue `epi`: B(n, k) = B[n * sn + k * sk]
inline void launch_bsplit_any(int epi, int64_t rows, const float* B, int64_t sn, int64_t sk, int N, int K, void* Bp, hipStream_t st) {
  if (x3q_shape(epi, N, K)) {
    const float* Bs[1] = {B};
    void* outs[1] = {Bp};
    launch_bsplitq3(Bs, sn, sk, N, K, outs, 1, st);
  } else {
    launch_bsplit(B, sn, sk, N, K, x3n_shape(epi, rows, N, K) ? X3N_NT : gemm_pick_nt(N), Bp, st);
  }
}
inline void launch_bsplit3_any(int epi, int64_t rows, const float* const* B, int64_t sn, int64_t sk, int N, int K, void* const* out, hipStream_t st,
                               const float* const* B2 = nullptr, int ksplit = 0) {
  if (x3q_shape(epi, N, K)) launch_bsplitq3(B, sn, sk, N, K, out, 3, st, B2, ksplit);
  else launch_bsplit3(B, sn, sk, N, K, x3n_shape(epi, rows, N, K) ? X3N_NT : gemm_pick_nt(N), out, st, B2, ksplit);
}

// the split-core row GEMM of the current mode for launches that prepared their own weight split (grouped launches, fused aggregators)
// with launch_bsplit_any / launch_bsplit3_any
template <int EPI>
inline int launch_rowgemm_x3_mode(GemmArgs a, const void* Bp, hipStream_t st) {
  if (x3q_shape(EPI, a.N, a.K1 + a.K2)) {
    if (!x3q_eligible(a)) return MRG_E_SHAPE;              // the split was prepared in this kernel's layout: no other kernel can read it
    return launch_rowgemm_x3q<EPI>(a, Bp, st);
  }
  if (x3n_shape(EPI, a.rows, a.N, a.K1 + a.K2)) {
    if (!x3_eligible(a)) return MRG_E_SHAPE;               // the split was prepared with two-tile column blocks
    return launch_rowgemm_x3<EPI>(a, Bp, st, X3N_NT);
  }
  if (gemm_mode() != 2 && x3s_eligible(a)) {
    // eight column tiles (D = 256) as ONE block where the epilogue fits (gemm_x3s8.hpp): the activation operand is read once
    if constexpr (EPI != EPI_GATE) {
      if (x3s8_eligible<EPI>(a)) return launch_rowgemm_x3s8<EPI>(a, Bp, st);
    }
    return launch_rowgemm_x3s<EPI>(a, Bp, st);
  }
  return launch_rowgemm_x3<EPI>(a, Bp, st);
}

inline size_t gemm_workspace_bytes(int K, int N) {
  const size_t split = bsplit_bytes_any(N, K);
  const size_t transp = (size_t)K * N * sizeof(float);
  return split > transp ? split : transp;
}

// B(n, k) = a.B[n * b_sn + k * b_sk] (a.ldb is ignored).  ws: gemm_workspace_bytes(K, N) bytes, may be NULL
// when b_sk == 1 (then only the exact-f32 core is available).
template <int EPI>
inline int launch_gemm(GemmArgs a, int64_t b_sn, int64_t b_sk, void* ws, hipStream_t st) {
  if (a.rows <= 0) return MRG_OK;
  if (!a.A2 || a.K2 == 0) { a.A2 = a.A1; a.K2 = 0; }
  const int K = a.K1 + a.K2;
  if (ws && gemm_mode() != 1 && x3_eligible(a)) {
    launch_bsplit_any(EPI, a.rows, a.B, b_sn, b_sk, a.N, K, ws, st);
    return launch_rowgemm_x3_mode<EPI>(a, ws, st);
  }
  if (b_sk != 1) {                                   // present B^T row-major to the f32 core
    if (!ws) return MRG_E_WORKSPACE;
    launch_transpose(a.B, (float*)ws, (int)(K), a.N, (int)b_sk, st);
    a.B = (const float*)ws;
    a.ldb = K;
  } else {
    a.ldb = (int)b_sn;
  }
  return launch_rowgemm<EPI>(a, st);
}

}  // namespace mrg
