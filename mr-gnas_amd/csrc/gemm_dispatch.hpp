// Dispatch between the two matrix cores of the tall-skinny GEMM (gemm.hpp: exact f32; gemm_x3.hpp / gemm_x3s.hpp: split bf16).
// Rounds 2-3 also carried a persistent transposed-accumulator kernel (mode 3) and a two-waves-per-SIMD kernel (mode 4) as tested
// comparison points; both lost to gemm_x3s.hpp and live in tools/lab/ since round 4 (tools/gemm_x3_lab.hip still times them).
#pragma once
#include "gemm_x3.hpp"
#include "gemm_x3s.hpp"
#include "gemm_x3s8.hpp"

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

// the split-core row GEMM of the current mode for launches that prepared their own weight split (grouped launches, fused aggregators)
template <int EPI>
inline int launch_rowgemm_x3_mode(GemmArgs a, const void* Bp, hipStream_t st) {
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
  const size_t split = x3_bsplit_bytes(N, K, gemm_pick_nt(N));
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
    launch_bsplit(a.B, b_sn, b_sk, a.N, K, gemm_pick_nt(a.N), ws, st);
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
