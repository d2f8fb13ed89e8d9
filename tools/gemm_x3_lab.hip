// Lab bench for the 3-way bf16 split GEMM core against the exact-f32 MFMA core: time + error vs float64.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mr-gnas_amd/csrc -I tools/lab -I include tools/gemm_x3_lab.hip -o tools/labbin/gemm_x3_lab
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "gemm_dispatch.hpp"
#include "gemm_x3p.hpp"        // tools/lab: the persistent and the two-waves-per-SIMD kernels (lab only since round 4)
using namespace mrg;

static float frand() { return (float)rand() / RAND_MAX - 0.5f; }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int64_t rows = argc > 1 ? atoll(argv[1]) : 558771;
  int K1 = argc > 2 ? atoi(argv[2]) : 200, K2 = argc > 3 ? atoi(argv[3]) : 200, N = argc > 4 ? atoi(argv[4]) : 200;
  int K = K1 + K2;
  const int reps = argc > 5 ? atoi(argv[5]) : 10;            // long runs for tools/clock_probe.sh
  float *A1, *A2, *B, *C, *C2; void* Bp;
  hipMalloc(&A1, rows * K1 * 4); hipMalloc(&A2, rows * (K2 ? K2 : 4) * 4); hipMalloc(&B, (size_t)N * K * 4);
  hipMalloc(&C, rows * N * 4); hipMalloc(&C2, rows * N * 4);
  std::vector<float> h1(rows * K1), h2(rows * (size_t)K2), hb((size_t)N * K), bias(N);
  for (auto& v : h1) v = frand() * 4; for (auto& v : h2) v = frand(); for (auto& v : hb) v = frand() * 0.2f; for (auto& v : bias) v = frand();
  hipMemcpy(A1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice);
  if (K2) hipMemcpy(A2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  float* dbias; hipMalloc(&dbias, N * 4); hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice);
  const int nt = gemm_pick_nt(N);
  hipMalloc(&Bp, x3_bsplit_bytes(N, K, nt));
  GemmArgs a{}; a.A1 = A1; a.A2 = K2 ? A2 : nullptr; a.K1 = K1; a.K2 = K2; a.B = B; a.ldb = K; a.C = C; a.ldc = N; a.N = N; a.rows = rows;
  a.bias = dbias; a.act = 0;
#if MRG_X3_DBG & 512
  // every launch of this build stamps: the buffer must exist before the first one (slots: 4 waves x workgroups of either kernel)
  const int64_t slots = 4 * (2 * ((rows + 255) / 256 + 8)) * 4;
  unsigned long long* tr = nullptr;
  if (hipMalloc(&tr, slots * 4 * 8) != hipSuccess || !tr) { printf("trace buffer allocation failed\n"); return 1; }
  if (hipMemcpyToSymbol(HIP_SYMBOL(mrg_x3_trace), &tr, sizeof(tr)) != hipSuccess) { printf("trace symbol not set\n"); return 1; }
  hipDeviceSynchronize();
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-28s %8.3f ms  %7.1f TF/s (f32-equivalent)  err=%s\n", name, ms, 2.0 * rows * K * N / ms * 1e-9, hipGetErrorString(hipGetLastError()));
  };
  timeit("x3 (split + gemm)", [&] { launch_bsplit(B, K, 1, N, K, nt, Bp, 0); launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0); });
  gemm_epi_lds() = 0;
  timeit("x3 (gemm only, acc-order stores)", [&] { launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0); });
  std::vector<float> c0(rows * N);
  hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
  hipMemset(C, 0, rows * N * 4);
  gemm_epi_lds() = 1;
  timeit("x3 (gemm only, row-order stores)", [&] { launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0); });
  {
    std::vector<float> cx(rows * N);
    hipMemcpy(cx.data(), C, cx.size() * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (size_t i = 0; i < cx.size(); ++i) bad += (cx[i] != c0[i]);
    printf("row-order vs acc-order stores: %lld of %lld outputs differ\n", (long long)bad, (long long)cx.size());
  }
  {   // accumulate epilogue (C += ...) both ways
    GemmArgs g = a; g.Cin = C2; g.ld_cin = N; g.bias = nullptr;
    hipMemset(C2, 0, rows * N * 4);
    gemm_epi_lds() = 0;
    timeit("x3 accumulate, acc-order", [&] { launch_rowgemm_x3<EPI_ACCUM>(g, Bp, 0); });
    gemm_epi_lds() = 1;
    timeit("x3 accumulate, row-order", [&] { launch_rowgemm_x3<EPI_ACCUM>(g, Bp, 0); });
    GemmArgs q = a; q.S = A1; q.ld_s = K1; q.aux = C2; q.scale = 1.f / 3; 
    if (K1 == N) {
      gemm_epi_lds() = 0;
      timeit("x3 gate (+aux), acc-order", [&] { launch_rowgemm_x3<EPI_GATE>(q, Bp, 0); });
      gemm_epi_lds() = 1;
      timeit("x3 gate (+aux), row-order", [&] { launch_rowgemm_x3<EPI_GATE>(q, Bp, 0); });
    }
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);      // C as the error check below expects it
  }
  {
    gemm_epi_lds() = 0;
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);
    std::vector<float> cref(rows * N), cs(rows * N);
    hipMemcpy(cref.data(), C, cref.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(C, 0, rows * N * 4);
    timeit("x3s LDS-B, 2 WG/CU (gemm only)", [&] { launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0); });
    hipMemcpy(cs.data(), C, cs.size() * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0; double worst = 0;
    for (size_t i = 0; i < cs.size(); ++i) { if (cs[i] != cref[i]) { ++bad; worst = std::max(worst, (double)fabs(cs[i] - cref[i])); } }
    printf("x3s vs x3: %lld of %lld outputs differ (max |diff| %.3e)\n", (long long)bad, (long long)cs.size(), worst);
    GemmArgs g = a; g.Cin = C2; g.ld_cin = N; g.bias = nullptr;
    timeit("x3s accumulate", [&] { launch_rowgemm_x3s<EPI_ACCUM>(g, Bp, 0); });
    if (K1 == N) {
      GemmArgs q = a; q.S = A1; q.ld_s = K1; q.aux = C2; q.scale = 1.f / 3;
      timeit("x3s gate (+aux)", [&] { launch_rowgemm_x3s<EPI_GATE>(q, Bp, 0); });
    }
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);
  }
  if (getenv("MRG_LAB_RING2")) gemm_wide8() = 2;   // seven-tile outputs on the ring-of-two kernel as well
  if (x3s8_eligible<EPI_BIAS_ACT>(a)) {   // round 4: eight column tiles as ONE block (gemm_x3s8.hpp) against two four-tile blocks
    gemm_epi_mode() = 0;
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);
    std::vector<float> cref(rows * N), cs(rows * N);
    hipMemcpy(cref.data(), C, cref.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(C, 0, rows * N * 4);
    timeit("x3s (N = 256: 2 x 4 tiles) (gemm only)", [&] { launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0); });
    hipMemset(C, 0, rows * N * 4);
    timeit("x3s8, ring of two, ONE block (gemm only)", [&] { launch_rowgemm_x3s8<EPI_BIAS_ACT>(a, Bp, 0); });
    hipMemcpy(cs.data(), C, cs.size() * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (size_t i = 0; i < cs.size(); ++i) bad += (cs[i] != cref[i]);
    printf("x3s8 vs x3: %lld of %lld outputs differ\n", (long long)bad, (long long)cs.size());
    GemmArgs g = a; g.Cin = C2; g.ld_cin = N; g.bias = nullptr;
    timeit("x3s accumulate, 2 x 4 tiles", [&] { launch_rowgemm_x3s<EPI_ACCUM>(g, Bp, 0); });
    timeit("x3s8 accumulate", [&] { launch_rowgemm_x3s8<EPI_ACCUM>(g, Bp, 0); });
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);
  }
  {   // round 4: transposed accumulators (16-byte epilogue accesses) against accumulator-order stores, every epilogue, bit for bit
    std::vector<float> r0(rows * N), r1(rows * N), x0(rows * N), x1(rows * N);
    float* AUX; hipMalloc(&AUX, rows * N * 4);
    auto cmp = [&](const char* what, std::vector<float>& u, std::vector<float>& v) {
      int64_t bad = 0; for (size_t i = 0; i < u.size(); ++i) bad += (u[i] != v[i]);
      printf("   %s: %lld of %lld outputs differ between the two accumulator layouts\n", what, (long long)bad, (long long)u.size());
    };
    for (int e = 0; e < 4; ++e) {
      if ((e == 2 || e == 3) && K1 != N) continue;
      GemmArgs g = a;
      const char* name = e == 0 ? "bias+act" : e == 1 ? "accumulate" : e == 2 ? "gate (+aux)" : "gate only";
      if (e == 1) { g.Cin = C2; g.ld_cin = N; g.bias = nullptr; }
      if (e == 2 || e == 3) { g.S = A1; g.ld_s = K1; g.aux = AUX; g.scale = 1.f / 3; if (e == 3) g.C = nullptr; }
      for (int mode = 0; mode <= 2; mode += 2) {
        gemm_epi_mode() = mode;
        hipMemset(C, 0, rows * N * 4); hipMemset(AUX, 0, rows * N * 4);
        char label[96]; snprintf(label, sizeof label, "x3s %s, %s", name, mode ? "transposed acc" : "acc-order");
        timeit(label, [&] {
          if (e == 0) launch_rowgemm_x3s<EPI_BIAS_ACT>(g, Bp, 0); else if (e == 1) launch_rowgemm_x3s<EPI_ACCUM>(g, Bp, 0);
          else launch_rowgemm_x3s<EPI_GATE>(g, Bp, 0);
        });
        hipMemcpy((mode ? r1 : r0).data(), C, r0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy((mode ? x1 : x0).data(), AUX, x0.size() * 4, hipMemcpyDeviceToHost);
      }
      cmp(name, r0, r1);
      if (e >= 2) cmp("its gate", x0, x1);
    }
    gemm_epi_mode() = 2;
    hipFree(AUX);
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0);
  }
  if (x3p_eligible(a)) {
    hipMemset(C, 0, rows * N * 4);
    timeit("x3 persistent (gemm only)", [&] { launch_rowgemm_x3p<EPI_BIAS_ACT>(a, Bp, 0); });
  }
  if (x3w_eligible(a)) {
    hipMemset(C, 0, rows * N * 4);
    timeit("x3 two waves/SIMD (gemm only)", [&] { launch_rowgemm_x3w<EPI_BIAS_ACT>(a, Bp, 0); });
  }
#if MRG_X3_DBG & 512
  {  // per-wave phase timestamps of ONE launch of each kernel (100 MHz clock: 10 ns units)
    std::vector<unsigned long long> h(slots * 4);
    auto report = [&](const char* name) {
      hipDeviceSynchronize();
      hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull, t3 = 0; int64_t n = 0;
      for (int64_t i = 0; i < slots; ++i) if (h[4 * i] && h[4 * i + 3]) { t0 = std::min(t0, h[4 * i]); t3 = std::max(t3, h[4 * i + 3]); ++n; }
      double sp = 0, sl = 0, se = 0; std::vector<double> starts;
      for (int64_t i = 0; i < slots; ++i) if (h[4 * i] && h[4 * i + 3]) {
        sp += (h[4 * i + 1] - h[4 * i]) * 0.01; sl += (h[4 * i + 2] - h[4 * i + 1]) * 0.01; se += (h[4 * i + 3] - h[4 * i + 2]) * 0.01;
        starts.push_back((h[4 * i] - t0) * 0.01);
      }
      std::sort(starts.begin(), starts.end());
      printf("%s: %lld waves, span %.1f us; mean per wave: prologue %.2f us, k-loop %.2f us, epilogue+drain %.2f us\n", name, (long long)n,
             (t3 - t0) * 0.01, sp / n, sl / n, se / n);
      printf("   wave start times (us) at deciles:");
      for (int d = 0; d <= 10; ++d) printf(" %.1f", starts[std::min<size_t>(starts.size() - 1, starts.size() * d / 10)]);
      printf("\n");
    };
    hipMemset(tr, 0, slots * 4 * 8);
    launch_rowgemm_x3<EPI_BIAS_ACT>(a, Bp, 0); report("one wave/SIMD");
    if (x3w_eligible(a)) { hipMemset(tr, 0, slots * 4 * 8); launch_rowgemm_x3w<EPI_BIAS_ACT>(a, Bp, 0); report("two waves/SIMD"); }
  }
#endif
  GemmArgs b = a; b.C = C2;
  timeit("f32 mfma", [&] { launch_rowgemm<EPI_BIAS_ACT>(b, 0); });
  // error vs float64 on a sample of rows
  std::vector<float> c1(rows * N), c2(rows * N);
  hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
  double e1m = 0, e2m = 0, e1s = 0, e2s = 0, ref_max = 0; int64_t cnt = 0;
  for (int64_t r = 0; r < rows; r += (r < 300 || r > rows - 300) ? 1 : 997) {
    for (int n = 0; n < N; ++n) {
      double s = bias[n];
      for (int k = 0; k < K1; ++k) s += (double)h1[r * K1 + k] * hb[(size_t)n * K + k];
      for (int k = 0; k < K2; ++k) s += (double)h2[r * K2 + k] * hb[(size_t)n * K + K1 + k];
      double d1 = fabs(c1[r * N + n] - s), d2 = fabs(c2[r * N + n] - s);
      e1m = fmax(e1m, d1); e2m = fmax(e2m, d2); e1s += d1 * d1; e2s += d2 * d2; ref_max = fmax(ref_max, fabs(s)); ++cnt;
    }
  }
  printf("vs float64 over %lld outputs (|ref| max %.3f): x3 max %.3e rms %.3e | f32 mfma max %.3e rms %.3e\n", (long long)cnt, ref_max,
         e1m, sqrt(e1s / cnt), e2m, sqrt(e2s / cnt));
  return 0;
}
