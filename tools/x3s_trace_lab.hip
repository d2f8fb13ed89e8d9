// Per-wave phase stamps of rowgemm_x3s_k (MRG_X3S_DBG = 16): where a workgroup's life goes, and how the two workgroups of a CU overlap.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMRG_X3S_DBG=16 -I mr-gnas_amd/csrc -I include tools/x3s_trace_lab.hip -o tools/labbin/x3s_trace
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "gemm_x3s.hpp"
using namespace mrg;
static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double pct(std::vector<double> v, double p) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }
int main(int argc, char** argv) {
  int64_t rows = argc > 1 ? atoll(argv[1]) : 272115; int K = argc > 2 ? atoi(argv[2]) : 200, N = argc > 3 ? atoi(argv[3]) : 200;
  float *A, *B, *C; void* Bp;
  hipMalloc(&A, rows * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, rows * N * 4);
  std::vector<float> h(rows * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hb((size_t)N * K); for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const int nt = gemm_pick_nt(N);
  hipMalloc(&Bp, x3_bsplit_bytes(N, K, nt));
  launch_bsplit(B, K, 1, N, K, nt, Bp, 0);
  const int64_t nwg = (rows + 127) / 128, nw = nwg * 4;
  unsigned long long* tr; hipMalloc(&tr, nw * 24 * 8); hipMemset(tr, 0, nw * 24 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(mrg_x3s_trace), &tr, sizeof(tr));
  GemmArgs a{}; a.A1 = A; a.K1 = K; a.B = B; a.ldb = K; a.C = C; a.ldc = N; a.N = N; a.rows = rows;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> t(nw * 24);
  hipMemcpy(t.data(), tr, nw * 24 * 8, hipMemcpyDeviceToHost);
  printf("x3s trace rows=%lld K=%d N=%d: %.3f ms (stamped launch)  %s\n", (long long)rows, K, N, ms, hipGetErrorString(hipGetLastError()));
  const int nslab = (K + 15) / 16;
  std::vector<double> pro, loop, epi, drain, tot, clk, slab[16];
  unsigned long long rt0 = ~0ull, rt1 = 0;
  for (int64_t w = 0; w < nw; ++w) {
    const unsigned long long* s = &t[w * 24];
    if (!s[20]) continue;
    pro.push_back((double)(s[1] - s[0])); loop.push_back((double)(s[18] - s[1])); epi.push_back((double)(s[19] - s[18]));
    drain.push_back((double)(s[20] - s[19])); tot.push_back((double)(s[20] - s[0]));
    if (s[22] > s[23]) clk.push_back((double)(s[20] - s[0]) / (double)(s[22] - s[23]) * 100.0);
    for (int i = 0; i < nslab && i < 16; ++i) slab[i].push_back((double)(s[2 + i] - (i ? s[1 + i] : s[1])));
    rt0 = std::min(rt0, s[23]); rt1 = std::max(rt1, s[22]);
  }
  printf("waves stamped %zu; launch span by the 100 MHz clock %.1f us; in-kernel clock (median over waves) %.0f MHz\n", tot.size(), (rt1 - rt0) / 100.0, med(clk));
  auto row = [&](const char* n, std::vector<double>& v) { printf("  %-28s median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", n, med(v), pct(v, 0.1), pct(v, 0.9)); };
  row("prologue (to first slab)", pro); row("k-loop", loop); row("epilogue (stores issued)", epi); row("store drain", drain); row("wave lifetime", tot);
  printf("  MFMA issue time of a wave's k-loop: %d cycles (%d slabs x %d MFMAs x 32)\n", nslab * nt * 6 * 32, nslab, nt * 6);
  for (int i = 0; i < nslab && i < 16; ++i) printf("  slab %2d: median %6.0f p10 %6.0f p90 %6.0f\n", i, med(slab[i]), pct(slab[i], 0.1), pct(slab[i], 0.9));
  // per CU: how much of the launch span at least one / both of its workgroups are inside their k-loop (100 MHz clock is chip-wide)
  // k-loop start/end in real time are interpolated from the wave's own cycle stamps.
  struct Iv { double a, b; };
  std::map<unsigned long long, std::vector<Iv>> cu;
  for (int64_t w = 0; w < nw; w += 4) {          // wave 0 of each workgroup
    const unsigned long long* s = &t[w * 24];
    if (!s[20] || s[22] <= s[23]) continue;
    const double f = (double)(s[22] - s[23]) / (double)(s[20] - s[0]);
    const double a0 = (double)(s[23] - rt0) + (double)(s[1] - s[0]) * f, b0 = (double)(s[23] - rt0) + (double)(s[18] - s[0]) * f;
    const unsigned long long hw = s[21] & 0xffffffffull, xcc = s[21] >> 32;
    const unsigned long long key = (xcc & 15) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 8 | ((hw >> 8) & 15);
    cu[key].push_back({a0, b0});
  }
  std::vector<double> any, both, none;
  for (auto& kv : cu) {
    std::vector<std::pair<double, int>> ev;
    for (auto& iv : kv.second) { ev.push_back({iv.a, 1}); ev.push_back({iv.b, -1}); }
    std::sort(ev.begin(), ev.end());
    double last = 0, t1 = 0, t2 = 0; int d = 0;
    for (auto& e : ev) { if (d >= 1) t1 += e.first - last; if (d >= 2) t2 += e.first - last; d += e.second; last = e.first; }
    const double span = (double)(rt1 - rt0);
    any.push_back(t1 / span); both.push_back(t2 / span);
  }
  {   // one CU's workgroups in time order (us since the first wave's start): start | k-loop from..to | stores issued | end
    int shown = 0;
    for (auto& kv : cu) {
      if (shown++ != 37) continue;
      struct Row { double t0, a, b, c, d; };
      std::vector<Row> rows;
      for (int64_t w = 0; w < nw; w += 4) {
        const unsigned long long* q = &t[w * 24];
        if (!q[20] || q[22] <= q[23]) continue;
        const unsigned long long hw = q[21] & 0xffffffffull, xcc = q[21] >> 32;
        const unsigned long long key = (xcc & 15) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 8 | ((hw >> 8) & 15);
        if (key != kv.first) continue;
        const double f = (double)(q[22] - q[23]) / (double)(q[20] - q[0]) / 100.0, b0 = (double)(q[23] - rt0) / 100.0;
        rows.push_back({b0, b0 + (q[1] - q[0]) * f, b0 + (q[18] - q[0]) * f, b0 + (q[19] - q[0]) * f, b0 + (q[20] - q[0]) * f});
      }
      std::sort(rows.begin(), rows.end(), [](const Row& x, const Row& y) { return x.t0 < y.t0; });
      printf("timeline of one CU (key %llx), %zu workgroups:\n", (unsigned long long)kv.first, rows.size());
      for (auto& r : rows) printf("   start %7.1f | k-loop %7.1f .. %7.1f | stores issued %7.1f | end %7.1f\n", r.t0, r.a, r.b, r.c, r.d);
    }
    // chip-wide: workgroups inside their k-loop / inside their epilogue, sampled every 10 us
    const double span_us = (double)(rt1 - rt0) / 100.0;
    printf("chip-wide occupancy (workgroups in k-loop | in epilogue) every 10 us:");
    for (double x = 5; x < span_us; x += 10) {
      int nl = 0, ne = 0;
      for (int64_t w = 0; w < nw; w += 4) {
        const unsigned long long* q = &t[w * 24];
        if (!q[20] || q[22] <= q[23]) continue;
        const double f = (double)(q[22] - q[23]) / (double)(q[20] - q[0]) / 100.0, b0 = (double)(q[23] - rt0) / 100.0;
        const double a = b0 + (q[1] - q[0]) * f, b = b0 + (q[18] - q[0]) * f, d = b0 + (q[20] - q[0]) * f;
        if (x >= a && x < b) ++nl; else if (x >= b && x < d) ++ne;
      }
      printf(" %d|%d", nl, ne);
    }
    printf("\n");
  }
  printf("CUs seen %zu; share of the launch span with >= 1 workgroup of the CU inside its k-loop: median %.2f, with 2: %.2f\n", cu.size(), med(any), med(both));
  return 0;
}
