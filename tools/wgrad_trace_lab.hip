// Per-wave phase stamps of wgrad_x3v_k (the split-core weight gradient): where the ~1.9 us of a 16-row tile go.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMRG_WGRAD_TRACE=1 -I mr-gnas_amd/csrc -I include tools/wgrad_trace_lab.hip -o tools/labbin/wgrad_trace
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "linear.hip"
static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main(int argc, char** argv) {
  int64_t rows = argc > 1 ? atoll(argv[1]) : 558771; int K1 = argc > 2 ? atoi(argv[2]) : 200, K2 = argc > 3 ? atoi(argv[3]) : 200, Nout = argc > 4 ? atoi(argv[4]) : 200;
  float *gY, *X1, *X2 = nullptr, *gW, *gb; void* ws;
  hipMalloc(&gY, rows * Nout * 4); hipMalloc(&X1, rows * K1 * 4); if (K2) hipMalloc(&X2, rows * K2 * 4);
  hipMalloc(&gW, (size_t)Nout * (K1 + K2) * 4); hipMalloc(&gb, Nout * 4);
  {
    std::vector<float> h(rows * std::max(Nout, std::max(K1, K2))); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(gY, h.data(), rows * Nout * 4, hipMemcpyHostToDevice); hipMemcpy(X1, h.data(), rows * K1 * 4, hipMemcpyHostToDevice);
    if (K2) hipMemcpy(X2, h.data(), rows * K2 * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&ws, (size_t)mrg_linear_bwd_weight_workspace_bytes(rows, K1 + K2, Nout));
  unsigned long long* tr; hipMalloc(&tr, 8 * 64 * 6 * 8); hipMemset(tr, 0, 8 * 64 * 6 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(mrg::mrg_wgrad_trace), &tr, sizeof(tr));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) mrg_linear_bwd_weight(gY, X1, X2, gW, gb, ws, rows, K1, K2, Nout, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  int rc = mrg_linear_bwd_weight(gY, X1, X2, gW, gb, ws, rows, K1, K2, Nout, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> t(8 * 64 * 6);
  hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost);
  printf("wgrad trace rows=%lld K=%d+%d Nout=%d: %.3f ms (stamped launch, incl. the reduction pass) rc=%d %s\n", (long long)rows, K1, K2, Nout, ms, rc, hipGetErrorString(hipGetLastError()));
  // clock: shader cycles per 100 MHz tick over tiles 8..56 of wave 0
  const unsigned long long* w0 = &t[0];
  double mhz = 0;
  if (w0[56 * 6 + 5] > w0[8 * 6 + 5]) mhz = (double)(w0[56 * 6 + 0] - w0[8 * 6 + 0]) / (double)(w0[56 * 6 + 5] - w0[8 * 6 + 5]) * 100.0;
  printf("in-kernel clock %.0f MHz (s_memtime counts at a fixed 100 MHz on some parts: then phases below are in 10 ns ticks)\n", mhz);
  printf("wave | tile period | multiply (reads + MFMAs issued) | split + write | issue next loads | barrier wait     (medians over tiles 8..56, shader cycles)\n");
  for (int w = 0; w < 8; ++w) {
    std::vector<double> per, mul, spl, ld, bar;
    for (int i = 8; i < 56; ++i) {
      const unsigned long long* s = &t[(w * 64 + i) * 6];
      const unsigned long long* n = &t[(w * 64 + i + 1) * 6];
      if (!s[4] || !n[0]) continue;
      per.push_back((double)(n[0] - s[0])); mul.push_back((double)(s[1] - s[0])); spl.push_back((double)(s[2] - s[1]));
      ld.push_back((double)(s[3] - s[2])); bar.push_back((double)(s[4] - s[3]));
    }
    printf("  %d  | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f   (%zu tiles)\n", w, med(per), med(mul), med(spl), med(ld), med(bar), per.size());
  }
  return 0;
}
