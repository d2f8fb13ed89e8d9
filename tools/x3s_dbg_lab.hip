// timing-only lab of rowgemm_x3s_k with the MRG_X3S_DBG switches
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_x3s.hpp"
using namespace mrg;
int main(int argc, char** argv) {
  int64_t rows = argc > 1 ? atoll(argv[1]) : 272115; int K = argc > 2 ? atoi(argv[2]) : 200, N = argc > 3 ? atoi(argv[3]) : 200;
  float *A, *B, *C; void* Bp;
  hipMalloc(&A, rows * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, rows * N * 4);
  std::vector<float> h(rows * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hb((size_t)N * K); for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  if (argc > 4) gemm_epi_mode() = atoi(argv[4]);
  const int nt = gemm_pick_nt(N);
  hipMalloc(&Bp, x3_bsplit_bytes(N, K, nt));
  launch_bsplit(B, K, 1, N, K, nt, Bp, 0);
  GemmArgs a{}; a.A1 = A; a.K1 = K; a.B = B; a.ldb = K; a.C = C; a.ldc = N; a.N = N; a.rows = rows;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch_rowgemm_x3s<EPI_BIAS_ACT>(a, Bp, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("x3s dbg=%d epilogue mode %d rows=%lld K=%d N=%d: %.3f ms  %s\n", MRG_X3S_DBG, gemm_epi_mode(), (long long)rows, K, N, ms / 20, hipGetErrorString(hipGetLastError()));
  return 0;
}
