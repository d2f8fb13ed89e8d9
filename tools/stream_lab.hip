// Lab: what a plain 2-reads-1-write float4 streaming kernel reaches on this box as a function of grid size, float4 in flight per
// lane and cache policy of loads / stores (compose_fwd_k 5.1 TB/s vs the guide's 6.29 TB/s copy).
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_lab.hip -o tools/labbin/stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4* p) { v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(float4 o, float4* p) { v4f v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }

template <int U, bool NTL, bool NTS, int MODE>   // MODE 0: out = a - b (2R1W); 1: copy (1R1W); 2: out = a+b+c+d (4R1W)
__global__ __launch_bounds__(256) void stream_k(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                const float4* __restrict__ d, float4* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 x[U], y[U], z[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride;
      x[u] = NTL ? ntload(a + j) : a[j];
      if (MODE != 1) y[u] = NTL ? ntload(b + j) : b[j];
      if (MODE == 2) { z[u] = NTL ? ntload(c + j) : c[j]; w[u] = NTL ? ntload(d + j) : d[j]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 o = x[u];
      if (MODE != 1) { o.x -= y[u].x; o.y -= y[u].y; o.z -= y[u].z; o.w -= y[u].w; }
      if (MODE == 2) { o.x += z[u].x + w[u].x; o.y += z[u].y + w[u].y; o.z += z[u].z + w[u].z; o.w += z[u].w + w[u].w; }
      if (NTS) ntstore(o, out + i + u * stride); else out[i + u * stride] = o;
    }
  }
  for (; i < n; i += stride) {
    float4 o = a[i];
    if (MODE != 1) { float4 y = b[i]; o.x -= y.x; o.y -= y.y; o.z -= y.z; o.w -= y.w; }
    if (MODE == 2) { float4 z = c[i], w = d[i]; o.x += z.x + w.x; o.y += z.y + w.y; o.z += z.z + w.z; o.w += z.w + w.w; }
    out[i] = o;
  }
}

// contiguous-chunk variant: every block owns one contiguous range (DRAM page locality), U float4 in flight per lane
template <int U, bool NTS>
__global__ __launch_bounds__(256) void chunk_k(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int64_t n) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  int64_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    float4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = a[i + u * 256]; y[u] = b[i + u * 256]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 o = make_float4(x[u].x - y[u].x, x[u].y - y[u].y, x[u].z - y[u].z, x[u].w - y[u].w);
      if (NTS) ntstore(o, out + i + u * 256); else out[i + u * 256] = o;
    }
  }
  for (; i < hi; i += 256) { float4 x = a[i], y = b[i]; out[i] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 558771; const int D = argc > 2 ? atoi(argv[2]) : 200;
  const int64_t n = rows * D / 4;
  float4 *a, *b, *c, *d, *o;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&d, n * 16); hipMalloc(&o, n * 16);
  hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16); hipMemset(d, 0, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int nbuf, auto launch) {
    launch(); hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; best = ms < best ? ms : best; sum += ms;
    }
    printf("%-58s best %7.1f us  %6.0f GB/s | mean %7.1f us %6.0f GB/s   %s\n", name, best * 1e3, nbuf * n * 16 / best * 1e-6, sum / 5 * 1e3,
           nbuf * n * 16 / (sum / 5) * 1e-6, hipGetErrorString(hipGetLastError()));
  };
  char nm[128];
#define SWEEP(U, NTL, NTS, MODE, NBUF)                                                                              \
  for (int bpc : {2, 4, 8, 16, 32}) {                                                                                 \
    int64_t g = 256 * bpc; int64_t need = (n + 256 * U - 1) / (256 * U); if (g > need) g = need;                      \
    snprintf(nm, sizeof nm, "mode %d U=%d ntl=%d nts=%d grid=256x%d", MODE, U, NTL, NTS, bpc);                       \
    run(nm, NBUF, [&] { hipLaunchKernelGGL((stream_k<U, NTL, NTS, MODE>), dim3((unsigned)g), dim3(256), 0, 0, a, b, c, d, o, n); }); \
  }
  SWEEP(1, false, false, 0, 3) SWEEP(2, false, false, 0, 3) SWEEP(4, false, false, 0, 3) SWEEP(8, false, false, 0, 3)
  SWEEP(2, false, true, 0, 3) SWEEP(4, false, true, 0, 3) SWEEP(4, true, true, 0, 3) SWEEP(4, true, false, 0, 3)
  SWEEP(1, false, false, 1, 2) SWEEP(4, false, false, 1, 2) SWEEP(4, false, true, 1, 2) SWEEP(4, true, true, 1, 2)
  SWEEP(1, false, false, 2, 5) SWEEP(2, false, false, 2, 5) SWEEP(2, false, true, 2, 5) SWEEP(2, true, true, 2, 5)
  {  // one block per ... exact grid (no grid-stride loop): every thread U float4
    for (int U : {1, 2, 4}) {
      int64_t g = (n + 256 * U - 1) / (256 * U);
      snprintf(nm, sizeof nm, "mode 0 exact grid U=%d (%lld blocks)", U, (long long)g);
      if (U == 1) run(nm, 3, [&] { hipLaunchKernelGGL((stream_k<1, false, false, 0>), dim3((unsigned)g), dim3(256), 0, 0, a, b, c, d, o, n); });
      if (U == 2) run(nm, 3, [&] { hipLaunchKernelGGL((stream_k<2, false, false, 0>), dim3((unsigned)g), dim3(256), 0, 0, a, b, c, d, o, n); });
      if (U == 4) run(nm, 3, [&] { hipLaunchKernelGGL((stream_k<4, false, false, 0>), dim3((unsigned)g), dim3(256), 0, 0, a, b, c, d, o, n); });
    }
  }
  for (int bpc : {4, 8, 16, 64}) {
    snprintf(nm, sizeof nm, "chunked U=4 nts=0 grid=256x%d", bpc);
    run(nm, 3, [&] { hipLaunchKernelGGL((chunk_k<4, false>), dim3(256 * bpc), dim3(256), 0, 0, a, b, o, n); });
    snprintf(nm, sizeof nm, "chunked U=4 nts=1 grid=256x%d", bpc);
    run(nm, 3, [&] { hipLaunchKernelGGL((chunk_k<4, true>), dim3(256 * bpc), dim3(256), 0, 0, a, b, o, n); });
  }
  run("hipMemcpyAsync D2D (1R1W)", 2, [&] { hipMemcpyAsync(o, a, n * 16, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
