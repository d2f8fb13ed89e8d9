// The tail of a search / train step -- clip_grad_norm_ + SGD with momentum (reference search/mr_lp_search.py:243-245,
// torch.optim.SGD at :118-119) -- over ALL parameter tensors in three launches.  torch runs it as ~30 multi_tensor_apply
// launches (a kernel-argument block holds ~100 tensor pointers) of ~16 us each plus the norm's stack / clamp kernels: 0.45 ms of
// a 7 ms sampled step.  Here the tensors are addressed through a device table of pointers and the work is cut into fixed
// chunks (host-built once: the parameter shapes never change), so the launch count does not depend on the tensor count.
//   1. multi_sqnorm_k      chunk b: sum of squares of its <= CHUNK gradient elements, in double -> partial[b]
//   2. multi_norm_final_k  one workgroup: partial[0..nb) in fixed order -> coef = min(1, max_norm / (sqrt(sum) + 1e-6))
//   3. multi_sgd_k         chunk b: g = coef * grad (+ wd * p);  buf = momentum * buf + g;  p -= lr * buf
// A tensor whose gradient pointer is null takes no part (torch skips parameters without a gradient).  Deterministic.
#include "common.hpp"
#include "../../include/mrgnas.h"

namespace mrg {

constexpr int OPT_CHUNK = 4096;          // elements per chunk (256 threads x 16)

__global__ __launch_bounds__(256) void multi_sqnorm_k(const float* const* __restrict__ grads, const int32_t* __restrict__ chunk_tensor,
                                                       const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ chunk_len,
                                                       double* __restrict__ partial) {
  __shared__ double red[256];
  const int b = blockIdx.x;
  const float* g = grads[chunk_tensor[b]];
  double acc = 0.0;
  if (g) {
    g += chunk_off[b];
    const int n = chunk_len[b];
    for (int i = threadIdx.x; i < n; i += 256) { const double v = g[i]; acc += v * v; }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[b] = red[0];
}

__global__ __launch_bounds__(256) void multi_norm_final_k(const double* __restrict__ partial, int nb, float max_norm, float* __restrict__ out2) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = max_norm / (norm + 1e-6f);                 // torch.nn.utils.clip_grad_norm_: clip_coef clamped to 1
    if (!(coef < 1.0f)) coef = 1.0f;
    out2[0] = norm;
    out2[1] = max_norm > 0.f ? coef : 1.0f;
  }
}

__global__ __launch_bounds__(256) void multi_sgd_k(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                    float* const* __restrict__ bufs, const int32_t* __restrict__ chunk_tensor,
                                                    const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ chunk_len,
                                                    const float* __restrict__ norm_coef, float lr, float momentum, float weight_decay) {
  const int b = blockIdx.x, t = chunk_tensor[b];
  const float* g = grads[t];
  if (!g) return;
  const int64_t off = chunk_off[b];
  g += off;
  float* p = params[t] + off;
  float* m = bufs[t] + off;
  const int n = chunk_len[b];
  const float coef = norm_coef[1];
  for (int i = threadIdx.x; i < n; i += 256) {
    float gi = g[i] * coef;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = momentum * m[i] + gi;
    m[i] = mi;
    p[i] = pi - lr * mi;
  }
}

}  // namespace mrg

extern "C" int mrg_optim_chunk(void) { return mrg::OPT_CHUNK; }

extern "C" int mrg_clip_sgd_step(void* const* params, const void* const* grads, void* const* bufs, const int32_t* chunk_tensor,
                                 const int64_t* chunk_off, const int32_t* chunk_len, int64_t n_chunks, double* partial, float* norm_coef,
                                 float max_norm, float lr, float momentum, float weight_decay, void* stream) {
  using namespace mrg;
  if (n_chunks <= 0) return MRG_OK;
  if (!params || !grads || !bufs || !chunk_tensor || !chunk_off || !chunk_len || !partial || !norm_coef) return MRG_E_NULLPTR;
  if (n_chunks > (int64_t)2147483647) return MRG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)n_chunks);
  hipLaunchKernelGGL(multi_sqnorm_k, grid, dim3(256), 0, st, (const float* const*)grads, chunk_tensor, chunk_off, chunk_len, partial);
  hipLaunchKernelGGL(multi_norm_final_k, dim3(1), dim3(256), 0, st, (const double*)partial, (int)n_chunks, max_norm, norm_coef);
  hipLaunchKernelGGL(multi_sgd_k, grid, dim3(256), 0, st, (float* const*)params, (const float* const*)grads, (float* const*)bufs, chunk_tensor,
                     chunk_off, chunk_len, (const float*)norm_coef, lr, momentum, weight_decay);
  return (int)hipGetLastError();
}
