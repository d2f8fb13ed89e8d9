// f4 / f3: the integer data-preparation steps either side of the hot path, on the device.
//   negative_sampling                 reference utils/utils_rgcn.py:191-204
//   node relabelling (np.unique(..., return_inverse=True))   reference utils/utils_rgcn.py:97-101
//   multi-hot (label-smoothed) targets   reference utils/process_data.py:4-31 + utils/data_set.py:15-33
//   filtered ranking                  reference train/mr_lp_train.py:290-299
// Random draws are INPUTS (the host side draws them with torch's device generator, the tests replay numpy's draws), so
// every function here is deterministic and bit-exact against the reference for the same draws.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include "common.hpp"

namespace mrg {

static inline int sblocks(int64_t n, int per = 256) {
  int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > 1048560 ? 1048560 : b));
}

// samples[0:B] = pos (label 1); samples[B + j] = pos[j % B] with subject (choices[j] > 0.5) or object replaced by values[j]
__global__ void negative_sampling_k(const int64_t* __restrict__ pos, int64_t B, int rate, const int64_t* __restrict__ values,
                                    const double* __restrict__ choices, int64_t* __restrict__ samples, float* __restrict__ labels) {
  const int64_t total = B * (rate + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i < B ? i : (i - B) % B;
    int64_t s = pos[3 * p], r = pos[3 * p + 1], o = pos[3 * p + 2];
    if (i >= B) {
      const int64_t j = i - B;
      if (choices[j] > 0.5) s = values[j]; else o = values[j];
    }
    samples[3 * i] = s; samples[3 * i + 1] = r; samples[3 * i + 2] = o;
    labels[i] = i < B ? 1.0f : 0.0f;
  }
}

__global__ void mark_nodes_k(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, int32_t* __restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { flags[a[i]] = 1; flags[b[i]] = 1; }
}
__global__ void compact_nodes_k(const int32_t* __restrict__ flags, const int32_t* __restrict__ rank, int64_t num_nodes, int64_t* __restrict__ uniq,
                                int32_t* __restrict__ count) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < num_nodes; v += (int64_t)gridDim.x * blockDim.x) {
    if (flags[v]) uniq[rank[v]] = v;
    if (v == num_nodes - 1) *count = rank[v] + flags[v];
  }
}
__global__ void remap_nodes_k(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, const int32_t* __restrict__ rank,
                              int64_t* __restrict__ na, int64_t* __restrict__ nb) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { na[i] = rank[a[i]]; nb[i] = rank[b[i]]; }
}

// one workgroup per query row: out[b, :] = v_zero, out[b, objs of key q[b]] = v_one
__global__ __launch_bounds__(256) void multi_hot_k(const int64_t* __restrict__ keys, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ objs,
                                                   const int64_t* __restrict__ q, int64_t U, int64_t num_ent, float v_zero, float v_one,
                                                   float* __restrict__ out) {
  const int64_t b = blockIdx.x;
  float* row = out + b * num_ent;
  for (int64_t c = threadIdx.x; c < num_ent; c += blockDim.x) row[c] = v_zero;
  __shared__ int lo_s, hi_s;
  if (threadIdx.x == 0) {
    const int64_t key = q[b];
    int64_t lo = 0, hi = U;                       // first index with keys[i] >= key
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < U && keys[lo] == key) { lo_s = rowptr[lo]; hi_s = rowptr[lo + 1]; } else { lo_s = 0; hi_s = 0; }
  }
  __syncthreads();                                 // also orders the fill above before the ones below
  for (int j = lo_s + threadIdx.x; j < hi_s; j += blockDim.x) row[objs[j]] = v_one;
}

// filtered rank of obj[b] in row b (train/mr_lp_train.py:290-299): entries with a non-zero label are pushed to -1e7, the
// target keeps its score; rank = 1 + #(greater) + #(equal, at a lower index) -- a stable descending sort's position
__global__ __launch_bounds__(256) void rank_filtered_k(const float* __restrict__ pred, const float* __restrict__ labels, const int64_t* __restrict__ obj,
                                                       int64_t N, int64_t* __restrict__ ranks) {
  const int64_t b = blockIdx.x;
  const float* p = pred + b * N;
  const float* l = labels + b * N;
  const int64_t o = obj[b];
  const float t = p[o];
  int cnt = 0;
  for (int64_t j = threadIdx.x; j < N; j += blockDim.x) {
    if (j == o) continue;
    const float m = ((unsigned char)l[j]) ? -10000000.0f : p[j];       // labels.byte(): float -> uint8, non-zero = filtered
    cnt += (m > t) || (m == t && j < o);
  }
  __shared__ int red[256];
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ranks[b] = 1 + red[0];
}

}  // namespace mrg

using namespace mrg;

extern "C" int mrg_negative_sampling(const int64_t* pos, int64_t B, int rate, const int64_t* values, const double* choices, int64_t* samples,
                                     float* labels, void* stream) {
  if (B < 0 || rate < 0) return MRG_E_SHAPE;
  if (B == 0) return MRG_OK;
  if (!pos || !samples || !labels || (rate > 0 && (!values || !choices))) return MRG_E_NULLPTR;
  hipLaunchKernelGGL(negative_sampling_k, dim3(sblocks(B * (rate + 1))), dim3(256), 0, (hipStream_t)stream, pos, B, rate, values, choices, samples, labels);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

static size_t relabel_scan_temp(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(n > 0 ? n : 1), rocprim::plus<int32_t>());
  return (b + 255) & ~(size_t)255;
}

extern "C" int64_t mrg_relabel_workspace_bytes(int64_t num_nodes) {
  if (num_nodes < 0) return 0;
  return (int64_t)(relabel_scan_temp(num_nodes) + 2 * (((size_t)(num_nodes + 1) * 4 + 255) & ~(size_t)255));
}

extern "C" int mrg_relabel_nodes(const int64_t* src, const int64_t* dst, int64_t n, int64_t num_nodes, int64_t* uniq, int64_t* new_src,
                                 int64_t* new_dst, int32_t* count, void* ws, int64_t ws_bytes, void* stream) {
  if (n < 0 || num_nodes < 0) return MRG_E_SHAPE;
  if (!count) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(count, 0, 4, st);
  if (e != hipSuccess) return (int)e;
  if (n == 0 || num_nodes == 0) return MRG_OK;
  if (!src || !dst || !uniq || !new_src || !new_dst) return MRG_E_NULLPTR;
  if (!ws || ws_bytes < mrg_relabel_workspace_bytes(num_nodes)) return MRG_E_WORKSPACE;
  const size_t tmp_bytes = relabel_scan_temp(num_nodes), arr = ((size_t)(num_nodes + 1) * 4 + 255) & ~(size_t)255;
  char* base = (char*)ws;
  int32_t* flags = (int32_t*)(base + tmp_bytes);
  int32_t* rank = (int32_t*)(base + tmp_bytes + arr);
  e = hipMemsetAsync(flags, 0, (size_t)num_nodes * 4, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(mark_nodes_k, dim3(sblocks(n)), dim3(256), 0, st, src, dst, n, flags);
  size_t tb = tmp_bytes;
  e = rocprim::exclusive_scan((void*)base, tb, flags, rank, 0, (size_t)num_nodes, rocprim::plus<int32_t>(), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(compact_nodes_k, dim3(sblocks(num_nodes)), dim3(256), 0, st, flags, rank, num_nodes, uniq, count);
  hipLaunchKernelGGL(remap_nodes_k, dim3(sblocks(n)), dim3(256), 0, st, src, dst, n, rank, new_src, new_dst);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_multi_hot_labels(const int64_t* keys, const int32_t* rowptr, const int32_t* objs, const int64_t* query, int64_t B, int64_t U,
                                    int64_t num_ent, float v_zero, float v_one, float* out, void* stream) {
  if (B < 0 || U < 0 || num_ent <= 0) return MRG_E_SHAPE;
  if (B == 0) return MRG_OK;
  if (!query || !out || (U > 0 && (!keys || !rowptr || !objs))) return MRG_E_NULLPTR;
  hipLaunchKernelGGL(multi_hot_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, keys, rowptr, objs, query, U, num_ent, v_zero, v_one, out);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_rank_filtered(const float* pred, const float* labels, const int64_t* obj, int64_t B, int64_t N, int64_t* ranks, void* stream) {
  if (B < 0 || N <= 0) return MRG_E_SHAPE;
  if (B == 0) return MRG_OK;
  if (!pred || !labels || !obj || !ranks) return MRG_E_NULLPTR;
  hipLaunchKernelGGL(rank_filtered_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, pred, labels, obj, N, ranks);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
