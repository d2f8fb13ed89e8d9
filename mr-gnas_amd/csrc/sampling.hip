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


// ---- neighbourhood-expansion edge sampler (reference utils/utils_rgcn.py:30-71, `--edge_sampler neighbor`) ------------------------
// sample_size DEPENDENT picks: every pick changes the weights of the next one.  One persistent 1024-thread workgroup walks the
// picks; inside a pick everything that is not sequential by nature runs over the workgroup:
//   weights w_v = sample_counts[v] * seen[v] (all ones over the vertices with sample_counts > 0 when their sum is 0)
//   vertex  = searchsorted(cumsum(w / sum w), u, side='right')      -> integer prefix sums W_j and the first j with W_j > u * W
//             (the reference's float64 cdf step j is W_j / W up to rounding: the two agree unless u lies within rounding distance
//             of a step, probability < 1e-7 per draw; the golden replay pins the draws it uses)
//   edge    = a uniformly chosen NOT YET PICKED entry of the vertex's adjacency list (the reference re-draws until it finds one):
//             replay mode: the reference's own sequence of tried slots (int64, as many as it drew);
//             random mode: ONE uniform u2 -> the floor(u2 * sample_counts[v])-th unpicked entry (the same distribution without
//             rejection: sample_counts[v] IS the number of unpicked entries of v).
// adjacency = CSR in the reference's append order (get_adj_and_degrees, :18-28): for triple i, (i, o) joins s's list, then (i, s) o's.
// status[0] = number of picks made (== sample_size on success), status[1] = tries consumed (replay mode).
constexpr int NS_THREADS = 1024;

__device__ __forceinline__ int64_t ns_block_scan(int64_t v, int64_t* sh, int64_t* total) {   // exclusive prefix of v over the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int w = 0; w < NS_THREADS / 64; ++w) { const int64_t t = sh[w]; sh[w] = run; run += t; }
    sh[NS_THREADS / 64] = run;
  }
  __syncthreads();
  const int64_t excl = sh[wave] + x - v;
  *total = sh[NS_THREADS / 64];
  __syncthreads();                                          // sh is reused by the next scan
  return excl;
}

__global__ __launch_bounds__(NS_THREADS) void sample_neighborhood_k(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ adj_edge,
                                                                    const int32_t* __restrict__ adj_other, const int32_t* __restrict__ degrees,
                                                                    int64_t N, int64_t T, int64_t S, const double* __restrict__ u_vertex,
                                                                    const int64_t* __restrict__ tries, int64_t n_tries,
                                                                    const double* __restrict__ u_edge, int32_t* __restrict__ edges,
                                                                    int32_t* __restrict__ counts, unsigned char* __restrict__ seen,
                                                                    unsigned char* __restrict__ picked, int64_t* __restrict__ status) {
  __shared__ int64_t sh[NS_THREADS / 64 + 1];
  __shared__ long long pick_v, pick_slot, try_pos;
  __shared__ int fail;
  const int tid = threadIdx.x;
  for (int64_t v = tid; v < N; v += NS_THREADS) { counts[v] = degrees[v]; seen[v] = 0; }
  for (int64_t e = tid; e < T; e += NS_THREADS) picked[e] = 0;
  if (tid == 0) { try_pos = 0; fail = 0; status[0] = 0; status[1] = 0; }
  __syncthreads();
  const int64_t chunk = (N + NS_THREADS - 1) / NS_THREADS;
  const int64_t lo = tid * chunk < N ? tid * chunk : N, hi = lo + chunk < N ? lo + chunk : N;
  for (int64_t i = 0; i < S; ++i) {
    // ---- the vertex
    int64_t mine = 0;
    for (int64_t v = lo; v < hi; ++v) mine += (int64_t)counts[v] * seen[v];
    int64_t W;
    int64_t before = ns_block_scan(mine, sh, &W);
    bool uniform = false;
    if (W == 0) {                                           // nothing seen yet (or every seen vertex is exhausted): uniform over the live vertices
      uniform = true;
      mine = 0;
      for (int64_t v = lo; v < hi; ++v) mine += counts[v] != 0 ? 1 : 0;
      before = ns_block_scan(mine, sh, &W);
    }
    if (W == 0) { if (tid == 0) fail = 1; __syncthreads(); break; }      // fewer incident edges than picks asked for
    const double x = u_vertex[i] * (double)W;
    if (tid == 0) pick_v = -1;
    __syncthreads();
    if (mine > 0 && (double)before <= x && x < (double)(before + mine)) {
      int64_t run = before;
      for (int64_t v = lo; v < hi; ++v) {
        run += uniform ? (counts[v] != 0 ? 1 : 0) : (int64_t)counts[v] * seen[v];
        if ((double)run > x) { pick_v = v; break; }
      }
    }
    __syncthreads();
    const bool none = pick_v < 0;                           // u * W rounded up to W: take the last vertex with weight
    __syncthreads();                                        // everybody has read pick_v before anybody changes it
    if (none) {
      long long cand = -1;
      for (int64_t v2 = lo; v2 < hi; ++v2) if (uniform ? counts[v2] != 0 : (counts[v2] != 0 && seen[v2])) cand = v2;
      if (cand >= 0) atomicMax((long long*)&pick_v, cand);
    }
    __syncthreads();
    const int64_t v = pick_v;
    const int32_t a0 = rowptr[v], deg = rowptr[v + 1] - a0;
    // ---- the edge
    if (tries) {                                            // replay: the reference's own sequence of tried slots
      if (tid == 0) {
        int64_t slot = -1;
        while (try_pos < n_tries) {
          const int64_t t = tries[try_pos++];
          if (t < 0 || t >= deg) { fail = 2; break; }
          if (!picked[adj_edge[a0 + t]]) { slot = t; break; }
        }
        if (slot < 0 && !fail) fail = 3;
        pick_slot = slot;
      }
      __syncthreads();
    } else {                                                // random: the k-th unpicked entry, k = floor(u2 * sample_counts[v])
      int64_t k = (int64_t)(u_edge[i] * (double)counts[v]);
      if (k >= counts[v]) k = counts[v] - 1;
      const int64_t per = ((int64_t)deg + NS_THREADS - 1) / NS_THREADS;
      const int64_t l2 = tid * per < deg ? tid * per : deg, h2 = l2 + per < deg ? l2 + per : deg;
      int64_t free_mine = 0;
      for (int64_t t = l2; t < h2; ++t) free_mine += picked[adj_edge[a0 + t]] ? 0 : 1;
      int64_t tot;
      const int64_t fb = ns_block_scan(free_mine, sh, &tot);
      if (tid == 0) pick_slot = -1;
      __syncthreads();
      if (free_mine > 0 && fb <= k && k < fb + free_mine) {
        int64_t run = fb;
        for (int64_t t = l2; t < h2; ++t) {
          if (!picked[adj_edge[a0 + t]]) { if (run == k) { pick_slot = t; break; } ++run; }
        }
      }
      __syncthreads();
      if (tid == 0 && pick_slot < 0) fail = 4;
      __syncthreads();
    }
    if (fail) break;
    if (tid == 0) {
      const int32_t e = adj_edge[a0 + pick_slot], o = adj_other[a0 + pick_slot];
      edges[i] = e;
      picked[e] = 1;
      seen[v] = 1;
      counts[v] -= 1;
      counts[o] -= 1;
      seen[o] = 1;
      status[0] = i + 1;
      status[1] = try_pos;
    }
    __syncthreads();
  }
  if (tid == 0 && fail) status[2] = fail;
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


// Workspace: sample_counts int32 [N], seen uint8 [N], picked uint8 [T].
extern "C" int64_t mrg_sample_neighborhood_workspace_bytes(int64_t N, int64_t T) {
  if (N < 0 || T < 0) return 0;
  return (int64_t)((((size_t)N * 4 + 255) & ~(size_t)255) + (((size_t)N + 255) & ~(size_t)255) + (((size_t)T + 255) & ~(size_t)255) + 256);
}

extern "C" int mrg_sample_edge_neighborhood(const int32_t* rowptr, const int32_t* adj_edge, const int32_t* adj_other, const int32_t* degrees,
                                            int64_t N, int64_t T, int64_t sample_size, const double* u_vertex, const int64_t* tries,
                                            int64_t n_tries, const double* u_edge, int32_t* edges, int64_t* status, void* ws, int64_t ws_bytes,
                                            void* stream) {
  if (N <= 0 || T < 0 || sample_size < 0 || n_tries < 0) return MRG_E_SHAPE;
  if (!status) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(status, 0, 3 * sizeof(int64_t), st) != hipSuccess) return MRG_E_WORKSPACE;
  if (sample_size == 0) return MRG_OK;
  if (!rowptr || !adj_edge || !adj_other || !degrees || !u_vertex || !edges) return MRG_E_NULLPTR;
  if (!tries && !u_edge) return MRG_E_NULLPTR;
  if (!ws || ws_bytes < mrg_sample_neighborhood_workspace_bytes(N, T)) return MRG_E_WORKSPACE;
  char* base = (char*)ws;
  int32_t* counts = (int32_t*)base;
  unsigned char* seen = (unsigned char*)(base + (((size_t)N * 4 + 255) & ~(size_t)255));
  unsigned char* picked = seen + (((size_t)N + 255) & ~(size_t)255);
  hipLaunchKernelGGL(sample_neighborhood_k, dim3(1), dim3(NS_THREADS), 0, st, rowptr, adj_edge, adj_other, degrees, N, T, sample_size, u_vertex,
                     tries, n_tries, u_edge, edges, counts, seen, picked, status);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
