// f2: graph construction, edge ordering and the index plans of the segmented kernels,
// on the device.  Integer work, bit-exact with the host formulation it replaces:
//   build_graph_from_triplets + comp_deg_norm   reference utils/utils_rgcn.py:120-158  (inverse edges appended,
//       `sorted(zip(rel, dst, src))`, norm = in_degree ** -0.5 with inf -> 0)
//   node_norm_to_edge_norm                       reference search/mr_lp_search.py:30-36  (norm[dst] * norm[src])
//   build_graph                                  reference train/mr_lp_train.py:77-89    (un-sorted halves)
// and the span / chunk plans that mr-gnas_amd/graph.py used to assemble from a dozen torch argsort / bincount /
// cumsum launches each (10.8 ms per 30 000-edge step graph in round 1; the reference samples a new graph every step).
//
// Every plan is: histogram -> exclusive scan -> stable sort by segment -> a few marking kernels -> scans.  The sort
// and scan primitives are rocPRIM's (ROCm's own device library, header-only); everything around them is here.
// Data-dependent sizes (number of hubs / slots / chunks) are written to a small `counts` array the host reads once.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "common.hpp"

namespace mrg {

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static inline unsigned bits_for(uint64_t max_value) {           // bits needed to represent values in [0, max_value]
  unsigned b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

// bump allocator over the caller's workspace
struct Arena {
  char* base; size_t off, cap;
  template <typename T> T* take(size_t n) {
    T* p = reinterpret_cast<T*>(base + off);
    off += align_up(n * sizeof(T));
    return p;
  }
  bool ok() const { return off <= cap; }
};

static size_t sort_pairs_temp(int64_t n) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                  (size_t)(n > 0 ? n : 1), 0, 32);
  return b;
}
static size_t sort_keys64_temp(int64_t n) {
  size_t b = 0;
  (void)rocprim::radix_sort_keys(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)(n > 0 ? n : 1), 0, 64);
  return b;
}
static size_t scan_temp(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(n > 0 ? n : 1), rocprim::plus<int32_t>());
  return b;
}

static inline int blocks_for(int64_t n, int per = 256) {
  int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// cnt[key[i]] += 1.  Round 3's one-global-atomic-per-element form took 1.2 ms on average and 7.2 ms at worst per call (3 M scoring
// triples into 475 relation bins: every lane of the chip on the same few words).  Two forms now:
//  * nbins <= HIST_LDS_BINS: a workgroup counts its contiguous chunk of elements in LDS (ds atomics) and flushes the non-zero
//    bins with ONE global atomic each;
//  * more bins (node-sized histograms, up to 1 M bins): a thread walks a contiguous run of HIST_RUN elements and merges equal
//    neighbours before it touches memory -- the keys arrive sorted or nearly so (relation blocks, destination lists), so a
//    run of equal keys costs one atomic.
// Integer adds commute: the counts are exact whatever the order.
constexpr int HIST_LDS_BINS = 12288;     // 48 KB of int32
constexpr int HIST_RUN = 16;
__global__ void hist_lds_k(const int32_t* __restrict__ key, int64_t n, int32_t* __restrict__ cnt, int nbins, int64_t per_block) {
  extern __shared__ int32_t hist_sm[];
  for (int b = threadIdx.x; b < nbins; b += blockDim.x) hist_sm[b] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int32_t k = key[i];
    if ((unsigned)k < (unsigned)nbins) atomicAdd(&hist_sm[k], 1);          // -1 padding / out-of-range ids are skipped, not counted into a neighbour's bin
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += blockDim.x) {
    const int32_t v = hist_sm[b];
    if (v) atomicAdd(&cnt[b], v);
  }
}
__global__ void hist_run_k(const int32_t* __restrict__ key, int64_t n, int32_t* __restrict__ cnt, int64_t nbins) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t i = t * HIST_RUN;
  if (i >= n) return;
  const int64_t hi = i + HIST_RUN < n ? i + HIST_RUN : n;
  int32_t cur = key[i], c = 1;
  for (++i; i < hi; ++i) {
    const int32_t k = key[i];
    if (k == cur) { ++c; continue; }
    if ((uint64_t)cur < (uint64_t)nbins) atomicAdd(&cnt[cur], c);           // -1 padding / out-of-range ids are skipped (advisor r4)
    cur = k; c = 1;
  }
  if ((uint64_t)cur < (uint64_t)nbins) atomicAdd(&cnt[cur], c);
}
static inline void launch_hist(const int32_t* key, int64_t n, int32_t* cnt, int64_t nbins, hipStream_t st) {
  if (n <= 0) return;
  if (nbins > 0 && nbins <= HIST_LDS_BINS) {
    int64_t blocks = (n + 16383) / 16384;                  // >= 16 K elements per workgroup: the flush is amortised
    if (blocks > 1024) blocks = 1024;
    const int64_t per_block = (n + blocks - 1) / blocks;
    hipLaunchKernelGGL(hist_lds_k, dim3((unsigned)blocks), dim3(256), (size_t)nbins * 4, st, key, n, cnt, (int)nbins, per_block);
  } else {
    const int64_t threads = (n + HIST_RUN - 1) / HIST_RUN;
    hipLaunchKernelGGL(hist_run_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, key, n, cnt, nbins > 0 ? nbins : ((int64_t)1 << 62));
  }
}
__global__ void iota_k(int32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}
__global__ void set_i32_k(int32_t* p, int32_t v) { *p = v; }

// flags of the first / last run of every span (graph.span_plan): a run is partial when its segment does not lie
// wholly inside the span
// Cut i between spans i-1 and i (graph.span_plan_torch has the same rule): nominally i * span; when that position falls INSIDE a
// segment, the cut moves to the NEARER end of that segment if it is at most `snap` (<= span / 4) elements away -- segments up to 2 * snap long
// are then never split over spans (no partial runs, no workspace slots, no hub-pass work for them) while every span keeps
// between 1/2 and 3/2 of the nominal length (balance).  Cuts stay strictly increasing (a move is < span / 2); only the LAST cut
// may reach E, which leaves the last span empty.
__device__ __forceinline__ int64_t span_cut(const int32_t* __restrict__ seg_s, const int32_t* __restrict__ segptr, int64_t E, int span,
                                           int snap, int64_t n_spans, int64_t i) {
  if (i <= 0) return 0;
  if (i >= n_spans) return E;
  const int64_t p = i * span, lim = snap;
  if (lim <= 0) return p;
  const int s = seg_s[p - 1];
  const int64_t b = segptr[s], e = segptr[s + 1];
  if (e <= p) return p;                                     // already a segment boundary
  const int64_t fwd = e - p, bwd = p - b;
  if (fwd <= bwd) return fwd <= lim ? e : p;
  return bwd <= lim ? b : p;
}
__global__ void span_flags_k(const int32_t* __restrict__ seg_s, const int32_t* __restrict__ segptr, int64_t E, int span, int snap, int64_t n_spans,
                             int32_t* __restrict__ flags, int32_t* __restrict__ runseg, int32_t* __restrict__ span_start) {
  for (int64_t sp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sp < n_spans; sp += (int64_t)gridDim.x * blockDim.x) {
    const int64_t start = span_cut(seg_s, segptr, E, span, snap, n_spans, sp), end = span_cut(seg_s, segptr, E, span, snap, n_spans, sp + 1);
    span_start[sp] = (int32_t)start;
    if (sp == n_spans - 1) span_start[n_spans] = (int32_t)E;
    if (start >= end) {                                     // the last cut moved to E (the last segment is short): an empty span, no runs
      flags[2 * sp] = 0; flags[2 * sp + 1] = 0;
      runseg[2 * sp] = -1; runseg[2 * sp + 1] = -1;
      continue;
    }
    const int f = seg_s[start], l = seg_s[end - 1];
    const bool fp = (segptr[f] < start) || (segptr[f + 1] > end);
    const bool lp = (l != f) && (segptr[l + 1] > end);
    flags[2 * sp] = fp; flags[2 * sp + 1] = lp;
    runseg[2 * sp] = f; runseg[2 * sp + 1] = l;
  }
}
// span_slot = slot id or -1; slot_seg[slot] = segment of the partial run; also the scan total
__global__ void span_slots_k(const int32_t* __restrict__ flags, const int32_t* __restrict__ slot_id, const int32_t* __restrict__ runseg,
                             int64_t n, int32_t* __restrict__ span_slot, int32_t* __restrict__ slot_seg, int32_t* __restrict__ n_slots) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int fl = flags[i], id = slot_id[i];
    span_slot[i] = fl ? id : -1;
    if (fl) slot_seg[id] = runseg[i];
    if (i == n - 1) *n_slots = id + fl;
  }
}
// head[j] = slot j starts a new hub (slots of one segment are consecutive)
__global__ void hub_heads_k(const int32_t* __restrict__ slot_seg, const int32_t* __restrict__ n_slots, int64_t cap, int32_t* __restrict__ head) {
  const int ns = *n_slots;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cap; j += (int64_t)gridDim.x * blockDim.x)
    head[j] = (j < ns && (j == 0 || slot_seg[j] != slot_seg[j - 1])) ? 1 : 0;
}
__global__ void hub_write_k(const int32_t* __restrict__ slot_seg, const int32_t* __restrict__ head, const int32_t* __restrict__ hub_idx,
                            const int32_t* __restrict__ n_slots, int64_t cap, int32_t* __restrict__ hub_seg, int32_t* __restrict__ hub_first,
                            int32_t* __restrict__ n_part) {
  const int ns = *n_slots;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cap; j += (int64_t)gridDim.x * blockDim.x) {
    if (j < ns && head[j]) { hub_seg[hub_idx[j]] = slot_seg[j]; hub_first[hub_idx[j]] = (int32_t)j; }
    if (j == cap - 1) *n_part = hub_idx[j] + head[j];
  }
  if (cap == 0 && blockIdx.x == 0 && threadIdx.x == 0) *n_part = 0;
}
__global__ void hub_counts_k(const int32_t* __restrict__ hub_first, const int32_t* __restrict__ n_part, const int32_t* __restrict__ n_slots,
                             int64_t cap, int32_t* __restrict__ hub_count) {
  const int np = *n_part, ns = *n_slots;
  for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < cap; h += (int64_t)gridDim.x * blockDim.x)
    if (h < np) hub_count[h] = (h + 1 < np ? hub_first[h + 1] : ns) - hub_first[h];
}
// segments without elements are appended as hubs with zero partials (the hub pass writes their zero rows)
__global__ void empty_flags_k(const int32_t* __restrict__ seg_len, int64_t nseg, int32_t* __restrict__ fl) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nseg; v += (int64_t)gridDim.x * blockDim.x) fl[v] = seg_len[v] == 0;
}
__global__ void empty_append_k(const int32_t* __restrict__ fl, const int32_t* __restrict__ idx, int64_t nseg, const int32_t* __restrict__ n_part,
                               int32_t* __restrict__ hub_seg, int32_t* __restrict__ hub_first, int32_t* __restrict__ hub_count,
                               const int32_t* __restrict__ n_slots, int32_t* __restrict__ counts) {
  const int np = *n_part;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nseg; v += (int64_t)gridDim.x * blockDim.x) {
    if (fl[v]) { const int h = np + idx[v]; hub_seg[h] = (int32_t)v; hub_first[h] = 0; hub_count[h] = 0; }
    if (v == nseg - 1) { counts[0] = np + idx[v] + fl[v]; counts[1] = *n_slots; }
  }
}
// int32x4 {seg, xi[perm], yi[perm] | 0, bits(scal[perm]) | bits(1.0f) | perm}
__global__ void meta_pack_k(const int32_t* __restrict__ perm, const int32_t* __restrict__ seg_s, const int32_t* __restrict__ xi,
                            const int32_t* __restrict__ yi, const float* __restrict__ scal, int w_is_index, int64_t E, int4* __restrict__ meta) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (int64_t)gridDim.x * blockDim.x) {
    const int p = perm[j];
    int4 m;
    m.x = seg_s[j];
    m.y = xi ? xi[p] : p;
    m.z = yi ? yi[p] : 0;
    m.w = w_is_index ? p : __float_as_int(scal ? scal[p] : 1.0f);
    meta[j] = m;
  }
}

// ---- chunk plan (graph.dst_csr_plan) ------------------------------------------------------------
__global__ void chunk_counts_k(const int32_t* __restrict__ deg, int64_t N, int chunk, int32_t* __restrict__ nch, int32_t* __restrict__ nslot,
                               int32_t* __restrict__ ishub) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
    int c = (deg[v] + chunk - 1) / chunk;
    c = c < 1 ? 1 : c;
    nch[v] = c; nslot[v] = c > 1 ? c : 0; ishub[v] = c > 1;
  }
}
__global__ void chunk_write_k(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nch, const int32_t* __restrict__ first,
                              const int32_t* __restrict__ slot_base, const int32_t* __restrict__ hub_idx, int64_t N, int chunk, int64_t E,
                              int32_t* __restrict__ c_node, int32_t* __restrict__ c_start, int32_t* __restrict__ c_end, int32_t* __restrict__ c_slot,
                              int32_t* __restrict__ hub_node, int32_t* __restrict__ hub_first, int32_t* __restrict__ hub_count,
                              int32_t* __restrict__ counts) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = nch[v], f = first[v], r0 = rowptr[v], r1 = (v + 1 < N) ? rowptr[v + 1] : (int)E;
    for (int k = 0; k < c; ++k) {
      const int s = r0 + k * chunk, e = s + chunk < r1 ? s + chunk : r1;
      c_node[f + k] = (int32_t)v; c_start[f + k] = s; c_end[f + k] = e;
      c_slot[f + k] = c > 1 ? slot_base[v] + k : -1;
    }
    if (c > 1) { const int h = hub_idx[v]; hub_node[h] = (int32_t)v; hub_first[h] = slot_base[v]; hub_count[h] = c; }
    if (v == N - 1) { counts[0] = f + c; counts[1] = hub_idx[v] + (c > 1); counts[2] = slot_base[v] + (c > 1 ? c : 0); }
  }
}

// ---- graph construction ---------------------------------------------------------------------------
// directed edges of T triples: e < T original (s -> o, r), e >= T inverse (o -> s, r + R)
__global__ void edges_from_triples_k(const int64_t* __restrict__ tri, int64_t T, int64_t N, int R, int sorted, uint64_t* __restrict__ key,
                                     int64_t* __restrict__ src, int64_t* __restrict__ dst, int64_t* __restrict__ et, int32_t* __restrict__ deg) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < 2 * T; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e < T ? e : e - T;
    const int64_t s = tri[3 * t], r = tri[3 * t + 1], o = tri[3 * t + 2];
    const int64_t u = e < T ? s : o, v = e < T ? o : s, rr = e < T ? r : r + R;
    atomicAdd(&deg[v], 1);
    if (sorted) key[e] = ((uint64_t)rr * (uint64_t)N + (uint64_t)v) * (uint64_t)N + (uint64_t)u;     // (rel, dst, src) lexicographic
    else { src[e] = u; dst[e] = v; et[e] = rr; }
  }
}
__global__ void edges_decode_k(const uint64_t* __restrict__ key, int64_t E, int64_t N, int64_t* __restrict__ src, int64_t* __restrict__ dst,
                               int64_t* __restrict__ et) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = key[e];
    src[e] = (int64_t)(k % (uint64_t)N);
    dst[e] = (int64_t)((k / (uint64_t)N) % (uint64_t)N);
    et[e] = (int64_t)(k / ((uint64_t)N * (uint64_t)N));
  }
}
// norm[e] = tab[deg[dst]] * tab[deg[src]] -- tab[d] = float32(d) ** float32(-0.5) (0 for d = 0), computed by the HOST's numpy so
// that the values are the reference's own (a device rsqrt / pow may differ in the last bit); the product is an IEEE float32 multiply
__global__ void edge_norm_k(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const int32_t* __restrict__ deg,
                            const float* __restrict__ tab, int64_t E, float* __restrict__ norm, int32_t* __restrict__ src32,
                            int32_t* __restrict__ dst32, int32_t* __restrict__ et32, const int64_t* __restrict__ et) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = src[e], v = dst[e];
    norm[e] = tab[deg[v]] * tab[deg[u]];
    if (src32) { src32[e] = (int32_t)u; dst32[e] = (int32_t)v; et32[e] = (int32_t)et[e]; }
  }
}
__global__ void max_i32_k(const int32_t* __restrict__ x, int64_t n, int32_t* __restrict__ out) {
  __shared__ int red[256];
  int m = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) m = x[i] > m ? x[i] : m;
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

}  // namespace mrg

using namespace mrg;

#define MRG_HIP(expr)                      \
  do {                                     \
    hipError_t e__ = (expr);               \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

extern "C" int64_t mrg_plan_workspace_bytes(int64_t E, int64_t nseg, int span) {
  if (E < 0 || nseg < 0 || span < 1) return 0;
  const int64_t n_spans = (E + span - 1) / span, cap = 2 * n_spans;
  size_t tmp = sort_pairs_temp(E);
  size_t t2 = scan_temp(nseg + 1 > cap ? nseg + 1 : cap);
  tmp = tmp > t2 ? tmp : t2;
  size_t b = align_up(tmp);
  b += align_up((size_t)E * 4);                         // iota
  b += align_up((size_t)(nseg + 1) * 4);                // segptr
  b += 4 * align_up((size_t)(cap + 1) * 4);             // flags, runseg, slot_id, slot_seg
  b += 2 * align_up((size_t)(cap + 1) * 4);             // head, hub_idx
  b += 2 * align_up((size_t)(nseg + 1) * 4);            // empty flags, empty idx
  b += align_up(64);                                    // scalars
  return (int64_t)b;
}

// hub_seg is filled with -1 over its whole capacity (2 * n_spans + nseg) first: a consumer may launch with the CAPACITY as
// the hub count (no device-to-host read of `counts`), the hub pass skips negative entries.
extern "C" int mrg_span_plan_build(const int32_t* seg, int64_t E, int64_t nseg, int span, int snap, int32_t* perm, int32_t* seg_sorted,
                                   int32_t* seg_len, int32_t* span_slot, int32_t* span_start, int32_t* hub_seg, int32_t* hub_first,
                                   int32_t* hub_count, int32_t* counts, void* ws, int64_t ws_bytes, void* stream) {
  if (E < 0 || nseg < 0 || span < 1 || snap < 0 || snap > span / 4) return MRG_E_SHAPE;
  if (!counts || (nseg > 0 && !seg_len)) return MRG_E_NULLPTR;
  if (E > 0 && (!seg || !perm || !seg_sorted || !span_slot || !span_start)) return MRG_E_NULLPTR;
  if (nseg > 0 && (!hub_seg || !hub_first || !hub_count)) return MRG_E_NULLPTR;
  if (!ws || ws_bytes < mrg_plan_workspace_bytes(E, nseg, span)) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_spans = (E + span - 1) / span, cap = 2 * n_spans;
  Arena A{(char*)ws, 0, (size_t)ws_bytes};
  size_t tmp_bytes = sort_pairs_temp(E);
  { size_t t2 = scan_temp(nseg + 1 > cap ? nseg + 1 : cap); tmp_bytes = tmp_bytes > t2 ? tmp_bytes : t2; }
  void* tmp = A.take<char>(tmp_bytes);
  int32_t* iota = A.take<int32_t>(E);
  int32_t* segptr = A.take<int32_t>(nseg + 1);
  int32_t* flags = A.take<int32_t>(cap + 1);
  int32_t* runseg = A.take<int32_t>(cap + 1);
  int32_t* slot_id = A.take<int32_t>(cap + 1);
  int32_t* slot_seg = A.take<int32_t>(cap + 1);
  int32_t* head = A.take<int32_t>(cap + 1);
  int32_t* hub_idx = A.take<int32_t>(cap + 1);
  int32_t* efl = A.take<int32_t>(nseg + 1);
  int32_t* eidx = A.take<int32_t>(nseg + 1);
  int32_t* scal = A.take<int32_t>(16);                  // [0] n_slots, [1] n_part
  if (!A.ok()) return MRG_E_WORKSPACE;
  int32_t* n_slots = scal, *n_part = scal + 1;
  MRG_HIP(hipMemsetAsync(scal, 0, 64, st));
  MRG_HIP(hipMemsetAsync(counts, 0, 8, st));
  if (nseg == 0) return MRG_OK;
  MRG_HIP(hipMemsetAsync(hub_seg, 0xFF, (size_t)(cap + nseg) * 4, st));
  MRG_HIP(hipMemsetAsync(seg_len, 0, (size_t)nseg * 4, st));
  if (E > 0) {
    launch_hist(seg, E, seg_len, nseg, st);
    hipLaunchKernelGGL(iota_k, dim3(blocks_for(E)), dim3(256), 0, st, iota, E);
  }
  size_t tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, seg_len, segptr, 0, (size_t)nseg, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(set_i32_k, dim3(1), dim3(1), 0, st, segptr + nseg, (int32_t)E);
  if (E > 0) {
    tb = tmp_bytes;
    MRG_HIP(rocprim::radix_sort_pairs(tmp, tb, seg, seg_sorted, (const int32_t*)iota, perm, (size_t)E, 0, bits_for((uint64_t)(nseg - 1)), st));
    hipLaunchKernelGGL(span_flags_k, dim3(blocks_for(n_spans)), dim3(256), 0, st, seg_sorted, segptr, E, span, snap, n_spans, flags, runseg, span_start);
    tb = tmp_bytes;
    MRG_HIP(rocprim::exclusive_scan(tmp, tb, flags, slot_id, 0, (size_t)cap, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(span_slots_k, dim3(blocks_for(cap)), dim3(256), 0, st, flags, slot_id, runseg, cap, span_slot, slot_seg, n_slots);
    hipLaunchKernelGGL(hub_heads_k, dim3(blocks_for(cap)), dim3(256), 0, st, slot_seg, n_slots, cap, head);
    tb = tmp_bytes;
    MRG_HIP(rocprim::exclusive_scan(tmp, tb, head, hub_idx, 0, (size_t)cap, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(hub_write_k, dim3(blocks_for(cap)), dim3(256), 0, st, slot_seg, head, hub_idx, n_slots, cap, hub_seg, hub_first, n_part);
    hipLaunchKernelGGL(hub_counts_k, dim3(blocks_for(cap)), dim3(256), 0, st, hub_first, n_part, n_slots, cap, hub_count);
  }
  hipLaunchKernelGGL(empty_flags_k, dim3(blocks_for(nseg)), dim3(256), 0, st, seg_len, nseg, efl);
  tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, efl, eidx, 0, (size_t)nseg, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(empty_append_k, dim3(blocks_for(nseg)), dim3(256), 0, st, efl, eidx, nseg, n_part, hub_seg, hub_first, hub_count, n_slots, counts);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int mrg_span_meta_pack(const int32_t* perm, const int32_t* seg_sorted, const int32_t* xi, const int32_t* yi, const float* scal,
                                  int w_is_index, void* meta, int64_t E, void* stream) {
  if (E < 0) return MRG_E_SHAPE;
  if (E == 0) return MRG_OK;
  if (!perm || !seg_sorted || !meta) return MRG_E_NULLPTR;
  hipLaunchKernelGGL(meta_pack_k, dim3(blocks_for(E)), dim3(256), 0, (hipStream_t)stream, perm, seg_sorted, xi, yi, scal, w_is_index, E, (int4*)meta);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int64_t mrg_chunk_plan_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  size_t tmp = sort_pairs_temp(E), t2 = scan_temp(N + 1);
  tmp = tmp > t2 ? tmp : t2;
  return (int64_t)(align_up(tmp) + 2 * align_up((size_t)E * 4) + 6 * align_up((size_t)(N + 1) * 4) + align_up(64));
}

extern "C" int mrg_chunk_plan_build(const int32_t* dst, int64_t E, int64_t N, int chunk, int32_t* eid, int32_t* rowptr, int32_t* in_degree,
                                    int32_t* chunk_node, int32_t* chunk_start, int32_t* chunk_end, int32_t* chunk_slot, int32_t* hub_node,
                                    int32_t* hub_first, int32_t* hub_count, int32_t* counts, void* ws, int64_t ws_bytes, void* stream) {
  if (E < 0 || N < 0 || chunk < 1) return MRG_E_SHAPE;
  if (!counts) return MRG_E_NULLPTR;
  hipStream_t st = (hipStream_t)stream;
  MRG_HIP(hipMemsetAsync(counts, 0, 12, st));
  if (N == 0) return MRG_OK;
  if (!rowptr || !in_degree || !chunk_node || !chunk_start || !chunk_end || !chunk_slot || !hub_node || !hub_first || !hub_count) return MRG_E_NULLPTR;
  if (E > 0 && (!dst || !eid)) return MRG_E_NULLPTR;
  if (!ws || ws_bytes < mrg_chunk_plan_workspace_bytes(E, N)) return MRG_E_WORKSPACE;
  Arena A{(char*)ws, 0, (size_t)ws_bytes};
  size_t tmp_bytes = sort_pairs_temp(E);
  { size_t t2 = scan_temp(N + 1); tmp_bytes = tmp_bytes > t2 ? tmp_bytes : t2; }
  void* tmp = A.take<char>(tmp_bytes);
  int32_t* iota = A.take<int32_t>(E);
  int32_t* dsts = A.take<int32_t>(E);
  int32_t* nch = A.take<int32_t>(N + 1);
  int32_t* nslot = A.take<int32_t>(N + 1);
  int32_t* ishub = A.take<int32_t>(N + 1);
  int32_t* first = A.take<int32_t>(N + 1);
  int32_t* slot_base = A.take<int32_t>(N + 1);
  int32_t* hub_idx = A.take<int32_t>(N + 1);
  if (!A.ok()) return MRG_E_WORKSPACE;
  MRG_HIP(hipMemsetAsync(in_degree, 0, (size_t)N * 4, st));
  MRG_HIP(hipMemsetAsync(chunk_node, 0xFF, (size_t)(N + E / chunk + 1) * 4, st));     // capacity-sized launches skip the -1 padding
  MRG_HIP(hipMemsetAsync(hub_node, 0xFF, (size_t)(E / chunk + 1) * 4, st));
  if (E > 0) {
    launch_hist(dst, E, in_degree, N, st);
    hipLaunchKernelGGL(iota_k, dim3(blocks_for(E)), dim3(256), 0, st, iota, E);
    size_t tb = tmp_bytes;
    MRG_HIP(rocprim::radix_sort_pairs(tmp, tb, dst, dsts, (const int32_t*)iota, eid, (size_t)E, 0, bits_for((uint64_t)(N - 1)), st));
  }
  size_t tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, in_degree, rowptr, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(set_i32_k, dim3(1), dim3(1), 0, st, rowptr + N, (int32_t)E);
  hipLaunchKernelGGL(chunk_counts_k, dim3(blocks_for(N)), dim3(256), 0, st, in_degree, N, chunk, nch, nslot, ishub);
  tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, nch, first, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, nslot, slot_base, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  tb = tmp_bytes;
  MRG_HIP(rocprim::exclusive_scan(tmp, tb, ishub, hub_idx, 0, (size_t)N, rocprim::plus<int32_t>(), st));
  hipLaunchKernelGGL(chunk_write_k, dim3(blocks_for(N)), dim3(256), 0, st, rowptr, nch, first, slot_base, hub_idx, N, chunk, E, chunk_node,
                     chunk_start, chunk_end, chunk_slot, hub_node, hub_first, hub_count, counts);
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}

extern "C" int64_t mrg_build_graph_workspace_bytes(int64_t T) {
  if (T < 0) return 0;
  return (int64_t)(align_up(sort_keys64_temp(2 * T)) + 2 * align_up((size_t)2 * T * 8) + align_up(64));
}

extern "C" int mrg_build_graph(const int64_t* triples, int64_t T, int64_t N, int R, int sorted, const float* deg_norm_table, int64_t table_len,
                               int64_t* src, int64_t* dst, int64_t* etype, float* norm, int32_t* in_degree, int32_t* src32, int32_t* dst32,
                               int32_t* etype32, int32_t* max_degree, void* ws, int64_t ws_bytes, void* stream) {
  if (T < 0 || N < 0 || R < 0 || table_len < 0) return MRG_E_SHAPE;
  if (T == 0) {                                   // an empty split still publishes its degree vector: all zeros
    if (N > 0 && in_degree) MRG_HIP(hipMemsetAsync(in_degree, 0, (size_t)N * 4, (hipStream_t)stream));
    if (max_degree) MRG_HIP(hipMemsetAsync(max_degree, 0, 4, (hipStream_t)stream));
    return MRG_OK;
  }
  if (N == 0) return MRG_E_SHAPE;
  if (!triples || !src || !dst || !etype || !in_degree) return MRG_E_NULLPTR;
  if (sorted && (bits_for((uint64_t)(2 * R)) + 2 * bits_for((uint64_t)(N - 1)) > 64)) return MRG_E_SHAPE;    // the 64-bit sort key does not fit
  if (!ws || ws_bytes < mrg_build_graph_workspace_bytes(T)) return MRG_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t E = 2 * T;
  Arena A{(char*)ws, 0, (size_t)ws_bytes};
  const size_t tmp_bytes = sort_keys64_temp(E);
  void* tmp = A.take<char>(tmp_bytes);
  uint64_t* key = A.take<uint64_t>(E);
  uint64_t* key_s = A.take<uint64_t>(E);
  MRG_HIP(hipMemsetAsync(in_degree, 0, (size_t)N * 4, st));
  hipLaunchKernelGGL(edges_from_triples_k, dim3(blocks_for(E)), dim3(256), 0, st, triples, T, N, R, sorted, key, src, dst, etype, in_degree);
  if (sorted) {
    size_t tb = tmp_bytes;
    const uint64_t kmax = ((uint64_t)(2 * R) * (uint64_t)N + (uint64_t)(N - 1)) * (uint64_t)N + (uint64_t)(N - 1);
    MRG_HIP(rocprim::radix_sort_keys(tmp, tb, (const uint64_t*)key, key_s, (size_t)E, 0, bits_for(kmax), st));
    hipLaunchKernelGGL(edges_decode_k, dim3(blocks_for(E)), dim3(256), 0, st, key_s, E, N, src, dst, etype);
  }
  if (max_degree) hipLaunchKernelGGL(max_i32_k, dim3(1), dim3(256), 0, st, in_degree, N, max_degree);
  if (norm) {
    if (!deg_norm_table) return MRG_E_NULLPTR;
    // table_len must exceed the largest in-degree; the host checks max_degree after the call when it cannot bound it before
    hipLaunchKernelGGL(edge_norm_k, dim3(blocks_for(E)), dim3(256), 0, st, src, dst, in_degree, deg_norm_table, E, norm, src32, dst32, etype32, etype);
  }
  MRG_LAUNCH_CHECK();
  return MRG_OK;
}
