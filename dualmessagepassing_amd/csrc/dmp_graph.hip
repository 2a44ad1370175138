// Integer graph-index kernels of the dual-message-passing hot path (gfx950).
//
//   csr_build          counting sort of the edge list by dst (or src); rows in
//                      ascending eid -> every later segment sum has one fixed order
//   incidence_build    in-edges and out-edges (flag flipped) per node, merged by edge id
//   degree_coef        2(1+log2(1+out_deg))
//   collate            dgl.batch: node-offset concatenation
//   add_reversed_edges train.py:299-327
//   line_graph_*       utils/graph.py:74-169 (directed line graph, emission order)
//   scans              exclusive prefix sums (int32 / int64)
//
// Every result here is an integer array and must equal the reference's bit for
// bit; nothing depends on atomic arrival order (rows are sorted after the
// atomic fill, hash-table winners are chosen by atomicMin on the item index).
#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kScanItems = 8;                       // items per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 items per workgroup

// ------------------------------------------------------------------ scans
// Three-phase exclusive scan: (A) per-tile scan + tile totals, (B) one
// workgroup scans the tile totals, (C) add tile offsets, write the grand total.
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *lds, T &total) {
  // v: this thread's value; returns exclusive prefix within the workgroup
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  T inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    T t = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += t;
  }
  if (lane == kWave - 1) lds[wave] = inc;
  __syncthreads();
  T wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    const T s = lds[w];
    if (w < wave) wave_off += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return wave_off + inc - v;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void scan_tiles(const TI *__restrict__ in, int64_t n,
                                                     TO *__restrict__ out, TO *__restrict__ tile_sum) {
  __shared__ TO lds[kBlock / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  TO v[kScanItems];
  TO s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? (TO)in[base + k] : (TO)0;
    s += v[k];
  }
  TO total;
  TO pre = block_exclusive_scan<TO>(s, lds, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

template <typename TO>
__global__ __launch_bounds__(kBlock) void scan_tile_sums(TO *__restrict__ tile_sum, int64_t ntiles,
                                                         TO *__restrict__ grand_total) {
  __shared__ TO lds[kBlock / kWave];
  TO carry = 0;
  for (int64_t base = 0; base < ntiles; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const TO v = i < ntiles ? tile_sum[i] : (TO)0;
    TO total;
    const TO pre = block_exclusive_scan<TO>(v, lds, total);
    if (i < ntiles) tile_sum[i] = carry + pre;
    carry += total;
  }
  if (threadIdx.x == 0) *grand_total = carry;
}

template <typename TO>
__global__ __launch_bounds__(kBlock) void scan_add_offsets(TO *__restrict__ out, int64_t n,
                                                           const TO *__restrict__ tile_sum) {
  const TO off = tile_sum[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) out[base + k] += off;
}

// out[0..n] = exclusive scan of in[0..n-1]; tile_ws holds ceil(n/2048) entries.
template <typename TI, typename TO>
int exclusive_scan(const TI *in, int64_t n, TO *out, TO *tile_ws, hipStream_t st) {
  if (n == 0) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(TO), st);
    if (e != hipSuccess) { set_last_hip_error(e); return DMP_ERR_HIP; }
    return DMP_OK;
  }
  const int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  scan_tiles<TI, TO><<<(unsigned)ntiles, kBlock, 0, st>>>(in, n, out, tile_ws);
  scan_tile_sums<TO><<<1, kBlock, 0, st>>>(tile_ws, ntiles, out + n);
  if (ntiles > 1) scan_add_offsets<TO><<<(unsigned)ntiles, kBlock, 0, st>>>(out, n, tile_ws);
  return check_launch();
}

// ------------------------------------------------------------------ CSR build
__global__ __launch_bounds__(kBlock) void csr_count(const int64_t *__restrict__ key, int64_t E,
                                                    int64_t N, int32_t *__restrict__ cnt,
                                                    int32_t *__restrict__ key32,
                                                    int32_t *__restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t k = key[e];
  const bool ok = k >= 0 && k < N;
  if (key32) key32[e] = ok ? (int32_t)k : 0;
  if (ok) atomicAdd(&cnt[k], 1);
  else atomicOr(status, 1);
}

__global__ __launch_bounds__(kBlock) void csr_fill(const int64_t *__restrict__ key,
                                                   const uint8_t *__restrict__ flag, int64_t E,
                                                   int64_t N, int32_t *__restrict__ cursor,
                                                   int32_t *__restrict__ ent) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t k = key[e];
  if (k < 0 || k >= N) return;
  const int pos = atomicAdd(&cursor[k], 1);
  ent[pos] = ((int32_t)e << 1) | (flag ? (flag[e] ? 1 : 0) : 0);
}

// Restore ascending eid inside each row (the atomic fill arrives almost sorted,
// so insertion sort is near linear); rows longer than kHeapFrom use heap sort.
constexpr int kHeapFrom = 96;
constexpr int kRegSort = 32;
__device__ void sift_down(int32_t *a, int start, int end) {
  int root = start;
  while (2 * root + 1 <= end) {
    int child = 2 * root + 1, sw = root;
    if (a[sw] < a[child]) sw = child;
    if (child + 1 <= end && a[sw] < a[child + 1]) sw = child + 1;
    if (sw == root) return;
    const int32_t t = a[root]; a[root] = a[sw]; a[sw] = t;
    root = sw;
  }
}
__device__ __forceinline__ void csr_sort_span(int32_t *__restrict__ a, const int n) {
  if (n <= kRegSort) {
    // Short rows (the common case): one batch of loads, a bitonic network in
    // registers, one batch of stores -- no dependent global round trips.
    int32_t v[kRegSort];
#pragma unroll
    for (int i = 0; i < kRegSort; ++i) v[i] = i < n ? a[i] : INT32_MAX;
#pragma unroll
    for (int k = 2; k <= kRegSort; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int i = 0; i < kRegSort; ++i) {
          const int l = i ^ j;
          if (l > i) {
            const int32_t lo = min(v[i], v[l]), hi = max(v[i], v[l]);
            const bool up = (i & k) == 0;
            v[i] = up ? lo : hi;
            v[l] = up ? hi : lo;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kRegSort; ++i)
      if (i < n) a[i] = v[i];
  } else if (n <= kHeapFrom) {
    for (int i = 1; i < n; ++i) {
      const int32_t x = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; }
      a[j + 1] = x;
    }
  } else {
    for (int s = (n - 2) / 2; s >= 0; --s) sift_down(a, s, n - 1);
    for (int e = n - 1; e > 0; --e) {
      const int32_t t = a[e]; a[e] = a[0]; a[0] = t;
      sift_down(a, 0, e - 1);
    }
  }
}
__device__ __forceinline__ void csr_sort_row(const int32_t *__restrict__ rowptr, int64_t N, int32_t *__restrict__ ent,
                                             int64_t *__restrict__ degree, int64_t r) {
  if (r >= N) return;
  const int beg = rowptr[r], end = rowptr[r + 1];
  if (degree) degree[r] = end - beg;
  csr_sort_span(ent + beg, end - beg);
}

__global__ __launch_bounds__(kBlock) void csr_sort_rows(const int32_t *__restrict__ rowptr, int64_t N,
                                                        int32_t *__restrict__ ent,
                                                        int64_t *__restrict__ degree) {
  csr_sort_row(rowptr, N, ent, degree, (int64_t)blockIdx.x * kBlock + threadIdx.x);
}

// ---- both CSRs of a graph (by destination and by source) built side by side: blockIdx.y picks the key array, so the
// nine dispatches of a build are paid once for the pair
struct CsrPair {
  const int64_t *key[2];
  int32_t *rowptr[2], *ent[2], *key32[2], *cnt[2], *tiles[2];
  int64_t *degree[2];
  int32_t *status;                                              // [2]
};
__global__ __launch_bounds__(kBlock) void csr_count_pair(const CsrPair p, int64_t E, int64_t N) {
  const int y = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t k = p.key[y][e];
  const bool ok = k >= 0 && k < N;
  if (p.key32[y]) p.key32[y][e] = ok ? (int32_t)k : 0;
  if (ok) atomicAdd(&p.cnt[y][k], 1);
  else atomicOr(p.status + y, 1);
}
__global__ __launch_bounds__(kBlock) void scan_tiles_pair(const CsrPair p, int64_t n) {
  __shared__ int32_t lds[kBlock / kWave];
  const int y = blockIdx.y;
  const int32_t *__restrict__ in = p.cnt[y];
  int32_t *__restrict__ out = p.rowptr[y];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t v[kScanItems];
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    sum += v[k];
  }
  int32_t total;
  int32_t pre = block_exclusive_scan<int32_t>(sum, lds, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) p.tiles[y][blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void scan_tile_sums_pair(const CsrPair p, int64_t ntiles, int64_t n) {
  __shared__ int32_t lds[kBlock / kWave];
  const int y = blockIdx.y;
  int32_t *__restrict__ tile_sum = p.tiles[y];
  int32_t carry = 0;
  for (int64_t base = 0; base < ntiles; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const int32_t v = i < ntiles ? tile_sum[i] : 0;
    int32_t total;
    const int32_t pre = block_exclusive_scan<int32_t>(v, lds, total);
    if (i < ntiles) tile_sum[i] = carry + pre;
    carry += total;
  }
  if (threadIdx.x == 0) p.rowptr[y][n] = carry;
}
// rowptr += tile offset; the counters become the fill cursors (= rowptr)
__global__ __launch_bounds__(kBlock) void scan_finish_pair(const CsrPair p, int64_t n) {
  const int y = blockIdx.y;
  const int32_t off = p.tiles[y][blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) {
      const int32_t v = p.rowptr[y][base + k] + off;
      p.rowptr[y][base + k] = v;
      p.cnt[y][base + k] = v;
    }
}
__global__ __launch_bounds__(kBlock) void csr_fill_pair(const CsrPair p, const uint8_t *__restrict__ flag, int64_t E, int64_t N) {
  const int y = blockIdx.y;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t k = p.key[y][e];
  if (k < 0 || k >= N) return;
  const int pos = atomicAdd(&p.cnt[y][k], 1);
  p.ent[y][pos] = ((int32_t)e << 1) | (flag ? (flag[e] ? 1 : 0) : 0);
}
__global__ __launch_bounds__(kBlock) void csr_sort_rows_pair(const CsrPair p, int64_t N) {
  const int y = blockIdx.y;
  csr_sort_row(p.rowptr[y], N, p.ent[y], p.degree[y], (int64_t)blockIdx.x * kBlock + threadIdx.x);
}

// ---- both CSRs of a BLOCK-DIAGONAL batch in ONE launch (dmp_csr_build_graphs): a workgroup per graph.  The edges of
// graph g are rows [edge_off[g], edge_off[g + 1]) of both entry arrays and touch only its nodes, so the degree counters, their
// prefix sums and the fill cursors of a graph live in LDS (a graph of at most kCsrGraphNodes nodes): count -> scan -> fill ->
// sort every row by edge id, with workgroup barriers between the phases instead of eight dispatches and a zeroed scratch.
constexpr int kCsrGraphNodes = 2048;
struct CsrGraphs {
  const int64_t *key[2]; const uint8_t *flag; const int64_t *node_off, *edge_off; int64_t B, N, E;
  int32_t *rowptr[2], *ent[2], *key32[2]; int64_t *degree[2]; int32_t *status;
};
__global__ __launch_bounds__(kBlock) void csr_graphs_k(const CsrGraphs p) {
  __shared__ int32_t cnt[2][kCsrGraphNodes];
  __shared__ int32_t lds[kBlock / kWave];
  const int64_t g = blockIdx.x;
  const int64_t n0 = p.node_off[g], e0 = p.edge_off[g], e1 = p.edge_off[g + 1];
  const int n = (int)(p.node_off[g + 1] - n0);
  for (int i = threadIdx.x; i < n; i += kBlock) { cnt[0][i] = 0; cnt[1][i] = 0; }
  __syncthreads();
  for (int64_t e = e0 + threadIdx.x; e < e1; e += kBlock) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int64_t k = p.key[y][e];
      const bool ok = k >= n0 && k < n0 + n;
      if (p.key32[y]) p.key32[y][e] = ok ? (int32_t)k : 0;
      if (ok) atomicAdd(&cnt[y][k - n0], 1);
      else atomicOr(p.status + y, 1);
    }
  }
  __syncthreads();
  const int ipt = (n + kBlock - 1) / kBlock, base = threadIdx.x * ipt;        // consecutive nodes per thread
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    int32_t sum = 0;
    for (int i = base; i < base + ipt && i < n; ++i) sum += cnt[y][i];
    int32_t total;
    int32_t pre = (int32_t)e0 + block_exclusive_scan<int32_t>(sum, lds, total);
    for (int i = base; i < base + ipt && i < n; ++i) {
      const int32_t v = cnt[y][i];
      p.rowptr[y][n0 + i] = pre;
      cnt[y][i] = pre;                                                        // the row's fill cursor
      pre += v;
    }
    if (g == p.B - 1 && threadIdx.x == 0) p.rowptr[y][p.N] = (int32_t)p.E;
  }
  __syncthreads();
  for (int64_t e = e0 + threadIdx.x; e < e1; e += kBlock) {
    const int32_t v = ((int32_t)e << 1) | (p.flag ? (p.flag[e] ? 1 : 0) : 0);
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int64_t k = p.key[y][e];
      if (k >= n0 && k < n0 + n) p.ent[y][atomicAdd(&cnt[y][k - n0], 1)] = v;
    }
  }
  __syncthreads();                                                            // the entries of this graph are in place (same workgroup)
  for (int i = threadIdx.x; i < n; i += kBlock) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int beg = i == 0 ? (int)e0 : cnt[y][i - 1], end = cnt[y][i];      // a cursor ends where the next row begins
      if (p.degree[y]) p.degree[y][n0 + i] = end - beg;
      csr_sort_span(p.ent[y] + beg, end - beg);
    }
  }
}

__global__ __launch_bounds__(kBlock) void copy_i32(const int32_t *__restrict__ a, int64_t n,
                                                   int32_t *__restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) b[i] = a[i];
}

// ------------------------------------------------------------------ incidence
__global__ __launch_bounds__(kBlock) void incidence_ptr(const int32_t *__restrict__ in_ptr,
                                                        const int32_t *__restrict__ out_ptr, int64_t N,
                                                        int32_t *__restrict__ inc_ptr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i <= N) inc_ptr[i] = in_ptr[i] + out_ptr[i];
}
// one thread per node: its in-entries and its out-entries (flag bit flipped) MERGED by ascending edge id (both lists are
// sorted; an edge that is both -- a self loop -- lists its in-entry first), so that every per-node, per-half sum over the
// incidence CSR runs in ascending eid: the order in which the one-pass kernel (csrc/dmp_segacc.hip) streams the edge rows.
// (A thread per entry with a binary search over the row offsets was 17 dependent round trips per entry: 23 us for 1.1 M
// entries.)
__global__ __launch_bounds__(kBlock) void incidence_fill(const int32_t *__restrict__ in_ptr,
                                                         const int32_t *__restrict__ in_ent,
                                                         const int32_t *__restrict__ out_ptr,
                                                         const int32_t *__restrict__ out_ent, int64_t N,
                                                         int64_t E2, int32_t *__restrict__ inc_ent) {
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= N) return;
  int a = in_ptr[w], b = out_ptr[w];
  const int a1 = in_ptr[w + 1], b1 = out_ptr[w + 1];
  int64_t o = (int64_t)a + b;
  int x = a < a1 ? in_ent[a] : 0, y = b < b1 ? out_ent[b] : 0;
  while ((a < a1 || b < b1) && o < E2) {
    if (b >= b1 || (a < a1 && (x >> 1) <= (y >> 1))) {
      inc_ent[o++] = x;
      if (++a < a1) x = in_ent[a];
    } else {
      inc_ent[o++] = y ^ 1;
      if (++b < b1) y = out_ent[b];
    }
  }
}

__global__ __launch_bounds__(kBlock) void degree_coef_k(const int64_t *__restrict__ deg, int64_t N,
                                                        float *__restrict__ coef) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  // dmpnn.py:144-146: d = out_deg.float(); d = (1 + d).log2(); 2 * (1 + d)
  const float d = log2f(1.0f + (float)deg[i]);
  coef[i] = 2.0f * (1.0f + d);
}

// ------------------------------------------------------------------ collate
__device__ __forceinline__ int64_t upper_graph(const int64_t *off, int64_t B, int64_t i) {
  // largest g in [0,B) with off[g] <= i   (off has B+1 entries, off[B] > i)
  int64_t lo = 0, hi = B;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}
// node and edge offsets in one launch (blockIdx.y: 0 = nodes, 1 = edges; B <= one scan tile), then the edge and the
// node pass in one launch: four dispatches of a collate become two
__global__ __launch_bounds__(kBlock) void collate_offsets(const int64_t *__restrict__ num_nodes, const int64_t *__restrict__ num_edges,
                                                          int64_t B, int64_t *__restrict__ node_off, int64_t *__restrict__ edge_off) {
  __shared__ int64_t lds[kBlock / kWave];
  const int64_t *__restrict__ in = blockIdx.y == 0 ? num_nodes : num_edges;
  int64_t *__restrict__ out = blockIdx.y == 0 ? node_off : edge_off;
  const int64_t base = (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems];
  int64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < B) ? in[base + k] : 0;
    sum += v[k];
  }
  int64_t total;
  int64_t pre = block_exclusive_scan<int64_t>(sum, lds, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < B) out[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) out[B] = total;
}
__global__ __launch_bounds__(kBlock) void collate_both(const int64_t *__restrict__ ls, const int64_t *__restrict__ ld,
                                                       const int64_t *__restrict__ node_off, const int64_t *__restrict__ edge_off,
                                                       int64_t B, int64_t E, int64_t N, int64_t *__restrict__ src,
                                                       int64_t *__restrict__ dst, int32_t *__restrict__ edge_graph,
                                                       int32_t *__restrict__ node_graph) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.y == 0) {
    if (i >= E) return;
    const int64_t g = upper_graph(edge_off, B, i);
    const int64_t o = node_off[g];
    src[i] = ls[i] + o;
    dst[i] = ld[i] + o;
    if (edge_graph) edge_graph[i] = (int32_t)g;
  } else {
    if (i >= N || !node_graph) return;
    node_graph[i] = (int32_t)upper_graph(node_off, B, i);
  }
}

// several batches (the pattern batch and the target batch of a step) in the same two launches: blockIdx.y = 2 job + part
struct CollateJobs { dmp_collate_job job[DMP_COLLATE_MAX_JOBS]; };
__global__ __launch_bounds__(kBlock) void collate_offsets_jobs(const CollateJobs t) {
  __shared__ int64_t lds[kBlock / kWave];
  const dmp_collate_job &j = t.job[blockIdx.y >> 1];
  const int64_t *__restrict__ in = (blockIdx.y & 1) == 0 ? j.num_nodes : j.num_edges;
  int64_t *__restrict__ out = (blockIdx.y & 1) == 0 ? j.node_off : j.edge_off;
  const int64_t base = (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems];
  int64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < j.B) ? in[base + k] : 0;
    sum += v[k];
  }
  int64_t total;
  int64_t pre = block_exclusive_scan<int64_t>(sum, lds, total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < j.B) out[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 0) out[j.B] = total;
}
__global__ __launch_bounds__(kBlock) void collate_both_jobs(const CollateJobs t) {
  const dmp_collate_job &j = t.job[blockIdx.y >> 1];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((blockIdx.y & 1) == 0) {
    if (i >= j.E) return;
    const int64_t g = upper_graph(j.edge_off, j.B, i);
    const int64_t o = j.node_off[g];
    j.src[i] = j.local_src[i] + o;
    j.dst[i] = j.local_dst[i] + o;
    if (j.edge_graph) j.edge_graph[i] = (int32_t)g;
  } else {
    if (i >= j.N || !j.node_graph) return;
    j.node_graph[i] = (int32_t)upper_graph(j.node_off, j.B, i);
  }
}

__global__ __launch_bounds__(kBlock) void add_rev_k(
    const int64_t *__restrict__ src, const int64_t *__restrict__ dst, const int64_t *__restrict__ eid,
    const int64_t *__restrict__ el, const int64_t *__restrict__ edge_off, int64_t B, int64_t E,
    int64_t max_ne, int64_t max_nel, int64_t *__restrict__ o_src, int64_t *__restrict__ o_dst,
    int64_t *__restrict__ o_eid, int64_t *__restrict__ o_el, uint8_t *__restrict__ o_rev) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t g = upper_graph(edge_off, B, e);
  const int64_t beg = edge_off[g], n = edge_off[g + 1] - beg, k = e - beg;
  const int64_t f = 2 * beg + k, r = 2 * beg + n + k;
  const int64_t u = src[e], v = dst[e];
  o_src[f] = u; o_dst[f] = v; o_eid[f] = eid[e]; o_el[f] = el[e]; o_rev[f] = 0;
  // train.py:307-318: add_edges(v, u, id = max_nge + arange(num_ge), label + max_ngel, rev = 1)
  o_src[r] = v; o_dst[r] = u; o_eid[r] = max_ne + k; o_el[r] = el[e] + max_nel; o_rev[r] = 1;
}

// ------------------------------------------------------------------ line graph
__global__ __launch_bounds__(kBlock) void lg_count(const int32_t *__restrict__ in_ptr,
                                                   const int64_t *__restrict__ src, int64_t E,
                                                   int64_t *__restrict__ cnt) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t s = src[e];
  cnt[e] = in_ptr[s + 1] - in_ptr[s];
}
// one wave per primal edge e: its in-degree(src e) dual edges are written in
// ascending eid of the in-edge = the reference's emission order (graph.py:130-133)
__global__ __launch_bounds__(kBlock) void lg_fill(const int32_t *__restrict__ in_ptr,
                                                  const int32_t *__restrict__ in_ent,
                                                  const int64_t *__restrict__ src,
                                                  const int64_t *__restrict__ off, int64_t E,
                                                  int64_t *__restrict__ dsrc, int64_t *__restrict__ ddst,
                                                  int64_t *__restrict__ payload) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const int64_t s = src[e];
  const int beg = in_ptr[s], n = in_ptr[s + 1] - beg;
  const int64_t o = off[e];
  for (int k = threadIdx.x % kWave; k < n; k += kWave) {
    dsrc[o + k] = in_ent[beg + k] >> 1;
    ddst[o + k] = e;
    payload[o + k] = s;
  }
}

__global__ __launch_bounds__(kBlock) void fill_i64(int64_t *__restrict__ a, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ __launch_bounds__(kBlock) void first_of_id_k(const int64_t *__restrict__ eid, int64_t E,
                                                        int64_t K, int64_t *__restrict__ first) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t k = eid[e];
  if (k >= 0 && k < K)
    atomicMin(reinterpret_cast<unsigned long long *>(&first[k]), (unsigned long long)e);
}
__global__ __launch_bounds__(kBlock) void first_fix_k(int64_t *__restrict__ first, int64_t K) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < K && first[k] == INT64_MAX) first[k] = -1;
}

// Open-addressing table of item indices; a slot's identity is the key of the
// item it holds, and only same-key items ever replace each other (atomicMin),
// so the winner of every key is its lowest item index regardless of timing.
constexpr unsigned long long kEmpty = ~0ull;
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__device__ __forceinline__ uint64_t key_hash(int64_t a, int64_t l, int64_t b) {
  return mix64((uint64_t)a * 0x9E3779B97F4A7C15ULL ^ mix64((uint64_t)b + 0x7F4A7C15ULL) ^
               mix64((uint64_t)l * 0xD6E8FEB86659FD93ULL));
}
__global__ __launch_bounds__(kBlock) void dedupe_insert(const int64_t *__restrict__ ka,
                                                        const int64_t *__restrict__ kl,
                                                        const int64_t *__restrict__ kb, int64_t M,
                                                        unsigned long long *__restrict__ table,
                                                        uint64_t mask) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  const int64_t a = ka[m], l = kl[m], b = kb[m];
  uint64_t h = key_hash(a, l, b) & mask;
  while (true) {
    unsigned long long s = atomicCAS(&table[h], kEmpty, (unsigned long long)m);
    if (s == kEmpty) return;
    if (ka[s] == a && kl[s] == l && kb[s] == b) {
      atomicMin(&table[h], (unsigned long long)m);
      return;
    }
    h = (h + 1) & mask;
  }
}
__global__ __launch_bounds__(kBlock) void dedupe_lookup(const int64_t *__restrict__ ka,
                                                        const int64_t *__restrict__ kl,
                                                        const int64_t *__restrict__ kb, int64_t M,
                                                        const unsigned long long *__restrict__ table,
                                                        uint64_t mask, uint8_t *__restrict__ keep) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  const int64_t a = ka[m], l = kl[m], b = kb[m];
  uint64_t h = key_hash(a, l, b) & mask;
  while (true) {
    const unsigned long long s = table[h];
    if (s == kEmpty) { keep[m] = 1; return; }  // unreachable after insert; fail open
    if (ka[s] == a && kl[s] == l && kb[s] == b) { keep[m] = (s == (unsigned long long)m); return; }
    h = (h + 1) & mask;
  }
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
inline int64_t scan_tiles_for(int64_t n) { return (n + kScanTile - 1) / kScanTile + 1; }
inline uint64_t table_size(int64_t M) {
  uint64_t c = 16;
  while (c < (uint64_t)M * 2) c <<= 1;
  return c;
}

// ---------------------------------------------------------------------------------------------
// Subisomorphism weights of GraphAdjDataset.batchify (dataset.py:54-107,1491-1520,1618-1634),
// whole batch at once.  sub: the batch's subisomorphism rows back to back (row = target node per
// pattern node, graph-local); sample_ptr[i] = first element of sample i.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t upper_slot(const int64_t *ptr, int64_t n, int64_t x) {
  // largest i in [0, n) with ptr[i] <= x   (ptr non-decreasing, ptr[0] <= x < ptr[n])
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (ptr[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void subiso_node_k(const int64_t *sub, int64_t T, const int64_t *sample_ptr, int64_t B,
                              const int64_t *g_node_off, unsigned long long *out, int32_t *status) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = upper_slot(sample_ptr, B, t);
    const int64_t v = sub[t], n = g_node_off[i + 1] - g_node_off[i];
    if (v < 0 || v >= n) { if (status) atomicOr(status, 1); continue; }
    atomicAdd(out + g_node_off[i] + v, 1ULL);
  }
}

// compute_edgeseq_subisoweights keeps one label list per (u, v) key, filled run by run of equal
// consecutive keys in edge-id order: a later run of the same key replaces the earlier one
// (dataset.py:79-88).  active[j] = 1 iff edge j belongs to the LAST run of its key.
__global__ void pattern_edge_active_k(const int64_t *p_src, const int64_t *p_dst, const int64_t *p_edge_off,
                                      const int32_t *p_edge_graph, int64_t PE, uint8_t *active) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= PE) return;
  const int64_t end = p_edge_off[p_edge_graph[j] + 1];
  const int64_t u = p_src[j], v = p_dst[j];
  bool left_run = false, alive = true;
  for (int64_t k = j + 1; k < end; ++k) {
    const bool same = p_src[k] == u && p_dst[k] == v;
    if (!same) left_run = true;
    else if (left_run) { alive = false; break; }
  }
  active[j] = alive ? 1 : 0;
}

__global__ void subiso_edge_k(const int64_t *sub, const int64_t *sample_ptr, const int64_t *work_ptr, int64_t B,
                              const int64_t *p_node_off, const int64_t *p_edge_off, const int64_t *p_src,
                              const int64_t *p_dst, const int64_t *p_label, const uint8_t *active,
                              const int64_t *g_node_off, const int32_t *out_ptr, const int32_t *out_ent,
                              const int32_t *g_dst, const int64_t *g_label, unsigned long long *out,
                              int32_t *status) {
  const int64_t W = work_ptr[B];
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < W; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = upper_slot(work_ptr, B, w);
    const int64_t pe = p_edge_off[i + 1] - p_edge_off[i], pn = p_node_off[i + 1] - p_node_off[i];
    const int64_t local = w - work_ptr[i];
    const int64_t r = local / pe, j = p_edge_off[i] + local % pe;
    if (!active[j]) continue;
    const int64_t *row = sub + sample_ptr[i] + r * pn;
    const int64_t gn = g_node_off[i + 1] - g_node_off[i];
    const int64_t mu = row[p_src[j] - p_node_off[i]], mv = row[p_dst[j] - p_node_off[i]];
    if (mu < 0 || mu >= gn || mv < 0 || mv >= gn) { if (status) atomicOr(status, 1); continue; }
    const int64_t u = g_node_off[i] + mu;
    const int32_t v = (int32_t)(g_node_off[i] + mv);
    const int64_t l = p_label[j];
    for (int32_t q = out_ptr[u]; q < out_ptr[u + 1]; ++q) {
      const int32_t e = out_ent[q] >> 1;
      if (g_dst[e] == v && g_label[e] == l) atomicAdd(out + e, 1ULL);
    }
  }
}

// UNC mini-batch samplers (UnsupervisedNodeClassification/Model/DMPNN/src/utils.py:279-349; DGL's random_walk /
// sample_neighbors there).  Random choices come from a counter-based generator -- a 32-bit mix of (seed, a, b) -- so a
// launch is reproducible from its seed and has a plain restatement (oracle/graph_oracle.py::rng_hash).
__host__ __device__ __forceinline__ uint32_t rng_hash(uint32_t seed, uint32_t a, uint32_t b) {
  uint32_t x = seed ^ (a * 0x9E3779B1u) ^ (((b * 0x85EBCA77u) << 15) | ((b * 0x85EBCA77u) >> 17));
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// walk w of seed s: trace[0] = the seed; step t follows one of the current node's out-edges, the k-th of its CSR row
// with k = (hash * out_degree) >> 32; a node without out-edges ends the walk (-1 from there on, as dgl.sampling.random_walk).
__global__ void random_walks_k(const int32_t *out_ptr, const int32_t *out_ent, const int32_t *dst, const int64_t *seeds,
                               int64_t S, int walks, int depth, uint32_t seed, int64_t *traces, uint8_t *visited) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= S * walks) return;
  int64_t cur = seeds[t / walks];
  int64_t *tr = traces ? traces + t * (depth + 1) : nullptr;
  if (tr) tr[0] = cur;
  if (visited) visited[cur] = 1;
  for (int step = 1; step <= depth; ++step) {
    if (cur >= 0) {
      const int32_t lo = out_ptr[cur], deg = out_ptr[cur + 1] - lo;
      if (deg == 0) {
        cur = -1;
      } else {
        const uint32_t h = rng_hash(seed, (uint32_t)t, (uint32_t)step);
        cur = dst[out_ent[lo + (int32_t)(((uint64_t)h * (uint64_t)deg) >> 32)] >> 1];
        if (visited) visited[cur] = 1;
      }
    }
    if (tr) tr[step] = cur;
  }
}

// sample_neighbors(graph, nodes, width, edge_dir="in"): every wanted node keeps all of its in-edges if it has at most
// `width`, else the `width` in-edges with the smallest keys hash(seed, edge id, 0) (ties: the smaller edge id) -- a uniform
// choice without replacement.  One thread per node, an insertion list of the `width` best keys.
constexpr int kMaxSampleWidth = 64;
__global__ void sample_in_edges_k(const int32_t *in_ptr, const int32_t *in_ent, const uint8_t *wanted, int64_t N, int width,
                                  uint32_t seed, uint8_t *mask) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= N || (wanted && !wanted[v])) return;
  const int32_t lo = in_ptr[v], hi = in_ptr[v + 1];
  if (hi - lo <= width) {
    for (int32_t q = lo; q < hi; ++q) mask[in_ent[q] >> 1] = 1;
    return;
  }
  uint64_t best[kMaxSampleWidth];                              // (key << 32) | edge id, ascending
  int n = 0;
  for (int32_t q = lo; q < hi; ++q) {
    const uint32_t e = (uint32_t)(in_ent[q] >> 1);
    const uint64_t k = ((uint64_t)rng_hash(seed, e, 0u) << 32) | e;
    if (n == width && k >= best[n - 1]) continue;
    int i = n < width ? n++ : n - 1;
    while (i > 0 && best[i - 1] > k) { best[i] = best[i - 1]; --i; }
    best[i] = k;
  }
  for (int i = 0; i < n; ++i) mask[(uint32_t)best[i]] = 1;
}

// The same choice for ANY width (the reference's default sample width is 128, main.py:294): a wave per node, no list.
// The width-th smallest key t of the row is found by bisection on the 32-bit key (33 counting passes over the row,
// lanes strided); every edge with key < t is kept, and of the edges with key == t the first `need` in row order (rows
// list edges in ascending edge id, so that is the tie rule "smaller edge id").
__global__ __launch_bounds__(kBlock) void sample_in_edges_wave_k(const int32_t *in_ptr, const int32_t *in_ent, const uint8_t *wanted,
                                                               int64_t N, int width, uint32_t seed, uint8_t *mask) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (v >= N || (wanted && !wanted[v])) return;                // wave-uniform
  const int32_t lo = in_ptr[v], hi = in_ptr[v + 1];
  if (hi - lo <= width) {
    for (int32_t q = lo + lane; q < hi; q += 64) mask[in_ent[q] >> 1] = 1;
    return;
  }
  auto count_le = [&](uint32_t t) {
    int c = 0;
    for (int32_t q = lo + lane; q < hi; q += 64) c += rng_hash(seed, (uint32_t)(in_ent[q] >> 1), 0u) <= t;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    return c;
  };
  uint32_t a = 0u, b = 0xFFFFFFFFu;                             // smallest t with count(key <= t) >= width
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (count_le(mid) >= width) b = mid; else a = mid + 1;
  }
  const uint32_t t = a;
  int need = width - (t ? count_le(t - 1) : 0);                // ties to take, >= 1
  for (int32_t q0 = lo; q0 < hi; q0 += 64) {
    const int32_t q = q0 + lane;
    uint32_t e = 0, k = 0xFFFFFFFFu;
    const bool in = q < hi;
    if (in) { e = (uint32_t)(in_ent[q] >> 1); k = rng_hash(seed, e, 0u); }
    const bool tie = in && k == t;
    const unsigned long long ties = __ballot(tie);
    const int before = __popcll(ties & ((1ull << lane) - 1ull));
    if (in && (k < t || (tie && before < need))) mask[e] = 1;
    need -= __popcll(ties);
    if (need < 0) need = 0;
  }
}

// Pooling index (ops.PoolIndex): the rows of every graph of a batch (contiguous ranges of `sizes[i]` rows) cut into chunks
// of `chunk` rows, so that the per-graph sums of the prediction heads run as two launches of the segment-sum kernel
// (rows -> chunk sums -> graph sums) with enough independent rows to fill the chip.  One single-workgroup scan for the
// two offset vectors, one fill launch for everything else -- instead of ~20 small tensor launches per index and batch.
//   off  [B+1] int64  first row of graph i              coff = gptr [B+1] int32  first chunk of graph i
//   vptr [V+1] int32  first row of chunk v (V = R / chunk + B bounds the chunk count; unused tail chunks are empty)
//   vent [R]   int32  (row << 1) | flag[row]            gent [V] int32  chunk << 1            seg [R] int32  graph of row
// `sizes` / `flag` come as two pieces each (pattern graphs then target graphs of a union pass) to spare the concatenations.
struct PoolArgs {
  const int64_t *sizes_a, *sizes_b; int64_t Ba, Bb;
  const uint8_t *flag_a, *flag_b; int64_t Ra;                // rows [0, Ra) take flag_a, the rest flag_b (NULL: 0)
  int64_t R, V; int chunk;
  int64_t *off; int32_t *gptr, *vptr, *vent, *gent, *seg;
  uint8_t *flag8; int32_t *rowmap; int64_t *sizes;           // optional: both pieces' flags / sizes back to back; seg or -1 if flagged
};
struct PoolJobs { PoolArgs job[DMP_POOL_MAX_JOBS]; };

// blockIdx.x = job.  A thread owns kPoolItems CONSECUTIVE graphs of a slice of kBlock * kPoolItems (all of a slice's sizes are in
// flight at once: one memory round trip per 2048 graphs instead of one per 256 plus sixteen barriers), scans them in registers,
// and the threads' totals are scanned wave by wave (shuffles) and across the waves in LDS.
constexpr int kPoolItems = 8;
__global__ __launch_bounds__(kBlock) void pool_offsets_k(const PoolJobs jobs) {
  const PoolArgs &a = jobs.job[blockIdx.x];
  constexpr int kWaves = kBlock / 64;
  __shared__ int64_t w_rows[kWaves], w_chunks[kWaves];
  const int64_t B = a.Ba + a.Bb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t base_rows = 0, base_chunks = 0;                     // (the same in every thread)
  if (threadIdx.x == 0) { a.off[0] = 0; a.gptr[0] = 0; }
  for (int64_t i0 = 0; i0 < B; i0 += (int64_t)kBlock * kPoolItems) {
    const int64_t first = i0 + (int64_t)threadIdx.x * kPoolItems;
    int64_t n[kPoolItems], r[kPoolItems], c[kPoolItems];
#pragma unroll
    for (int u = 0; u < kPoolItems; ++u) {
      const int64_t i = first + u;
      n[u] = i < B ? (i < a.Ba ? a.sizes_a[i] : a.sizes_b[i - a.Ba]) : 0;
    }
    int64_t tr = 0, tc = 0;
#pragma unroll
    for (int u = 0; u < kPoolItems; ++u) {
      if (first + u < B && a.sizes) a.sizes[first + u] = n[u];
      tr += n[u]; tc += (n[u] + a.chunk - 1) / a.chunk;
      r[u] = tr; c[u] = tc;                                   // inclusive inside the thread
    }
    int64_t xr = tr, xc = tc;                                 // inclusive scan of the threads' totals inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t yr = __shfl_up(xr, off), yc = __shfl_up(xc, off);
      if (lane >= off) { xr += yr; xc += yc; }
    }
    __syncthreads();                                          // (the previous slice's totals have been read)
    if (lane == 63) { w_rows[wave] = xr; w_chunks[wave] = xc; }
    __syncthreads();
    int64_t pr = base_rows, pc = base_chunks, sr = 0, sc = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      if (w < wave) { pr += w_rows[w]; pc += w_chunks[w]; }
      sr += w_rows[w]; sc += w_chunks[w];
    }
    pr += xr - tr; pc += xc - tc;                             // everything before this thread's first graph
#pragma unroll
    for (int u = 0; u < kPoolItems; ++u)
      if (first + u < B) {
        a.off[first + u + 1] = pr + r[u];
        a.gptr[first + u + 1] = (int32_t)(pc + c[u]);
      }
    base_rows += sr; base_chunks += sc;
  }
}

// blockIdx.y = job
__global__ void pool_fill_k(const PoolJobs jobs) {
  const PoolArgs &a = jobs.job[blockIdx.y];
  const int64_t B = a.Ba + a.Bb;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < a.R) {
    const uint8_t *f = t < a.Ra ? a.flag_a : a.flag_b;
    const int64_t fi = t < a.Ra ? t : t - a.Ra;
    const int fl = f ? (f[fi] != 0) : 0;
    a.vent[t] = (int32_t)((t << 1) | fl);
    if (a.flag8) a.flag8[t] = (uint8_t)fl;
    if (a.seg || a.rowmap) {
      const int32_t g = (int32_t)upper_slot(a.off, B, t);     // off[i] <= t < off[i+1] (empty graphs are skipped)
      if (a.seg) a.seg[t] = g;
      if (a.rowmap) a.rowmap[t] = fl ? -1 : g;
    }
  }
  if (t <= a.V) {
    int64_t first = a.R;                                      // unused tail chunks (and the closing entry): empty at the end
    if (t < a.V && t < a.gptr[B]) {
      int64_t lo = 0, hi = B;                                 // graph of chunk t: largest i with gptr[i] <= t
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.gptr[mid] <= t) lo = mid; else hi = mid;
      }
      first = a.off[lo] + (t - a.gptr[lo]) * a.chunk;
    }
    a.vptr[t] = (int32_t)first;
    if (t < a.V) a.gent[t] = (int32_t)(t << 1);
  }
}

// sum of a row weight per graph (and flag half): a wave per graph
__global__ __launch_bounds__(kBlock) void pool_weight_sums_k(const float *__restrict__ w, const uint8_t *__restrict__ flag8,
                                                             const int64_t *__restrict__ off, int64_t B, int halves,
                                                             float *__restrict__ out) {
  const int64_t g = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  if (g >= B) return;
  float s0 = 0.f, s1 = 0.f;
  for (int64_t r = off[g] + lane; r < off[g + 1]; r += kWave) {
    const float v = w ? w[r] : 1.f;
    if (halves == 2 && flag8[r]) s1 += v; else s0 += v;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, kWave); s1 += __shfl_xor(s1, o, kWave); }
  if (lane == 0) {
    out[g * halves] = s0;
    if (halves == 2) out[g * halves + 1] = s1;
  }
}

// get_dual_subisomorphisms (utils/graph.py:277-316 as convert_to_dual_data drives it, train.py:417-446): a node map
// of a sample (one row of its `subisomorphisms`) -> for every KEY of the pattern the id of the graph edge it lands on.
// Keys: maximal runs of consecutive pattern edges (edge-id order) with equal (src, dst); a later run of a key that
// exists already REPLACES that key's label list but keeps its position (a Python dict, graph.py:293-300).  Entry k of a
// row: among the graph edges mapped-src -> mapped-dst whose label occurs in key k's list, the LAST one in the
// (src, dst)-sorted order with ties in edge-id order -- i.e. the one with the largest edge id.  Rows are p_len wide;
// the entries past the number of keys, and keys without a matching edge, keep the reference's initial 0, which its
// caller then sends through g_eid[0]: the id of the FIRST edge of the sorted order.
// One thread per sample builds the key table; one thread per output entry does the lookup in the CSR by source.
__global__ void dual_keys_k(const int64_t *p_src, const int64_t *p_dst, const int64_t *p_edge_off, int64_t B,
                            const int64_t *g_node_off, const int64_t *g_edge_off, const int32_t *out_ptr,
                            const int32_t *out_ent, const int32_t *g_dst, int32_t *key_lo, int32_t *key_hi,
                            int32_t *num_keys, int64_t *first_sorted) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int64_t lo = p_edge_off[i], hi = p_edge_off[i + 1];
  int nk = 0;
  for (int64_t a = lo; a < hi;) {
    int64_t b = a + 1;
    while (b < hi && p_src[b] == p_src[a] && p_dst[b] == p_dst[a]) ++b;
    int slot = -1;                                            // an earlier key with the same endpoints?
    for (int k = 0; k < nk; ++k) {
      const int64_t f = lo + key_lo[lo + k];
      if (p_src[f] == p_src[a] && p_dst[f] == p_dst[a]) { slot = k; break; }
    }
    if (slot < 0) slot = nk++;
    key_lo[lo + slot] = (int32_t)(a - lo);                   // the label list of the key: pattern edges [a, b)
    key_hi[lo + slot] = (int32_t)(b - lo);
    a = b;
  }
  num_keys[i] = nk;
  // first edge of the graph in (src, dst, edge id) order, as a LOCAL edge id
  int64_t first = 0;
  for (int64_t n = g_node_off[i]; n < g_node_off[i + 1]; ++n) {
    if (out_ptr[n + 1] > out_ptr[n]) {
      int32_t best = out_ent[out_ptr[n]] >> 1;               // rows list ascending edge ids: the first minimum wins
      for (int32_t q = out_ptr[n] + 1; q < out_ptr[n + 1]; ++q) {
        const int32_t e = out_ent[q] >> 1;
        if (g_dst[e] < g_dst[best]) best = e;
      }
      first = best - g_edge_off[i];
      break;
    }
  }
  first_sorted[i] = first;
}

__global__ void dual_match_k(const int64_t *sub, const int64_t *sample_ptr, const int64_t *work_ptr, int64_t B,
                             const int64_t *p_node_off, const int64_t *p_edge_off, const int64_t *p_src,
                             const int64_t *p_dst, const int64_t *p_label, const int32_t *key_lo, const int32_t *key_hi,
                             const int32_t *num_keys, const int64_t *first_sorted, const int64_t *g_node_off,
                             const int64_t *g_edge_off, const int32_t *out_ptr, const int32_t *out_ent,
                             const int32_t *g_dst, const int64_t *g_label, int64_t *out, int32_t *status) {
  const int64_t W = work_ptr[B];
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < W; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = upper_slot(work_ptr, B, w);
    const int64_t pe = p_edge_off[i + 1] - p_edge_off[i], pn = p_node_off[i + 1] - p_node_off[i];
    const int64_t local = w - work_ptr[i];
    const int64_t r = local / pe, k = local % pe;
    int64_t res = first_sorted[i];
    if (k < num_keys[i]) {
      const int64_t e0 = p_edge_off[i];
      const int64_t a = e0 + key_lo[e0 + k], b = e0 + key_hi[e0 + k];
      const int64_t *row = sub + sample_ptr[i] + r * pn;
      const int64_t gn = g_node_off[i + 1] - g_node_off[i];
      const int64_t mu = row[p_src[a] - p_node_off[i]], mv = row[p_dst[a] - p_node_off[i]];
      if (mu < 0 || mu >= gn || mv < 0 || mv >= gn) {
        if (status) atomicOr(status, 1);
      } else {
        const int64_t u = g_node_off[i] + mu;
        const int32_t v = (int32_t)(g_node_off[i] + mv);
        for (int32_t q = out_ptr[u]; q < out_ptr[u + 1]; ++q) {   // ascending edge ids: the last hit is the largest
          const int32_t e = out_ent[q] >> 1;
          if (g_dst[e] != v) continue;
          const int64_t l = g_label[e];
          for (int64_t j = a; j < b; ++j)
            if (p_label[j] == l) { res = e - g_edge_off[i]; break; }
        }
      }
    }
    out[w] = res;
  }
}

// ---------------------------------------------------------------------------------------------
// Degree-class tile list of the typed edge kernels (csrc/dmp_typed.hip), built on the device in five
// launches without atomics deciding any order: slot order = class ascending, nodes of a class in
// ascending id, the in-edges of a node in ascending edge id (= its in-CSR row).
//   deg[v]   : the integer degree the layer's coefficient is computed from (class key), clamped to C-1
//   cnt[c]   : in-edges whose destination has degree c
//   tile_off : exclusive prefix of ceil(cnt[c] / 32) (tiles never mix classes)
// ---------------------------------------------------------------------------------------------
constexpr int kCtLdsBins = 4096;

// in-edges of node v that count: all of its in-CSR row, or (row_cnt: the gated build) those a 0 / 1 edge gate keeps
__device__ __forceinline__ int ct_rows(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ row_cnt, int64_t v) {
  return row_cnt ? row_cnt[v] : in_ptr[v + 1] - in_ptr[v];
}

__global__ __launch_bounds__(kBlock) void ct_live_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                    const float *__restrict__ gate, int64_t N, int32_t *__restrict__ row_cnt) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v >= N) return;
  int c = 0;
  for (int q = in_ptr[v], hi = in_ptr[v + 1]; q < hi; ++q) c += gate[in_ent[q] >> 1] != 0.f;
  row_cnt[v] = c;
}

// ---- the in-CSR of the edges a 0 / 1 gate keeps (dmp_csr_keep): (1) every node counts the kept entries of its row, every
// block adds its nodes' counts up; (2) every block sums the block totals before it, ranks its nodes with a block scan, and
// every node copies the kept entries of its row in order.  No atomics: the layout is a function of the inputs alone.
__global__ __launch_bounds__(kBlock) void keep_count_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                       const float *__restrict__ gate, int64_t N, int32_t *__restrict__ row_cnt,
                                                       int32_t *__restrict__ blk) {
  __shared__ int red[kBlock / 64];
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int c = 0;
  if (v < N) {
    for (int q = in_ptr[v], hi = in_ptr[v + 1]; q < hi; ++q) c += gate[in_ent[q] >> 1] != 0.f;
    row_cnt[v] = c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; blk[blockIdx.x] = t; }
}
__global__ __launch_bounds__(kBlock) void keep_fill_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                      const float *__restrict__ gate, const int32_t *__restrict__ row_cnt,
                                                      const int32_t *__restrict__ blk, int64_t N, int32_t *__restrict__ keep_ptr,
                                                      int32_t *__restrict__ keep_ent) {
  __shared__ int red[kBlock / 64], wtot[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) before += blk[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) red[wave] = before;
  __syncthreads();
  before = 0;
  for (int i = 0; i < kBlock / 64; ++i) before += red[i];
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int c = v < N ? row_cnt[v] : 0;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int s = before + incl - c;
  for (int w = 0; w < wave; ++w) s += wtot[w];
  if (v < N) {
    keep_ptr[v] = s;
    for (int q = in_ptr[v], hi = in_ptr[v + 1]; q < hi; ++q) {
      const int e = in_ent[q];
      if (gate[e >> 1] != 0.f) keep_ent[s++] = e;
    }
    if (v == N - 1) keep_ptr[N] = s;
  }
}

// ... the same two passes with a GROUP of G lanes per row, for CSRs whose rows are long (a pooling index's chunk table: ~50
// entries per row -- a thread per row walks them one dependent, uncoalesced load at a time: 15 + 30-45 us at bench.py's shape):
// the group strides through its row G entries at a time, counts with a group reduction, and places the kept entries with a
// ballot + prefix popcount.  Same arrays, same bits as the thread-per-row kernels.
template <int G>
__global__ __launch_bounds__(kBlock) void keep_count_g_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                         const float *__restrict__ gate, int64_t N, int32_t *__restrict__ row_cnt,
                                                         int32_t *__restrict__ blk) {
  constexpr int RPB = kBlock / G;
  __shared__ int red[kBlock / 64];
  const int64_t v = (int64_t)blockIdx.x * RPB + threadIdx.x / G;
  const int gl = threadIdx.x % G;
  int c = 0;
  if (v < N)
    for (int q = in_ptr[v] + gl, hi = in_ptr[v + 1]; q < hi; q += G) c += gate[in_ent[q] >> 1] != 0.f;
  int rc = c;
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) rc += __shfl_xor(rc, off, G);
  if (v < N && gl == 0) row_cnt[v] = rc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < kBlock / 64; ++i) t += red[i]; blk[blockIdx.x] = t; }
}
template <int G>
__global__ __launch_bounds__(kBlock) void keep_fill_g_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                        const float *__restrict__ gate, const int32_t *__restrict__ row_cnt,
                                                        const int32_t *__restrict__ blk, int64_t N, int32_t *__restrict__ keep_ptr,
                                                        int32_t *__restrict__ keep_ent) {
  constexpr int RPB = kBlock / G;
  static_assert(RPB <= 64 && G <= 32, "one wave scans the block's rows; a group's ballot bits fit a word");
  __shared__ int red[kBlock / 64], rstart[RPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) before += blk[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) red[wave] = before;
  if (wave == 0) {                                               // exclusive scan of the block's row counts (RPB <= 64 rows: one wave)
    const int64_t vr = (int64_t)blockIdx.x * RPB + lane;
    const int c = (lane < RPB && vr < N) ? row_cnt[vr] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
    if (lane < RPB) rstart[lane] = incl - c;
  }
  __syncthreads();
  before = 0;
  for (int i = 0; i < kBlock / 64; ++i) before += red[i];
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  const int64_t v = (int64_t)blockIdx.x * RPB + g;
  if (v >= N) return;                                            // (whole groups: the ballots below see whole groups of their wave)
  int s = before + rstart[g];
  if (gl == 0) keep_ptr[v] = s;
  const int lo = in_ptr[v], hi = in_ptr[v + 1];
  const int shift = (lane / G) * G;                              // this group's bits of the wave's ballot
  for (int q0 = lo; q0 < hi; q0 += G) {
    const int q = q0 + gl;
    int e = 0;
    bool keep = false;
    if (q < hi) { e = in_ent[q]; keep = gate[e >> 1] != 0.f; }
    const unsigned bits = (unsigned)((__ballot(keep) >> shift) & ((1ull << G) - 1ull));
    if (keep) keep_ent[s + __popc(bits & ((1u << gl) - 1u))] = e;
    s += __popc(bits);
  }
  if (v == N - 1 && gl == 0) keep_ptr[N] = s;
}

// ---- the INCIDENCE CSR over the kept edges, for the kept nodes only (dmp_incidence_keep): row i belongs to node list[i]
// (i < *count: the nodes a 0 / 1 node gate keeps, ascending) and lists the node's in-entries and out-entries (flag flipped)
// whose edge a 0 / 1 edge gate keeps, MERGED by ascending edge id as incidence_fill merges them.  Same two passes as the
// kept in-CSR above: (1) kept entries per row, block totals; (2) block offsets, ranks, the merge.  keep_ptr is indexed by the
// POSITION i, not by the node.
__global__ __launch_bounds__(kBlock) void inc_keep_count_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                           const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_ent,
                                                           const float *__restrict__ gate, const int32_t *__restrict__ list,
                                                           const int32_t *__restrict__ count, int32_t *__restrict__ row_cnt,
                                                           int32_t *__restrict__ blk) {
  __shared__ int red[kBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t n = *count;
  int c = 0;
  if (i < n) {
    const int v = list[i];
    for (int q = in_ptr[v], hi = in_ptr[v + 1]; q < hi; ++q) c += gate[in_ent[q] >> 1] != 0.f;
    if (out_ptr)          // (NULL: the in-entries alone -- the kept edges' CSR by destination with a row per list position)
      for (int q = out_ptr[v], hi = out_ptr[v + 1]; q < hi; ++q) c += gate[out_ent[q] >> 1] != 0.f;
    row_cnt[i] = c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kBlock / 64; ++k) t += red[k]; blk[blockIdx.x] = t; }
}
__global__ __launch_bounds__(kBlock) void inc_keep_fill_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                          const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_ent,
                                                          const float *__restrict__ gate, const int32_t *__restrict__ list,
                                                          const int32_t *__restrict__ count, const int32_t *__restrict__ row_cnt,
                                                          const int32_t *__restrict__ blk, int32_t *__restrict__ keep_ptr,
                                                          int32_t *__restrict__ keep_ent) {
  __shared__ int red[kBlock / 64], wtot[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) before += blk[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) red[wave] = before;
  __syncthreads();
  before = 0;
  for (int k = 0; k < kBlock / 64; ++k) before += red[k];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t n = *count;
  const int c = i < n ? row_cnt[i] : 0;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int s = before + incl - c;
  for (int w = 0; w < wave; ++w) s += wtot[w];
  if (i < n) {
    keep_ptr[i] = s;
    const int v = list[i];
    int a = in_ptr[v], b = out_ptr ? out_ptr[v] : 0;
    const int a1 = in_ptr[v + 1], b1 = out_ptr ? out_ptr[v + 1] : 0;
    int x = a < a1 ? in_ent[a] : 0, y = b < b1 ? out_ent[b] : 0;
    while (a < a1 || b < b1) {                                  // incidence_fill's merge (an edge that is both lists its in-entry first)
      if (b >= b1 || (a < a1 && (x >> 1) <= (y >> 1))) {
        if (gate[x >> 1] != 0.f) keep_ent[s++] = x;
        if (++a < a1) x = in_ent[a];
      } else {
        if (gate[y >> 1] != 0.f) keep_ent[s++] = y ^ 1;
        if (++b < b1) y = out_ent[b];
      }
    }
    if (i == n - 1) keep_ptr[n] = s;
  }
  if (i == 0 && n == 0) keep_ptr[0] = 0;
}

__global__ __launch_bounds__(kBlock) void ct_hist_k(const int64_t *__restrict__ deg, const int32_t *__restrict__ in_ptr,
                                                    const int32_t *__restrict__ row_cnt,
                                                    int64_t N, int C, unsigned long long *cnt, int32_t *status) {
  __shared__ unsigned int h[kCtLdsBins];
  __shared__ int top;                                          // largest class in use in this block
  for (int i = threadIdx.x; i < kCtLdsBins; i += kBlock) h[i] = 0u;
  if (threadIdx.x == 0) top = 0;
  __syncthreads();
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v < N) {
    const unsigned int d = (unsigned int)ct_rows(in_ptr, row_cnt, v);
    int64_t k = deg[v];
    if (k < 0) k = 0;
    if (k >= C - 1) {
      if (k > C - 1 && d > 0 && status) atomicOr(status, 1);   // classes beyond the table share (and poison) the last one
      k = C - 1;
    }
    if (d > 0) {
      if (k < kCtLdsBins) atomicAdd(&h[k], d);                 // integer sums / maxima: the result does not depend on the order
      else atomicAdd(cnt + k, (unsigned long long)d);
      atomicMax(&top, (int)k);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCtLdsBins && i < C; i += kBlock)
    if (h[i]) atomicAdd(cnt + i, (unsigned long long)h[i]);
  if (threadIdx.x == 0 && top > 0 && status) atomicMax(status + 1, top);   // status[1]: largest class in use (the scan stops there)
}

// one block: tile_off[c] = sum_{c' < c} ceil(cnt[c'] / 32); tile_off[C] = tiles in use (also to num_tiles).
// Class c = j * 1024 + t sits in row j (<= 64 rows: C <= 65536), column t = thread: coalesced loads all in
// flight at once, a wave scan per row, then one scan over the (row, wave) totals in row-major order.
__global__ __launch_bounds__(1024) void ct_scan_k(const unsigned long long *__restrict__ cnt, int C, const int32_t *__restrict__ status,
                                                  int32_t *tile_off, int32_t *num_tiles) {
  constexpr int kRows = 64;
  __shared__ int part[kRows * 16];
  // classes above the largest one in use (status[1], from ct_hist_k) are empty and never looked up: only the rows
  // up to tile_off[top + 1] are scanned -- one or two of the 64 for the degrees of real graphs
  const int top = status[1] < C - 1 ? status[1] : C - 1;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, rows = ((top + 1) >> 10) + 1;
  int v[kRows];
#pragma unroll
  for (int j = 0; j < kRows; ++j) {
    const int c = j * 1024 + t;
    v[j] = (j < rows && c < C) ? (int)((cnt[c] + 31) >> 5) : 0;
  }
#pragma unroll
  for (int j = 0; j < kRows; ++j) {                            // inclusive scan of row j inside the wave
    if (j >= rows) { if (lane == 63) part[j * 16 + wave] = 0; continue; }
    int x = v[j];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    v[j] = x;
    if (lane == 63) part[j * 16 + wave] = x;
  }
  __syncthreads();
  const int mine = part[t];
  const int live = rows * 16;                                  // the (row, wave) totals past the scanned rows are zero
  for (int off = 1; off < live; off <<= 1) {                   // inclusive Hillis-Steele scan of the live totals
    const int add = (t >= off && t < live) ? part[t - off] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const int total = part[live - 1];
  __syncthreads();
  part[t] -= mine;                                             // exclusive
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRows; ++j) {
    const int c = j * 1024 + t;
    if (j < rows && c < C) {
      const int incl = v[j] + part[j * 16 + wave];
      tile_off[c] = incl - (int)((cnt[c] + 31) >> 5);
    }
  }
  if (t == 0) { tile_off[C] = total; *num_tiles = total; }
}

// node_base[v] = first slot of v's in-edges = 32 * tile_off[class] + in-edges of the class's nodes with a lower id.
// One wave per (class, node segment) walks its nodes in ascending id, 64 at a time, with a wave scan;
// the common classes (degree < kCtFast) split the node range into kCtSegs segments whose totals are
// taken in a first pass (ct_seg_k), rarer classes (hubs) walk the whole range with one wave.
// The class's wave of segment 0 also writes the class's coefficient into its tiles.
constexpr int kCtFast = 1024;
constexpr int kCtSegs = 128;

__device__ __forceinline__ int ct_key(const int64_t *deg, int64_t v, int C) {
  const int64_t k = deg[v];
  return (int)(k < 0 ? 0 : (k > C - 1 ? C - 1 : k));
}

__device__ __forceinline__ void ct_class_seg(int w, int C, int64_t N, int &c, int64_t &lo, int64_t &hi) {
  // waves [0, kCtFast * kCtSegs): (class, node segment) of the common classes
  c = w / kCtSegs;
  const int64_t per = ((N + kCtSegs - 1) / kCtSegs + 63) & ~(int64_t)63;
  lo = (int64_t)(w % kCtSegs) * per;
  hi = lo + per < N ? lo + per : N;
  if (lo > N) lo = N;
}

__global__ __launch_bounds__(kBlock) void ct_seg_k(const int64_t *__restrict__ deg, const int32_t *__restrict__ in_ptr,
                                                   const int32_t *__restrict__ row_cnt,
                                                   int64_t N, int C, const unsigned long long *__restrict__ cnt,
                                                   int32_t *segsum) {
  const int w = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  const int fast = C < kCtFast ? C : kCtFast;
  if (w >= fast * kCtSegs) return;
  int c;
  int64_t lo, hi;
  ct_class_seg(w, C, N, c, lo, hi);
  if (cnt[c] == 0) return;
  int sum = 0;
  for (int64_t v0 = lo + lane; v0 < hi; v0 += 256) {           // 4 independent chunks of 64 nodes in flight
    int k[4], d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = v0 + 64 * j;
      const bool ok = v < hi;
      k[j] = ok ? ct_key(deg, v, C) : -1;
      d[j] = ok ? ct_rows(in_ptr, row_cnt, v) : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += k[j] == c ? d[j] : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);   // integer: order-free
  if (lane == 0) segsum[w] = sum;
}

constexpr int kCtSlowWaves = 256;   // workers that share the rare classes >= kCtFast among themselves

__device__ __forceinline__ void ct_walk(const int64_t *__restrict__ deg, const int32_t *__restrict__ in_ptr,
                                        const int32_t *__restrict__ row_cnt, int C, int c,
                                        int64_t lo, int64_t hi, int running, int lane, int32_t *node_base) {
  auto fetch = [&](int64_t v, int &k, int &d) {
    k = -1; d = 0;
    if (v < hi) { k = ct_key(deg, v, C); d = ct_rows(in_ptr, row_cnt, v); }
  };
  int kn, dn;
  fetch(lo + lane, kn, dn);
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t v = base + lane;
    const bool m = kn == c;
    const int d = m ? dn : 0;
    fetch(v + 64, kn, dn);                                     // next 64 nodes: in flight under the scan
    int incl = d;                                              // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (m) node_base[v] = running + incl - d;
    running += __shfl(incl, 63, 64);
  }
}

__device__ __forceinline__ void ct_write_scale(int c, int C, const int32_t *status, int t0, int t1, int lane, float *tile_scale) {
  float scale = 2.0f * (1.0f + log2f(1.0f + (float)c));        // == degree_coef_k for this degree
  if (c == C - 1 && status && *status) scale = __builtin_nanf("");
  for (int t = t0 + lane; t < t1; t += 64) tile_scale[t] = scale;
}

__global__ __launch_bounds__(kBlock) void ct_base_k(const int64_t *__restrict__ deg, const int32_t *__restrict__ in_ptr,
                                                    const int32_t *__restrict__ row_cnt,
                                                    int64_t N, int C, const unsigned long long *__restrict__ cnt,
                                                    const int32_t *__restrict__ tile_off, const int32_t *__restrict__ status,
                                                    const int32_t *__restrict__ segsum, int32_t *node_base, float *tile_scale) {
  const int w = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  const int fast = C < kCtFast ? C : kCtFast;
  if (w < fast * kCtSegs) {
    int c;
    int64_t lo, hi;
    ct_class_seg(w, C, N, c, lo, hi);
    if (cnt[c] == 0) return;
    const int s = w % kCtSegs;
    int running = tile_off[c] * 32;
    for (int j = 0; j < s; ++j) running += segsum[w - s + j];
    if (s == 0) ct_write_scale(c, C, status, tile_off[c], tile_off[c + 1], lane, tile_scale);
    ct_walk(deg, in_ptr, row_cnt, C, c, lo, hi, running, lane, node_base);
  } else if (w < fast * kCtSegs + kCtSlowWaves) {
    // this worker's contiguous share of the rare classes; 64 of them are tested at a time (one per lane)
    const int per = (C - fast + kCtSlowWaves - 1) / kCtSlowWaves;
    const int c_lo = fast + (w - fast * kCtSegs) * per, c_hi = c_lo + per < C ? c_lo + per : C;
    for (int cb = c_lo; cb < c_hi; cb += 64) {
      const int cl = cb + lane;
      unsigned long long live = __ballot(cl < c_hi && cnt[cl] != 0);
      while (live) {
        const int c = cb + __builtin_ctzll(live);
        live &= live - 1;
        ct_write_scale(c, C, status, tile_off[c], tile_off[c + 1], lane, tile_scale);
        ct_walk(deg, in_ptr, row_cnt, C, c, 0, N, tile_off[c] * 32, lane, node_base);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void ct_fill_k(const int32_t *__restrict__ in_ptr, const int32_t *__restrict__ in_ent,
                                                    const int32_t *__restrict__ node_base, int64_t N, int32_t *slot_edge,
                                                    const float *__restrict__ gate, const int32_t *__restrict__ row_cnt) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v >= N) return;
  const int lo = in_ptr[v], hi = in_ptr[v + 1];
  if (hi == lo || (row_cnt && row_cnt[v] == 0)) return;        // (node_base is only written for nodes with rows that count)
  int s = node_base[v];
  for (int q = lo; q < hi; ++q) {
    const int e = in_ent[q] >> 1;
    if (!gate || gate[e] != 0.f) slot_edge[s++] = e;
  }
}

// ---- out = [a | b (+ add)] for several array pairs: the structure arrays of the union of two batches
struct ConcatJobs {
  dmp_concat_job job[DMP_CONCAT_MAX_JOBS];
};
__global__ __launch_bounds__(kBlock) void concat_pairs_k(const ConcatJobs t) {
  const dmp_concat_job &j = t.job[blockIdx.y];
  const int64_t n = j.na + j.nb;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const bool first = i < j.na;
    const int64_t k = first ? i : i - j.na;
    if (j.elem_size == 8) {
      const int64_t v = first ? static_cast<const int64_t *>(j.a)[k] : static_cast<const int64_t *>(j.b)[k] + j.add_b;
      static_cast<int64_t *>(j.out)[i] = v;
    } else if (j.elem_size == 4) {
      float v;
      if (first) v = j.a ? static_cast<const float *>(j.a)[k] : j.fill_a;
      else v = static_cast<const float *>(j.b)[k];
      static_cast<float *>(j.out)[i] = v;                   // bit copy for any 4-byte type (no arithmetic on it)
    } else {
      static_cast<uint8_t *>(j.out)[i] = first ? static_cast<const uint8_t *>(j.a)[k] : static_cast<const uint8_t *>(j.b)[k];
    }
  }
}

// ---- enc = table[ids] for several small tables (embed.py:199-224): one thread per output element
struct LookupJobs {
  dmp_lookup_job job[DMP_LOOKUP_MAX_JOBS];
};
__global__ __launch_bounds__(kBlock) void table_rows_k(const LookupJobs t) {
  const dmp_lookup_job &j = t.job[blockIdx.y];
  if (j.width <= 32) {
    // narrow rows (the multi-hot codes: 8-14 floats): a thread per ROW -- one index load, no 64-bit division per element, a wave
    // writes 64 consecutive rows; pairs of floats where the row starts allow it
    const bool pairs = !(j.width & 1) && !(j.ld & 1) && !(reinterpret_cast<uintptr_t>(j.table) & 7u) && !(reinterpret_cast<uintptr_t>(j.out) & 7u);
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < j.rows; r += (int64_t)gridDim.x * kBlock) {
      const int64_t id = j.idx[r];
      const bool ok = id >= 0 && id < j.table_rows;
      const float *src = j.table + (ok ? id : 0) * j.ld;
      float *dst = j.out + r * j.width;
      if (pairs) {
        for (int c = 0; c < j.width; c += 2) {
          float2 v = *reinterpret_cast<const float2 *>(src + c);
          if (!ok) v = make_float2(__builtin_nanf(""), __builtin_nanf(""));
          *reinterpret_cast<float2 *>(dst + c) = v;
        }
      } else {
        for (int c = 0; c < j.width; ++c) dst[c] = ok ? src[c] : __builtin_nanf("");
      }
    }
    return;
  }
  const int64_t total = j.rows * j.width;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / j.width;
    const int c = (int)(i - r * j.width);
    const int64_t id = j.idx[r];
    j.out[i] = (id >= 0 && id < j.table_rows) ? j.table[id * j.ld + c] : __builtin_nanf("");
  }
}

// ---- pre-padding masks + counts (utils/dl.py:113-127): one workgroup per (graph, job)
struct MaskJobs {
  dmp_mask_job job[DMP_MASK_MAX_JOBS];
};
__global__ __launch_bounds__(kBlock) void len_masks_k(const MaskJobs t) {
  const dmp_mask_job &j = t.job[blockIdx.y];
  const int64_t b = blockIdx.x, L = j.max_len;
  const int64_t size = j.sizes[b], pad = L - size;
  const int64_t start = j.off ? j.off[b] : b * L;
  int cnt = 0;
  for (int64_t x = threadIdx.x; x < L; x += kBlock) {
    bool m = x >= pad;
    if (m && j.rev) m = j.rev[start + x - pad] == 0;
    j.mask[b * L + x] = m ? 1 : 0;
    cnt += m ? 1 : 0;
  }
  __shared__ int part[kBlock / 64];
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < kBlock / 64; ++w) total += part[w];
    j.count[b] = (float)total;
  }
}

// ---- ScalarFilter gates (filter.py:6-16): mark the labels of every pattern, then look the target rows up
struct FilterJobs {
  dmp_filter_job job[DMP_FILTER_MAX_JOBS];
  int n;
};
__global__ __launch_bounds__(kBlock) void filter_mark_k(const FilterJobs t, int64_t B, uint8_t *__restrict__ present) {
  const dmp_filter_job &j = t.job[blockIdx.y];
  uint8_t *__restrict__ pr = present + j.present_off;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < j.num_p) {
    const int64_t b = j.seg_is_i32 ? (int64_t)static_cast<const int32_t *>(j.p_seg)[i] : static_cast<const int64_t *>(j.p_seg)[i];
    const int64_t l = j.p_label[i];
    if (b >= 0 && b < B && l >= 0 && l < j.num_labels) pr[b * j.num_labels + l] = 1;
  }
  // a pattern shorter than the longest one is pre-padded with zeros: label 0 takes part in the comparison
  if (j.p_sizes && i < B && j.p_sizes[i] < j.p_max) pr[i * j.num_labels] = 1;
}
__global__ __launch_bounds__(kBlock) void filter_gate_k(const FilterJobs t, int64_t B, const uint8_t *__restrict__ present) {
  const dmp_filter_job &j = t.job[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= j.num_g) return;
  const int64_t b = j.seg_is_i32 ? (int64_t)static_cast<const int32_t *>(j.g_seg)[i] : static_cast<const int64_t *>(j.g_seg)[i];
  const int64_t l = j.g_label[i];
  const bool in = b >= 0 && b < B && l >= 0 && l < j.num_labels;
  j.gate[i] = (in && present[j.present_off + b * j.num_labels + l]) ? 1.f : 0.f;
}

}  // namespace
}  // namespace dmp

using namespace dmp;

#define DMP_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) { set_last_hip_error(e_); return DMP_ERR_HIP; } \
  } while (0)

extern "C" {

size_t dmp_csr_workspace_words(int64_t N, int64_t E) {
  (void)E;
  if (N < 0) return 0;
  // cnt/cursor [N] + scan tile sums
  return (size_t)N + (size_t)scan_tiles_for(N) + 8;
}

size_t dmp_csr_pair_workspace_words(int64_t N) {
  if (N < 0) return 0;
  return 2 * ((size_t)N + (size_t)scan_tiles_for(N)) + 8;
}

int dmp_csr_build_pair(const int64_t *dst, const int64_t *src, const uint8_t *flag, int64_t E, int64_t N,
                       int32_t *in_ptr, int32_t *in_ent, int32_t *dst32, int64_t *in_deg,
                       int32_t *out_ptr, int32_t *out_ent, int32_t *src32, int64_t *out_deg,
                       int32_t *status, int32_t *ws, void *stream) {
  if (E < 0 || N < 0 || !in_ptr || !out_ptr || !status || !ws) return DMP_ERR_BAD_ARG;
  if (E > 0 && (!dst || !src || !in_ent || !out_ent)) return DMP_ERR_BAD_ARG;
  if (E >= ((int64_t)1 << 30) || N >= ((int64_t)1 << 31) - 1) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = scan_tiles_for(N);
  CsrPair p;
  p.key[0] = dst; p.key[1] = src;
  p.rowptr[0] = in_ptr; p.rowptr[1] = out_ptr;
  p.ent[0] = in_ent; p.ent[1] = out_ent;
  p.key32[0] = dst32; p.key32[1] = src32;
  p.degree[0] = in_deg; p.degree[1] = out_deg;
  p.cnt[0] = ws; p.cnt[1] = ws + N;
  p.tiles[0] = ws + 2 * N; p.tiles[1] = ws + 2 * N + ntiles;
  p.status = status;
  DMP_HIP_TRY(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st));
  if (N == 0) {
    DMP_HIP_TRY(hipMemsetAsync(in_ptr, 0, sizeof(int32_t), st));
    DMP_HIP_TRY(hipMemsetAsync(out_ptr, 0, sizeof(int32_t), st));
    if (E > 0) DMP_HIP_TRY(hipMemsetAsync(status, 0x01, 2 * sizeof(int32_t), st));   // every endpoint is out of range
    return DMP_OK;
  }
  DMP_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(int32_t) * (size_t)(2 * N), st));
  const int64_t nt = (N + kScanTile - 1) / kScanTile;           // tiles in use (the layout keeps one spare per array)
  if (E > 0) csr_count_pair<<<dim3(nblk(E), 2), kBlock, 0, st>>>(p, E, N);
  scan_tiles_pair<<<dim3((unsigned)nt, 2), kBlock, 0, st>>>(p, N);
  scan_tile_sums_pair<<<dim3(1, 2), kBlock, 0, st>>>(p, nt, N);
  scan_finish_pair<<<dim3((unsigned)nt, 2), kBlock, 0, st>>>(p, N);
  if (E > 0) csr_fill_pair<<<dim3(nblk(E), 2), kBlock, 0, st>>>(p, flag, E, N);
  csr_sort_rows_pair<<<dim3(nblk(N), 2), kBlock, 0, st>>>(p, N);
  return check_launch();
}

int dmp_csr_build_graphs_max_nodes(void) { return kCsrGraphNodes; }

int dmp_csr_build_graphs(const int64_t *dst, const int64_t *src, const uint8_t *flag, const int64_t *node_off,
                         const int64_t *edge_off, int64_t B, int64_t E, int64_t N, int32_t *in_ptr, int32_t *in_ent,
                         int32_t *dst32, int64_t *in_deg, int32_t *out_ptr, int32_t *out_ent, int32_t *src32,
                         int64_t *out_deg, int32_t *status, void *stream) {
  if (E < 0 || N < 0 || B <= 0 || !in_ptr || !out_ptr || !status || !node_off || !edge_off) return DMP_ERR_BAD_ARG;
  if (E > 0 && (!dst || !src || !in_ent || !out_ent)) return DMP_ERR_BAD_ARG;
  if (E >= ((int64_t)1 << 30) || N >= ((int64_t)1 << 31) - 1 || B > 0x7fffffff) return DMP_ERR_UNSUPPORTED;
  CsrGraphs p;
  p.key[0] = dst; p.key[1] = src; p.flag = flag; p.node_off = node_off; p.edge_off = edge_off; p.B = B; p.N = N; p.E = E;
  p.rowptr[0] = in_ptr; p.rowptr[1] = out_ptr; p.ent[0] = in_ent; p.ent[1] = out_ent;
  p.key32[0] = dst32; p.key32[1] = src32; p.degree[0] = in_deg; p.degree[1] = out_deg; p.status = status;
  csr_graphs_k<<<(unsigned)B, kBlock, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int dmp_csr_build(const int64_t *key, const uint8_t *flag, int64_t E, int64_t N, int32_t *rowptr,
                  int32_t *ent, int32_t *key32, int64_t *degree, int32_t *status, int32_t *ws,
                  void *stream) {
  if (E < 0 || N < 0 || !rowptr || !status || !ws) return DMP_ERR_BAD_ARG;
  if (E > 0 && (!key || !ent)) return DMP_ERR_BAD_ARG;
  if (E >= ((int64_t)1 << 30) || N >= ((int64_t)1 << 31) - 1) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int32_t *cnt = ws, *tiles = ws + N;
  DMP_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  if (N > 0) DMP_HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)N, st));
  if (E > 0) csr_count<<<nblk(E), kBlock, 0, st>>>(key, E, N, cnt, key32, status);
  int rc = exclusive_scan<int32_t, int32_t>(cnt, N, rowptr, tiles, st);
  if (rc != DMP_OK) return rc;
  if (N > 0 && E > 0) {
    copy_i32<<<nblk(N), kBlock, 0, st>>>(rowptr, N, cnt);  // cnt becomes the fill cursor
    csr_fill<<<nblk(E), kBlock, 0, st>>>(key, flag, E, N, cnt, ent);
  }
  if (N > 0) csr_sort_rows<<<nblk(N), kBlock, 0, st>>>(rowptr, N, ent, degree);
  return check_launch();
}

int dmp_incidence_build(const int32_t *in_ptr, const int32_t *in_ent, const int32_t *out_ptr,
                        const int32_t *out_ent, int64_t N, int64_t E, int32_t *inc_ptr,
                        int32_t *inc_ent, void *stream) {
  if (N < 0 || E < 0 || !in_ptr || !out_ptr || !inc_ptr) return DMP_ERR_BAD_ARG;
  if (E > 0 && (!in_ent || !out_ent || !inc_ent)) return DMP_ERR_BAD_ARG;
  if (E >= ((int64_t)1 << 29)) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  incidence_ptr<<<nblk(N + 1), kBlock, 0, st>>>(in_ptr, out_ptr, N, inc_ptr);
  if (E > 0 && N > 0)
    incidence_fill<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, out_ptr, out_ent, N, 2 * E, inc_ent);
  return check_launch();
}

int dmp_degree_coef(const int64_t *out_deg, int64_t N, float *coef, void *stream) {
  if (N < 0) return DMP_ERR_BAD_ARG;
  if (N == 0) return DMP_OK;
  if (!out_deg || !coef) return DMP_ERR_BAD_ARG;
  degree_coef_k<<<nblk(N), kBlock, 0, (hipStream_t)stream>>>(out_deg, N, coef);
  return check_launch();
}

size_t dmp_scan_workspace_words(int64_t n) { return n < 0 ? 0 : (size_t)scan_tiles_for(n) + 8; }

int dmp_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, int64_t *ws, void *stream) {
  if (n < 0 || !out || !ws || (n > 0 && !in)) return DMP_ERR_BAD_ARG;
  return exclusive_scan<int64_t, int64_t>(in, n, out, ws, (hipStream_t)stream);
}

int dmp_collate(const int64_t *local_src, const int64_t *local_dst, const int64_t *num_nodes,
                const int64_t *num_edges, int64_t B, int64_t N, int64_t E, int64_t *node_off,
                int64_t *edge_off, int64_t *src, int64_t *dst, int32_t *edge_graph,
                int32_t *node_graph, void *stream) {
  if (B < 0 || N < 0 || E < 0 || !node_off || !edge_off) return DMP_ERR_BAD_ARG;
  if (B > 0 && (!num_nodes || !num_edges)) return DMP_ERR_BAD_ARG;
  if (E > 0 && (!local_src || !local_dst || !src || !dst || B == 0)) return DMP_ERR_BAD_ARG;
  if (B > kScanTile) {
    // sizes are scanned by single-tile launches below (no scratch in the signature)
    return DMP_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    DMP_HIP_TRY(hipMemsetAsync(node_off, 0, sizeof(int64_t), st));
    DMP_HIP_TRY(hipMemsetAsync(edge_off, 0, sizeof(int64_t), st));
    return DMP_OK;
  }
  // B <= 2048: one tile each; the tile total lands in off[B] twice (harmless)
  collate_offsets<<<dim3(1, 2), kBlock, 0, st>>>(num_nodes, num_edges, B, node_off, edge_off);
  const bool nodes = node_graph && N > 0;
  if (E > 0 || nodes) {
    const int64_t most = (nodes && N > E) ? N : (E > 0 ? E : N);
    collate_both<<<dim3(nblk(most), nodes ? 2 : 1), kBlock, 0, st>>>(local_src, local_dst, node_off, edge_off, B, E, N, src, dst,
                                                                   edge_graph, node_graph);
  }
  return check_launch();
}

int dmp_collate_jobs(const dmp_collate_job *jobs, int n, void *stream) {
  if (!jobs || n < 1 || n > DMP_COLLATE_MAX_JOBS) return DMP_ERR_BAD_ARG;
  CollateJobs t;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const dmp_collate_job &j = jobs[i];
    if (j.B <= 0 || j.N < 0 || j.E < 0 || !j.node_off || !j.edge_off || !j.num_nodes || !j.num_edges) return DMP_ERR_BAD_ARG;
    if (j.E > 0 && (!j.local_src || !j.local_dst || !j.src || !j.dst)) return DMP_ERR_BAD_ARG;
    if (j.B > kScanTile) return DMP_ERR_UNSUPPORTED;         // sizes are scanned by single-tile launches, as in dmp_collate
    t.job[i] = j;
    const int64_t m = j.E > j.N ? j.E : j.N;
    if (m > most) most = m;
  }
  hipStream_t st = (hipStream_t)stream;
  collate_offsets_jobs<<<dim3(1, 2 * n), kBlock, 0, st>>>(t);
  if (most > 0) collate_both_jobs<<<dim3(nblk(most), 2 * n), kBlock, 0, st>>>(t);
  return check_launch();
}

int dmp_add_reversed_edges(const int64_t *src, const int64_t *dst, const int64_t *eid,
                           const int64_t *elabel, const int64_t *edge_off, int64_t B, int64_t E,
                           int64_t max_ne, int64_t max_nel, int64_t *o_src, int64_t *o_dst,
                           int64_t *o_eid, int64_t *o_elabel, uint8_t *o_rev, void *stream) {
  if (B < 0 || E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!src || !dst || !eid || !elabel || !edge_off || !o_src || !o_dst || !o_eid || !o_elabel ||
      !o_rev || B == 0)
    return DMP_ERR_BAD_ARG;
  add_rev_k<<<nblk(E), kBlock, 0, (hipStream_t)stream>>>(src, dst, eid, elabel, edge_off, B, E, max_ne,
                                                         max_nel, o_src, o_dst, o_eid, o_elabel, o_rev);
  return check_launch();
}

int dmp_line_graph_count(const int32_t *in_ptr, const int64_t *src, int64_t E, int64_t *cnt,
                         void *stream) {
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!in_ptr || !src || !cnt) return DMP_ERR_BAD_ARG;
  lg_count<<<nblk(E), kBlock, 0, (hipStream_t)stream>>>(in_ptr, src, E, cnt);
  return check_launch();
}

int dmp_line_graph_fill(const int32_t *in_ptr, const int32_t *in_ent, const int64_t *src,
                        const int64_t *off, int64_t E, int64_t *dual_src, int64_t *dual_dst,
                        int64_t *payload, void *stream) {
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!in_ptr || !in_ent || !src || !off || !dual_src || !dual_dst || !payload) return DMP_ERR_BAD_ARG;
  lg_fill<<<(unsigned)((E + 3) / 4), kBlock, 0, (hipStream_t)stream>>>(in_ptr, in_ent, src, off, E,
                                                                      dual_src, dual_dst, payload);
  return check_launch();
}

int dmp_first_edge_of_id(const int64_t *eid, int64_t E, int64_t K, int64_t *first, void *stream) {
  if (E < 0 || K < 0) return DMP_ERR_BAD_ARG;
  if (K == 0) return DMP_OK;
  if (!first || (E > 0 && !eid)) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  fill_i64<<<nblk(K), kBlock, 0, st>>>(first, K, INT64_MAX);
  if (E > 0) first_of_id_k<<<nblk(E), kBlock, 0, st>>>(eid, E, K, first);
  first_fix_k<<<nblk(K), kBlock, 0, st>>>(first, K);
  return check_launch();
}

size_t dmp_dedupe_table_words(int64_t M) { return M < 0 ? 0 : (size_t)table_size(M); }

int dmp_dedupe_first(const int64_t *key_a, const int64_t *key_l, const int64_t *key_b, int64_t M,
                     int64_t *table, uint8_t *keep, void *stream) {
  if (M < 0) return DMP_ERR_BAD_ARG;
  if (M == 0) return DMP_OK;
  if (!key_a || !key_l || !key_b || !table || !keep) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t cap = table_size(M);
  DMP_HIP_TRY(hipMemsetAsync(table, 0xFF, sizeof(int64_t) * cap, st));
  unsigned long long *t = reinterpret_cast<unsigned long long *>(table);
  dedupe_insert<<<nblk(M), kBlock, 0, st>>>(key_a, key_l, key_b, M, t, cap - 1);
  dedupe_lookup<<<nblk(M), kBlock, 0, st>>>(key_a, key_l, key_b, M, t, cap - 1, keep);
  return check_launch();
}

int dmp_random_walks(const int32_t *out_ptr, const int32_t *out_ent, const int32_t *dst, const int64_t *seeds,
                     int64_t num_seeds, int walks, int depth, uint64_t seed, int64_t *traces, uint8_t *visited, void *stream) {
  if (num_seeds < 0 || walks < 0 || depth < 0) return DMP_ERR_BAD_ARG;
  if (num_seeds == 0 || walks == 0) return DMP_OK;
  if (!out_ptr || !seeds || (!traces && !visited) || (depth > 0 && (!out_ent || !dst))) return DMP_ERR_BAD_ARG;
  if (num_seeds * walks >= ((int64_t)1 << 32)) return DMP_ERR_UNSUPPORTED;
  random_walks_k<<<nblk(num_seeds * walks), kBlock, 0, (hipStream_t)stream>>>(out_ptr, out_ent, dst, seeds, num_seeds, walks, depth,
                                                                          (uint32_t)(seed ^ (seed >> 32)), traces, visited);
  return check_launch();
}

int dmp_sample_in_edges(const int32_t *in_ptr, const int32_t *in_ent, const uint8_t *wanted, int64_t N, int64_t E, int width,
                        uint64_t seed, uint8_t *mask, void *stream) {
  if (N < 0 || E < 0 || width < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!in_ptr || !in_ent || !mask) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DMP_HIP_TRY(hipMemsetAsync(mask, 0, (size_t)E, st));
  const uint32_t s32 = (uint32_t)(seed ^ (seed >> 32));
  if (N > 0 && width > 0) {
    if (width <= kMaxSampleWidth) sample_in_edges_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, wanted, N, width, s32, mask);
    else sample_in_edges_wave_k<<<nblk(N * 64), kBlock, 0, st>>>(in_ptr, in_ent, wanted, N, width, s32, mask);   // any width
  }
  return check_launch();
}

int dmp_pool_index_jobs(const dmp_pool_job *jobs, int n, void *stream) {
  if (!jobs || n < 1 || n > DMP_POOL_MAX_JOBS) return DMP_ERR_BAD_ARG;
  PoolJobs t;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const dmp_pool_job &j = jobs[i];
    if (j.Ba < 0 || j.Bb < 0 || j.R < 0 || j.rows_a < 0 || j.rows_a > j.R || j.chunk <= 0) return DMP_ERR_BAD_ARG;
    if (!j.off || !j.gptr || !j.vptr || (j.Ba > 0 && !j.sizes_a) || (j.Bb > 0 && !j.sizes_b)) return DMP_ERR_BAD_ARG;
    if (j.R >= ((int64_t)1 << 30)) return DMP_ERR_UNSUPPORTED;     // (row << 1) | flag in 32 bits
    PoolArgs a{j.sizes_a, j.sizes_b, j.Ba, j.Bb, j.flag_a, j.flag_b, j.rows_a, j.R, j.R / j.chunk + j.Ba + j.Bb, j.chunk, j.off,
               j.gptr, j.vptr, j.vent, j.gent, j.seg, j.flag8, j.rowmap, j.sizes};
    if ((a.R > 0 && !a.vent) || (a.V > 0 && !a.gent)) return DMP_ERR_BAD_ARG;
    t.job[i] = a;
    const int64_t m = a.R > a.V + 1 ? a.R : a.V + 1;
    if (m > most) most = m;
  }
  hipStream_t st = (hipStream_t)stream;
  pool_offsets_k<<<(unsigned)n, kBlock, 0, st>>>(t);
  pool_fill_k<<<dim3((unsigned)nblk(most), (unsigned)n), kBlock, 0, st>>>(t);
  return check_launch();
}

int dmp_pool_index(const int64_t *sizes_a, int64_t Ba, const int64_t *sizes_b, int64_t Bb, const uint8_t *flag_a,
                   const uint8_t *flag_b, int64_t rows_a, int64_t R, int chunk, int64_t *off, int32_t *gptr, int32_t *vptr,
                   int32_t *vent, int32_t *gent, int32_t *seg, void *stream) {
  dmp_pool_job j{sizes_a, sizes_b, Ba, Bb, flag_a, flag_b, rows_a, R, chunk, off, gptr, vptr, vent, gent, seg, nullptr, nullptr,
                 nullptr};
  return dmp_pool_index_jobs(&j, 1, stream);
}

int dmp_pool_weight_sums(const float *w, const uint8_t *flag8, const int64_t *off, int64_t B, float *out, void *stream) {
  if (B < 0) return DMP_ERR_BAD_ARG;
  if (B == 0) return DMP_OK;
  if (!off || !out) return DMP_ERR_BAD_ARG;
  pool_weight_sums_k<<<nblk(B * kWave), kBlock, 0, (hipStream_t)stream>>>(w, flag8, off, B, flag8 ? 2 : 1, out);
  return check_launch();
}

int dmp_dual_subisomorphisms(const int64_t *sub, int64_t T, const int64_t *sample_ptr, const int64_t *work_ptr,
                             int64_t work_total, int64_t B, const int64_t *p_node_off, const int64_t *p_edge_off,
                             const int64_t *p_src, const int64_t *p_dst, const int64_t *p_label, int64_t PE,
                             const int64_t *g_node_off, const int64_t *g_edge_off, const int32_t *g_out_ptr,
                             const int32_t *g_out_ent, const int32_t *g_dst, const int64_t *g_label, int32_t *workspace,
                             int64_t *first_sorted, int64_t *out, int32_t *status, void *stream) {
  if (T < 0 || B < 0 || PE < 0 || work_total < 0) return DMP_ERR_BAD_ARG;
  if (B == 0) return DMP_OK;
  if (!sample_ptr || !work_ptr || !p_node_off || !p_edge_off || !g_node_off || !g_edge_off || !g_out_ptr || !workspace ||
      !first_sorted || (PE > 0 && (!p_src || !p_dst || !p_label)) || (work_total > 0 && (!sub || !out)))
    return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int32_t *key_lo = workspace, *key_hi = workspace + PE, *num_keys = workspace + 2 * PE;   // 2 PE + B words
  dual_keys_k<<<nblk(B), kBlock, 0, st>>>(p_src, p_dst, p_edge_off, B, g_node_off, g_edge_off, g_out_ptr, g_out_ent, g_dst,
                                         key_lo, key_hi, num_keys, first_sorted);
  if (work_total > 0) {
    const int64_t blocks = (work_total + kBlock - 1) / kBlock;
    dual_match_k<<<(unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks), kBlock, 0, st>>>(
        sub, sample_ptr, work_ptr, B, p_node_off, p_edge_off, p_src, p_dst, p_label, key_lo, key_hi, num_keys, first_sorted,
        g_node_off, g_edge_off, g_out_ptr, g_out_ent, g_dst, g_label, out, status);
  }
  return check_launch();
}

int dmp_subiso_node_weights(const int64_t *sub, int64_t T, const int64_t *sample_ptr, int64_t B,
                            const int64_t *g_node_off, int64_t *out, int64_t N, int32_t *status, void *stream) {
  if (T < 0 || B < 0 || N < 0) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N > 0) {
    if (!out) return DMP_ERR_BAD_ARG;
    DMP_HIP_TRY(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)N, st));
  }
  if (status) DMP_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  if (T == 0) return DMP_OK;
  if (B == 0 || N == 0 || !sub || !sample_ptr || !g_node_off) return DMP_ERR_BAD_ARG;
  const int64_t blocks = (T + kBlock - 1) / kBlock;
  subiso_node_k<<<(unsigned)(blocks < 65536 ? blocks : 65536), kBlock, 0, st>>>(
      sub, T, sample_ptr, B, g_node_off, reinterpret_cast<unsigned long long *>(out), status);
  return check_launch();
}

int dmp_pattern_edge_active(const int64_t *p_src, const int64_t *p_dst, const int64_t *p_edge_off,
                            const int32_t *p_edge_graph, int64_t PE, uint8_t *active, void *stream) {
  if (PE < 0) return DMP_ERR_BAD_ARG;
  if (PE == 0) return DMP_OK;
  if (!p_src || !p_dst || !p_edge_off || !p_edge_graph || !active) return DMP_ERR_BAD_ARG;
  pattern_edge_active_k<<<nblk(PE), kBlock, 0, (hipStream_t)stream>>>(p_src, p_dst, p_edge_off, p_edge_graph, PE, active);
  return check_launch();
}

int dmp_subiso_edge_weights(const int64_t *sub, int64_t T, const int64_t *sample_ptr, const int64_t *work_ptr,
                            int64_t B, const int64_t *p_node_off, const int64_t *p_edge_off,
                            const int64_t *p_src, const int64_t *p_dst, const int64_t *p_label,
                            const uint8_t *active, const int64_t *g_node_off, const int32_t *g_out_ptr,
                            const int32_t *g_out_ent, const int32_t *g_dst, const int64_t *g_label,
                            int64_t *out, int64_t E, int64_t work_hint, int32_t *status, void *stream) {
  if (T < 0 || B < 0 || E < 0) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (E > 0) {
    if (!out) return DMP_ERR_BAD_ARG;
    DMP_HIP_TRY(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)E, st));
  }
  if (status) DMP_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  if (T == 0 || E == 0) return DMP_OK;
  if (B == 0 || !sub || !sample_ptr || !work_ptr || !p_node_off || !p_edge_off || !p_src || !p_dst || !p_label ||
      !active || !g_node_off || !g_out_ptr || !g_out_ent || !g_dst || !g_label)
    return DMP_ERR_BAD_ARG;
  // the exact amount of work (rows x pattern edges, summed) lives on the device in work_ptr[B];
  // work_hint only sizes the grid (grid-stride loop), so no host sync is needed
  int64_t blocks = ((work_hint > 0 ? work_hint : T) + kBlock - 1) / kBlock;
  if (blocks > 65536) blocks = 65536;
  subiso_edge_k<<<(unsigned)blocks, kBlock, 0, st>>>(sub, sample_ptr, work_ptr, B, p_node_off, p_edge_off, p_src,
                                                    p_dst, p_label, active, g_node_off, g_out_ptr, g_out_ent,
                                                    g_dst, g_label, reinterpret_cast<unsigned long long *>(out),
                                                    status);
  return check_launch();
}

static int64_t dmp_class_tiles_segsum_words(int num_classes) {
  return (int64_t)(num_classes < kCtFast ? num_classes : kCtFast) * kCtSegs;
}

size_t dmp_class_tiles_workspace_words(int64_t num_nodes, int num_classes) {
  // int32 words: [cnt 2C | status 1 | pad 1 | tile_off C+1 | segsum | node_base N]
  if (num_nodes < 0 || num_classes < 2) return 0;
  return (size_t)(2 * (int64_t)num_classes + 2 + num_classes + 1 + dmp_class_tiles_segsum_words(num_classes) + num_nodes + 8);
}

int dmp_concat_pairs(const dmp_concat_job *jobs, int num_jobs, void *stream) {
  if (num_jobs < 0 || num_jobs > DMP_CONCAT_MAX_JOBS) return DMP_ERR_BAD_ARG;
  if (num_jobs == 0) return DMP_OK;
  if (!jobs) return DMP_ERR_BAD_ARG;
  ConcatJobs t;
  int64_t most = 0;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_concat_job &j = jobs[i];
    if (j.na < 0 || j.nb < 0 || (j.elem_size != 1 && j.elem_size != 4 && j.elem_size != 8)) return DMP_ERR_BAD_ARG;
    if (j.na > 0 && !j.a && j.elem_size != 4) return DMP_ERR_BAD_ARG;
    if ((j.nb > 0 && !j.b) || (j.na + j.nb > 0 && !j.out)) return DMP_ERR_BAD_ARG;
    if (j.add_b != 0 && j.elem_size != 8) return DMP_ERR_BAD_ARG;
    t.job[i] = j;
    if (j.na + j.nb > most) most = j.na + j.nb;
  }
  if (most == 0) return DMP_OK;
  int64_t nb = (most + kBlock - 1) / kBlock;
  if (nb > 4096) nb = 4096;
  concat_pairs_k<<<dim3((unsigned)nb, (unsigned)num_jobs), kBlock, 0, (hipStream_t)stream>>>(t);
  DMP_HIP_TRY(hipGetLastError());
  return DMP_OK;
}

int dmp_table_rows(const dmp_lookup_job *jobs, int num_jobs, void *stream) {
  if (num_jobs < 0 || num_jobs > DMP_LOOKUP_MAX_JOBS) return DMP_ERR_BAD_ARG;
  if (num_jobs == 0) return DMP_OK;
  if (!jobs) return DMP_ERR_BAD_ARG;
  LookupJobs t;
  int64_t most = 0;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_lookup_job &j = jobs[i];
    if (j.rows < 0 || j.width < 1 || j.table_rows < 0 || j.ld < j.width) return DMP_ERR_BAD_ARG;
    if (j.rows > 0 && (!j.table || !j.idx || !j.out)) return DMP_ERR_BAD_ARG;
    t.job[i] = j;
    const int64_t units = j.width <= 32 ? j.rows : j.rows * j.width;      // threads of work: a row each where rows are narrow
    if (units > most) most = units;
  }
  if (most == 0) return DMP_OK;
  int64_t nb = (most + kBlock - 1) / kBlock;
  if (nb > 8192) nb = 8192;
  table_rows_k<<<dim3((unsigned)nb, (unsigned)num_jobs), kBlock, 0, (hipStream_t)stream>>>(t);
  DMP_HIP_TRY(hipGetLastError());
  return DMP_OK;
}

int dmp_len_masks(const dmp_mask_job *jobs, int num_jobs, int64_t B, void *stream) {
  if (num_jobs < 0 || num_jobs > DMP_MASK_MAX_JOBS || B < 0 || B > 0x7fffffffLL) return DMP_ERR_BAD_ARG;
  if (num_jobs == 0 || B == 0) return DMP_OK;
  if (!jobs) return DMP_ERR_BAD_ARG;
  MaskJobs t;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_mask_job &j = jobs[i];
    if (j.max_len < 0 || !j.sizes || !j.count || (j.max_len > 0 && !j.mask)) return DMP_ERR_BAD_ARG;
    t.job[i] = j;
  }
  len_masks_k<<<dim3((unsigned)B, (unsigned)num_jobs), kBlock, 0, (hipStream_t)stream>>>(t);
  DMP_HIP_TRY(hipGetLastError());
  return DMP_OK;
}

int dmp_scalar_filter_gates(const dmp_filter_job *jobs, int num_jobs, int64_t B, uint8_t *present,
                            int64_t present_bytes, void *stream) {
  if (num_jobs < 0 || num_jobs > DMP_FILTER_MAX_JOBS || B < 0 || present_bytes < 0) return DMP_ERR_BAD_ARG;
  if (num_jobs == 0) return DMP_OK;
  if (!jobs) return DMP_ERR_BAD_ARG;
  FilterJobs t;
  t.n = num_jobs;
  int64_t most_p = 0, most_g = 0;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_filter_job &j = jobs[i];
    if (j.num_p < 0 || j.num_g < 0 || j.num_labels < 1 || j.present_off < 0 || j.p_max < 0) return DMP_ERR_BAD_ARG;
    if (j.present_off + B * j.num_labels > present_bytes) return DMP_ERR_BAD_ARG;
    if ((j.num_p > 0 && (!j.p_seg || !j.p_label)) || (j.num_g > 0 && (!j.g_seg || !j.g_label || !j.gate)))
      return DMP_ERR_BAD_ARG;
    t.job[i] = j;
    const int64_t mp = j.num_p > (j.p_sizes ? B : 0) ? j.num_p : (j.p_sizes ? B : 0);
    if (mp > most_p) most_p = mp;
    if (j.num_g > most_g) most_g = j.num_g;
  }
  if (present_bytes > 0) {
    if (!present) return DMP_ERR_BAD_ARG;
    DMP_HIP_TRY(hipMemsetAsync(present, 0, (size_t)present_bytes, (hipStream_t)stream));
  }
  if (most_p > 0) {
    filter_mark_k<<<dim3((unsigned)((most_p + kBlock - 1) / kBlock), (unsigned)num_jobs), kBlock, 0,
                    (hipStream_t)stream>>>(t, B, present);
    DMP_HIP_TRY(hipGetLastError());
  }
  if (most_g > 0) {
    filter_gate_k<<<dim3((unsigned)((most_g + kBlock - 1) / kBlock), (unsigned)num_jobs), kBlock, 0,
                    (hipStream_t)stream>>>(t, B, present);
    DMP_HIP_TRY(hipGetLastError());
  }
  return DMP_OK;
}

static int class_tiles_impl(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int32_t *row_cnt,
                            int64_t N, int64_t E, int num_classes, int64_t tiles_bound, int32_t *ws, int32_t *slot_edge,
                            float *tile_scale, int32_t *num_tiles, void *stream);

int dmp_class_tiles(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, int64_t N, int64_t E,
                    int num_classes, int64_t tiles_bound, int32_t *ws, int32_t *slot_edge, float *tile_scale,
                    int32_t *num_tiles, void *stream) {
  return class_tiles_impl(deg, in_ptr, in_ent, nullptr, nullptr, N, E, num_classes, tiles_bound, ws, slot_edge, tile_scale, num_tiles,
                          stream);
}

constexpr int kKeepGroup = 16;          // lanes per row of the long-row form
int64_t dmp_csr_keep_scratch_words(int64_t N) { return N + (N + kBlock / kKeepGroup - 1) / (kBlock / kKeepGroup) + 1; }

int dmp_csr_keep(const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int64_t N, int64_t num_entries, int32_t *row_cnt,
                 int32_t *keep_ptr, int32_t *keep_ent, void *stream) {
  if (N < 0) return DMP_ERR_BAD_ARG;
  if (!keep_ptr) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return hipMemsetAsync(keep_ptr, 0, sizeof(int32_t), st) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!in_ptr || !in_ent || !gate || !row_cnt || !keep_ent) return DMP_ERR_BAD_ARG;
  int32_t *blk = row_cnt + N;                                  // row_cnt: N counts + one total per block of rows
  if (num_entries >= 12 * N) {                                 // long rows (num_entries: a host-side hint, 0 = unknown): a lane group per row
    constexpr int RPB = kBlock / kKeepGroup;
    const unsigned nb = (unsigned)((N + RPB - 1) / RPB);
    keep_count_g_k<kKeepGroup><<<nb, kBlock, 0, st>>>(in_ptr, in_ent, gate, N, row_cnt, blk);
    keep_fill_g_k<kKeepGroup><<<nb, kBlock, 0, st>>>(in_ptr, in_ent, gate, row_cnt, blk, N, keep_ptr, keep_ent);
    return check_launch();
  }
  keep_count_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, gate, N, row_cnt, blk);
  keep_fill_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, gate, row_cnt, blk, N, keep_ptr, keep_ent);
  return check_launch();
}

int dmp_incidence_keep(const int32_t *in_ptr, const int32_t *in_ent, const int32_t *out_ptr, const int32_t *out_ent, const float *gate,
                       const int32_t *list, const int32_t *count, int64_t N, int32_t *row_cnt, int32_t *keep_ptr, int32_t *keep_ent,
                       void *stream) {
  if (N < 0 || !keep_ptr) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return hipMemsetAsync(keep_ptr, 0, sizeof(int32_t), st) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!in_ptr || !in_ent || (out_ptr && !out_ent) || !gate || !list || !count || !row_cnt || !keep_ent) return DMP_ERR_BAD_ARG;
  int32_t *blk = row_cnt + N;                                  // row_cnt: N counts + one total per block (dmp_csr_keep_scratch_words)
  inc_keep_count_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, out_ptr, out_ent, gate, list, count, row_cnt, blk);
  inc_keep_fill_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, out_ptr, out_ent, gate, list, count, row_cnt, blk, keep_ptr, keep_ent);
  return check_launch();
}

int dmp_class_tiles_gated(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int32_t *row_cnt,
                          int64_t N, int64_t E, int num_classes, int64_t tiles_bound, int32_t *ws, int32_t *slot_edge,
                          float *tile_scale, int32_t *num_tiles, void *stream) {
  if (!gate || (N > 0 && !row_cnt)) return DMP_ERR_BAD_ARG;
  return class_tiles_impl(deg, in_ptr, in_ent, gate, row_cnt, N, E, num_classes, tiles_bound, ws, slot_edge, tile_scale, num_tiles,
                          stream);
}

static int class_tiles_impl(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int32_t *row_cnt,
                            int64_t N, int64_t E, int num_classes, int64_t tiles_bound, int32_t *ws, int32_t *slot_edge,
                            float *tile_scale, int32_t *num_tiles, void *stream) {
  if (N < 0 || E < 0 || num_classes < 2 || num_classes > 65536 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!ws || !slot_edge || !tile_scale || !num_tiles) return DMP_ERR_BAD_ARG;
  if (N > 0 && (!deg || !in_ptr)) return DMP_ERR_BAD_ARG;
  if (E > 0 && !in_ent) return DMP_ERR_BAD_ARG;
  if (tiles_bound < E / 32 + num_classes) return DMP_ERR_BAD_ARG;      // every class may add one partial tile
  if ((reinterpret_cast<uintptr_t>(ws) & 7u) != 0) return DMP_ERR_BAD_ARG;
  unsigned long long *cnt = reinterpret_cast<unsigned long long *>(ws);
  int32_t *status = ws + 2 * (int64_t)num_classes;
  int32_t *tile_off = status + 2;
  int32_t *segsum = tile_off + num_classes + 1;
  int32_t *node_base = segsum + dmp_class_tiles_segsum_words(num_classes);
  hipStream_t st = (hipStream_t)stream;
  DMP_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(int32_t) * (size_t)(2 * (int64_t)num_classes + 2), st));    // counters + status
  DMP_HIP_TRY(hipMemsetAsync(slot_edge, 0xFF, sizeof(int32_t) * (size_t)tiles_bound * 32, st));      // -1 = padding
  if (N > 0 && gate) ct_live_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, gate, N, row_cnt);
  if (N > 0) ct_hist_k<<<nblk(N), kBlock, 0, st>>>(deg, in_ptr, row_cnt, N, num_classes, cnt, status);
  ct_scan_k<<<1, 1024, 0, st>>>(cnt, num_classes, status, tile_off, num_tiles);
  if (N > 0) {
    const int fast = num_classes < kCtFast ? num_classes : kCtFast;
    const int64_t waves = (int64_t)fast * kCtSegs + kCtSlowWaves;
    ct_seg_k<<<nblk((int64_t)fast * kCtSegs * 64), kBlock, 0, st>>>(deg, in_ptr, row_cnt, N, num_classes, cnt, segsum);
    ct_base_k<<<nblk(waves * 64), kBlock, 0, st>>>(deg, in_ptr, row_cnt, N, num_classes, cnt, tile_off, status, segsum, node_base,
                                                  tile_scale);
    ct_fill_k<<<nblk(N), kBlock, 0, st>>>(in_ptr, in_ent, node_base, N, slot_edge, gate, row_cnt);
  }
  return check_launch();
}

}  // extern "C"
