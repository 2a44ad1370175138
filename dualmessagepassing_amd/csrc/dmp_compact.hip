// Gate compaction of a block-diagonal batch (gfx950 / MI355X): the edges a 0 / 1 filter gate keeps, graph by graph, in a
// batch of FIXED size.
//
// The reference multiplies the target graph's label embeddings by a filter gate before the rep-net and every layer's
// update by the same gate (SubgraphCountingMatching/models/basemodel.py:1515-1531, dmpnn.py:215-277: `zn = z + g (...)`):
// an edge with gate 0 enters as a zero row, stays a zero row through every layer and adds nothing to any node sum, pooled
// sum or gradient -- only its endpoints' DEGREES see it (dmpnn.py:101,144-146).  This transform hands the rep-net the
// kept edges only: same order (ascending eid inside every graph, so every fixed-order sum keeps its order), the degrees of
// the whole graph beside it, and `capacity - kept` padding edges (gate 0, self-loops spread over the graphs and their
// nodes: inert by the same argument) so that the compacted batch has `capacity` edges whatever the labels were -- a
// recorded step (HIP graph) replays it.  More kept edges than `capacity` ORs bit 0 into the caller's status word (the
// outputs are truncated then: the caller runs such a batch as it stands); a padded graph without nodes ORs bit 1.  The
// status word is only ever OR-ed into, so one word can watch over many steps; the caller clears it.
//
// Two launches, a workgroup per graph: (1) kept edges per graph, the whole graph's out-degrees (LDS histogram over the
// graph's node range, global atomics for graphs above 2,048 nodes); (2) every workgroup sums the per-graph counts before it
// (B <= a few thousand words out of L2), ranks its kept edges with wave ballots and writes them, then its padding edges.
#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kHistNodes = 2048;

__device__ __forceinline__ int64_t block_sum64(int64_t v, int64_t *lds) {   // every thread gets the sum; lds: kBlock / kWave words
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
  __syncthreads();
  int64_t s = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) s += lds[w];
  return s;
}

__global__ __launch_bounds__(kBlock) void gate_count_k(const float *__restrict__ gate, const int64_t *__restrict__ src,
                                                       const int64_t *__restrict__ node_off, const int64_t *__restrict__ edge_off,
                                                       int32_t *__restrict__ kept, unsigned long long *__restrict__ out_deg) {
  __shared__ int64_t red[kBlock / kWave];
  __shared__ int hist[kHistNodes];
  const int64_t g = blockIdx.x, beg = edge_off[g], end = edge_off[g + 1];
  const int64_t n0 = node_off[g], n = node_off[g + 1] - n0;
  const bool in_lds = out_deg && n <= kHistNodes;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += kBlock) hist[i] = 0;
    __syncthreads();
  }
  int64_t c = 0;
  for (int64_t e = beg + threadIdx.x; e < end; e += kBlock) {
    c += gate[e] != 0.f;
    if (out_deg) {
      const int64_t u = src[e];
      if (in_lds) {
        if (u >= n0 && u < n0 + n) atomicAdd(&hist[u - n0], 1);
      } else {
        atomicAdd(&out_deg[u], 1ull);                      // zeroed by the host for such batches
      }
    }
  }
  c = block_sum64(c, red);
  if (threadIdx.x == 0) kept[g] = (int32_t)c;
  if (in_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kBlock) out_deg[n0 + i] = (unsigned long long)hist[i];
  }
}

struct GateFill {
  const float *gate; const int64_t *src, *dst; const uint8_t *rev; const int64_t *node_off, *edge_off;
  const int32_t *kept; int64_t B, cap;
  int64_t *src_c, *dst_c; uint8_t *rev_c; int64_t *eid_map; float *gate_c; int64_t *num_edges_c, *edge_off_c; int32_t *status;
};

__global__ __launch_bounds__(kBlock) void gate_fill_k(const GateFill a) {
  __shared__ int64_t red[kBlock / kWave];
  __shared__ int wcnt[kBlock / kWave];
  const int64_t g = blockIdx.x;
  int64_t before = 0, total = 0;
  for (int64_t i = threadIdx.x; i < a.B; i += kBlock) {
    const int64_t k = a.kept[i];
    total += k;
    if (i < g) before += k;
  }
  before = block_sum64(before, red);
  total = block_sum64(total, red);
  int64_t P = a.cap - total;
  const bool over = P < 0;
  if (over) P = 0;
  const int64_t q = P / a.B, r = P % a.B;
  const int64_t pad = q + (g < r ? 1 : 0), pad_before = g * q + (g < r ? g : r);
  const int64_t kept_g = a.kept[g], off = before + pad_before;
  const int64_t n0 = a.node_off[g], n = a.node_off[g + 1] - n0;
  if (threadIdx.x == 0) {
    const int64_t lo = off < a.cap ? off : a.cap, hi = off + kept_g + pad < a.cap ? off + kept_g + pad : a.cap;
    a.edge_off_c[g] = lo;
    a.num_edges_c[g] = hi - lo;
    if (g == a.B - 1) a.edge_off_c[a.B] = hi;
    const int bits = (over ? 1 : 0) | (pad > 0 && n <= 0 ? 2 : 0);
    if (bits) atomicOr(a.status, bits);
  }
  // ---- the kept edges, in eid order: rank = kept edges before it in the graph
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t beg = a.edge_off[g], end = a.edge_off[g + 1];
  int64_t running = 0;
  for (int64_t base = beg; base < end; base += kBlock) {
    const int64_t e = base + threadIdx.x;
    const float gv = e < end ? a.gate[e] : 0.f;
    const bool f = gv != 0.f;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int wave_off = 0, chunk = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
      const int cnt = wcnt[w];
      if (w < wave) wave_off += cnt;
      chunk += cnt;
    }
    const int64_t pos = off + running + wave_off + __popcll(m & ((1ull << lane) - 1ull));
    if (f && pos < a.cap) {
      a.src_c[pos] = a.src[e];
      a.dst_c[pos] = a.dst[e];
      if (a.rev_c) a.rev_c[pos] = a.rev ? a.rev[e] : (uint8_t)0;
      a.eid_map[pos] = e;
      a.gate_c[pos] = gv;
    }
    running += chunk;
    __syncthreads();
  }
  // ---- the padding edges: gate 0, self-loops over the graph's nodes in turn, row 0 as their (unread) source row
  for (int64_t j = threadIdx.x; j < pad; j += kBlock) {
    const int64_t pos = off + kept_g + j;
    if (pos >= a.cap) break;
    // (a graph without nodes cannot hold its share of the padding: status bit 1 is up -- the guarded optimizer drops the step,
    // harness.fit arms the veto on both bits -- and the endpoint is clamped to a node of the batch so that nothing indexes past it)
    const int64_t node = n > 0 ? n0 + j % n : (n0 > 0 ? n0 - 1 : 0);
    a.src_c[pos] = node;
    a.dst_c[pos] = node;
    if (a.rev_c) a.rev_c[pos] = 0;
    a.eid_map[pos] = 0;
    a.gate_c[pos] = 0.f;
  }
}

__global__ __launch_bounds__(kBlock) void out_degrees_k(const int64_t *__restrict__ src, int64_t E, int64_t N,
                                                        unsigned long long *__restrict__ deg) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += (int64_t)gridDim.x * kBlock) {
    const int64_t u = src[e];
    if (u >= 0 && u < N) atomicAdd(&deg[u], 1ull);
  }
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

int dmp_gate_compact_hist_nodes(void) { return kHistNodes; }

int dmp_gate_compact(const float *gate, const int64_t *src, const int64_t *dst, const uint8_t *rev, const int64_t *node_off,
                     const int64_t *edge_off, int64_t B, int64_t N, int64_t E, int64_t cap, int zero_deg, int32_t *kept,
                     int64_t *out_deg, int64_t *src_c, int64_t *dst_c, uint8_t *rev_c, int64_t *eid_map, float *gate_c,
                     int64_t *num_edges_c, int64_t *edge_off_c, int32_t *status, void *stream) {
  if (B <= 0 || N < 0 || E <= 0 || cap <= 0) return DMP_ERR_BAD_ARG;
  if (!gate || !src || !dst || !node_off || !edge_off || !kept || !src_c || !dst_c || !eid_map || !gate_c || !num_edges_c ||
      !edge_off_c || !status)
    return DMP_ERR_BAD_ARG;
  if (B > 0x7fffffff || E >= (int64_t)1 << 31) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (out_deg && zero_deg && N > 0) DMP_HIP_TRY(hipMemsetAsync(out_deg, 0, sizeof(int64_t) * (size_t)N, st));
  gate_count_k<<<(unsigned)B, kBlock, 0, st>>>(gate, src, node_off, edge_off, kept, reinterpret_cast<unsigned long long *>(out_deg));
  GateFill a{gate, src, dst, rev, node_off, edge_off, kept, B, cap, src_c, dst_c, rev_c, eid_map, gate_c, num_edges_c, edge_off_c,
             status};
  gate_fill_k<<<(unsigned)B, kBlock, 0, st>>>(a);
  return check_launch();
}

int dmp_out_degrees(const int64_t *src, int64_t E, int64_t N, int64_t *deg, void *stream) {
  if (E < 0 || N < 0) return DMP_ERR_BAD_ARG;
  if (N == 0) return DMP_OK;
  if (!deg || (E > 0 && !src)) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DMP_HIP_TRY(hipMemsetAsync(deg, 0, sizeof(int64_t) * (size_t)N, st));
  if (E > 0) {
    const int64_t blocks = (E + kBlock - 1) / kBlock;
    out_degrees_k<<<(unsigned)(blocks < 4096 ? blocks : 4096), kBlock, 0, st>>>(src, E, N, reinterpret_cast<unsigned long long *>(deg));
  }
  return check_launch();
}

}  // extern "C"
