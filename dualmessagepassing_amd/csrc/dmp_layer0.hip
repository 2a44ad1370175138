// The FIRST layer of a rep-net whose edge input rows are a label embedding (gfx950).
//
// The reference embeds the multi-hot edge label code ([E, K], K ~ 10) into hid_dim and hands the [E, H] rows to the first
// DMPLayer (basemodel.py:1393-1420, dmpnn.py:111-156).  Every product of that layer with its edge input is therefore a
// product with  z0 = enc W  of rank K:
//     z0 (A + c B)                    = enc (W A) + c enc (W B)                  (the edge pre-activation)
//     sum over a node's edges of z0   = (sum of enc) W                           (the node aggregate)
//     z0^T [dPre | c dPre]            = W^T (enc^T [dPre | c dPre])              (the class-typed weight gradient)
//     enc^T dz0                       = enc^T dzn + (enc^T [dPre | c dPre]) [A|B]^T + (sum of enc)^T dS   (the embedding's gradient)
// so the E-row MFMA kernels of the general layer (one product forward, two backward, a scatter-add and the [E, H] input
// gradient) become two streaming passes with K fused multiply-adds per element: no matrix pipe, no class tiles, the
// edge rows in their natural (graph-major) order.
//
//   l0_pack_k        encU[r] = r < n ? enc_p[r] : gate[r-n] enc_g[r-n]   zero-padded to Kpad (16-byte rows for the segment sum);
//                    with a column offset for the second kind of rows, two tables read as ONE of 2K rows (node rows)
//   l0_edge_fwd_k    H1[r]   = act(encU[r] MA + c_r encU[r] MB + P[a_r, 0:H] - P[b_r, H:2H] + bias)
//   l0_bwd_w_k       partial sums of  encU^T dPre,  (c encU)^T dPre,  encU^T dZn
//
// A wave owns a row at a time (H / 64 values per lane); the row's K inputs, its coefficient and its two node rows are
// wave-uniform: one vector load, v_readlane, SGPR operands (as smallk_* in dmp_fused.hip).
#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kL0K = 16;              // widest label code
constexpr int kL0Rows = 4;            // rows in flight per wave
constexpr int kFwdRows = 4;
constexpr bool kBwdWide = true;          // four accumulators per combine round (-6 us at E = 549 k)
constexpr bool kBwdNoPrefetch = true;    // 8 rows per batch, no second register set (-5 us)
constexpr int kL0MaxPartials = 1024;
constexpr int kLaneCoef = 16, kLaneA = 17, kLaneB = 18;   // lanes that fetch the row's coefficient and node rows

typedef float f2_t __attribute__((ext_vector_type(2)));
template <int VW> struct Vec;
template <> struct Vec<1> { float v; };
template <> struct Vec<2> { f2_t v; };
__device__ __forceinline__ Vec<1> vload1(const float *row, int lane) { Vec<1> o; o.v = row[lane]; return o; }
template <int VW> __device__ __forceinline__ Vec<VW> vload(const float *row, int lane) {
  Vec<VW> o;
  if constexpr (VW == 2) o.v = *reinterpret_cast<const f2_t *>(row + lane * 2);
  else o.v = row[lane];
  return o;
}
template <int VW> __device__ __forceinline__ void vstore(float *row, int lane, const Vec<VW> &o) {
  if constexpr (VW == 2) *reinterpret_cast<f2_t *>(row + lane * 2) = o.v;
  else row[lane] = o.v;
}
template <int VW> __device__ __forceinline__ Vec<VW> vzero() {
  Vec<VW> o;
  if constexpr (VW == 2) o.v = f2_t{0.f, 0.f};
  else o.v = 0.f;
  return o;
}
// acc += s * v with the wave-uniform s straight from an SGPR.  H = 128: both of the lane's columns in ONE packed FMA (the
// scalar is the low half of an SGPR pair, read for both halves: op_sel_hi 0) -- these kernels are VALU-bound otherwise.
__device__ __forceinline__ void fmac_s(Vec<2> &acc, float s, const Vec<2> &v) {
  const unsigned long long sp = (unsigned long long)__builtin_bit_cast(unsigned, s);
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc.v) : "s"(sp), "v"(v.v));
}
__device__ __forceinline__ void fmac_s(Vec<1> &acc, float s, const Vec<1> &v) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(acc.v) : "s"(s), "v"(v.v));
}
template <int VW> __device__ __forceinline__ float &at(Vec<VW> &a, int c) {
  if constexpr (VW == 2) return reinterpret_cast<float *>(&a.v)[c];
  else return a.v;
}
template <int VW> __device__ __forceinline__ float at(const Vec<VW> &a, int c) {
  if constexpr (VW == 2) return a.v[c];
  else return a.v;
}
__device__ __forceinline__ float lane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int lane_i(float v, int l) { return __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l); }

struct PackArgs {
  const float *encp; int64_t ldp; int64_t n; const float *encg; int64_t ldg; const float *gate; int64_t rows_g;
  int K, Kpad, goff; float *out;
};

// A thread per ROW (its gate once, its <= 16 codes, 16-byte stores: Kpad % 4 == 0 and `out` comes 16-byte aligned from the host side).
struct PackJobs { PackArgs job[2]; };     // blockIdx.y = job (the edge rows' codes and the node rows' codes of a step: one launch)
__global__ __launch_bounds__(kBlock) void l0_pack_k(const PackJobs js) {
  const PackArgs &p = js.job[blockIdx.y];
  const int64_t rows = p.n + p.rows_g;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kBlock) {
    const bool pat = r < p.n;
    const float *src = pat ? p.encp + r * p.ldp : p.encg + (r - p.n) * p.ldg;
    const int c0 = pat ? 0 : p.goff;                       // the row's codes sit in columns c0 .. c0 + K - 1
    const float g = (!pat && p.gate) ? p.gate[r - p.n] : 1.f;
    float *dst = p.out + r * p.Kpad;
    for (int c = 0; c < p.Kpad; c += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = c + u - c0;
        v[u] = (k >= 0 && k < p.K) ? src[k] : 0.f;
        if (!pat && p.gate) v[u] *= g;
      }
      *reinterpret_cast<float4 *>(dst + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

struct FwdArgs {
  const float *enc; int64_t lde;           // [R, >= K] gated label codes of the union's edge rows
  const float *M; int64_t ldm;             // [K, 2H] = W [A | B]
  const float *P; int64_t ldp;             // node projections [N, >= 2H]: P[a, 0:H] - P[b, H:2H]
  const float *bias; const float *coef_e; const int32_t *sel_a, *sel_b;
  int64_t R; float slope; float *out; int64_t ldo;
  const uint32_t *rowmask;                 // bit r of rowmask[t] == 0: row 32 t + r is a DEAD row (its consumers all multiply it by a zero
                                           // gate and skip it): nothing is gathered, computed or stored for it.  NULL: every row
  const int32_t *list, *count;             // LIST form: the rows to do, ascending (dmp_kept_rows: the rows whose mask bit is set), *count of them
};

// LIST: the kernel walks a list of row ids instead of the rows 0 .. R-1 -- the rows a 0 / 1 gate keeps; a batch of rows is then
// kRows KEPT rows (the masked form spends a batch's issue slots on its dead rows too).
template <int K, int VW, bool LIST = false>
__global__ __launch_bounds__(kBlock) void l0_edge_fwd_k(const FwdArgs p) {
  constexpr int WPB = kBlock / 64, kRows = kFwdRows, H = 64 * VW;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  Vec<VW> ma[K], mb[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { ma[k] = vload<VW>(p.M + k * p.ldm, lane); mb[k] = vload<VW>(p.M + k * p.ldm + H, lane); }
  Vec<VW> bias = vzero<VW>();
  if (p.bias) bias = vload<VW>(p.bias, lane);
  const int64_t stride = (int64_t)gridDim.x * WPB * kRows;
  // the row's scalars: lanes 0..K-1 its label code, three more lanes its coefficient and its two node rows -- one load
  // instruction per row, each lane with its own base and step
  const char *base = reinterpret_cast<const char *>(p.enc + lane);
  int64_t step = p.lde * 4;
  if (lane == kLaneCoef) { base = reinterpret_cast<const char *>(p.coef_e); step = 4; }
  if (lane == kLaneA) { base = reinterpret_cast<const char *>(p.sel_a); step = 4; }
  if (lane == kLaneB) { base = reinterpret_cast<const char *>(p.sel_b); step = 4; }
  const bool on = lane < K || (lane >= kLaneCoef && lane <= kLaneB);
  static_assert(32 % kRows == 0, "a batch of rows lies inside one mask word");
  const int64_t limit = LIST ? (int64_t)*p.count : p.R;    // batches of rows: positions in the list / rows
  auto fetch = [&](int64_t r0, float (&mm)[kRows], uint32_t &lv, int (&id)[kRows]) {
    // the batch's row-mask bits with its scalars, a whole batch ahead of their use (r0 is a multiple of kRows: one word)
    lv = (!LIST && p.rowmask && r0 < p.R) ? (p.rowmask[r0 >> 5] >> (r0 & 31)) : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t q = r0 + u;                             // wave-uniform
      const int64_t r = LIST ? (q < limit ? (int64_t)p.list[q] : -1) : q;
      float m = 0.f;                                        // rows past the end: code 0, node rows 0 (loaded, never stored)
      if (on && (LIST ? r >= 0 : r < p.R)) m = *reinterpret_cast<const float *>(base + r * step);
      mm[u] = m;
      id[u] = (int)r;
    }
  };
  // Measured (knob builds of round 3): 135 us at E = 549 k, of which the scalars + epilogue 25, the FMAs 30, the
  // stores 32-49 and the two gathers 30-41 -- they add up whatever the occupancy (3..6 waves per SIMD), the rows per batch,
  // a two-batch software pipeline with the gathers issued ahead of the stores, contiguous row runs per workgroup or an
  // XCD-aware order (all within 135-145 us); PMC: 281 MB written, 260 MB fetched (the node rows 3 x).
  float mine[kRows], next[kRows];
  int rid[kRows], ridn[kRows];
  uint32_t lv = 0xffffffffu, lvn = 0xffffffffu;
  int64_t r0 = ((int64_t)blockIdx.x * WPB + wave) * kRows;
  if (r0 < limit) fetch(r0, mine, lv, rid);
  for (; r0 < limit; r0 += stride) {
    fetch(r0 + stride, next, lvn, ridn);                    // the next batch's scalars: a dependent round trip less per batch
    Vec<VW> pa[kRows], pb[kRows];
    // a dead row (mask bit 0) gathers node row 0 instead of its own two (one cached line: no traffic, and no branch in the
    // load / compute sequence -- control flow here costs the loop its overlapped loads: measured 120 -> 170 us) and is not stored
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const bool live = (lv >> u) & 1u;                      // wave-uniform: scalar selects
      const int ia = live ? lane_i(mine[u], kLaneA) : 0, ib = live ? lane_i(mine[u], kLaneB) : 0;
      pa[u] = vload<VW>(p.P + (int64_t)ia * p.ldp, lane);
      pb[u] = vload<VW>(p.P + (int64_t)ib * p.ldp + H, lane);
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (r0 + u >= limit) break;
      const float cf = lane_f(mine[u], kLaneCoef);
      Vec<VW> g0 = vzero<VW>(), g1 = vzero<VW>();
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float x = lane_f(mine[u], k);
        fmac_s(g0, x, ma[k]);
        fmac_s(g1, x, mb[k]);
      }
      Vec<VW> t;
#pragma unroll
      for (int c = 0; c < VW; ++c) {
        // ((G0 + coef G1) + (P[a] - P[b])) + bias: the order of the general layer (dmp_agg.hip::edge_combine)
        float y = at<VW>(g1, c) * cf;
        y += at<VW>(g0, c);
        y += at<VW>(pa[u], c) - at<VW>(pb[u], c);
        y += at<VW>(bias, c);
        at<VW>(t, c) = act_fwd(y, p.slope);
      }
      if ((lv >> u) & 1u) vstore<VW>(p.out + (LIST ? (int64_t)rid[u] : r0 + u) * p.ldo, lane, t);
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) { mine[u] = next[u]; rid[u] = ridn[u]; }
    lv = lvn;
  }
}

struct BwdArgs {
  const float *enc; int64_t lde; const float *coef_e;
  const float *dPre; int64_t ldd;          // [R, >= H]
  const float *dZn; int64_t ldz;           // [R, >= H] or NULL (no residual connection)
  int64_t R; float *partial;               // [blocks, K, (dZn ? 3 : 2) * H]
  const uint32_t *rowmask;                 // bit r of rowmask[t] == 0: row 32 t + r has an all-zero code row (a zero gate went into
                                           // l0_pack): its dPre / dZn rows are not fetched (they would be multiplied by zeros).  NULL: all
  const int32_t *list, *count;             // LIST form (see l0_edge_fwd_k): the rows to add, *count of them
};

template <int K, int VW, bool RES, bool LIST = false>
__global__ __launch_bounds__(kBlock) void l0_bwd_w_k(const BwdArgs p) {
  constexpr int WPB = kBlock / 64, kRows = kBwdNoPrefetch ? 8 : kL0Rows, H = 64 * VW, NACC = (RES ? 3 : 2) * K;
  __shared__ float red4[(kBwdWide ? kBlock / 64 : 1) * kBlock * VW];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  Vec<VW> acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = vzero<VW>();
  const int64_t stride = (int64_t)gridDim.x * WPB * kRows;
  Vec<VW> d[kRows], dn[kRows], z[RES ? kRows : 1], zn[RES ? kRows : 1];
  float mine[kRows], minen[kRows];
  const int64_t limit = LIST ? (int64_t)*p.count : p.R;
  auto load_batch = [&](int64_t r0, Vec<VW> (&dd)[kRows], Vec<VW> (&zz)[RES ? kRows : 1], float (&mm)[kRows]) {
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t q = r0 + u;                             // wave-uniform
      const int64_t r = LIST ? (q < limit ? (int64_t)p.list[q] : 0) : q;
      const bool ok = q < limit;
      const bool live = ok && (LIST || !p.rowmask || ((p.rowmask[r >> 5] >> (r & 31)) & 1u));   // scalar load, scalar branch
      if (live) {
        dd[u] = vload<VW>(p.dPre + r * p.ldd, lane);
        if (RES) zz[u] = vload<VW>(p.dZn + r * p.ldz, lane);
      } else {
        dd[u] = vzero<VW>();
        if (RES) zz[u] = vzero<VW>();
      }
      const float *src = lane == kLaneCoef ? p.coef_e + r : p.enc + r * p.lde + lane;
      float m = 0.f;
      if (ok && (lane < K || lane == kLaneCoef)) m = *src;
      mm[u] = m;
    }
  };
  int64_t r0 = ((int64_t)blockIdx.x * WPB + wave) * kRows;
  if (!kBwdNoPrefetch && r0 < limit) load_batch(r0, d, z, mine);
  for (; r0 < limit; r0 += stride) {
    if (kBwdNoPrefetch) load_batch(r0, d, z, mine);
    else load_batch(r0 + stride, dn, zn, minen);            // rows past the end read as zeros
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (!LIST && p.rowmask) {                              // an all-zero code row: nothing to add (wave-uniform test)
        const int64_t r = r0 + u;
        if (r < p.R && !((p.rowmask[r >> 5] >> (r & 31)) & 1u)) continue;
      }
      const float cf = lane_f(mine[u], kLaneCoef);
      Vec<VW> cd;                                           // c_r dPre[r]: (c enc)^T dPre = enc^T (c dPre), one scalar per input
#pragma unroll
      for (int c = 0; c < VW; ++c) at<VW>(cd, c) = at<VW>(d[u], c) * cf;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float x = lane_f(mine[u], k);
        fmac_s(acc[k], x, d[u]);
        fmac_s(acc[K + k], x, cd);
        if (RES) fmac_s(acc[2 * K + k], x, z[u]);
      }
    }
    if (!kBwdNoPrefetch) {
#pragma unroll
      for (int u = 0; u < kRows; ++u) { d[u] = dn[u]; mine[u] = minen[u]; if (RES) z[u] = zn[u]; }
    }
  }
  // fixed-order combine of the 4 waves.  kBwdWide: four accumulators per round, wave w sums (and stores) the w-th of them.
  if constexpr (kBwdWide) {
#pragma unroll
    for (int k0 = 0; k0 < NACC; k0 += WPB) {
#pragma unroll
      for (int j = 0; j < WPB; ++j)
        if (k0 + j < NACC) {
#pragma unroll
          for (int c = 0; c < VW; ++c) red4[((j * WPB + wave) * 64 + lane) * VW + c] = at<VW>(acc[k0 + j], c);
        }
      __syncthreads();
      const int k = k0 + wave;
      if (k < NACC) {
        Vec<VW> t;
#pragma unroll
        for (int c = 0; c < VW; ++c) {
          float a = red4[((wave * WPB) * 64 + lane) * VW + c];
#pragma unroll
          for (int w = 1; w < WPB; ++w) a += red4[((wave * WPB + w) * 64 + lane) * VW + c];
          at<VW>(t, c) = a;
        }
        vstore<VW>(p.partial + ((int64_t)blockIdx.x * NACC + (k % K) * (NACC / K) + k / K) * H, lane, t);
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
#pragma unroll
      for (int c = 0; c < VW; ++c) red4[threadIdx.x * VW + c] = at<VW>(acc[k], c);
      __syncthreads();
      if (wave == 0) {
        Vec<VW> t;
#pragma unroll
        for (int c = 0; c < VW; ++c) {
          float a = red4[lane * VW + c];
#pragma unroll
          for (int w = 1; w < WPB; ++w) a += red4[(w * 64 + lane) * VW + c];
          at<VW>(t, c) = a;
        }
        // row k of the [K, (2 or 3) H] result: [enc^T dPre | (c enc)^T dPre | enc^T dZn]
        vstore<VW>(p.partial + ((int64_t)blockIdx.x * NACC + (k % K) * (NACC / K) + k / K) * H, lane, t);
      }
      __syncthreads();
    }
  }
}

inline unsigned l0_blocks(int64_t R, int64_t cap) {
  const int64_t chunk = (int64_t)(kBlock / 64) * (kBwdNoPrefetch ? 8 : kL0Rows), nb = (R + chunk - 1) / chunk;
  return (unsigned)(nb < cap ? (nb > 0 ? nb : 1) : cap);
}

template <int K>
void launch_fwd(const FwdArgs &p, int H, hipStream_t st) {
  const int64_t chunk = (int64_t)(kBlock / 64) * kFwdRows, want = (p.R + chunk - 1) / chunk;
  const unsigned nb = (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
  if (p.list) {
    if (H == 128) l0_edge_fwd_k<K, 2, true><<<nb, kBlock, 0, st>>>(p);
    else l0_edge_fwd_k<K, 1, true><<<nb, kBlock, 0, st>>>(p);
  } else if (H == 128) l0_edge_fwd_k<K, 2><<<nb, kBlock, 0, st>>>(p);
  else l0_edge_fwd_k<K, 1><<<nb, kBlock, 0, st>>>(p);
}

template <int K>
void launch_bwd(const BwdArgs &p, int H, hipStream_t st) {
  const unsigned nb = l0_blocks(p.R, kL0MaxPartials);
  if (p.list) {
    if (H == 128) { if (p.dZn) l0_bwd_w_k<K, 2, true, true><<<nb, kBlock, 0, st>>>(p); else l0_bwd_w_k<K, 2, false, true><<<nb, kBlock, 0, st>>>(p); }
    else { if (p.dZn) l0_bwd_w_k<K, 1, true, true><<<nb, kBlock, 0, st>>>(p); else l0_bwd_w_k<K, 1, false, true><<<nb, kBlock, 0, st>>>(p); }
    return;
  }
  if (H == 128) { if (p.dZn) l0_bwd_w_k<K, 2, true><<<nb, kBlock, 0, st>>>(p); else l0_bwd_w_k<K, 2, false><<<nb, kBlock, 0, st>>>(p); }
  else { if (p.dZn) l0_bwd_w_k<K, 1, true><<<nb, kBlock, 0, st>>>(p); else l0_bwd_w_k<K, 1, false><<<nb, kBlock, 0, st>>>(p); }
}

// ---- the rows whose mask bit is set, ascending (dmp_kept_rows): per-block counts, then every block ranks and writes its rows
constexpr int kKeptWords = 256;          // mask words per block
__global__ __launch_bounds__(kKeptWords) void kept_count_k(const uint32_t *__restrict__ mask, int64_t R, int32_t *__restrict__ blk) {
  __shared__ int red[kKeptWords / 64];
  const int64_t W = (R + 31) / 32, w = (int64_t)blockIdx.x * kKeptWords + threadIdx.x;
  uint32_t m = w < W ? mask[w] : 0u;
  if (w == W - 1 && (R & 31)) m &= (1u << (R & 31)) - 1u;      // bits past the last row do not count
  int c = __popc(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int i = 0; i < kKeptWords / 64; ++i) t += red[i]; blk[blockIdx.x] = t; }
}
// TILES: the list doubles as a slot list of 32-row tiles for the tile kernels (csrc/dmp_typed.hip): the entries from *count up
// to the next multiple of 32 are set to -1 (padding slots) and count[1] = the number of tiles, (*count + 31) / 32.
template <bool TILES>
__global__ __launch_bounds__(kKeptWords) void kept_fill_k(const uint32_t *__restrict__ mask, int64_t R, const int32_t *__restrict__ blk,
                                                           int32_t *__restrict__ list, int32_t *__restrict__ count) {
  __shared__ int red[kKeptWords / 64], wtot[kKeptWords / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows kept by the blocks before this one (a few dozen to a few hundred words out of L2); the last block also has the total
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kKeptWords) before += blk[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) red[wave] = before;
  __syncthreads();
  before = 0;
  for (int i = 0; i < kKeptWords / 64; ++i) before += red[i];
  const int64_t W = (R + 31) / 32, w = (int64_t)blockIdx.x * kKeptWords + threadIdx.x;
  uint32_t m = w < W ? mask[w] : 0u;
  if (w == W - 1 && (R & 31)) m &= (1u << (R & 31)) - 1u;
  const int c = __popc(m);
  int incl = c;                                                // inclusive scan inside the wave, then over the waves
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int pos = before + incl - c;
  for (int i = 0; i < wave; ++i) pos += wtot[i];
  const int base = (int)(w * 32);
  while (m) { const int b = __builtin_ctz(m); m &= m - 1; list[pos++] = base + b; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kKeptWords - 1) {   // (the last thread's pos: everything before it + its own)
    *count = pos;
    if (TILES) {
      for (int i = pos; i < ((pos + 31) & ~31); ++i) list[i] = -1;
      count[1] = (pos + 31) >> 5;
    }
  }
}

// The first layer's NODE side from the label codes, one pass:  with node rows x = venc WV0 and edge rows z0 = enc W0 (so a
// node's aggregates are S_h = S0_h W0, S0_h = the sums of its in / out edges' codes) everything the layer computes per node
// before its second Linear is linear in ~36 code columns per row:
//     H1n[r] = act( venc[r] (WV0 Wx_0) + S0_in[r] (W0 Bn_in) + S0_out[r] (W0 Bn_out) + bn )        blockIdx.y == 0
//     P[r]   = venc[r] (WV0 [Wx_1 | Wx_2])                                                         blockIdx.y == 1, 2
// (dmpnn.py:113,121,125,129-140) -- instead of four N-row library products, an N x 3H product from the codes and a row pass
// (111 us of launches at N = 73 k).  A wave owns a row at a time; the row's codes are wave-uniform (one load per lane, read
// back by v_readlane as SGPR operands), the matrices' rows live in registers.  ``rowmask`` (the kept nodes of a 0 / 1 node
// gate): a dead row's H1n is left unwritten (every consumer walks the kept nodes' tiles); its P rows are written as ZEROS
// (its code row is zero) -- the first layer's edge kernel gathers P through the plain selectors.
constexpr int kNodeKT = 40;             // code columns per row the registers hold: VK + 2 K0 (the packed matrix has kNodeKT rows)
constexpr int kNodeRows = 16;           // rows in flight per wave
struct NodeFwdArgs {
  const float *venc; int64_t ldv; int VK;
  const float *S0; int64_t lds; int K0, Kp;
  const float *W; int64_t ldw;            // [kNodeKT, 3H] packed: rows 0 .. VK-1 = WV0 Wx (all three blocks), then K0 rows W0 Bn_in and K0 rows
                                          // W0 Bn_out in the FIRST block only; every other entry zero
  const float *bias; float slope;
  const uint32_t *rowmask;                // bit r of word t: node 32 t + r is kept (absolute node ids); NULL: all
  const int32_t *list, *count;            // the kept nodes' ids, ascending (dmp_kept_rows over ALL nodes), *count of them; NULL: rows n0 .. n1-1
  int64_t n0, n1;                         // the node rows of this table
  float *h1; int64_t ldh; float *P; int64_t ldp;
  int64_t q_begin;                        // list form: the first list position that can hold a row of this table (host lower bound)
};

// blockIdx.y: 0 = H1n, 1 / 2 = the projection blocks, 3 = zero rows of P for the dead nodes (with a mask).  Under a list the
// launch walks the list's positions from q_begin on, sixteen per wave, and acts on the rows of its table (n0 <= r < n1): every
// batch is sixteen LIVE rows.  A wave's life is five dependent round trips (count, list, codes + matrix, stores) whatever it
// does, so the launch lasts (waves / resident waves) lifetimes: measured stage by stage at bench.py's shape, a launch that
// walked every list position with eight rows per wave spent 12 of its 43 us in waves that found no row of their table and ran
// 2-5 rounds of waves; the fixed kNodeKT-row matrix (zero rows past the columns in use) keeps the instruction stream free of
// branches (a run-time column count: ~2,000 scalar instructions of address selection per wave).
template <int VW>
__global__ __launch_bounds__(kBlock) void l0_node_fwd_k(const NodeFwdArgs p) {
  constexpr int WPB = kBlock / 64, kRows = kNodeRows, H = 64 * VW;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int blk = blockIdx.y;
  if (blk == 3) {                                                 // a dead node's projections: zeros (its code row is zero)
    if (!p.rowmask) return;
    const Vec<VW> zero = vzero<VW>();
    // a wave per mask word: one load, then stores only (no wait between them)
    const int64_t w0 = p.n0 >> 5, w1 = (p.n1 + 31) >> 5;
    for (int64_t wi = w0 + (int64_t)blockIdx.x * WPB + wave; wi < w1; wi += (int64_t)gridDim.x * WPB) {
      uint32_t deadbits = ~p.rowmask[wi];
      deadbits = __builtin_amdgcn_readfirstlane(deadbits);
      while (deadbits) {
        const int b = __builtin_ctz(deadbits);
        deadbits &= deadbits - 1;
        const int64_t r = wi * 32 + b;
        if (r < p.n0 || r >= p.n1) continue;
        vstore<VW>(p.P + r * p.ldp, lane, zero);
        vstore<VW>(p.P + r * p.ldp + H, lane, zero);
      }
    }
    return;
  }
  // positions of the list (a row below n1 cannot sit at position n1 or later) / rows of the range
  const int64_t limit = p.list ? ((int64_t)*p.count < p.n1 ? (int64_t)*p.count : p.n1) : p.n1 - p.n0;
  const int64_t stride = (int64_t)gridDim.x * WPB * kRows;
  int64_t q0 = p.q_begin + ((int64_t)blockIdx.x * WPB + wave) * kRows;
  if (q0 >= limit) return;
  Vec<VW> w[kNodeKT];
#pragma unroll
  for (int k = 0; k < kNodeKT; ++k) w[k] = vload<VW>(p.W + k * p.ldw + (int64_t)blk * H, lane);
  Vec<VW> bias = vzero<VW>();
  if (blk == 0 && p.bias) bias = vload<VW>(p.bias, lane);
  // this lane's code column: lanes 0 .. VK-1 the node's own code, the next K0 its in-sums, the next K0 its out-sums
  const float *base = p.venc + lane;
  int64_t step = p.ldv;
  if (lane >= p.VK) { base = p.S0 + (lane - p.VK < p.K0 ? lane - p.VK : p.Kp + lane - p.VK - p.K0); step = p.lds; }
  const bool on = lane < p.VK + 2 * p.K0;
  float *const outb = blk == 0 ? p.h1 : p.P + (int64_t)(blk - 1) * H;
  const int64_t ldo = blk == 0 ? p.ldh : p.ldp;
  for (; q0 < limit; q0 += stride) {
    // (every load below is issued unconditionally at a clamped index and its result selected afterwards: a wave-uniform
    // condition around a load is a scalar branch -- a basic block and a wait per row)
    int rid[kRows], raw[kRows];
    uint32_t word[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t qc = q0 + u < limit ? q0 + u : limit - 1;     // (limit >= 1 here)
      raw[u] = p.list ? p.list[qc] : (int)(p.n0 + qc);
    }
    if (!p.list && p.rowmask) {
#pragma unroll
      for (int u = 0; u < kRows; ++u) word[u] = p.rowmask[raw[u] >> 5];
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      int r = raw[u];
      const bool dead = !p.list && p.rowmask && !((word[u] >> (r & 31)) & 1u);
      if (q0 + u >= limit || r < p.n0 || r >= p.n1 || dead) r = -1;      // (a list over all nodes: the other table's rows)
      rid[u] = __builtin_amdgcn_readfirstlane(r);
    }
    float mine[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const float m = on ? base[(int64_t)(rid[u] >= 0 ? rid[u] : (int)p.n0) * step] : 0.f;
      mine[u] = rid[u] >= 0 ? m : 0.f;
    }
    // all eight code loads land HERE, before the first store: gfx9 counts loads and stores in one in-order counter, so a wait
    // for a later row's codes placed after an earlier row's store also waits for that store's acknowledgement
    static_assert(kRows == 16, "the pins below list sixteen values");
    asm volatile("" : "+v"(mine[0]), "+v"(mine[1]), "+v"(mine[2]), "+v"(mine[3]), "+v"(mine[4]), "+v"(mine[5]), "+v"(mine[6]), "+v"(mine[7]));
    asm volatile("" : "+v"(mine[8]), "+v"(mine[9]), "+v"(mine[10]), "+v"(mine[11]), "+v"(mine[12]), "+v"(mine[13]), "+v"(mine[14]), "+v"(mine[15]));
    // four rows' sums side by side: a row's kNodeKT multiply-adds are one dependent chain (and every scalar operand a
    // v_readlane the next instruction waits for) -- 6.4 us for 8 rows one after the other
#pragma unroll
    for (int u0 = 0; u0 < kRows; u0 += 4) {
      if (rid[u0] < 0 && rid[u0 + 1] < 0 && rid[u0 + 2] < 0 && rid[u0 + 3] < 0) continue;
      Vec<VW> e[4] = {bias, bias, bias, bias};
#pragma unroll
      for (int k = 0; k < kNodeKT; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fmac_s(e[j], lane_f(mine[u0 + j], k), w[k]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (rid[u0 + j] < 0) continue;
        if (blk == 0) {
#pragma unroll
          for (int c = 0; c < VW; ++c) at<VW>(e[j], c) = act_fwd(at<VW>(e[j], c), p.slope);
        }
        vstore<VW>(outb + (int64_t)rid[u0 + j] * ldo, lane, e[j]);
      }
    }
  }
}

#define L0_SWITCH(K, CALL)                                                                          \
  switch (K) {                                                                                      \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; \
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; default: CALL(16); break; \
  }

inline bool al8(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; }

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_l0_pack_jobs(const dmp_l0_pack_job *jobs, int num_jobs, void *stream) {
  if (!jobs || num_jobs < 1 || num_jobs > 2) return DMP_ERR_BAD_ARG;
  PackJobs js{};
  int64_t most = 0;
  int n = 0;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_l0_pack_job &q = jobs[i];
    if (q.rows_p < 0 || q.rows_g < 0 || q.K <= 0 || q.goff < 0 || q.Kpad < q.goff + q.K || !q.out) return DMP_ERR_BAD_ARG;
    if ((q.rows_p > 0 && (!q.enc_p || q.ldp < q.K)) || (q.rows_g > 0 && (!q.enc_g || q.ldg < q.K))) return DMP_ERR_BAD_ARG;
    const int64_t total = q.rows_p + q.rows_g;             // a thread per row
    if (total == 0) continue;
    if (q.Kpad % 4 || (reinterpret_cast<uintptr_t>(q.out) & 15u)) return DMP_ERR_UNSUPPORTED;
    js.job[n++] = PackArgs{q.enc_p, q.ldp, q.rows_p, q.enc_g, q.ldg, q.gate, q.rows_g, q.K, q.Kpad, q.goff, q.out};
    if (total > most) most = total;
  }
  if (n == 0) return DMP_OK;
  const int64_t nb = (most + kBlock - 1) / kBlock;
  l0_pack_k<<<dim3((unsigned)(nb < 8192 ? nb : 8192), (unsigned)n), kBlock, 0, (hipStream_t)stream>>>(js);
  return check_launch();
}

int dmp_l0_pack(const float *enc_p, int64_t ldp, int64_t rows_p, const float *enc_g, int64_t ldg, const float *gate,
                int64_t rows_g, int K, int Kpad, int goff, float *out, void *stream) {
  const dmp_l0_pack_job j{enc_p, ldp, rows_p, enc_g, ldg, gate, rows_g, K, Kpad, goff, out};
  return dmp_l0_pack_jobs(&j, 1, stream);
}

int dmp_l0_edge_fwd_masked(const float *enc, int64_t lde, int K, const float *M, int64_t ldm, const float *P, int64_t ldp,
                           const float *bias, const float *coef_e, const int32_t *sel_a, const int32_t *sel_b,
                           const uint32_t *rowmask, int64_t R, int H, float slope, float *out, int64_t ldo, void *stream) {
  if (R < 0 || K <= 0) return DMP_ERR_BAD_ARG;
  if ((H != 128 && H != 64) || K > kL0K || !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return DMP_OK;
  if (!enc || !M || !P || !coef_e || !sel_a || !sel_b || !out || lde < K || ldm < 2 * H || ldp < 2 * H || ldo < H) return DMP_ERR_BAD_ARG;
  if (ldm % 2 || ldp % 2 || ldo % 2 || !al8(M) || !al8(P) || !al8(out) || !al8(bias)) return DMP_ERR_UNSUPPORTED;
  FwdArgs p{enc, lde, M, ldm, P, ldp, bias, coef_e, sel_a, sel_b, R, slope, out, ldo, rowmask, nullptr, nullptr};
  hipStream_t st = (hipStream_t)stream;
#define L0_CALL(KK) launch_fwd<KK>(p, H, st)
  L0_SWITCH(K, L0_CALL)
#undef L0_CALL
  return check_launch();
}

// ---- several kept-row lists in ONE pair of launches (dmp_kept_rows_jobs): blockIdx.y = the job; and, riding in the count
// launch, the per-edge selectors with dead nodes as -1 (dmp_edge_select_nodes: needs the node mask only, as the lists do)
struct KeptJobs {
  const uint32_t *mask[DMP_KEPT_MAX_JOBS]; int64_t R[DMP_KEPT_MAX_JOBS]; int tiles[DMP_KEPT_MAX_JOBS]; int nb[DMP_KEPT_MAX_JOBS];
  int32_t *scratch[DMP_KEPT_MAX_JOBS], *list[DMP_KEPT_MAX_JOBS], *count[DMP_KEPT_MAX_JOBS];
  int n;
  const int32_t *src, *dst; const uint8_t *flag; const uint32_t *nodemask; int64_t E; int32_t *selA, *selB, *dstM;
};
__global__ __launch_bounds__(kKeptWords) void kept_count_jobs_k(const KeptJobs t) {
  const int j = blockIdx.y;
  if (j == t.n) {                                   // the selectors: a grid-stride pass over the edges
    for (int64_t e = (int64_t)blockIdx.x * kKeptWords + threadIdx.x; e < t.E; e += (int64_t)gridDim.x * kKeptWords) {
      int u = t.src[e], v = t.dst[e];
      if (!((t.nodemask[u >> 5] >> (u & 31)) & 1u)) u = -1;
      if (!((t.nodemask[v >> 5] >> (v & 31)) & 1u)) v = -1;
      const bool f = t.flag && t.flag[e];
      t.selA[e] = f ? u : v;
      t.selB[e] = f ? v : u;
      t.dstM[e] = v;
    }
    return;
  }
  if ((int)blockIdx.x >= t.nb[j]) return;
  __shared__ int red[kKeptWords / 64];
  const int64_t R = t.R[j], W = (R + 31) / 32, w = (int64_t)blockIdx.x * kKeptWords + threadIdx.x;
  uint32_t m = w < W ? t.mask[j][w] : 0u;
  if (w == W - 1 && (R & 31)) m &= (1u << (R & 31)) - 1u;
  int c = __popc(m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int s = 0; for (int i = 0; i < kKeptWords / 64; ++i) s += red[i]; t.scratch[j][blockIdx.x] = s; }
}
__global__ __launch_bounds__(kKeptWords) void kept_fill_jobs_k(const KeptJobs t) {
  const int j = blockIdx.y;
  if ((int)blockIdx.x >= t.nb[j]) return;
  __shared__ int red[kKeptWords / 64], wtot[kKeptWords / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t *blk = t.scratch[j];
  int32_t *list = t.list[j], *count = t.count[j];
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kKeptWords) before += blk[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) red[wave] = before;
  __syncthreads();
  before = 0;
  for (int i = 0; i < kKeptWords / 64; ++i) before += red[i];
  const int64_t R = t.R[j], W = (R + 31) / 32, w = (int64_t)blockIdx.x * kKeptWords + threadIdx.x;
  uint32_t m = w < W ? t.mask[j][w] : 0u;
  if (w == W - 1 && (R & 31)) m &= (1u << (R & 31)) - 1u;
  const int c = __popc(m);
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int pos = before + incl - c;
  for (int i = 0; i < wave; ++i) pos += wtot[i];
  const int base = (int)(w * 32);
  while (m) { const int b = __builtin_ctz(m); m &= m - 1; list[pos++] = base + b; }
  if ((int)blockIdx.x == t.nb[j] - 1 && threadIdx.x == kKeptWords - 1) {
    *count = pos;
    if (t.tiles[j]) {
      for (int i = pos; i < ((pos + 31) & ~31); ++i) list[i] = -1;
      count[1] = (pos + 31) >> 5;
    }
  }
}

int dmp_kept_rows_jobs(const dmp_kept_job *jobs, int n, const int32_t *src, const int32_t *dst, const uint8_t *flag,
                       const uint32_t *nodemask, int64_t E, int32_t *selA, int32_t *selB, int32_t *dstM, void *stream) {
  if (n < 0 || n > DMP_KEPT_MAX_JOBS || (n > 0 && !jobs) || E < 0) return DMP_ERR_BAD_ARG;
  const bool sel = selA != nullptr;
  if (sel && (E > 0) && (!src || !dst || !nodemask || !selB || !dstM)) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  KeptJobs t{};
  int most = 0;
  t.n = n;
  for (int j = 0; j < n; ++j) {
    const dmp_kept_job &q = jobs[j];
    if (q.R <= 0 || q.R >= ((int64_t)1 << 31) - 32 || !q.mask || !q.scratch || !q.list || !q.count) return DMP_ERR_BAD_ARG;   // (empty lists: dmp_kept_rows)
    t.mask[j] = q.mask; t.R[j] = q.R; t.tiles[j] = q.tiles ? 1 : 0; t.scratch[j] = q.scratch; t.list[j] = q.list; t.count[j] = q.count;
    t.nb[j] = (int)(((q.R + 31) / 32 + kKeptWords - 1) / kKeptWords);
    if (t.nb[j] > most) most = t.nb[j];
  }
  t.src = src; t.dst = dst; t.flag = flag; t.nodemask = nodemask; t.E = sel ? E : 0; t.selA = selA; t.selB = selB; t.dstM = dstM;
  const bool with_sel = sel && E > 0;
  if (n == 0 && !with_sel) return DMP_OK;
  int gx = most;
  if (with_sel) {
    const int64_t eb = (E + kKeptWords - 1) / kKeptWords;
    const int want = (int)(eb < 1024 ? eb : 1024);
    if (want > gx) gx = want;
  }
  kept_count_jobs_k<<<dim3((unsigned)gx, (unsigned)(n + (with_sel ? 1 : 0))), kKeptWords, 0, st>>>(t);
  if (n > 0) kept_fill_jobs_k<<<dim3((unsigned)most, (unsigned)n), kKeptWords, 0, st>>>(t);
  return check_launch();
}

int64_t dmp_kept_rows_scratch_words(int64_t R) { return ((R + 31) / 32 + kKeptWords - 1) / kKeptWords + 1; }

int dmp_kept_rows(const uint32_t *rowmask, int64_t R, int tiles, int32_t *scratch, int32_t *list, int32_t *count, void *stream) {
  if (R < 0 || !count) return DMP_ERR_BAD_ARG;
  if (R >= ((int64_t)1 << 31) - 32) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (R == 0) return hipMemsetAsync(count, 0, (tiles ? 2 : 1) * sizeof(int32_t), st) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!rowmask || !scratch || !list) return DMP_ERR_BAD_ARG;
  const unsigned nb = (unsigned)(((R + 31) / 32 + kKeptWords - 1) / kKeptWords);
  kept_count_k<<<nb, kKeptWords, 0, st>>>(rowmask, R, scratch);
  if (tiles) kept_fill_k<true><<<nb, kKeptWords, 0, st>>>(rowmask, R, scratch, list, count);
  else kept_fill_k<false><<<nb, kKeptWords, 0, st>>>(rowmask, R, scratch, list, count);
  return check_launch();
}

int dmp_l0_edge_fwd_rows(const float *enc, int64_t lde, int K, const float *M, int64_t ldm, const float *P, int64_t ldp,
                         const float *bias, const float *coef_e, const int32_t *sel_a, const int32_t *sel_b,
                         const int32_t *list, const int32_t *count, int64_t R, int H, float slope, float *out, int64_t ldo, void *stream) {
  if (R < 0 || K <= 0 || !list || !count) return DMP_ERR_BAD_ARG;
  if ((H != 128 && H != 64) || K > kL0K || !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return DMP_OK;
  if (!enc || !M || !P || !coef_e || !sel_a || !sel_b || !out || lde < K || ldm < 2 * H || ldp < 2 * H || ldo < H) return DMP_ERR_BAD_ARG;
  if (ldm % 2 || ldp % 2 || ldo % 2 || !al8(M) || !al8(P) || !al8(out) || !al8(bias)) return DMP_ERR_UNSUPPORTED;
  FwdArgs p{enc, lde, M, ldm, P, ldp, bias, coef_e, sel_a, sel_b, R, slope, out, ldo, nullptr, list, count};
  hipStream_t st = (hipStream_t)stream;
#define L0_CALL(KK) launch_fwd<KK>(p, H, st)
  L0_SWITCH(K, L0_CALL)
#undef L0_CALL
  return check_launch();
}

int dmp_l0_bwd_w_rows(const float *enc, int64_t lde, int K, const float *coef_e, const float *dPre, int64_t ldd, const float *dZn,
                      int64_t ldz, const int32_t *list, const int32_t *count, int64_t R, int H, float *partial, void *stream) {
  if (R < 0 || K <= 0 || !partial || !list || !count) return DMP_ERR_BAD_ARG;
  if ((H != 128 && H != 64) || K > kL0K) return DMP_ERR_UNSUPPORTED;
  const int nacc = (dZn ? 3 : 2) * K;
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)nacc * H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!enc || !coef_e || !dPre || lde < K || ldd < H || (dZn && ldz < H)) return DMP_ERR_BAD_ARG;
  if (ldd % 2 || ldz % 2 || !al8(dPre) || !al8(dZn) || !al8(partial)) return DMP_ERR_UNSUPPORTED;
  BwdArgs p{enc, lde, coef_e, dPre, ldd, dZn, ldz, R, partial, nullptr, list, count};
  hipStream_t st = (hipStream_t)stream;
#define L0_CALL(KK) launch_bwd<KK>(p, H, st)
  L0_SWITCH(K, L0_CALL)
#undef L0_CALL
  return check_launch();
}

int64_t dmp_l0_node_pack_rows(void) { return kNodeKT; }

int dmp_l0_node_fwd(const float *venc, int64_t ldv, int VK, const float *S0, int64_t lds, int K0, int Kp, const float *W, int64_t ldw,
                    const float *bias, float slope, const uint32_t *rowmask, const int32_t *list, const int32_t *count, int64_t list_bound,
                    int64_t q_begin, int64_t n0, int64_t n1, int H, float *h1, int64_t ldh, float *P, int64_t ldp, void *stream) {
  if (n0 < 0 || n1 < n0 || VK <= 0 || K0 <= 0 || Kp < K0 || (list && (!count || !rowmask || list_bound < 0 || q_begin < 0 || q_begin > n0)))
    return DMP_ERR_BAD_ARG;
  if (n1 >= ((int64_t)1 << 31)) return DMP_ERR_UNSUPPORTED;
  if ((H != 128 && H != 64) || VK + 2 * K0 > kNodeKT || !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (n1 == n0) return DMP_OK;
  if (!venc || !S0 || !W || !h1 || !P || ldv < VK || lds < 2 * Kp || ldw < 3 * H || ldh < H || ldp < 2 * H) return DMP_ERR_BAD_ARG;
  if (ldw % 2 || ldh % 2 || ldp % 2 || !al8(W) || !al8(h1) || !al8(P) || !al8(bias)) return DMP_ERR_UNSUPPORTED;
  NodeFwdArgs p{venc, ldv, VK, S0, lds, K0, Kp, W, ldw, bias, slope, rowmask, list, count, n0, n1, h1, ldh, P, ldp, list ? q_begin : 0};
  // (list form: the positions q_begin .. that can hold rows below n1: at most n1 - q_begin of them, and no more than the list has)
  const int64_t chunk = (int64_t)(kBlock / 64) * kNodeRows;
  const int64_t walk = list ? ((list_bound < n1 ? list_bound : n1) - q_begin > 0 ? (list_bound < n1 ? list_bound : n1) - q_begin : 0) : n1 - n0;
  const int64_t nb = (walk + chunk - 1) / chunk;
  const dim3 grid((unsigned)(nb < 4096 ? (nb > 0 ? nb : 1) : 4096), rowmask ? 4u : 3u);
  if (H == 128) l0_node_fwd_k<2><<<grid, kBlock, 0, (hipStream_t)stream>>>(p);
  else l0_node_fwd_k<1><<<grid, kBlock, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int64_t dmp_l0_bwd_w_blocks(int64_t rows) { return (int64_t)l0_blocks(rows, kL0MaxPartials); }

int dmp_l0_bwd_w_masked(const float *enc, int64_t lde, int K, const float *coef_e, const float *dPre, int64_t ldd, const float *dZn,
                        int64_t ldz, const uint32_t *rowmask, int64_t R, int H, float *partial, void *stream) {
  if (R < 0 || K <= 0 || !partial) return DMP_ERR_BAD_ARG;
  if ((H != 128 && H != 64) || K > kL0K) return DMP_ERR_UNSUPPORTED;
  const int nacc = (dZn ? 3 : 2) * K;
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)nacc * H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!enc || !coef_e || !dPre || lde < K || ldd < H || (dZn && ldz < H)) return DMP_ERR_BAD_ARG;
  if (ldd % 2 || ldz % 2 || !al8(dPre) || !al8(dZn) || !al8(partial)) return DMP_ERR_UNSUPPORTED;
  BwdArgs p{enc, lde, coef_e, dPre, ldd, dZn, ldz, R, partial, rowmask, nullptr, nullptr};
  hipStream_t st = (hipStream_t)stream;
#define L0_CALL(KK) launch_bwd<KK>(p, H, st)
  L0_SWITCH(K, L0_CALL)
#undef L0_CALL
  return check_launch();
}

}  // extern "C"
