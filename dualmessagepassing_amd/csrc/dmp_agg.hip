// Aggregation kernels of the dual-message-passing hot path (gfx950 / MI355X).
//
// All of them are HBM-bound row movers over fp32 feature rows of H floats
// (512 B at H = 128).  Common shape: a group of G = H/4 lanes (16, 32 or 64)
// owns one row and moves it as one float4 (16 B) per lane, so every
// wave-instruction moves 1 KiB in whole 128-B lines; 256-thread workgroups,
// XCD-aware row->workgroup mapping (dmp_common.h) so the node rows a batched
// graph's edges re-read are served by one XCD's L2.
//
//   seg_sum / seg_sum2   CSR segment sum by destination, atomics-free, fixed
//                        (ascending eid) order  -> run-to-run bit-stable
//   gather_rows          out[e] = X[idx[e]]
//   gather_select        backward of seg_sum2
//   edge_combine         fused DMPLayer edge pre-activation
//   edge_combine_bwd_g   its backward w.r.t. the per-edge GEMM output
//   compgcn_agg(+bwd)    CompGCN composition fused into the segment sum
//
// Reference call sites are cited in include/dmp_hip.h.
#include "dmp_common.h"

namespace dmp {

static thread_local char g_last_err[256] = "";
void set_last_hip_error(hipError_t e) {
  const char *s = hipGetErrorString(e);
  int i = 0;
  for (; s && s[i] && i < 255; ++i) g_last_err[i] = s[i];
  g_last_err[i] = 0;
}

namespace {

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void st4(float *p, const float4 &v) {
  *reinterpret_cast<float4 *>(p) = v;
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void add4(float4 &a, const float4 &b) {
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}
__device__ __forceinline__ float4 mul4(const float4 &a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float4 sub4(const float4 &a, const float4 &b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 had4(const float4 &a, const float4 &b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
// a float4 as two complex numbers (re, im | re, im):  conj(a) b,  a b,  conj(a) b with the roles used by the backward
__device__ __forceinline__ float4 cmulc4(const float4 &a, const float4 &b) {     // conj(a) * b
  return make_float4(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x, a.z * b.z + a.w * b.w, a.z * b.w - a.w * b.z);
}
__device__ __forceinline__ float4 cmul4(const float4 &a, const float4 &b) {      // a * b
  return make_float4(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x, a.z * b.z - a.w * b.w, a.z * b.w + a.w * b.z);
}

// ---------------------------------------------------------------------------
// Segment sum.  One G-lane group per destination row.  The group's lanes first
// fetch up to G CSR entries with one coalesced load, then the entries are
// broadcast by shuffle and the source rows are fetched four at a time (four
// independent 16-B loads in flight per lane) and added in CSR order.
// ---------------------------------------------------------------------------
// KIND is not used by the code: 1 tags the launches over an incidence CSR (every source row listed under two destinations),
// so that a profiler's per-kernel statistics keep them apart from the launches over a CSR by destination.
template <int G, bool SPLIT, bool WEIGHTED, bool REMAP, int KIND = 0, int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK) void seg_sum_vec(
    const float *__restrict__ M, int64_t ldm, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ ent, const float *__restrict__ ew, int N, int H,
    float s0, float s1, float *__restrict__ out, int64_t ldo,
    const int32_t *__restrict__ rowlist = nullptr, const int32_t *__restrict__ rowcount = nullptr, int ptr_by_pos = 0) {
  constexpr int RPB = BLOCK / G;
  constexpr int U = 8;  // independent 16-B row loads in flight per lane (4 and 16 measured slower over the incidence CSR)
  // REMAP (XCD-local rows): needed when rows are shared between destinations (incidence
  // CSR) and also faster when M was just written by the previous kernel (59 vs 70 us);
  // plain dispatch order only wins (~3 us) on a cold read-once stream (scripts/kbench.py).
  // (under a row list the launch is sized for all N rows but only the first *rowcount list positions have work: the remap runs
  // over THOSE workgroups -- over the whole grid the list's positions would all land on the first XCDs, 3.2 of 8 at 40 % kept rows.
  // Measured and left out, round 5: a kernel of its own for the list with 2 or 3 rows per lane group and the index loads of all of
  // them requested together -- ONE round of resident waves instead of 1.8 -- is slower, 17.6 -> 20.4 / 27 us: with ~3 entries per
  // row the launch lives on the number of waves that have a request in flight, not on the length of a wave's dependent chain)
  int nwg = gridDim.x;
  if (rowlist) {
    nwg = (*rowcount + RPB - 1) / RPB;
    if ((int)blockIdx.x >= nwg) return;
  }
  int row = (REMAP ? xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x) * RPB + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  if (row >= N) return;
  int prow = row;       // the row of the CSR
  if (rowlist) {        // the destination rows to do, ascending (the nodes a 0 / 1 node gate keeps): the others are left unwritten
    if (row >= *rowcount) return;
    row = rowlist[row];
    if (!ptr_by_pos) prow = row;     // (ptr_by_pos: the CSR has one row per list position, dmp_incidence_keep)
  }
  const int beg = rowptr[prow], end = rowptr[prow + 1];
  for (int c0 = 0; c0 < H; c0 += G * 4) {
    const int c = c0 + lane * 4;
    const bool act = c < H;
    float4 a0 = zero4(), a1 = zero4();
    for (int base = beg; base < end; base += G) {
      const int cnt = min(G, end - base);
      int my = 0;
      float myw = 1.f;
      if (lane < cnt) {
        my = ent[base + lane];
        if (WEIGHTED) myw = ew[my >> 1];
      }
      for (int j = 0; j < cnt; j += U) {
        float4 v[U];
        int e[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          e[k] = __shfl(my, min(j + k, G - 1), G);
          float w = 1.f;
          if (WEIGHTED) w = __shfl(myw, min(j + k, G - 1), G);
          if (act && j + k < cnt) {
            if (WEIGHTED && w == 0.f) {
              v[k] = zero4();                                  // a row with weight 0 (a gated-out edge) is not fetched: 0 * row = 0
            } else {
              v[k] = ld4(M + (int64_t)(e[k] >> 1) * ldm + c);
              if (WEIGHTED) v[k] = mul4(v[k], w);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (act && j + k < cnt) {
            if (SPLIT && (e[k] & 1)) add4(a1, v[k]); else add4(a0, v[k]);
          }
        }
      }
    }
    if (act) {
      float *o = out + (int64_t)row * ldo + c;
      if (SPLIT) {   // (non-temporal stores of the sums: -5.6 % stand-alone over the incidence CSR, nothing inside the step)
        st4(o, mul4(a0, s0));
        st4(o + H, mul4(a1, s1));
      } else {
        st4(o, a0);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Split segment sum for block-diagonal batches whose CSR rows SHARE source rows -- the incidence CSR of the layer's
// backward: every edge row is summed into both of its endpoints' rows.  The plain kernel reads the row twice and the
// second read misses L2 for a third of the rows (PMC: 1.29 x the algorithmic bytes, profiles/r02_pmc_seg_sum2.json).
// Here a workgroup owns a TILE -- a few whole graphs: their node rows and ALL the edge rows those nodes refer to, at
// most kTileRows of them -- and one slice of kTileCols columns: it stages the slice of the tile's edge rows in LDS once
// (one 128-byte line per row) and feeds every node sum from there, in CSR order (same summation order, same bits as
// seg_sum_vec).  Rows in LDS are padded to 36 floats so that the 8 row groups of a wave read disjoint banks.
// Tiles: graphs [0, Ba) in groups of ka, then graphs [Ba, Ba + Bb) in groups of kb (the pattern and the target graphs
// of a union pass); node_off / edge_off: first node / edge row of every graph (+ the totals).
// ---------------------------------------------------------------------------
constexpr int kTileRows = 512, kTileCols = 32, kTilePad = 36, kTileThreads = 512;
struct TileSpec { const int64_t *node_off, *edge_off; int64_t Ba, Bb; int ka, kb; };

// One 512-thread workgroup per (tile, column slice), two resident per CU (72 KB of LDS each); the slice is the fastest
// grid index, so the four workgroups that read the four 128-byte lines of a tile's rows run side by side (one DRAM page
// activation per row).  A group of 8 lanes owns a node row.  Order inside a workgroup: request the slice's row pieces
// (8 x 16 B per lane), then the CSR bounds and the first 16 entries of the group's row (in flight together), write
// the pieces to LDS, barrier, node sums from LDS.
__global__ __launch_bounds__(kTileThreads, 2) void seg_sum_tiled(
    const float *__restrict__ M, int64_t ldm, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ent,
    const TileSpec ts, int H, float s0, float s1, float *__restrict__ out, int64_t ldo, int nslice) {
  __shared__ float rows[kTileRows * kTilePad];
  constexpr int kPieces = kTileRows * 8 / kTileThreads, kRowStep = kTileThreads / 8;
  const int64_t tile = blockIdx.x / nslice;
  const int c0 = (int)(blockIdx.x % nslice) * kTileCols;
  const int64_t tiles_a = ts.Ba > 0 ? (ts.Ba + ts.ka - 1) / ts.ka : 0;
  int64_t g0, g1;
  if (tile < tiles_a) { g0 = tile * ts.ka; g1 = g0 + ts.ka < ts.Ba ? g0 + ts.ka : ts.Ba; }
  else { g0 = ts.Ba + (tile - tiles_a) * ts.kb; g1 = g0 + ts.kb < ts.Ba + ts.Bb ? g0 + ts.kb : ts.Ba + ts.Bb; }
  const int64_t n0 = ts.node_off[g0], n1 = ts.node_off[g1], e0 = ts.edge_off[g0];
  const int R = (int)(ts.edge_off[g1] - e0);
  const int q = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  float4 v[kPieces];
  {
    const float *src = M + (e0 + r0) * ldm + c0 + q * 4;
#pragma unroll
    for (int k = 0; k < kPieces; ++k)
      if (r0 + k * kRowStep < R) v[k] = ld4(src + (int64_t)k * kRowStep * ldm);
  }
  const int64_t row0 = n0 + r0;
  int beg0 = 0, end0 = 0, my0 = 0, my1 = 0;
  if (row0 < n1) {
    beg0 = rowptr[row0]; end0 = rowptr[row0 + 1];
    if (beg0 + q < end0) my0 = ent[beg0 + q];
    if (beg0 + 8 + q < end0) my1 = ent[beg0 + 8 + q];
  }
#pragma unroll
  for (int k = 0; k < kPieces; ++k)
    if (r0 + k * kRowStep < R) *reinterpret_cast<float4 *>(&rows[(r0 + k * kRowStep) * kTilePad + q * 4]) = v[k];
  __syncthreads();
  auto add_entry = [&](int e, float4 &a0, float4 &a1) {
    const float4 x = *reinterpret_cast<const float4 *>(&rows[((e >> 1) - (int)e0) * kTilePad + q * 4]);
    if (e & 1) add4(a1, x); else add4(a0, x);
  };
  for (int64_t row = row0; row < n1; row += kRowStep) {
    float4 a0 = zero4(), a1 = zero4();
    int beg, end;
    if (row == row0) {
      beg = beg0; end = end0;
      const int c1 = min(8, end - beg), c2 = min(8, end - beg - 8);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < c1) add_entry(__shfl(my0, j, 8), a0, a1);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < c2) add_entry(__shfl(my1, j, 8), a0, a1);
      beg += 16;
    } else {
      beg = rowptr[row]; end = rowptr[row + 1];
    }
    for (int base = beg; base < end; base += 8) {
      const int cnt = min(8, end - base);
      const int my = q < cnt ? ent[base + q] : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) add_entry(__shfl(my, j, 8), a0, a1);
    }
    float *o = out + row * ldo + c0 + q * 4;
    st4(o, mul4(a0, s0));
    st4(o + H, mul4(a1, s1));
  }
}

// Any H / any alignment: one wave per row, scalar column-strided accesses.
template <bool SPLIT>
__global__ __launch_bounds__(kBlock) void seg_sum_scalar(
    const float *__restrict__ M, int64_t ldm, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ ent, const float *__restrict__ ew, int N, int H,
    float s0, float s1, float *__restrict__ out, int64_t ldo) {
  const int row = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (row >= N) return;
  const int beg = rowptr[row], end = rowptr[row + 1];
  for (int c = lane; c < H; c += kWave) {
    float a0 = 0.f, a1 = 0.f;
    for (int i = beg; i < end; ++i) {
      const int e = ent[i];
      float v = M[(int64_t)(e >> 1) * ldm + c];
      if (ew) v *= ew[e >> 1];
      if (SPLIT && (e & 1)) a1 += v; else a0 += v;
    }
    float *o = out + (int64_t)row * ldo + c;
    if (SPLIT) { o[0] = a0 * s0; o[H] = a1 * s1; } else { o[0] = a0; }
  }
}

// ---------------------------------------------------------------------------
// Per-edge streaming kernels: a G-lane group per R consecutive edge rows.  The first R
// lanes of the group fetch the R rows' indices / flags with one load each and broadcast
// them by shuffle, then all loads of the R rows are issued before the first store.
// Plain dispatch order (no XCD remap): measured faster for pure streams (kbench).
// ---------------------------------------------------------------------------
template <int G, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void gather_rows_vec(
    const float *__restrict__ X, int64_t ldx, const int32_t *__restrict__ idx,
    const float *__restrict__ ew, int64_t E, int H, float *__restrict__ out, int64_t ldo) {
  constexpr int GPB = kBlock / G;
  const int64_t e0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / G) * R;
  const int lane = threadIdx.x % G;
  if (e0 >= E) return;
  int mi = 0;
  float mw = 1.f;
  if (lane < R && e0 + lane < E) {
    mi = idx[e0 + lane];
    if (WEIGHTED) mw = ew[e0 + lane];
  }
  int r[R];
  float w[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {  // broadcasts outside the column loop: every lane takes part
    r[k] = __shfl(mi, k, G);
    w[k] = WEIGHTED ? __shfl(mw, k, G) : 1.f;
  }
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (e0 + k < E) {
        v[k] = ld4(X + (int64_t)r[k] * ldx + c);
        if (WEIGHTED) v[k] = mul4(v[k], w[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (e0 + k < E) st4(out + (e0 + k) * ldo + c, v[k]);
  }
}

__global__ __launch_bounds__(kBlock) void gather_rows_scalar(
    const float *__restrict__ X, int64_t ldx, const int32_t *__restrict__ idx,
    const float *__restrict__ ew, int64_t E, int H, float *__restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const float w = ew ? ew[e] : 1.f;
  const int64_t r = idx[e];
  for (int c = threadIdx.x % kWave; c < H; c += kWave) out[e * ldo + c] = X[r * ldx + c] * w;
}

template <int G, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void gather_select_vec(
    const float *__restrict__ D, int64_t ldd, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ flag, const float *__restrict__ ew, const float *__restrict__ base,
    int64_t ldb, int64_t E, int H, float s0, float s1, float *__restrict__ out, int64_t ldo) {
  constexpr int GPB = kBlock / G;
  const int64_t e0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / G) * R;
  const int lane = threadIdx.x % G;
  if (e0 >= E) return;
  int md = 0, mf = 0;
  float ms = s0;
  if (lane < R && e0 + lane < E) {
    md = dst[e0 + lane];
    mf = flag ? flag[e0 + lane] : 0;
    ms = mf ? s1 : s0;
    if (WEIGHTED) ms *= ew[e0 + lane];
  }
  int64_t off[R];
  float sc[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int d = __shfl(md, k, G), f = __shfl(mf, k, G);
    off[k] = (int64_t)d * ldd + (f ? H : 0);
    sc[k] = __shfl(ms, k, G);
  }
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (e0 + k < E) {
        v[k] = mul4(ld4(D + off[k] + c), sc[k]);
        if (base) {  // out = base + gathered (gradient accumulation fused into the gather)
          float4 t = ld4(base + (e0 + k) * ldb + c);
          add4(t, v[k]);
          v[k] = t;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (e0 + k < E) st4(out + (e0 + k) * ldo + c, v[k]);
  }
}

__global__ __launch_bounds__(kBlock) void gather_select_scalar(
    const float *__restrict__ D, int64_t ldd, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ flag, const float *__restrict__ ew, const float *__restrict__ base,
    int64_t ldb, int64_t E, int H, float s0, float s1, float *__restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const bool f = flag && flag[e];
  float s = f ? s1 : s0;
  if (ew) s *= ew[e];
  const float *d = D + (int64_t)dst[e] * ldd + (f ? H : 0);
  for (int c = threadIdx.x % kWave; c < H; c += kWave) {
    const float v = d[c] * s;
    out[e * ldo + c] = base ? base[e * ldb + c] + v : v;
  }
}

// Y[e] = ((G0 + coef*G1) + (P[a,0:H] - P[b,H:2H])) + bias     (reference order,
// dmpnn.py:147: matmul(Z,eloop) + add + agg, then + ebias)
template <int G, int R, bool RELU>
__global__ __launch_bounds__(kBlock) void edge_combine_vec(
    const float *__restrict__ Gm, int64_t ldg, const float *__restrict__ P, int64_t ldp,
    const float *__restrict__ coef, const float *__restrict__ bias,
    const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ flag, int64_t E, int H, float slope, float *__restrict__ Y, int64_t ldy) {
  constexpr int GPB = kBlock / G;
  const int64_t e0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / G) * R;
  const int lane = threadIdx.x % G;
  if (e0 >= E) return;
  int ma = 0, mb = 0;
  float mc = 0.f;
  if (lane < R && e0 + lane < E) {
    const int u = src[e0 + lane], v = dst[e0 + lane];
    const bool f = flag && flag[e0 + lane];
    ma = f ? u : v;  // row of P[:, 0:H]  (W_dst side)
    mb = f ? v : u;  // row of P[:, H:2H] (W_src side)
    mc = coef[v];
  }
  int ra[R], rb[R];
  float cf[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    ra[k] = __shfl(ma, k, G);
    rb[k] = __shfl(mb, k, G);
    cf[k] = __shfl(mc, k, G);
  }
  for (int c = lane * 4; c < H; c += G * 4) {
    const float4 bi = bias ? ld4(bias + c) : zero4();
    float4 g0[R], g1[R], pa[R], pb[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (e0 + k < E) {
        g0[k] = ld4(Gm + (e0 + k) * ldg + c);
        g1[k] = ld4(Gm + (e0 + k) * ldg + H + c);
        pa[k] = ld4(P + (int64_t)ra[k] * ldp + c);
        pb[k] = ld4(P + (int64_t)rb[k] * ldp + H + c);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (e0 + k < E) {
        float4 t = mul4(g1[k], cf[k]);
        add4(t, g0[k]);
        add4(t, sub4(pa[k], pb[k]));
        add4(t, bi);
        if (RELU) t = make_float4(act_fwd(t.x, slope), act_fwd(t.y, slope), act_fwd(t.z, slope), act_fwd(t.w, slope));
        st4(Y + (e0 + k) * ldy + c, t);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void edge_combine_scalar(
    const float *__restrict__ Gm, int64_t ldg, const float *__restrict__ P, int64_t ldp,
    const float *__restrict__ coef, const float *__restrict__ bias,
    const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ flag, int64_t E, int H, int relu, float slope, float *__restrict__ Y, int64_t ldy) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const int u = src[e], v = dst[e];
  const bool f = flag && flag[e];
  const float cf = coef[v];
  const int64_t a = f ? u : v, b = f ? v : u;
  for (int c = threadIdx.x % kWave; c < H; c += kWave) {
    float t = Gm[e * ldg + H + c] * cf;
    t += Gm[e * ldg + c];
    t += P[a * ldp + c] - P[b * ldp + H + c];
    if (bias) t += bias[c];
    if (relu) t = act_fwd(t, slope);
    Y[e * ldy + c] = t;
  }
}

template <int G, int R>
__global__ __launch_bounds__(kBlock) void edge_combine_bwd_g_vec(
    const float *__restrict__ dY, int64_t ldy, const float *__restrict__ coef,
    const int32_t *__restrict__ dst, int64_t E, int H, float *__restrict__ dG, int64_t ldg) {
  constexpr int GPB = kBlock / G;
  const int64_t e0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / G) * R;
  const int lane = threadIdx.x % G;
  if (e0 >= E) return;
  float mc = 0.f;
  if (lane < R && e0 + lane < E) mc = coef[dst[e0 + lane]];
  float cf[R];
#pragma unroll
  for (int k = 0; k < R; ++k) cf[k] = __shfl(mc, k, G);
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (e0 + k < E) v[k] = ld4(dY + (e0 + k) * ldy + c);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (e0 + k < E) {
        st4(dG + (e0 + k) * ldg + c, v[k]);
        st4(dG + (e0 + k) * ldg + H + c, mul4(v[k], cf[k]));
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void edge_combine_bwd_g_scalar(
    const float *__restrict__ dY, int64_t ldy, const float *__restrict__ coef,
    const int32_t *__restrict__ dst, int64_t E, int H, float *__restrict__ dG, int64_t ldg) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const float cf = coef[dst[e]];
  for (int c = threadIdx.x % kWave; c < H; c += kWave) {
    const float v = dY[e * ldy + c];
    dG[e * ldg + c] = v;
    dG[e * ldg + H + c] = v * cf;
  }
}

// ---------------------------------------------------------------------------
// CompGCN: out[v] = [ sum_{flag=0} n_e comp(X[src e], Z[e]) | sum_{flag=1} ... ]
// ---------------------------------------------------------------------------
template <int G, int COMP>
__global__ __launch_bounds__(kBlock) void compgcn_agg_vec(
    const float *__restrict__ X, int64_t ldx, const float *__restrict__ Z, int64_t ldz,
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ent,
    const int32_t *__restrict__ src, const float *__restrict__ norm, int N, int H,
    float *__restrict__ out, int64_t ldo) {
  constexpr int RPB = kBlock / G;
  const int row = xcd_remap(blockIdx.x, gridDim.x) * RPB + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  if (row >= N) return;
  const int beg = rowptr[row], end = rowptr[row + 1];
  for (int c0 = 0; c0 < H; c0 += G * 4) {
    const int c = c0 + lane * 4;
    const bool act = c < H;
    float4 a0 = zero4(), a1 = zero4();
    for (int base = beg; base < end; base += G) {
      const int cnt = min(G, end - base);
      int my = 0, mys = 0;
      float myw = 1.f;
      if (lane < cnt) {
        my = ent[base + lane];
        mys = src[my >> 1];
        if (norm) myw = norm[my >> 1];
      }
      int j = 0;
      for (; j + 2 <= cnt; j += 2) {
        const int e0 = __shfl(my, j, G), e1 = __shfl(my, j + 1, G);
        const int u0 = __shfl(mys, j, G), u1 = __shfl(mys, j + 1, G);
        const float w0 = __shfl(myw, j, G), w1 = __shfl(myw, j + 1, G);
        if (act) {
          const float4 x0 = ld4(X + (int64_t)u0 * ldx + c), z0 = ld4(Z + (int64_t)(e0 >> 1) * ldz + c);
          const float4 x1 = ld4(X + (int64_t)u1 * ldx + c), z1 = ld4(Z + (int64_t)(e1 >> 1) * ldz + c);
          const float4 m0 = mul4(COMP == 0 ? sub4(x0, z0) : COMP == 1 ? had4(x0, z0) : cmulc4(x0, z0), w0);
          const float4 m1 = mul4(COMP == 0 ? sub4(x1, z1) : COMP == 1 ? had4(x1, z1) : cmulc4(x1, z1), w1);
          if (e0 & 1) add4(a1, m0); else add4(a0, m0);
          if (e1 & 1) add4(a1, m1); else add4(a0, m1);
        }
      }
      for (; j < cnt; ++j) {
        const int e0 = __shfl(my, j, G), u0 = __shfl(mys, j, G);
        const float w0 = __shfl(myw, j, G);
        if (act) {
          const float4 x0 = ld4(X + (int64_t)u0 * ldx + c), z0 = ld4(Z + (int64_t)(e0 >> 1) * ldz + c);
          const float4 m0 = mul4(COMP == 0 ? sub4(x0, z0) : COMP == 1 ? had4(x0, z0) : cmulc4(x0, z0), w0);
          if (e0 & 1) add4(a1, m0); else add4(a0, m0);
        }
      }
    }
    if (act) {
      float *o = out + (int64_t)row * ldo + c;
      st4(o, a0);
      st4(o + H, a1);
    }
  }
}

template <int COMP>
__global__ __launch_bounds__(kBlock) void compgcn_agg_scalar(
    const float *__restrict__ X, int64_t ldx, const float *__restrict__ Z, int64_t ldz,
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ent,
    const int32_t *__restrict__ src, const float *__restrict__ norm, int N, int H,
    float *__restrict__ out, int64_t ldo) {
  const int row = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (row >= N) return;
  const int beg = rowptr[row], end = rowptr[row + 1];
  for (int c = threadIdx.x % kWave; c < H; c += kWave) {
    float a0 = 0.f, a1 = 0.f;
    for (int i = beg; i < end; ++i) {
      const int e = ent[i], eid = e >> 1;
      const float x = X[(int64_t)src[eid] * ldx + c], z = Z[(int64_t)eid * ldz + c];
      float m = COMP == 0 ? x - z : x * z;
      if (norm) m *= norm[eid];
      if (e & 1) a1 += m; else a0 += m;
    }
    out[(int64_t)row * ldo + c] = a0;
    out[(int64_t)row * ldo + H + c] = a1;
  }
}

template <int G, int COMP>
__global__ __launch_bounds__(kBlock) void compgcn_agg_bwd_vec(
    const float *__restrict__ D, int64_t ldd, const float *__restrict__ X, int64_t ldx,
    const float *__restrict__ Z, int64_t ldz, const int32_t *__restrict__ src,
    const int32_t *__restrict__ dst, const uint8_t *__restrict__ flag,
    const float *__restrict__ norm, int64_t E, int H, float *__restrict__ dZ, int64_t lddz,
    float *__restrict__ dXe, int64_t lddxe) {
  constexpr int RPB = kBlock / G;
  const int64_t e = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * RPB + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  if (e >= E) return;
  const bool f = flag && flag[e];
  const float w = norm ? norm[e] : 1.f;
  const int u = src[e], v = dst[e];
  for (int c = lane * 4; c < H; c += G * 4) {
    const float4 g = mul4(ld4(D + (int64_t)v * ldd + (f ? H : 0) + c), w);
    if (COMP == 0) {
      st4(dZ + e * lddz + c, mul4(g, -1.f));
      st4(dXe + e * lddxe + c, g);
    } else if (COMP == 1) {
      const float4 x = ld4(X + (int64_t)u * ldx + c), z = ld4(Z + e * ldz + c);
      st4(dZ + e * lddz + c, had4(g, x));
      st4(dXe + e * lddxe + c, had4(g, z));
    } else {
      // m = conj(x) z as pairs of reals: dz = x g (complex product), dx = conj(g) z ... written out:
      //   dz_re = g_re x_re - g_im x_im, dz_im = g_re x_im + g_im x_re;  dx_re = g_re z_re + g_im z_im, dx_im = g_re z_im - g_im z_re
      const float4 x = ld4(X + (int64_t)u * ldx + c), z = ld4(Z + e * ldz + c);
      st4(dZ + e * lddz + c, cmul4(g, x));
      st4(dXe + e * lddxe + c, cmulc4(g, z));
    }
  }
}

template <int COMP>
__global__ __launch_bounds__(kBlock) void compgcn_agg_bwd_scalar(
    const float *__restrict__ D, int64_t ldd, const float *__restrict__ X, int64_t ldx,
    const float *__restrict__ Z, int64_t ldz, const int32_t *__restrict__ src,
    const int32_t *__restrict__ dst, const uint8_t *__restrict__ flag,
    const float *__restrict__ norm, int64_t E, int H, float *__restrict__ dZ, int64_t lddz,
    float *__restrict__ dXe, int64_t lddxe) {
  const int64_t e = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (e >= E) return;
  const bool f = flag && flag[e];
  const float w = norm ? norm[e] : 1.f;
  const int64_t u = src[e], v = dst[e];
  for (int c = threadIdx.x % kWave; c < H; c += kWave) {
    const float g = D[v * ldd + (f ? H : 0) + c] * w;
    if (COMP == 0) {
      dZ[e * lddz + c] = -g;
      dXe[e * lddxe + c] = g;
    } else {
      dZ[e * lddz + c] = g * X[u * ldx + c];
      dXe[e * lddxe + c] = g * Z[e * ldz + c];
    }
  }
}

// ------------------------------------------------------------------ dispatch
inline int group_lanes(int H) { return H <= 64 ? 16 : (H <= 128 ? 32 : 64); }
inline unsigned blocks_for(int64_t rows, int rows_per_block) {
  return (unsigned)((rows + rows_per_block - 1) / rows_per_block);
}
inline bool vec_ok(int H, std::initializer_list<int64_t> lds, std::initializer_list<const void *> ps) {
  if (H % 4) return false;
  for (int64_t l : lds) if (l % 4) return false;
  for (const void *p : ps) if (p && !aligned16(p)) return false;
  return true;
}
constexpr int64_t kMaxRows = (int64_t)1 << 30;  // eid << 1 must fit int32

}  // namespace
}  // namespace dmp

using namespace dmp;

#define DMP_DISPATCH_G(H, ...)                                   \
  do {                                                           \
    const int g_ = group_lanes(H);                               \
    if (g_ == 16) { constexpr int G = 16; __VA_ARGS__; }         \
    else if (g_ == 32) { constexpr int G = 32; __VA_ARGS__; }    \
    else { constexpr int G = 64; __VA_ARGS__; }                  \
  } while (0)

extern "C" {

int dmp_abi_version(void) { return DMP_ABI_VERSION; }
const char *dmp_last_hip_error(void) { return g_last_err; }

#define DMP_SS(SP, WT, RM) \
  seg_sum_vec<G, SP, WT, RM><<<nb, kBlock, 0, st>>>(M, ldm, rowptr, ent, ew, (int)N, H, s0, s1, out, ldo, rowlist, rowcount, ptr_by_pos)
#define DMP_SS_INC() \
  seg_sum_vec<G, true, false, true, 1><<<nb, kBlock, 0, st>>>(M, ldm, rowptr, ent, ew, (int)N, H, s0, s1, out, ldo, rowlist, rowcount, ptr_by_pos)
#define DMP_SS_TAG2() \
  seg_sum_vec<G, true, false, true, 2><<<nb, kBlock, 0, st>>>(M, ldm, rowptr, ent, ew, (int)N, H, s0, s1, out, ldo)

static int seg_sum_impl(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent,
                        const float *ew, int64_t N, int H, bool split, float s0, float s1,
                        float *out, int64_t ldo, int rows_shared, void *stream,
                        const int32_t *rowlist = nullptr, const int32_t *rowcount = nullptr, int ptr_by_pos = 0) {
  if (N < 0 || H <= 0 || ldm < H || ldo < (split ? 2 * H : H)) return DMP_ERR_BAD_ARG;
  if (N == 0) return DMP_OK;
  if (!rowptr || !out) return DMP_ERR_BAD_ARG;
  if (rowlist && (!rowcount || !M || !ent || !vec_ok(H, {ldm, ldo}, {M, out}) || rows_shared == 3 || (rows_shared == 2 && !split) || ew)) return DMP_ERR_UNSUPPORTED;
  if (N >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (!M || !ent) {  // a graph without edges: every sum is empty
    const size_t width = sizeof(float) * (size_t)(split ? 2 * H : H);
    hipError_t e = hipMemset2DAsync(out, sizeof(float) * (size_t)ldo, 0, width, (size_t)N, st);
    if (e != hipSuccess) { set_last_hip_error(e); return DMP_ERR_HIP; }
    return DMP_OK;
  }
  if (vec_ok(H, {ldm, ldo}, {M, out})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(N, kBlock / G);
      if (rows_shared) {
        // (rows_shared 3: the same code under its own kernel name -- a measurement launch a profile must keep apart from the step's)
        if (split) { if (ew) DMP_SS(true, true, true); else if (rows_shared == 2) DMP_SS_INC(); else if (rows_shared == 3) DMP_SS_TAG2(); else DMP_SS(true, false, true); }
        else { if (ew) DMP_SS(false, true, true); else DMP_SS(false, false, true); }
      } else {
        if (split) { if (ew) DMP_SS(true, true, false); else DMP_SS(true, false, false); }
        else { if (ew) DMP_SS(false, true, false); else DMP_SS(false, false, false); }
      }
    });
  } else {
    const unsigned nb = blocks_for(N, kBlock / kWave);
    if (split) seg_sum_scalar<true><<<nb, kBlock, 0, st>>>(M, ldm, rowptr, ent, ew, (int)N, H, s0, s1, out, ldo);
    else seg_sum_scalar<false><<<nb, kBlock, 0, st>>>(M, ldm, rowptr, ent, ew, (int)N, H, s0, s1, out, ldo);
  }
  return check_launch();
}

int dmp_seg_sum(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent,
                const float *edge_w, int64_t num_nodes, int H, float *out, int64_t ldo,
                int rows_shared, void *stream) {
  return seg_sum_impl(M, ldm, rowptr, ent, edge_w, num_nodes, H, false, 1.f, 1.f, out, ldo, rows_shared, stream);
}

int dmp_seg_sum2(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent,
                 const float *edge_w, int64_t num_nodes, int H, float s0, float s1, float *out,
                 int64_t ldo, int rows_shared, void *stream) {
  return seg_sum_impl(M, ldm, rowptr, ent, edge_w, num_nodes, H, true, s0, s1, out, ldo, rows_shared, stream);
}

int dmp_seg_sum2_rows(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent, const int32_t *rowlist,
                      const int32_t *rowcount, int ptr_by_pos, int incidence, int64_t num_nodes, int H, float s0, float s1, float *out,
                      int64_t ldo, void *stream) {
  if (!rowlist || !rowcount) return DMP_ERR_BAD_ARG;
  // (incidence: said of an incidence CSR -- the kernel instantiation tagged so, so that a profile keeps the backward's sums apart)
  return seg_sum_impl(M, ldm, rowptr, ent, nullptr, num_nodes, H, true, s0, s1, out, ldo, incidence ? 2 : 1, stream, rowlist, rowcount,
                      ptr_by_pos ? 1 : 0);
}

int dmp_seg_sum2_tiled(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent, const int64_t *node_off,
                       const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb, int H, float s0, float s1,
                       float *out, int64_t ldo, void *stream) {
  if (Ba < 0 || Bb < 0 || H <= 0 || ldm < H || ldo < 2 * H || (Ba > 0 && ka < 1) || (Bb > 0 && kb < 1)) return DMP_ERR_BAD_ARG;
  if (Ba + Bb == 0) return DMP_OK;
  if (!M || !rowptr || !ent || !node_off || !edge_off || !out) return DMP_ERR_BAD_ARG;
  if (H % kTileCols || ldm % 4 || ldo % 4 || !aligned16(M) || !aligned16(out)) return DMP_ERR_UNSUPPORTED;
  const int64_t tiles = (Ba > 0 ? (Ba + ka - 1) / ka : 0) + (Bb > 0 ? (Bb + kb - 1) / kb : 0);
  TileSpec ts{node_off, edge_off, Ba, Bb, ka > 0 ? ka : 1, kb > 0 ? kb : 1};
  const int nslice = H / kTileCols;
  if (tiles * nslice >= ((int64_t)1 << 31)) return DMP_ERR_UNSUPPORTED;
  seg_sum_tiled<<<(unsigned)(tiles * nslice), kTileThreads, 0, (hipStream_t)stream>>>(M, ldm, rowptr, ent, ts, H, s0, s1, out, ldo, nslice);
  return check_launch();
}

int dmp_gather_rows(const float *X, int64_t ldx, const int32_t *idx, const float *edge_w,
                    int64_t E, int H, float *out, int64_t ldo, void *stream) {
  if (E < 0 || H <= 0 || ldx < H || ldo < H) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!X || !idx || !out) return DMP_ERR_BAD_ARG;
  if (E >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldx, ldo}, {X, out})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(E, (kBlock / G) * 4);
      if (edge_w) gather_rows_vec<G, 4, true><<<nb, kBlock, 0, st>>>(X, ldx, idx, edge_w, E, H, out, ldo);
      else gather_rows_vec<G, 4, false><<<nb, kBlock, 0, st>>>(X, ldx, idx, edge_w, E, H, out, ldo);
    });
  } else {
    gather_rows_scalar<<<blocks_for(E, kBlock / kWave), kBlock, 0, st>>>(X, ldx, idx, edge_w, E, H, out, ldo);
  }
  return check_launch();
}

int dmp_gather_select(const float *D, int64_t ldd, const int32_t *dst, const uint8_t *flag,
                      const float *edge_w, const float *base, int64_t ldb, int64_t E, int H,
                      float s0, float s1, float *out, int64_t ldo, void *stream) {
  if (E < 0 || H <= 0 || ldd < 2 * H || ldo < H || (base && ldb < H)) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!D || !dst || !out) return DMP_ERR_BAD_ARG;
  if (E >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldd, ldo, base ? ldb : 0}, {D, out, base})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(E, (kBlock / G) * 4);
      if (edge_w) gather_select_vec<G, 4, true><<<nb, kBlock, 0, st>>>(D, ldd, dst, flag, edge_w, base, ldb, E, H, s0, s1, out, ldo);
      else gather_select_vec<G, 4, false><<<nb, kBlock, 0, st>>>(D, ldd, dst, flag, edge_w, base, ldb, E, H, s0, s1, out, ldo);
    });
  } else {
    gather_select_scalar<<<blocks_for(E, kBlock / kWave), kBlock, 0, st>>>(D, ldd, dst, flag, edge_w, base, ldb, E, H, s0, s1, out, ldo);
  }
  return check_launch();
}

int dmp_edge_combine(const float *Gm, int64_t ldg, const float *P, int64_t ldp, const float *coef,
                     const float *bias, const int32_t *src, const int32_t *dst,
                     const uint8_t *flag, int64_t E, int H, int relu, float slope, float *Y, int64_t ldy,
                     void *stream) {
  if (E < 0 || H <= 0 || ldg < 2 * H || ldp < 2 * H || ldy < H) return DMP_ERR_BAD_ARG;
  if (relu && !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Gm || !P || !coef || !src || !dst || !Y) return DMP_ERR_BAD_ARG;
  if (E >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldg, ldp, ldy}, {Gm, P, Y, bias})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(E, (kBlock / G) * 2);
      if (relu) edge_combine_vec<G, 2, true><<<nb, kBlock, 0, st>>>(Gm, ldg, P, ldp, coef, bias, src, dst, flag, E, H, slope, Y, ldy);
      else edge_combine_vec<G, 2, false><<<nb, kBlock, 0, st>>>(Gm, ldg, P, ldp, coef, bias, src, dst, flag, E, H, slope, Y, ldy);
    });
  } else {
    edge_combine_scalar<<<blocks_for(E, kBlock / kWave), kBlock, 0, st>>>(Gm, ldg, P, ldp, coef, bias, src, dst, flag, E, H, relu, slope, Y, ldy);
  }
  return check_launch();
}

int dmp_edge_combine_bwd_g(const float *dY, int64_t ldy, const float *coef, const int32_t *dst,
                           int64_t E, int H, float *dG, int64_t ldg, void *stream) {
  if (E < 0 || H <= 0 || ldy < H || ldg < 2 * H) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!dY || !coef || !dst || !dG) return DMP_ERR_BAD_ARG;
  if (E >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldy, ldg}, {dY, dG})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(E, kBlock / G);
      edge_combine_bwd_g_vec<G, 1><<<nb, kBlock, 0, st>>>(dY, ldy, coef, dst, E, H, dG, ldg);
    });
  } else {
    edge_combine_bwd_g_scalar<<<blocks_for(E, kBlock / kWave), kBlock, 0, st>>>(dY, ldy, coef, dst, E, H, dG, ldg);
  }
  return check_launch();
}

int dmp_compgcn_agg(const float *X, int64_t ldx, const float *Z, int64_t ldz, const int32_t *rowptr,
                    const int32_t *ent, const int32_t *src, const float *norm, int64_t N, int H,
                    int comp, float *out, int64_t ldo, void *stream) {
  if (N < 0 || H <= 0 || ldx < H || ldz < H || ldo < 2 * H || comp < 0 || comp > 2) return DMP_ERR_BAD_ARG;
  if (N == 0) return DMP_OK;
  if (!X || !Z || !rowptr || !ent || !src || !out) return DMP_ERR_BAD_ARG;
  if (N >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldx, ldz, ldo}, {X, Z, out})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(N, kBlock / G);
      if (comp == 0) compgcn_agg_vec<G, 0><<<nb, kBlock, 0, st>>>(X, ldx, Z, ldz, rowptr, ent, src, norm, (int)N, H, out, ldo);
      else if (comp == 1) compgcn_agg_vec<G, 1><<<nb, kBlock, 0, st>>>(X, ldx, Z, ldz, rowptr, ent, src, norm, (int)N, H, out, ldo);
      else compgcn_agg_vec<G, 2><<<nb, kBlock, 0, st>>>(X, ldx, Z, ldz, rowptr, ent, src, norm, (int)N, H, out, ldo);
    });
  } else {
    if (comp == 2) return DMP_ERR_UNSUPPORTED;               // complex pairs: the float4 path only (H % 4 == 0, aligned rows)
    const unsigned nb = blocks_for(N, kBlock / kWave);
    if (comp == 0) compgcn_agg_scalar<0><<<nb, kBlock, 0, st>>>(X, ldx, Z, ldz, rowptr, ent, src, norm, (int)N, H, out, ldo);
    else compgcn_agg_scalar<1><<<nb, kBlock, 0, st>>>(X, ldx, Z, ldz, rowptr, ent, src, norm, (int)N, H, out, ldo);
  }
  return check_launch();
}

int dmp_compgcn_agg_bwd(const float *D, int64_t ldd, const float *X, int64_t ldx, const float *Z,
                        int64_t ldz, const int32_t *src, const int32_t *dst, const uint8_t *flag,
                        const float *norm, int64_t E, int H, int comp, float *dZ, int64_t lddz,
                        float *dXe, int64_t lddxe, void *stream) {
  if (E < 0 || H <= 0 || ldd < 2 * H || ldx < H || ldz < H || lddz < H || lddxe < H || comp < 0 || comp > 2)
    return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!D || !X || !Z || !src || !dst || !dZ || !dXe) return DMP_ERR_BAD_ARG;
  if (E >= kMaxRows) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(H, {ldd, ldx, ldz, lddz, lddxe}, {D, X, Z, dZ, dXe})) {
    DMP_DISPATCH_G(H, {
      const unsigned nb = blocks_for(E, kBlock / G);
      if (comp == 0) compgcn_agg_bwd_vec<G, 0><<<nb, kBlock, 0, st>>>(D, ldd, X, ldx, Z, ldz, src, dst, flag, norm, E, H, dZ, lddz, dXe, lddxe);
      else if (comp == 1) compgcn_agg_bwd_vec<G, 1><<<nb, kBlock, 0, st>>>(D, ldd, X, ldx, Z, ldz, src, dst, flag, norm, E, H, dZ, lddz, dXe, lddxe);
      else compgcn_agg_bwd_vec<G, 2><<<nb, kBlock, 0, st>>>(D, ldd, X, ldx, Z, ldz, src, dst, flag, norm, E, H, dZ, lddz, dXe, lddxe);
    });
  } else {
    if (comp == 2) return DMP_ERR_UNSUPPORTED;
    const unsigned nb = blocks_for(E, kBlock / kWave);
    if (comp == 0) compgcn_agg_bwd_scalar<0><<<nb, kBlock, 0, st>>>(D, ldd, X, ldx, Z, ldz, src, dst, flag, norm, E, H, dZ, lddz, dXe, lddxe);
    else compgcn_agg_bwd_scalar<1><<<nb, kBlock, 0, st>>>(D, ldd, X, ldx, Z, ldz, src, dst, flag, norm, E, H, dZ, lddz, dXe, lddxe);
  }
  return check_launch();
}

}  // extern "C"
