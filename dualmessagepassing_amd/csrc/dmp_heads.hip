// The pooled prediction heads (SubgraphCountingMatching/models/pred.py:93-156 on per-graph sums, the node and the
// edge head of basemodel.py:1477-1498) as three launches for ALL heads: forward, backward over the rows, backward
// weight gradients.  The tensors are tiny ([B, 128..516], B = pairs per batch): the algebra is ~60 launches per
// step when done op by op, i.e. pure launch latency at the end of forward and the start of backward.
//
//   p = ps Wp^T + sp bp;  g = gs Wg^T + sg bg;  s = [pl, gl, 1/pl, 1/gl]
//   f = [p | g | g - p | g * p | s]  (4h + 4);  y1 = relu(f W1^T + b1);  y = [y1 | s] W2^T + b2
//
// h = input width = hidden width = 128 or 64.  fp32 MFMA products through LDS tiles (as csrc/dmp_fold.hip): 16 rows per
// workgroup, thread (r, cg) owns h / 16 consecutive columns of row r.
#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kThreads = 256, kRows = 16;
constexpr int kMaxHeads = DMP_HEADS_MAX;

// widths derived from the head's input = hidden width h (128, or the reference's shipped 64)
template <int kH> struct HeadGeom {
  static constexpr int kF = 4 * kH + 4, kYS = kH + 4;      // feature row [p | g | g - p | g * p | s], saved [y1 | s]
  static constexpr int kPad = kH + 4, kFPad = 4 * kH + 8;  // LDS row strides
  static constexpr int CW = kH / 16;                       // thread (r, cg) owns CW consecutive columns of row r
  static constexpr int NT = kH / 64, NKB = kH / 16;        // 16-column MFMA tiles per wave, k blocks of 16
};

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

struct Heads {
  dmp_head_weights w[kMaxHeads];
  dmp_head_io io[kMaxHeads];
  dmp_head_grads gr[kMaxHeads];
  int B;
  float slope;   // negative slope of the heads' activation (0 = ReLU; pred_act_func, pred.py:36)
};

// acc[CW] += A_lds[r][0..h-1] . op(W)[.., CW cg ..]   for one h-wide contraction; W element (j, k) at W[j * ldw + k]
//   wT = 1: out col j, contraction k: out[j] = sum_k a[k] W[j][k]     (x W^T, W in nn.Linear layout)
//   wT = 0: out col k, contraction j: out[k] = sum_j a[j] W[j][k]     (x W)
// A: the 16 operand rows in LDS (row i at A + i * lda, >= h values at stride 1, 16-byte aligned rows).
// The 16 x h product runs on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32: wave w owns output columns (h/4) w .. (h/4)(w+1) - 1):
// every lane requests its share of the whole h-deep weight panel in ONE batch of loads straight in the operand layout
// (one L2 round trip per product instead of one per 32-deep LDS slice), the A operand comes from LDS as float4 (lane
// (m, kq) takes k = 16 kb + 4 kq .. + 3, the weights follow the same order); the 16 x h result goes through Ws
// (>= 16 * kPad floats) into the (row r = tid / 16, CW columns cg) accumulators of the callers' epilogues.
template <int kH>
__device__ __forceinline__ void gemm_h(float (&acc)[kH / 16], const float *A, int lda, const float *W, int ldw, int wT, int wcol0,
                                       float *Ws, int tid, int cg) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int kPad = HeadGeom<kH>::kPad, CW = HeadGeom<kH>::CW, NT = HeadGeom<kH>::NT, NKB = HeadGeom<kH>::NKB;
  const int lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
  float4 bw[NT][NKB];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = wave * (16 * NT) + t * 16 + m;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int k = kb * 16 + 4 * kq;
      if (wT) {
        bw[t][kb] = ldg4(W + (int64_t)n * ldw + wcol0 + k);
      } else {
        const float *c = W + (int64_t)k * ldw + wcol0 + n;
        bw[t][kb] = make_float4(c[0], c[ldw], c[2 * (int64_t)ldw], c[3 * (int64_t)ldw]);
      }
    }
  }
  __syncthreads();                                          // the callers' operand rows are in LDS; Ws is free again
  f32x4 d[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const float4 a = *reinterpret_cast<const float4 *>(A + m * lda + kb * 16 + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw[t][kb].x, d[t], 0, 0, 0);
      d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw[t][kb].y, d[t], 0, 0, 0);
      d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw[t][kb].z, d[t], 0, 0, 0);
      d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw[t][kb].w, d[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) Ws[(4 * kq + i) * kPad + wave * (16 * NT) + t * 16 + m] = d[t][i];
  __syncthreads();
  const int r = tid >> 4;
#pragma unroll
  for (int v = 0; v < CW / 4; ++v) {
    const float4 o = *reinterpret_cast<const float4 *>(&Ws[r * kPad + cg * CW + 4 * v]);
    acc[4 * v] += o.x; acc[4 * v + 1] += o.y; acc[4 * v + 2] += o.z; acc[4 * v + 3] += o.w;
  }
}

template <int kH>
__device__ __forceinline__ void stage_rows(float *As, const float *X, int64_t ldx, int i0, int B, int tid) {
  constexpr int kQ = kH / 4, kPass = kThreads / kQ;
#pragma unroll
  for (int m = 0; m < kRows / kPass; ++m) {
    const int row = tid / kQ + kPass * m, c4 = (tid % kQ) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + row < B) a = ldg4(X + (int64_t)(i0 + row) * ldx + c4);
    *reinterpret_cast<float4 *>(&As[row * HeadGeom<kH>::kPad + c4]) = a;
  }
}

// CW consecutive floats of a thread's row: registers -> memory
template <int CW>
__device__ __forceinline__ void store_cw(float *o, const float (&v)[CW]) {
#pragma unroll
  for (int q = 0; q < CW / 4; ++q) *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

template <int kH>
__global__ __launch_bounds__(kThreads) void heads_fwd_k(const Heads t) {
  constexpr int kF = HeadGeom<kH>::kF, kYS = HeadGeom<kH>::kYS, kPad = HeadGeom<kH>::kPad, kFPad = HeadGeom<kH>::kFPad, CW = HeadGeom<kH>::CW;
  __shared__ float As[kRows * kPad];
  __shared__ float Fs[kRows * kFPad];
  __shared__ float Ws[16 * kPad];
  const dmp_head_weights &w = t.w[blockIdx.y];
  const dmp_head_io &io = t.io[blockIdx.y];
  const int tid = threadIdx.x, r = tid >> 4, cg = tid & 15, i0 = (int)blockIdx.x * kRows, b = i0 + r;
  const bool live = b < t.B;
  float p[CW], g[CW];
  // p = ps Wp^T + sp bp,  g = gs Wg^T + sg bg
#pragma unroll
  for (int e = 0; e < CW; ++e) { p[e] = 0.f; g[e] = 0.f; }
  stage_rows<kH>(As, io.ps, io.ld_ps, i0, t.B, tid);
  gemm_h<kH>(p, As, kPad, w.Wp, kH, 1, 0, Ws, tid, cg);
  __syncthreads();
  stage_rows<kH>(As, io.gs, io.ld_gs, i0, t.B, tid);
  gemm_h<kH>(g, As, kPad, w.Wg, kH, 1, 0, Ws, tid, cg);
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    p[e] += io.scale_p * w.bp[cg * CW + e];
    g[e] += io.scale_g * w.bg[cg * CW + e];
  }
  // f = [p | g | g - p | g * p | s]
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) { s4[0] = io.pl[b]; s4[1] = io.gl[b]; s4[2] = 1.0f / s4[0]; s4[3] = 1.0f / s4[1]; }
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    const int c = cg * CW + e;
    Fs[r * kFPad + c] = p[e];
    Fs[r * kFPad + kH + c] = g[e];
    Fs[r * kFPad + 2 * kH + c] = g[e] - p[e];
    Fs[r * kFPad + 3 * kH + c] = g[e] * p[e];
  }
  if (cg < 4) Fs[r * kFPad + 4 * kH + cg] = s4[cg];
  __syncthreads();
  if (live) {
    float *F = io.F + (int64_t)b * kF;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < CW / 4; ++v)
        *reinterpret_cast<float4 *>(F + q * kH + cg * CW + 4 * v) = *reinterpret_cast<const float4 *>(&Fs[r * kFPad + q * kH + cg * CW + 4 * v]);
    if (cg < 4) F[4 * kH + cg] = s4[cg];
  }
  // y1 = act(f W1^T + b1): four h-wide contractions + the four scalars
  float y1[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) y1[e] = w.b1[cg * CW + e];
#pragma unroll 1
  for (int q = 0; q < 4; ++q) gemm_h<kH>(y1, Fs + q * kH, kFPad, w.W1, kF, 1, q * kH, Ws, tid, cg);
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    const float *wr = w.W1 + (int64_t)(cg * CW + e) * kF + 4 * kH;
    y1[e] += s4[0] * wr[0] + s4[1] * wr[1] + s4[2] * wr[2] + s4[3] * wr[3];
    y1[e] = act_fwd(y1[e], t.slope);
  }
  // y = [y1 | s] W2^T + b2
  float part = 0.f;
#pragma unroll
  for (int e = 0; e < CW; ++e) part += y1[e] * w.W2[cg * CW + e];
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) part += __shfl_xor(part, off, 16);
  if (live) {
    float *ys = io.Y1S + (int64_t)b * kYS;
    store_cw<CW>(ys + cg * CW, y1);
    if (cg < 4) ys[kH + cg] = s4[cg];
    if (cg == 0)
      io.y[b] = part + s4[0] * w.W2[kH] + s4[1] * w.W2[kH + 1] + s4[2] * w.W2[kH + 2] + s4[3] * w.W2[kH + 3] + w.b2[0];
  }
}

// Backward over the rows: dy -> dy1 (activation mask) -> df = dy1 W1 -> dp, dg -> dps = dp Wp, dgs = dg Wg.
// Writes what the weight gradients need (dY1, dP, dG) and the input gradients.
template <int kH>
__global__ __launch_bounds__(kThreads) void heads_bwd_rows_k(const Heads t) {
  constexpr int kF = HeadGeom<kH>::kF, kYS = HeadGeom<kH>::kYS, kPad = HeadGeom<kH>::kPad, CW = HeadGeom<kH>::CW;
  __shared__ float As[kRows * kPad];
  __shared__ float Ws[16 * kPad];
  const dmp_head_weights &w = t.w[blockIdx.y];
  const dmp_head_io &io = t.io[blockIdx.y];
  const dmp_head_grads &gr = t.gr[blockIdx.y];
  const int tid = threadIdx.x, r = tid >> 4, cg = tid & 15, i0 = (int)blockIdx.x * kRows, b = i0 + r;
  const bool live = b < t.B;
  const float dy = live ? gr.dy[b] * (gr.dy_scale ? gr.dy_scale[b] : 1.0f) : 0.f;
  // dy1 = y1 > 0 ? dy W2 : slope * dy W2
  float d1[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    const float y1 = live ? io.Y1S[(int64_t)b * kYS + cg * CW + e] : 0.f;
    d1[e] = act_bwd(y1, dy * w.W2[cg * CW + e], t.slope);
    As[r * kPad + cg * CW + e] = d1[e];
  }
  if (live) store_cw<CW>(gr.dY1 + (int64_t)b * kH + cg * CW, d1);
  // df blocks: [dfp | dfg | dfd | dfm] = dy1 W1[:, 0:4h]
  float df[4][CW];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < CW; ++e) df[q][e] = 0.f;
    gemm_h<kH>(df[q], As, kPad, w.W1, kF, 0, q * kH, Ws, tid, cg);
  }
  float dp[CW], dg[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) {
    const float p = live ? io.F[(int64_t)b * kF + cg * CW + e] : 0.f;
    const float g = live ? io.F[(int64_t)b * kF + kH + cg * CW + e] : 0.f;
    dp[e] = (df[0][e] - df[2][e]) + df[3][e] * g;
    dg[e] = (df[1][e] + df[2][e]) + df[3][e] * p;
  }
  if (live) {
    store_cw<CW>(gr.dP + (int64_t)b * kH + cg * CW, dp);
    store_cw<CW>(gr.dG + (int64_t)b * kH + cg * CW, dg);
  }
  // input gradients: dps = dp Wp, dgs = dg Wg
  float out[CW];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < CW; ++e) { As[r * kPad + cg * CW + e] = dp[e]; out[e] = 0.f; }
  gemm_h<kH>(out, As, kPad, w.Wp, kH, 0, 0, Ws, tid, cg);
  if (live && gr.dps) store_cw<CW>(gr.dps + (int64_t)b * gr.ld_dps + cg * CW, out);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < CW; ++e) { As[r * kPad + cg * CW + e] = dg[e]; out[e] = 0.f; }
  gemm_h<kH>(out, As, kPad, w.Wg, kH, 0, 0, Ws, tid, cg);
  if (live && gr.dgs) store_cw<CW>(gr.dgs + (int64_t)b * gr.ld_dgs + cg * CW, out);
}

// Weight gradients: out[j, k] = sum_i dC[i, j] M[i, k]  (j < nj <= 128, k < ncols), bias[j] = bscale * sum_i dC[i, j].
// One workgroup per (job, 16 output rows, 16 output columns) on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32 contracts four
// batch rows per instruction): the four waves take interleaved groups of four batch rows, every lane requests its A / B
// operands of kSteps instructions in one batch straight in the operand layout, a batch ahead of the one being
// multiplied (B = 1024: four dependent L2 round trips per workgroup instead of sixteen staged 64-row chunks), and the
// four partial tiles meet in LDS.
struct WJob { const float *dC; const float *M; const float *row_scale; float *out; float *bias; int64_t ldc, ldm; int nj, ncols, ldo; float bscale; };
constexpr int kMaxWJobs = 4 * kMaxHeads;
struct WJobs { WJob job[kMaxWJobs]; int blk0[kMaxWJobs + 1]; int n; int B; };
constexpr int kColW = 16, kSteps = 16;

__global__ __launch_bounds__(kThreads) void heads_bwd_w_k(const WJobs t) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[4 * 16 * 17];
  __shared__ float bred[4 * 64];
  int q = 0;
  while (q + 1 < t.n && (int)blockIdx.x >= t.blk0[q + 1]) ++q;
  const WJob &jb = t.job[q];
  const int kblocks = (jb.ncols + kColW - 1) / kColW;
  const int sub = (int)blockIdx.x - t.blk0[q];
  const int j0 = (sub / kblocks) * kRows, k0 = (sub % kblocks) * kColW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
  const bool a_on = j0 + m < jb.nj, b_on = k0 + m < jb.ncols;
  const float *ap = jb.dC + j0 + m, *bp = jb.M + k0 + m;
  const int groups = (t.B + 3) / 4;                         // groups of four batch rows; wave w takes w, w + 4, ...
  float an[kSteps], sn[kSteps], bn[kSteps];
  auto fetch = [&](int g0) {
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const int64_t i = 4 * (int64_t)(g0 + wave + 4 * st) + kq;
      const bool in = i < t.B;
      an[st] = (in && a_on) ? ap[i * jb.ldc] : 0.f;
      sn[st] = (in && jb.row_scale) ? jb.row_scale[i] : 1.0f;
      bn[st] = (in && b_on) ? bp[i * jb.ldm] : 0.f;
    }
  };
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  fetch(0);
  for (int g0 = 0; g0 < groups; g0 += 4 * kSteps) {
    float a[kSteps], b[kSteps];
#pragma unroll
    for (int st = 0; st < kSteps; ++st) { a[st] = an[st] * sn[st]; b[st] = bn[st]; }
    if (g0 + 4 * kSteps < groups) fetch(g0 + 4 * kSteps);
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], b[st], d, 0, 0, 0);
      bsum += a[st];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * kq + i) * 17 + m] = d[i];
  bred[wave * 64 + lane] = bsum;
  __syncthreads();
  const int jr = tid >> 4, kc = tid & 15;
  if (j0 + jr < jb.nj && k0 + kc < jb.ncols)
    jb.out[(int64_t)(j0 + jr) * jb.ldo + k0 + kc] =
        (red[(0 * 16 + jr) * 17 + kc] + red[(1 * 16 + jr) * 17 + kc]) + (red[(2 * 16 + jr) * 17 + kc] + red[(3 * 16 + jr) * 17 + kc]);
  if (jb.bias && k0 == 0 && tid < 16 && j0 + tid < jb.nj) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += bred[w * 64 + c * 16 + tid];
    jb.bias[j0 + tid] = jb.bscale * sum;
  }
}

struct BlendArgs { const float *y[kMaxHeads]; const float *gl[kMaxHeads]; float *w[kMaxHeads]; float *out; int n, B; };
__global__ __launch_bounds__(kThreads) void heads_blend_k(const BlendArgs a) {
  const int b = (int)blockIdx.x * kThreads + threadIdx.x;
  if (b >= a.B) return;
  float total = a.gl[0][b];
  for (int i = 1; i < a.n; ++i) total = total + a.gl[i][b];
  float out = 0.f;
  for (int i = 0; i < a.n; ++i) {
    const float w = a.gl[i][b] / total;
    a.w[i][b] = w;
    const float term = __fmul_rn(a.y[i][b], w);             // separate multiply and add, as the op-by-op blend rounds
    out = i == 0 ? term : __fadd_rn(out, term);
  }
  a.out[b] = out;
}

inline void add_wjob(WJobs &t, int &blocks, const float *dC, int64_t ldc, int nj, const float *row_scale, const float *M,
                     int64_t ldm, int ncols, float *out, int ldo, float *bias, float bscale) {
  WJob &j = t.job[t.n];
  j.dC = dC; j.ldc = ldc; j.nj = nj; j.row_scale = row_scale; j.M = M; j.ldm = ldm; j.ncols = ncols; j.out = out; j.ldo = ldo;
  j.bias = bias; j.bscale = bscale;
  t.blk0[t.n] = blocks;
  blocks += ((nj + kRows - 1) / kRows) * ((ncols + kColW - 1) / kColW);
  t.n += 1;
  t.blk0[t.n] = blocks;
}

inline bool ok16p(const void *q) { return q && (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

bool heads_valid(const dmp_head_weights *w, const dmp_head_io *io, int n, int B, int H) {
  const int kH = H;
  if (n < 1 || n > kMaxHeads || B < 0 || (H != 128 && H != 64) || !w || !io) return false;
  for (int i = 0; i < n; ++i) {
    if (!ok16p(w[i].Wp) || !ok16p(w[i].bp) || !ok16p(w[i].Wg) || !ok16p(w[i].bg) || !ok16p(w[i].W1) || !ok16p(w[i].b1) ||
        !ok16p(w[i].W2) || !w[i].b2)
      return false;
    if (!ok16p(io[i].ps) || !ok16p(io[i].gs) || !io[i].pl || !io[i].gl || !ok16p(io[i].F) || !ok16p(io[i].Y1S) || !io[i].y ||
        io[i].ld_ps < kH || io[i].ld_gs < kH || io[i].ld_ps % 4 || io[i].ld_gs % 4)
      return false;
  }
  return true;
}

// The count loss of a training step and its seed in ONE launch (train.py:624-628: bp_crit(leaky_relu(pred_c, slope), counts) with
// reduction 'mean', then loss.backward()): around a thousand numbers, five torch launches (criterion, mean, two fills, the
// criterion's backward).  One workgroup: loss[0] = mean_i crit(act(pred_i) - target_i), dpred[i] = d loss / d pred_i; fixed
// summation order (thread-strided partial sums, a tree over the threads): the same bits on every launch.
//   kind 0: MSE d^2      1: MAE |d|      2: smooth L1 (beta 1): d^2 / 2 inside |d| < 1, |d| - 1/2 outside
constexpr int kLossThreads = 1024;
__global__ __launch_bounds__(kLossThreads) void count_loss_k(const float *__restrict__ pred, const float *__restrict__ target, int64_t n,
                                                             int kind, float slope, float *__restrict__ loss, float *__restrict__ dpred) {
  __shared__ float red[kLossThreads];
  const float inv_n = 1.f / (float)n;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kLossThreads) {
    const float p = pred[i];
    const float da = p > 0.f ? 1.f : slope;                  // leaky_relu'(p) (torch: the negative slope at p <= 0)
    const float d = p * da - target[i];
    float v, g;
    if (kind == 0) { v = d * d; g = 2.f * d; }
    else if (kind == 1) { v = fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    else { const float a = fabsf(d); v = a < 1.f ? 0.5f * d * d : a - 0.5f; g = a < 1.f ? d : (d > 0.f ? 1.f : -1.f); }
    acc += v;
    dpred[i] = g * da * inv_n;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = kLossThreads / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_n;
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_heads_forward(const dmp_head_weights *w, const dmp_head_io *io, int num_heads, int B, int H, float slope,
                      void *stream) {
  if ((H != 128 && H != 64 && H > 0) || !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (!heads_valid(w, io, num_heads, B, H)) return DMP_ERR_BAD_ARG;
  if (B == 0) return DMP_OK;
  Heads t;
  for (int i = 0; i < num_heads; ++i) { t.w[i] = w[i]; t.io[i] = io[i]; t.gr[i] = dmp_head_grads{}; }
  t.B = B; t.slope = slope;
  const dim3 grid((unsigned)((B + kRows - 1) / kRows), (unsigned)num_heads);
  if (H == 128) heads_fwd_k<128><<<grid, kThreads, 0, (hipStream_t)stream>>>(t);
  else heads_fwd_k<64><<<grid, kThreads, 0, (hipStream_t)stream>>>(t);
  return check_launch();
}

int dmp_heads_blend(const float *const *y, const float *const *gl, float *const *w, int num_heads, int B, float *out,
                    void *stream) {
  if (num_heads < 1 || num_heads > kMaxHeads || B < 0 || !y || !gl || !w) return DMP_ERR_BAD_ARG;
  if (B == 0) return DMP_OK;
  if (!out) return DMP_ERR_BAD_ARG;
  BlendArgs a;
  for (int i = 0; i < num_heads; ++i) {
    if (!y[i] || !gl[i] || !w[i]) return DMP_ERR_BAD_ARG;
    a.y[i] = y[i]; a.gl[i] = gl[i]; a.w[i] = w[i];
  }
  a.out = out; a.n = num_heads; a.B = B;
  heads_blend_k<<<(unsigned)((B + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(a);
  return check_launch();
}

int dmp_heads_backward(const dmp_head_weights *w, const dmp_head_io *io, const dmp_head_grads *g, int num_heads, int B,
                       int H, float slope, void *stream) {
  if ((H != 128 && H != 64 && H > 0) || !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (!heads_valid(w, io, num_heads, B, H) || !g) return DMP_ERR_BAD_ARG;
  const int kH = H, kF = 4 * H + 4, kYS = H + 4;
  for (int i = 0; i < num_heads; ++i) {
    if (!g[i].dy || !ok16p(g[i].dY1) || !ok16p(g[i].dP) || !ok16p(g[i].dG) || (g[i].dps && (!ok16p(g[i].dps) || g[i].ld_dps % 4)) ||
        (g[i].dgs && (!ok16p(g[i].dgs) || g[i].ld_dgs % 4)) || !g[i].dWp || !g[i].dbp || !g[i].dWg || !g[i].dbg || !g[i].dW1 ||
        !g[i].db1 || !g[i].dW2 || !g[i].db2)
      return DMP_ERR_BAD_ARG;
  }
  if (B == 0) return DMP_OK;   // the caller zero-fills the weight gradients of an empty batch
  Heads t;
  WJobs wj;
  wj.n = 0; wj.B = B;
  int blocks = 0;
  for (int i = 0; i < num_heads; ++i) {
    t.w[i] = w[i]; t.io[i] = io[i]; t.gr[i] = g[i];
    add_wjob(wj, blocks, g[i].dY1, kH, kH, nullptr, io[i].F, kF, kF, g[i].dW1, kF, g[i].db1, 1.0f);
    add_wjob(wj, blocks, g[i].dP, kH, kH, nullptr, io[i].ps, io[i].ld_ps, kH, g[i].dWp, kH, g[i].dbp, io[i].scale_p);
    add_wjob(wj, blocks, g[i].dG, kH, kH, nullptr, io[i].gs, io[i].ld_gs, kH, g[i].dWg, kH, g[i].dbg, io[i].scale_g);
    add_wjob(wj, blocks, g[i].dy, 1, 1, g[i].dy_scale, io[i].Y1S, kYS, kYS, g[i].dW2, kYS, g[i].db2, 1.0f);
  }
  t.B = B; t.slope = slope;
  const dim3 grid((unsigned)((B + kRows - 1) / kRows), (unsigned)num_heads);
  if (H == 128) heads_bwd_rows_k<128><<<grid, kThreads, 0, (hipStream_t)stream>>>(t);
  else heads_bwd_rows_k<64><<<grid, kThreads, 0, (hipStream_t)stream>>>(t);
  int rc = check_launch();
  if (rc != DMP_OK) return rc;
  heads_bwd_w_k<<<(unsigned)blocks, kThreads, 0, (hipStream_t)stream>>>(wj);
  return check_launch();
}

int dmp_count_loss(const float *pred, const float *target, int64_t n, int kind, float neg_slope, float *loss, float *dpred, void *stream) {
  if (n < 1 || kind < 0 || kind > 2 || !pred || !target || !loss || !dpred) return DMP_ERR_BAD_ARG;
  if (n > ((int64_t)1 << 22) || !(neg_slope >= 0.f && neg_slope <= 1.f)) return DMP_ERR_UNSUPPORTED;     // (one workgroup: a batch's counts)
  count_loss_k<<<1, kLossThreads, 0, (hipStream_t)stream>>>(pred, target, n, kind, neg_slope, loss, dpred);
  return check_launch();
}

}  // extern "C"
