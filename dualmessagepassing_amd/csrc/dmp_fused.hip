// Row-wise epilogue kernels of the fused DMPLayer (gfx950): the elementwise steps between
// the GEMMs and the aggregation kernels, each a single streaming pass that also emits the
// column sums the bias gradients need (as per-workgroup partial rows, reduced by
// reduce_partials in a fixed order -> bit-stable, no atomics).
//
//   gate_residual        out  = prev + gate (.) upd            (dmpnn.py:263-273: v*gate, residual add)
//   scale_rows_colsum    dUpd = gate (.) dOut ; colsum(dUpd)   (its backward + d b2 of the MLP)
//   relu_bwd_colsum      dPre = act>0 ? dH : 0 ; colsum(dPre)  (ReLU backward + d b0 of the MLP)
//   bwd_g_colsum         dG   = [dY | coef[dst] dY] ; colsum(dY)  (edge_combine backward + d ebias)
//   colsum               colsum(A)
//   reduce_partials      out[l] = sum_s partial[s, l]          (also the split-K dW reduction)
//
// All HBM-bound; G = H/4 lanes per row, float4 per lane, U rows in flight per group,
// persistent workgroups (<= kMaxPartials) walking row chunks in a grid-stride loop.
#include <math.h>

#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kMaxPartials = 1024;  // partial rows written by one launch (4 workgroups per CU)
constexpr int kU = 4;               // rows per group per iteration

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void add4(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float4 mul4(const float4 &a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

enum { OP_GATE_RES = 0, OP_SCALE_CS = 1, OP_RELU_BWD_CS = 2, OP_BWD_G_CS = 3, OP_COLSUM = 4, OP_RELU_BWD_G_CS = 5,
       OP_ADD_BIAS_RELU = 6, OP_RELU_BWD_GATHER_CS = 7 };

struct RowArgs {
  const float *a; int64_t lda;   // first input  (prev | dOut | dH  | dY | A)
  const float *b; int64_t ldb;   // second input (upd  |  -   | act | -  | -)
  const float *rowscale;         // gate [R] or coef [N] (OP_BWD_G_CS, indexed through dst)
  const int32_t *dst;            // OP_BWD_G_CS only
  float *out; int64_t ldo;       // output (may be NULL for OP_SCALE_CS without gate / OP_COLSUM)
  float *partial;                // [gridDim.x, H] column-sum partials (NULL for OP_GATE_RES)
  int64_t R; int H;
  float slope;                   // negative slope of the activation (0 = ReLU): OP_ADD_BIAS_RELU, OP_RELU_BWD_*
};

template <int G, int OP>
__global__ __launch_bounds__(kBlock) void rowop_kernel(RowArgs p) {
  constexpr int GPB = kBlock / G;
  __shared__ float4 red[kBlock];
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int64_t chunk = (int64_t)GPB * kU;
  for (int c0 = 0; c0 < p.H; c0 += G * 4) {
    const int c = c0 + lane * 4;
    const bool act = c < p.H;
    float4 cs = zero4();
    const float4 colv = (OP == OP_ADD_BIAS_RELU && act && p.rowscale) ? ld4(p.rowscale + c) : zero4();   // the bias of this lane's columns
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < p.R; base += (int64_t)gridDim.x * chunk) {
      const int64_t r0 = base + (int64_t)grp * kU;
      // per-row scalars: lanes 0..kU-1 fetch, then broadcast (all lanes of the group take part)
      float ms = 1.f;
      if (lane < kU && r0 + lane < p.R) {
        if (OP == OP_GATE_RES || OP == OP_SCALE_CS) ms = p.rowscale ? p.rowscale[r0 + lane] : 1.f;
        if (OP == OP_BWD_G_CS || OP == OP_RELU_BWD_G_CS) ms = p.rowscale[p.dst[r0 + lane]];
      }
      int mi = -1;                                   // OP_RELU_BWD_GATHER_CS: the table row of this row's upstream gradient
      if (OP == OP_RELU_BWD_GATHER_CS && lane < kU && r0 + lane < p.R) {
        ms = p.rowscale ? p.rowscale[r0 + lane] : 1.f;
        mi = p.dst[r0 + lane];
      }
      float sc[kU];
      int tr[kU];
#pragma unroll
      for (int k = 0; k < kU; ++k) {
        sc[k] = __shfl(ms, k, G);
        tr[k] = OP == OP_RELU_BWD_GATHER_CS ? __shfl(mi, k, G) : 0;
      }
      if (!act) continue;
      float4 x[kU], y[kU];
#pragma unroll
      for (int k = 0; k < kU; ++k) {
        const int64_t r = r0 + k;
        if (r < p.R) {
          if (OP == OP_GATE_RES) {
            x[k] = p.a ? ld4(p.a + r * p.lda + c) : zero4();
            y[k] = ld4(p.b + r * p.ldb + c);
          } else if (OP == OP_RELU_BWD_GATHER_CS) {     // upstream row = rowscale[r] * table[map[r]] (map < 0: zero)
            x[k] = tr[k] >= 0 ? mul4(ld4(p.a + (int64_t)tr[k] * p.lda + c), sc[k]) : zero4();
            y[k] = ld4(p.b + r * p.ldb + c);
          } else if (OP == OP_RELU_BWD_CS || OP == OP_RELU_BWD_G_CS || OP == OP_ADD_BIAS_RELU) {
            x[k] = ld4(p.a + r * p.lda + c);
            y[k] = ld4(p.b + r * p.ldb + c);
          } else {
            x[k] = ld4(p.a + r * p.lda + c);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kU; ++k) {
        const int64_t r = r0 + k;
        if (r >= p.R) continue;
        if (OP == OP_GATE_RES) {
          float4 t = p.rowscale ? mul4(y[k], sc[k]) : y[k];  // (upd * gate), then + prev: reference order
          if (p.a) add4(t, x[k]);
          st4(p.out + r * p.ldo + c, t);
        } else if (OP == OP_ADD_BIAS_RELU) {
          float4 t = x[k];
          add4(t, y[k]);
          add4(t, colv);
          st4(p.out + r * p.ldo + c, make_float4(act_fwd(t.x, p.slope), act_fwd(t.y, p.slope), act_fwd(t.z, p.slope), act_fwd(t.w, p.slope)));
        } else if (OP == OP_SCALE_CS) {
          float4 t = p.rowscale ? mul4(x[k], sc[k]) : x[k];
          if (p.out) st4(p.out + r * p.ldo + c, t);
          add4(cs, t);
        } else if (OP == OP_RELU_BWD_CS || OP == OP_RELU_BWD_GATHER_CS) {
          float4 t = make_float4(act_bwd(y[k].x, x[k].x, p.slope), act_bwd(y[k].y, x[k].y, p.slope),
                                 act_bwd(y[k].z, x[k].z, p.slope), act_bwd(y[k].w, x[k].w, p.slope));
          st4(p.out + r * p.ldo + c, t);
          add4(cs, t);
        } else if (OP == OP_BWD_G_CS) {
          st4(p.out + r * p.ldo + c, x[k]);
          st4(p.out + r * p.ldo + p.H + c, mul4(x[k], sc[k]));
          add4(cs, x[k]);
        } else if (OP == OP_RELU_BWD_G_CS) {
          float4 t = make_float4(act_bwd(y[k].x, x[k].x, p.slope), act_bwd(y[k].y, x[k].y, p.slope),
                                 act_bwd(y[k].z, x[k].z, p.slope), act_bwd(y[k].w, x[k].w, p.slope));
          st4(p.out + r * p.ldo + c, t);
          st4(p.out + r * p.ldo + p.H + c, mul4(t, sc[k]));
          add4(cs, t);
        } else {
          add4(cs, x[k]);
        }
      }
    }
    if (OP != OP_GATE_RES && OP != OP_ADD_BIAS_RELU) {
      // fixed-order combine of the groups' column partials
      red[threadIdx.x] = cs;
      __syncthreads();
      if (grp == 0 && act) {
        float4 t = red[lane];
#pragma unroll
        for (int g = 1; g < GPB; ++g) add4(t, red[g * G + lane]);
        st4(p.partial + (int64_t)blockIdx.x * p.H + c, t);
      }
      __syncthreads();
    }
  }
}

// out[l..l+3] = sum_s partial[s, l..l+3], s ascending within a group, groups combined in order.
template <bool ACCUM>
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float *__restrict__ partial, int64_t S,
                                                                 int64_t L, float *__restrict__ out) {
  constexpr int G = 32, GPB = kBlock / G;
  __shared__ float4 red[kBlock];
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int64_t l = ((int64_t)blockIdx.x * G + lane) * 4;
  float4 acc = zero4();
  if (l < L) {
    // group g owns the contiguous slice of S/GPB rows -> one fixed summation order
    const int64_t per = (S + GPB - 1) / GPB;
    const int64_t s0 = grp * per, s1 = min(S, s0 + per);
    int64_t s = s0;
    for (; s + 4 <= s1; s += 4) {
      const float4 a = ld4(partial + s * L + l), b = ld4(partial + (s + 1) * L + l);
      const float4 c = ld4(partial + (s + 2) * L + l), d = ld4(partial + (s + 3) * L + l);
      add4(acc, a); add4(acc, b); add4(acc, c); add4(acc, d);
    }
    for (; s < s1; ++s) add4(acc, ld4(partial + s * L + l));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp == 0 && l < L) {
    float4 t = red[lane];
#pragma unroll
    for (int g = 1; g < GPB; ++g) add4(t, red[g * G + lane]);
    if (ACCUM) add4(t, ld4(out + l));
    st4(out + l, t);
  }
}

// Several independent reductions in one launch: workgroup b serves segment i with blk0[i] <= b < blk0[i+1];
// the order of the additions of every segment is that of reduce_partials_kernel.
constexpr int kMaxSegments = DMP_REDUCE_MAX_SEGMENTS;
struct ReduceSegs {
  const float *partial[kMaxSegments];
  float *out[kMaxSegments];
  int64_t S[kMaxSegments], L[kMaxSegments];
  int blk0[kMaxSegments + 1];
  int n;
};

// (measured, round 5: an XCD-contiguous order of the column chunks -- xcd_remap of the block index -- is SLOWER, 32.8 -> 39.7 us at
// the layer backward's 133 MB: the plain order spreads every partial row over all XCDs' memory channels at once)
__global__ __launch_bounds__(kBlock) void reduce_partials_multi_kernel(const ReduceSegs a) {
  constexpr int G = 32, GPB = kBlock / G;
  __shared__ float4 red[kBlock];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.blk0[i + 1]) ++i;
  const float *__restrict__ partial = a.partial[i];
  const int64_t S = a.S[i], L = a.L[i];
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int64_t l = ((int64_t)((int)blockIdx.x - a.blk0[i]) * G + lane) * 4;
  float4 acc = zero4();
  if (l < L) {
    const int64_t per = (S + GPB - 1) / GPB;
    const int64_t s0 = grp * per, s1 = min(S, s0 + per);
    int64_t s = s0;
    for (; s + 8 <= s1; s += 8) {                           // eight rows in flight per thread: the pass is latency-bound otherwise
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(partial + (s + u) * L + l);
#pragma unroll
      for (int u = 0; u < 8; ++u) add4(acc, v[u]);
    }
    for (; s + 4 <= s1; s += 4) {
      const float4 x = ld4(partial + s * L + l), y = ld4(partial + (s + 1) * L + l);
      const float4 z = ld4(partial + (s + 2) * L + l), w = ld4(partial + (s + 3) * L + l);
      add4(acc, x); add4(acc, y); add4(acc, z); add4(acc, w);
    }
    for (; s < s1; ++s) add4(acc, ld4(partial + s * L + l));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp == 0 && l < L) {
    float4 t = red[lane];
#pragma unroll
    for (int g = 1; g < GPB; ++g) add4(t, red[g * G + lane]);
    st4(a.out[i] + l, t);
  }
}

// AdamW (decoupled weight decay, optional AMSGrad) over one flat fp32 buffer: torch.optim.AdamW's
// single-tensor update, element by element, in one launch.
struct AdamArgs {
  float *p; const float *g; float *m, *v, *vmax;
  int64_t n;
  float decay, beta1_c, beta2, beta2_c, step_size, inv_bc2_sqrt, eps;
  int nskip;                                                  // ranges [skip_lo, skip_hi) left untouched (4-float aligned)
  int64_t skip_lo[DMP_ADAMW_MAX_SKIP], skip_hi[DMP_ADAMW_MAX_SKIP];
  // device-resident step count and learning rate (a step captured in a HIP graph: nothing per-step in the launch arguments)
  const double *state;                                        // [step, lr] or NULL (the fields above hold the host's values)
  double lr_wd, beta1, beta2d;                                // state != NULL: weight decay, betas for the bias corrections
  const int32_t *veto;                                        // dmp_adamw_step_guarded: veto[2] != 0 -> this step is dropped
  // dmp_adamw_step_segments: the buffer is P parameter tensors back to back (segment s = elements [seg_off[s], seg_off[s + 1])),
  // each with its OWN step count as torch.optim.AdamW keeps it; seg_tab[2 s] = lr / (1 - beta1^step_s) or < 0 for a segment
  // without a gradient this step, seg_tab[2 s + 1] = 1 / sqrt(1 - beta2^step_s) (written by adamw_tick_segments)
  const int64_t *seg_off; const float *seg_tab; int P;
};

// step += 1 -- unless the step is vetoed (dmp_adamw_step_guarded): veto[0] = flags raised since the last optimizer step
// (e.g. by dmp_gate_compact: a batch that did not fit its capacity), veto[1] = every flag ever raised (for the host),
// veto[2] = was this step dropped, veto[3] = steps dropped so far
__global__ void adamw_tick(double *state, int32_t *veto, int32_t mask) {
  bool drop = false;
  if (veto) {
    const int32_t f = veto[0];
    drop = (f & mask) != 0;
    veto[1] |= f;
    veto[0] = 0;
    veto[2] = drop ? 1 : 0;
    if (drop) veto[3] += 1;
  }
  if (!drop) state[0] += 1.0;
}

// per-segment step counts (state[2 + s]) and step sizes: one block; the veto logic of adamw_tick first
struct SegTick { double *state; float *tab; int P; double beta1, beta2; int32_t *veto; int32_t mask; uint64_t live[DMP_ADAMW_MAX_SEGMENTS / 64]; int all;
                 const float *live_dev; };   // live_dev [P] (device, or NULL): != 0 = the tensor has a gradient on SOME rank (dp.FlatGradSync)
__global__ __launch_bounds__(kBlock) void adamw_tick_segments(const SegTick t) {
  __shared__ int drop_s;
  if (threadIdx.x == 0) {
    bool drop = false;
    if (t.veto) {
      const int32_t f = t.veto[0];
      drop = (f & t.mask) != 0;
      t.veto[1] |= f;
      t.veto[0] = 0;
      t.veto[2] = drop ? 1 : 0;
      if (drop) t.veto[3] += 1;
    }
    if (!drop) t.state[0] += 1.0;                              // optimizer steps taken (what a single-count state reports)
    drop_s = drop ? 1 : 0;
  }
  __syncthreads();
  const double lr = t.state[1];
  for (int s = threadIdx.x; s < t.P; s += kBlock) {
    const bool live = t.live_dev ? t.live_dev[s] != 0.f : (t.all || ((t.live[s >> 6] >> (s & 63)) & 1ull));
    if (live && !drop_s) {
      const double step = t.state[2 + s] + 1.0;
      t.state[2 + s] = step;
      t.tab[2 * s] = (float)(lr / (1.0 - pow(t.beta1, step)));
      t.tab[2 * s + 1] = (float)(1.0 / sqrt(1.0 - pow(t.beta2, step)));
    } else {
      t.tab[2 * s] = -1.f;
    }
  }
}

struct PackSegs {
  const float *src[DMP_PACK_MAX_SEGMENTS];
  int64_t off[DMP_PACK_MAX_SEGMENTS];
  int64_t len[DMP_PACK_MAX_SEGMENTS];
  int pad;         // != 0: the floats between a segment's end and the next multiple of 4 are cleared too (a buffer laid out in 16-byte pieces)
};

// blockIdx.y = segment, blockIdx.x strides over it in float4 steps (scalar loads for unaligned sources and tails)
__global__ __launch_bounds__(kBlock) void pack_segments_kernel(const PackSegs a, float *__restrict__ dst) {
  const int s = blockIdx.y;
  const float *__restrict__ src = a.src[s];
  float *__restrict__ out = dst + a.off[s];
  const int64_t n = a.len[s];
  const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * kBlock * 4) {
    if (!src) {                                               // a segment without a source: zeros (a parameter without a gradient)
      if (i + 4 <= n || a.pad) *reinterpret_cast<float4 *>(out + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      else for (int64_t j = i; j < n; ++j) out[j] = 0.f;
    } else if (vec && i + 4 <= n) {
      *reinterpret_cast<float4 *>(out + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
      for (int64_t j = i; j < i + 4 && (j < n || a.pad); ++j) out[j] = j < n ? src[j] : 0.f;
    }
  }
}

// (nothing of ``a`` is written and its arrays are indexed by unrolled constants only: a by-value argument that is modified or
// indexed by a run-time value is copied to scratch by every lane -- 408 bytes per lane, 71 MB per launch, 42 instead of 9 us)
__global__ __launch_bounds__(kBlock) void adamw_kernel(const AdamArgs a) {
  if (a.veto && a.veto[2]) return;                             // a dropped step: parameters and moments stay as they are
  float decay = a.decay, step_size = a.step_size, inv_bc2_sqrt = a.inv_bc2_sqrt;
  if (a.seg_tab) {
    decay = (float)(1.0 - a.state[1] * a.lr_wd);
  } else if (a.state) {                                        // the host formulas of dmp_adamw_step_skip, evaluated on the device
    const double step = a.state[0], lr = a.state[1];
    const double bc1 = 1.0 - pow(a.beta1, step), bc2 = 1.0 - pow(a.beta2d, step);
    decay = (float)(1.0 - lr * a.lr_wd);
    step_size = (float)(lr / bc1);
    inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  }
  const int64_t stride = (int64_t)gridDim.x * kBlock * 4;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; i < a.n; i += stride) {
    bool skip = false;                                         // parameters without a gradient this step (torch.optim.AdamW skips them)
    if (a.nskip > 0) {
#pragma unroll
      for (int s = 0; s < DMP_ADAMW_MAX_SKIP; ++s) skip |= (s < a.nskip && i >= a.skip_lo[s] && i < a.skip_hi[s]);
    }
    if (skip) continue;
    if (a.seg_tab) {                                           // this element's parameter tensor: its own bias corrections
      int lo = 0, hi = a.P;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.seg_off[mid] <= i) lo = mid; else hi = mid;
      }
      step_size = a.seg_tab[2 * lo];
      if (step_size < 0.f) continue;                           // no gradient this step: untouched, its step count stands
      inv_bc2_sqrt = a.seg_tab[2 * lo + 1];
    }
    float p[4], g[4], m[4], v[4], vm[4];
    const bool full = i + 4 <= a.n;
    const int cnt = full ? 4 : (int)(a.n - i);
    if (full) {
      *reinterpret_cast<float4 *>(p) = ld4(a.p + i); *reinterpret_cast<float4 *>(g) = ld4(a.g + i);
      *reinterpret_cast<float4 *>(m) = ld4(a.m + i); *reinterpret_cast<float4 *>(v) = ld4(a.v + i);
      if (a.vmax) *reinterpret_cast<float4 *>(vm) = ld4(a.vmax + i);
    } else {
      for (int k = 0; k < cnt; ++k) { p[k] = a.p[i + k]; g[k] = a.g[i + k]; m[k] = a.m[i + k]; v[k] = a.v[i + k]; if (a.vmax) vm[k] = a.vmax[i + k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= cnt) break;
      p[k] = p[k] * decay;
      m[k] = m[k] + a.beta1_c * (g[k] - m[k]);
      v[k] = v[k] * a.beta2 + a.beta2_c * g[k] * g[k];
      float d = v[k];
      if (a.vmax) { vm[k] = fmaxf(vm[k], v[k]); d = vm[k]; }
      const float denom = __fsqrt_rn(d) * inv_bc2_sqrt + a.eps;
      p[k] = p[k] - step_size * (m[k] / denom);
    }
    if (full) {
      st4(a.p + i, *reinterpret_cast<float4 *>(p)); st4(a.m + i, *reinterpret_cast<float4 *>(m));
      st4(a.v + i, *reinterpret_cast<float4 *>(v));
      if (a.vmax) st4(a.vmax + i, *reinterpret_cast<float4 *>(vm));
    } else {
      for (int k = 0; k < cnt; ++k) { a.p[i + k] = p[k]; a.m[i + k] = m[k]; a.v[i + k] = v[k]; if (a.vmax) a.vmax[i + k] = vm[k]; }
    }
  }
}

// Weight gradient of a narrow input layer fused with a row gate:  out[k, :] = sum_r X[r, k] * gate[r] * D[r, :]
// (X [R, K <= 16]: the multihot label encodings, D [R, 128]: the upstream gradient of gate * (X W)).  One pass
// over D instead of materialising gate * D and running a [K, R] x [R, 128] product over it.  A wave owns a row
// at a time (float2 per lane), so the row's K inputs and its gate are wave-uniform: one vector load, v_readlane, SGPR operands
// of the FMAs; 4 rows in flight per wave.
constexpr int kSmallK = 16;
constexpr int kSmallRows = 4;                                 // rows in flight per wave (the loop is one memory round trip per batch)
struct SmallKArgs { const float *X; int64_t ldx; int K; const float *D; int64_t ldd; const float *gate; int64_t R; float *partial;
                    int ncols; const float *D2; int64_t ldd2;      // blockIdx.y = column block j: D[:, jH:(j+1)H], or D2 for j == ncols
                    const uint32_t *rowmask;                       // bit r of rowmask[t] == 0: row 32 t + r of X is all zeros (or its gate is 0): its D rows are not fetched
                    const int32_t *list, *count; };                // the rows to add, ascending (dmp_kept_rows), *count of them: a batch is then kSmallRows LIVE rows

// VW = H / 64 values per lane: 2 (H = 128, one float2 per lane) or 1 (H = 64)
template <int VW> struct LaneVec { float v[VW]; };
template <int VW> __device__ __forceinline__ LaneVec<VW> lane_load(const float *row, int lane) {
  LaneVec<VW> o;
  if (VW == 2) { const float2 t = *reinterpret_cast<const float2 *>(row + lane * 2); o.v[0] = t.x; o.v[VW - 1] = t.y; }
  else o.v[0] = row[lane];
  return o;
}
template <int VW> __device__ __forceinline__ void lane_store(float *row, int lane, const LaneVec<VW> &o) {
  if (VW == 2) *reinterpret_cast<float2 *>(row + lane * 2) = make_float2(o.v[0], o.v[VW - 1]);
  else row[lane] = o.v[0];
}

// (nbx: the workgroups of this product -- gridDim.x, or fewer when several products of different lengths share a launch)
template <int K, int VW>
__device__ __forceinline__ void smallk_atb_body(const SmallKArgs &p, const unsigned nbx, float *red) {
  constexpr int WPB = kBlock / 64, kRows = kSmallRows, H = 64 * VW;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int colblk = blockIdx.y;                            // one launch for several column blocks of the upstream gradient
  const float *Dj = colblk < p.ncols ? p.D + (int64_t)colblk * H : p.D2;
  const int64_t ldj = colblk < p.ncols ? p.ldd : p.ldd2;
  float *partial = p.partial + (int64_t)colblk * nbx * K * H;
  LaneVec<VW> acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < VW; ++c) acc[k].v[c] = 0.f;
  const int64_t stride = (int64_t)nbx * WPB * kRows;
  // lanes 0..K-1 fetch a row's inputs, lane K its gate (one load each); the values are read back lane by lane
  // into wave-uniform operands (v_readlane: no memory traffic, no LDS).  The next batch's loads are issued
  // before the current batch's FMAs.
  LaneVec<VW> d[kRows], dn[kRows];
  float mine[kRows], minen[kRows];
  static_assert(32 % kSmallRows == 0, "a batch of rows lies inside one mask word");
  const int64_t limit = p.list ? (int64_t)*p.count : p.R;  // positions of the list / rows
  auto load_batch = [&](int64_t r0, LaneVec<VW> (&dd)[kRows], float (&mm)[kRows]) {
    // the batch's mask bits from ONE word; a dead row (or one past the end) loads row 0 of D instead of its own -- one cached
    // line, no traffic, and no branch around the load: a wave-uniform condition around a load is a scalar branch, a basic block
    // and a wait per row (finding (t) of DESIGN 8); its products are skipped below
    const uint32_t word = (!p.list && p.rowmask && r0 < p.R) ? (p.rowmask[r0 >> 5] >> (r0 & 31)) : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t q = r0 + u;                             // wave-uniform: a row, or a position of the list
      const bool ok = q < limit;
      const int64_t r = p.list ? (int64_t)p.list[ok ? q : (limit > 0 ? limit - 1 : 0)] : q;      // (clamped: the load is issued either way)
      const bool live = ok && ((word >> u) & 1u);
      dd[u] = lane_load<VW>(Dj + (live ? r : (int64_t)0) * ldj, lane);
      // one load instruction for both: lanes < K point into the row of X, lane K at the row's gate
      const float *src = lane < K ? p.X + r * p.ldx + lane : p.gate + r;
      float m = (lane == K && !p.gate) ? 1.f : 0.f;
      if (ok && (lane < K || (lane == K && p.gate))) m = *src;
      mm[u] = m;
    }
  };
  int64_t r0 = ((int64_t)blockIdx.x * WPB + wave) * kRows;
  if (r0 < limit) load_batch(r0, d, mine);
  for (; r0 < limit; r0 += stride) {
    load_batch(r0 + stride, dn, minen);                     // rows past the end read as zeros
    const uint32_t wcur = (!p.list && p.rowmask) ? (p.rowmask[r0 >> 5] >> (r0 & 31)) : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (r0 + u >= limit || !((wcur >> u) & 1u)) continue;  // a masked row / a row past the end: all of its products are zero (wave-uniform)
      const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine[u]), K));
      const float scaled = mine[u] * g;                     // lane k: gate * X[r, k]
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, scaled), k));
        // v_fmac with the wave-uniform operand straight from its SGPR (the compiler would broadcast it into a
        // VGPR pair for a packed FMA: two extra moves per FMA in a VALU-bound loop); a sum: no order to keep
#pragma unroll
        for (int c = 0; c < VW; ++c) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[k].v[c]) : "s"(x), "v"(d[u].v[c]));
      }
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) { d[u] = dn[u]; mine[u] = minen[u]; }
  }
  // fixed-order combine of the 4 waves, one k at a time
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int c = 0; c < VW; ++c) red[threadIdx.x * VW + c] = acc[k].v[c];
    __syncthreads();
    if (wave == 0) {
      LaneVec<VW> t;
#pragma unroll
      for (int c = 0; c < VW; ++c) {
        t.v[c] = red[lane * VW + c];
#pragma unroll
        for (int w = 1; w < WPB; ++w) t.v[c] += red[(w * 64 + lane) * VW + c];
      }
      lane_store<VW>(partial + ((int64_t)blockIdx.x * K + k) * H, lane, t);
    }
    __syncthreads();
  }
}

template <int K, int VW>
__global__ __launch_bounds__(kBlock) void smallk_atb_k(const SmallKArgs p) {
  __shared__ float red[kBlock * VW];
  smallk_atb_body<K, VW>(p, gridDim.x, red);
}

// Several such products of the same K and H in ONE launch (blockIdx.z = the product; one column block each): the first layer's
// node-code weight gradients -- two tables x two halves of the code sums, 5-12 us apiece as launches of their own.
constexpr int kSmallKJobs = 4;
struct SmallKJobs { SmallKArgs job[kSmallKJobs]; unsigned nb[kSmallKJobs]; };
template <int K, int VW>
__global__ __launch_bounds__(kBlock) void smallk_atb_jobs_k(const SmallKJobs js) {
  __shared__ float red[kBlock * VW];
  const unsigned nbx = js.nb[blockIdx.z];
  if (blockIdx.x >= nbx) return;
  smallk_atb_body<K, VW>(js.job[blockIdx.z], nbx, red);
}

// The forward of the same narrow layer with its gate:  out[r, :] = gate[r] * sum_k X[r, k] W[k, :]  -- the gated
// embedding rows written straight into their place (the union buffer of the joint rep-net pass) from the K
// inputs per row instead of from the [R, H] embedding.  W lives in registers (H / 64 values per lane and input).
struct SmallKFwdArgs { const float *X; int64_t ldx; const float *W; int64_t ldw; const float *gate; int64_t R; float *out; int64_t ldo;   // blockIdx.y: column block of W / out
                       int skip_zero; };   // rows whose gate is 0 are not stored (DEAD rows: every reader leaves them out)

template <int K, int VW>
__global__ __launch_bounds__(kBlock) void smallk_embed_k(const SmallKFwdArgs p) {
  constexpr int WPB = kBlock / 64, kRows = kSmallRows;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int64_t coloff = (int64_t)blockIdx.y * (64 * VW);
  LaneVec<VW> w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = lane_load<VW>(p.W + k * p.ldw + coloff, lane);
  const int64_t stride = (int64_t)gridDim.x * WPB * kRows;
  for (int64_t r0 = ((int64_t)blockIdx.x * WPB + wave) * kRows; r0 < p.R; r0 += stride) {
    float mine[kRows];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t r = r0 + u;
      const bool ok = r < p.R;
      const float *src = lane < K ? p.X + r * p.ldx + lane : p.gate + r;
      float m = (lane == K && !p.gate) ? 1.f : 0.f;
      if (ok && (lane < K || (lane == K && p.gate))) m = *src;
      mine[u] = m;
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int64_t r = r0 + u;
      if (r >= p.R) break;
      const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine[u]), K));
      if (p.skip_zero && g == 0.f) continue;                  // wave-uniform
      LaneVec<VW> e;
#pragma unroll
      for (int c = 0; c < VW; ++c) e.v[c] = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {                         // sum over k ascending, then the gate: (X W) * gate as the reference
        const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine[u]), k));
#pragma unroll
        for (int c = 0; c < VW; ++c) e.v[c] += x * w[k].v[c];
      }
#pragma unroll
      for (int c = 0; c < VW; ++c) e.v[c] *= g;
      lane_store<VW>(p.out + r * p.ldo + coloff, lane, e);
    }
  }
}

template <int K>
void launch_smallk_fwd(const SmallKFwdArgs &p, int H, int ncols, hipStream_t st) {
  const int64_t chunk = (int64_t)(kBlock / 64) * kSmallRows, nb = (p.R + chunk - 1) / chunk;
  const dim3 grid((unsigned)(nb < 4096 ? nb : 4096), (unsigned)ncols);
  if (H == 128) smallk_embed_k<K, 2><<<grid, kBlock, 0, st>>>(p);
  else smallk_embed_k<K, 1><<<grid, kBlock, 0, st>>>(p);
}

inline unsigned smallk_blocks(int64_t R) {
  const int64_t chunk = (int64_t)(kBlock / 64) * kSmallRows, nb = (R + chunk - 1) / chunk;
  return (unsigned)(nb < kMaxPartials ? (nb > 0 ? nb : 1) : kMaxPartials);
}

template <int K>
void launch_smallk(const SmallKArgs &p, int H, hipStream_t st) {
  const dim3 grid(smallk_blocks(p.R), (unsigned)(p.ncols + (p.D2 ? 1 : 0)));
  if (H == 128) smallk_atb_k<K, 2><<<grid, kBlock, 0, st>>>(p);
  else smallk_atb_k<K, 1><<<grid, kBlock, 0, st>>>(p);
}

template <int K>
static void launch_smallk_jobs(const SmallKJobs &js, unsigned nbmax, int n, int H, hipStream_t st) {
  const dim3 grid(nbmax, 1u, (unsigned)n);
  if (H == 128) smallk_atb_jobs_k<K, 2><<<grid, kBlock, 0, st>>>(js);
  else smallk_atb_jobs_k<K, 1><<<grid, kBlock, 0, st>>>(js);
}

inline int group_lanes(int H) { return H <= 64 ? 16 : (H <= 128 ? 32 : 64); }

inline unsigned grid_for(int64_t R, int G) {
  const int64_t chunk = (int64_t)(kBlock / G) * kU;
  const int64_t nb = (R + chunk - 1) / chunk;
  return (unsigned)(nb < kMaxPartials ? (nb > 0 ? nb : 1) : kMaxPartials);
}


// The last layer's activation backward with its pooled companion (see dmp_relu_bwd_gathered_colsum): one pass over the saved
// activation H1 [R, H] walks the pooling index's chunks (vptr / vent: <= 64 rows of one graph each, (row << 1) | flag) and
// produces  dPre[r] = act'(H1[r]) (.) (gate[r] table[rowmap[r]])  (rowmap < 0: zero),  the column sums of dPre, AND the
// gated per-chunk sums  Qc[c] = [sum_{flag=0} gate[r] H1[r] | sum_{flag=1} ...]  that the weight gradient of the layer's
// second Linear needs (dW2 = T^T Q): H1 is read once for both.  A G-lane group per chunk, four rows in flight.
struct PoolBwdArgs {
  const float *h1; int64_t ldh; const float *table; int64_t ldt; const int32_t *rowmap; const float *gate;
  const int32_t *vptr, *vent; int64_t V; int H; float slope;
  float *dpre; int64_t ldp; float *qc; float *partial;
};

template <int G>
__global__ __launch_bounds__(kBlock) void pool_relu_bwd_k(PoolBwdArgs p) {
  constexpr int GPB = kBlock / G;
  __shared__ float4 red[kBlock];
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const int c = lane * 4;
  const bool act = c < p.H;
  float4 cs = zero4();
  for (int64_t ch = (int64_t)blockIdx.x * GPB + grp; ch < p.V; ch += (int64_t)gridDim.x * GPB) {
    const int beg = p.vptr[ch], end = p.vptr[ch + 1];
    float4 q0 = zero4(), q1 = zero4();
    for (int base = beg; base < end; base += kU) {
      // per-row scalars of up to kU rows: lanes 0..kU-1 fetch, then broadcast
      int ent = 0, mi = -1;
      float gt = 0.f;
      if (lane < kU && base + lane < end) {
        ent = p.vent[base + lane];
        const int r = ent >> 1;
        mi = p.rowmap[r];
        gt = p.gate ? p.gate[r] : 1.f;
      }
      float4 hv[kU], uv[kU];
      int en[kU], tr[kU];
      float sc[kU];
#pragma unroll
      for (int k = 0; k < kU; ++k) { en[k] = __shfl(ent, k, G); tr[k] = __shfl(mi, k, G); sc[k] = __shfl(gt, k, G); }
      if (!act) continue;
#pragma unroll
      for (int k = 0; k < kU; ++k)
        if (base + k < end) {
          // a row under a zero gate: dPre = act'(.) (0 u) = 0 and 0 h1 = 0 whatever h1 holds -- neither row is fetched
          const bool live = sc[k] != 0.f;
          hv[k] = live ? ld4(p.h1 + (int64_t)(en[k] >> 1) * p.ldh + c) : zero4();
          uv[k] = (live && tr[k] >= 0) ? ld4(p.table + (int64_t)tr[k] * p.ldt + c) : zero4();
        }
#pragma unroll
      for (int k = 0; k < kU; ++k)
        if (base + k < end) {
          const float4 u = mul4(uv[k], sc[k]);
          const float4 t = make_float4(act_bwd(hv[k].x, u.x, p.slope), act_bwd(hv[k].y, u.y, p.slope),
                                       act_bwd(hv[k].z, u.z, p.slope), act_bwd(hv[k].w, u.w, p.slope));
          st4(p.dpre + (int64_t)(en[k] >> 1) * p.ldp + c, t);
          add4(cs, t);
          const float4 gh = mul4(hv[k], sc[k]);
          if (en[k] & 1) add4(q1, gh); else add4(q0, gh);
        }
    }
    if (act) {
      st4(p.qc + ch * 2 * p.H + c, q0);
      st4(p.qc + ch * 2 * p.H + p.H + c, q1);
    }
  }
  red[threadIdx.x] = cs;
  __syncthreads();
  if (grp == 0 && act) {
    float4 t = red[lane];
#pragma unroll
    for (int g = 1; g < GPB; ++g) add4(t, red[g * G + lane]);
    st4(p.partial + (int64_t)blockIdx.x * p.H + c, t);
  }
}

template <int OP>
int launch_rowop(const RowArgs &p, hipStream_t st) {
  const int g = group_lanes(p.H);
  if (g == 16) rowop_kernel<16, OP><<<grid_for(p.R, 16), kBlock, 0, st>>>(p);
  else if (g == 32) rowop_kernel<32, OP><<<grid_for(p.R, 32), kBlock, 0, st>>>(p);
  else rowop_kernel<64, OP><<<grid_for(p.R, 64), kBlock, 0, st>>>(p);
  return check_launch();
}

inline bool ok16(const void *q) { return !q || aligned16(q); }

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int64_t dmp_colsum_partial_rows(int64_t rows, int H) {
  if (rows <= 0 || H <= 0) return 1;
  return (int64_t)grid_for(rows, group_lanes(H));
}

#define DMP_ROW_CHECK(cond) \
  if (!(cond)) return DMP_ERR_BAD_ARG

static int vec_shape_ok(int H, int64_t l0, int64_t l1, int64_t l2) {
  return H > 0 && H % 4 == 0 && l0 % 4 == 0 && l1 % 4 == 0 && l2 % 4 == 0;
}

int dmp_gate_residual(const float *prev, int64_t ldp, const float *upd, int64_t ldu, const float *gate,
                      int64_t R, int H, float *out, int64_t ldo, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0);
  if (R == 0) return DMP_OK;
  DMP_ROW_CHECK(upd && out && ldu >= H && ldo >= H && (!prev || ldp >= H));
  if (!vec_shape_ok(H, prev ? ldp : 0, ldu, ldo) || !ok16(prev) || !ok16(upd) || !ok16(out)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{prev, ldp, upd, ldu, gate, nullptr, out, ldo, nullptr, R, H};
  return launch_rowop<OP_GATE_RES>(p, (hipStream_t)stream);
}

int dmp_add_bias_relu(const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias, int64_t R, int H,
                      float slope, float *out, int64_t ldo, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0);
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return DMP_OK;
  DMP_ROW_CHECK(a && b && out && lda >= H && ldb >= H && ldo >= H);
  if (!vec_shape_ok(H, lda, ldb, ldo) || !ok16(a) || !ok16(b) || !ok16(out) || !ok16(bias)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{a, lda, b, ldb, bias, nullptr, out, ldo, nullptr, R, H, slope};
  return launch_rowop<OP_ADD_BIAS_RELU>(p, (hipStream_t)stream);
}

int dmp_scale_rows_colsum(const float *dOut, int64_t ldd, const float *gate, int64_t R, int H, float *dUpd,
                          int64_t ldu, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0 && partial);
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(dOut && ldd >= H && (!dUpd || ldu >= H) && (!gate || dUpd));
  if (!vec_shape_ok(H, ldd, dUpd ? ldu : 0, 0) || !ok16(dOut) || !ok16(dUpd) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{dOut, ldd, nullptr, 0, gate, nullptr, dUpd, ldu, partial, R, H};
  return launch_rowop<OP_SCALE_CS>(p, (hipStream_t)stream);
}

int dmp_relu_bwd_colsum(const float *dH, int64_t ldh, const float *act, int64_t lda, int64_t R, int H, float slope,
                        float *dPre, int64_t ldp, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0 && partial);
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(dH && act && dPre && ldh >= H && lda >= H && ldp >= H);
  if (!vec_shape_ok(H, ldh, lda, ldp) || !ok16(dH) || !ok16(act) || !ok16(dPre) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{dH, ldh, act, lda, nullptr, nullptr, dPre, ldp, partial, R, H, slope};
  return launch_rowop<OP_RELU_BWD_CS>(p, (hipStream_t)stream);
}

int dmp_relu_bwd_gathered_colsum(const float *table, int64_t ldt, const int32_t *rowmap, const float *gate, const float *act,
                                 int64_t lda, int64_t R, int H, float slope, float *dPre, int64_t ldp, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0 && partial);
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(table && rowmap && act && dPre && ldt >= H && lda >= H && ldp >= H);
  if (!vec_shape_ok(H, ldt, lda, ldp) || !ok16(table) || !ok16(act) || !ok16(dPre) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{table, ldt, act, lda, gate, rowmap, dPre, ldp, partial, R, H, slope};
  return launch_rowop<OP_RELU_BWD_GATHER_CS>(p, (hipStream_t)stream);
}

int64_t dmp_pool_relu_bwd_blocks(int64_t num_chunks, int H) {
  const int g = group_lanes(H);
  const int64_t nb = (num_chunks + kBlock / g - 1) / (kBlock / g);
  return nb < 1 ? 1 : (nb < kMaxPartials ? nb : kMaxPartials);
}

int dmp_pool_relu_bwd(const float *h1, int64_t ldh, const float *table, int64_t ldt, const int32_t *rowmap, const float *gate,
                      const int32_t *vptr, const int32_t *vent, int64_t num_chunks, int64_t R, int H, float slope, float *dPre,
                      int64_t ldp, float *chunk_sums, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && num_chunks >= 0 && H > 0 && partial);
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  const int64_t nb = dmp_pool_relu_bwd_blocks(num_chunks, H);
  if (R == 0 || num_chunks == 0)
    return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H * (size_t)nb, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(h1 && table && rowmap && vptr && vent && dPre && chunk_sums && ldh >= H && ldt >= H && ldp >= H);
  if (!vec_shape_ok(H, ldh, ldt, ldp) || H > 256 || !ok16(h1) || !ok16(table) || !ok16(dPre) || !ok16(chunk_sums) || !ok16(partial))
    return DMP_ERR_UNSUPPORTED;
  PoolBwdArgs p{h1, ldh, table, ldt, rowmap, gate, vptr, vent, num_chunks, H, slope, dPre, ldp, chunk_sums, partial};
  hipStream_t st = (hipStream_t)stream;
  const int g = group_lanes(H);
  if (g == 16) pool_relu_bwd_k<16><<<(unsigned)nb, kBlock, 0, st>>>(p);
  else if (g == 32) pool_relu_bwd_k<32><<<(unsigned)nb, kBlock, 0, st>>>(p);
  else pool_relu_bwd_k<64><<<(unsigned)nb, kBlock, 0, st>>>(p);
  return check_launch();
}

int dmp_edge_combine_bwd_g_colsum(const float *dY, int64_t ldy, const float *coef, const int32_t *dst,
                                  int64_t E, int H, float *dG, int64_t ldg, float *partial, void *stream) {
  DMP_ROW_CHECK(E >= 0 && H > 0 && partial);
  if (E == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(dY && coef && dst && dG && ldy >= H && ldg >= 2 * H);
  if (!vec_shape_ok(H, ldy, ldg, 0) || !ok16(dY) || !ok16(dG) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{dY, ldy, nullptr, 0, coef, dst, dG, ldg, partial, E, H};
  return launch_rowop<OP_BWD_G_CS>(p, (hipStream_t)stream);
}

int dmp_relu_bwd_g_colsum(const float *dH, int64_t ldh, const float *act, int64_t lda, const float *coef,
                          const int32_t *dst, int64_t E, int H, float slope, float *dG, int64_t ldg, float *partial,
                          void *stream) {
  DMP_ROW_CHECK(E >= 0 && H > 0 && partial);
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(dH && act && coef && dst && dG && ldh >= H && lda >= H && ldg >= 2 * H);
  if (!vec_shape_ok(H, ldh, lda, ldg) || !ok16(dH) || !ok16(act) || !ok16(dG) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{dH, ldh, act, lda, coef, dst, dG, ldg, partial, E, H, slope};
  return launch_rowop<OP_RELU_BWD_G_CS>(p, (hipStream_t)stream);
}

int dmp_colsum_partials(const float *A, int64_t lda, int64_t R, int H, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && H > 0 && partial);
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(A && lda >= H);
  if (!vec_shape_ok(H, lda, 0, 0) || !ok16(A) || !ok16(partial)) return DMP_ERR_UNSUPPORTED;
  RowArgs p{A, lda, nullptr, 0, nullptr, nullptr, nullptr, 0, partial, R, H};
  return launch_rowop<OP_COLSUM>(p, (hipStream_t)stream);
}

int64_t dmp_smallk_atb_blocks(int64_t rows) { return (int64_t)smallk_blocks(rows); }

static int dmp_smallk_atb_cols(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2, int64_t ldd2,
                        const float *gate, int64_t R, int H, float *partial, void *stream) {
  return dmp_smallk_atb_cols_masked(X, ldx, K, D, ldd, ncols, D2, ldd2, gate, nullptr, R, H, partial, stream);
}

static int smallk_atb_cols_impl(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2,
                                int64_t ldd2, const float *gate, const uint32_t *rowmask, const int32_t *list, const int32_t *count,
                                int64_t R, int H, float *partial, void *stream) {
  DMP_ROW_CHECK(R >= 0 && K > 0 && partial && ncols >= 0 && ncols + (D2 ? 1 : 0) >= 1 && ncols <= 8);
  if ((H != 128 && H != 64) || K > kSmallK) return DMP_ERR_UNSUPPORTED;
  const int nblk = ncols + (D2 ? 1 : 0);
  if (R == 0) return hipMemsetAsync(partial, 0, sizeof(float) * (size_t)nblk * K * H, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  DMP_ROW_CHECK(X && (D || ncols == 0) && ldx >= K && (ncols == 0 || ldd >= (int64_t)ncols * H) && (!D2 || ldd2 >= H));
  if (ldd % 2 || ldd2 % 2 || (reinterpret_cast<uintptr_t>(D) & 7u) || (reinterpret_cast<uintptr_t>(D2) & 7u) || !ok16(partial))
    return DMP_ERR_UNSUPPORTED;
  SmallKArgs p{X, ldx, K, D, ldd, gate, R, partial, ncols, D2, ldd2, rowmask, list, count};
  hipStream_t st = (hipStream_t)stream;
  switch (K) {   // K is a compile-time constant of the kernel: the accumulators live in registers
    case 1: launch_smallk<1>(p, H, st); break;   case 2: launch_smallk<2>(p, H, st); break;
    case 3: launch_smallk<3>(p, H, st); break;   case 4: launch_smallk<4>(p, H, st); break;
    case 5: launch_smallk<5>(p, H, st); break;   case 6: launch_smallk<6>(p, H, st); break;
    case 7: launch_smallk<7>(p, H, st); break;   case 8: launch_smallk<8>(p, H, st); break;
    case 9: launch_smallk<9>(p, H, st); break;   case 10: launch_smallk<10>(p, H, st); break;
    case 11: launch_smallk<11>(p, H, st); break; case 12: launch_smallk<12>(p, H, st); break;
    case 13: launch_smallk<13>(p, H, st); break; case 14: launch_smallk<14>(p, H, st); break;
    case 15: launch_smallk<15>(p, H, st); break; default: launch_smallk<16>(p, H, st); break;
  }
  return check_launch();
}

int dmp_smallk_atb_cols_masked(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2,
                               int64_t ldd2, const float *gate, const uint32_t *rowmask, int64_t R, int H, float *partial,
                               void *stream) {
  return smallk_atb_cols_impl(X, ldx, K, D, ldd, ncols, D2, ldd2, gate, rowmask, nullptr, nullptr, R, H, partial, stream);
}

int dmp_smallk_atb_jobs(const dmp_smallk_job *jobs, int num_jobs, int K, int H, void *stream) {
  DMP_ROW_CHECK(jobs && num_jobs >= 1 && num_jobs <= kSmallKJobs && K > 0);
  if ((H != 128 && H != 64) || K > kSmallK) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  SmallKJobs js{};
  unsigned nbmax = 0;
  int n = 0;
  for (int j = 0; j < num_jobs; ++j) {
    const dmp_smallk_job &q = jobs[j];
    DMP_ROW_CHECK(q.R >= 0 && q.partial);
    if (!ok16(q.partial)) return DMP_ERR_UNSUPPORTED;
    if (q.R == 0) {    // an empty product: its one partial row block is zeros
      if (hipMemsetAsync(q.partial, 0, sizeof(float) * (size_t)K * H, st) != hipSuccess) return DMP_ERR_HIP;
      continue;
    }
    DMP_ROW_CHECK(q.X && q.D && q.ldx >= K && q.ldd >= H);
    if (q.ldd % 2 || (reinterpret_cast<uintptr_t>(q.D) & 7u)) return DMP_ERR_UNSUPPORTED;
    js.job[n] = SmallKArgs{q.X, q.ldx, K, q.D, q.ldd, q.gate, q.R, q.partial, 1, nullptr, 0, q.rowmask, nullptr, nullptr};
    js.nb[n] = smallk_blocks(q.R);
    nbmax = js.nb[n] > nbmax ? js.nb[n] : nbmax;
    ++n;
  }
  if (n == 0) return DMP_OK;
  switch (K) {
    case 1: launch_smallk_jobs<1>(js, nbmax, n, H, st); break;   case 2: launch_smallk_jobs<2>(js, nbmax, n, H, st); break;
    case 3: launch_smallk_jobs<3>(js, nbmax, n, H, st); break;   case 4: launch_smallk_jobs<4>(js, nbmax, n, H, st); break;
    case 5: launch_smallk_jobs<5>(js, nbmax, n, H, st); break;   case 6: launch_smallk_jobs<6>(js, nbmax, n, H, st); break;
    case 7: launch_smallk_jobs<7>(js, nbmax, n, H, st); break;   case 8: launch_smallk_jobs<8>(js, nbmax, n, H, st); break;
    case 9: launch_smallk_jobs<9>(js, nbmax, n, H, st); break;   case 10: launch_smallk_jobs<10>(js, nbmax, n, H, st); break;
    case 11: launch_smallk_jobs<11>(js, nbmax, n, H, st); break; case 12: launch_smallk_jobs<12>(js, nbmax, n, H, st); break;
    case 13: launch_smallk_jobs<13>(js, nbmax, n, H, st); break; case 14: launch_smallk_jobs<14>(js, nbmax, n, H, st); break;
    case 15: launch_smallk_jobs<15>(js, nbmax, n, H, st); break; default: launch_smallk_jobs<16>(js, nbmax, n, H, st); break;
  }
  return check_launch();
}

int dmp_smallk_atb_cols_rows(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2,
                             int64_t ldd2, const int32_t *list, const int32_t *count, int64_t R, int H, float *partial, void *stream) {
  if (!list || !count) return DMP_ERR_BAD_ARG;
  return smallk_atb_cols_impl(X, ldx, K, D, ldd, ncols, D2, ldd2, nullptr, nullptr, list, count, R, H, partial, stream);
}

int dmp_smallk_atb(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, const float *gate, int64_t R, int H,
                   float *partial, void *stream) {
  DMP_ROW_CHECK(R == 0 || (D && ldd >= H));
  return dmp_smallk_atb_cols(X, ldx, K, D, ldd, 1, nullptr, 0, gate, R, H, partial, stream);
}

static int smallk_embed_launch(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate, int64_t R,
                               int H, int ncols, float *out, int64_t ldo, int live_only, void *stream) {
  DMP_ROW_CHECK(R >= 0 && K > 0 && ncols >= 1 && ncols <= 8);
  if ((H != 128 && H != 64) || K > kSmallK) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return DMP_OK;
  DMP_ROW_CHECK(X && W && out && ldx >= K && ldw >= (int64_t)ncols * H && ldo >= (int64_t)ncols * H);
  if (ldw % 2 || ldo % 2 || (reinterpret_cast<uintptr_t>(W) & 7u) || (reinterpret_cast<uintptr_t>(out) & 7u)) return DMP_ERR_UNSUPPORTED;
  SmallKFwdArgs p{X, ldx, W, ldw, gate, R, out, ldo, (live_only && gate) ? 1 : 0};
  hipStream_t st = (hipStream_t)stream;
  switch (K) {
    case 1: launch_smallk_fwd<1>(p, H, ncols, st); break;   case 2: launch_smallk_fwd<2>(p, H, ncols, st); break;
    case 3: launch_smallk_fwd<3>(p, H, ncols, st); break;   case 4: launch_smallk_fwd<4>(p, H, ncols, st); break;
    case 5: launch_smallk_fwd<5>(p, H, ncols, st); break;   case 6: launch_smallk_fwd<6>(p, H, ncols, st); break;
    case 7: launch_smallk_fwd<7>(p, H, ncols, st); break;   case 8: launch_smallk_fwd<8>(p, H, ncols, st); break;
    case 9: launch_smallk_fwd<9>(p, H, ncols, st); break;   case 10: launch_smallk_fwd<10>(p, H, ncols, st); break;
    case 11: launch_smallk_fwd<11>(p, H, ncols, st); break; case 12: launch_smallk_fwd<12>(p, H, ncols, st); break;
    case 13: launch_smallk_fwd<13>(p, H, ncols, st); break; case 14: launch_smallk_fwd<14>(p, H, ncols, st); break;
    case 15: launch_smallk_fwd<15>(p, H, ncols, st); break; default: launch_smallk_fwd<16>(p, H, ncols, st); break;
  }
  return check_launch();
}

int dmp_smallk_embed_cols(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate, int64_t R,
                          int H, int ncols, float *out, int64_t ldo, void *stream) {
  return smallk_embed_launch(X, ldx, K, W, ldw, gate, R, H, ncols, out, ldo, 0, stream);
}

int dmp_smallk_embed_gate(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate, int64_t R,
                          int H, float *out, int64_t ldo, void *stream) {
  return smallk_embed_launch(X, ldx, K, W, ldw, gate, R, H, 1, out, ldo, 0, stream);
}

int dmp_smallk_embed_live(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate, int64_t R,
                          int H, float *out, int64_t ldo, void *stream) {
  if (!gate) return DMP_ERR_BAD_ARG;
  return smallk_embed_launch(X, ldx, K, W, ldw, gate, R, H, 1, out, ldo, 1, stream);
}

int dmp_reduce_partials(const float *partial, int64_t S, int64_t L, float *out, int accumulate, void *stream) {
  DMP_ROW_CHECK(S >= 0 && L > 0 && out);
  if (L % 4 || !ok16(partial) || !ok16(out)) return DMP_ERR_UNSUPPORTED;
  DMP_ROW_CHECK(S == 0 || partial);
  const unsigned nb = (unsigned)((L / 4 + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
  if (accumulate) reduce_partials_kernel<true><<<nb, kBlock, 0, st>>>(partial, S, L, out);
  else reduce_partials_kernel<false><<<nb, kBlock, 0, st>>>(partial, S, L, out);
  return check_launch();
}

int dmp_reduce_partials_multi(const float *const *partials, const int64_t *S, const int64_t *L, float *const *outs,
                              int n, void *stream) {
  DMP_ROW_CHECK(n >= 0 && n <= kMaxSegments);
  if (n == 0) return DMP_OK;
  DMP_ROW_CHECK(partials && S && L && outs);
  ReduceSegs a;
  int64_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    DMP_ROW_CHECK(S[i] >= 0 && L[i] > 0 && outs[i] && (S[i] == 0 || partials[i]));
    if (L[i] % 4 || !ok16(partials[i]) || !ok16(outs[i])) return DMP_ERR_UNSUPPORTED;
    a.partial[i] = partials[i]; a.out[i] = outs[i]; a.S[i] = S[i]; a.L[i] = L[i];
    a.blk0[i] = (int)blocks;
    blocks += (L[i] / 4 + 31) / 32;
    if (blocks > 0x7fffffff) return DMP_ERR_UNSUPPORTED;
  }
  a.blk0[n] = (int)blocks;
  a.n = n;
  reduce_partials_multi_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(a);
  return check_launch();
}

int dmp_pack_segments(const float *const *src, const int64_t *dst_off, const int64_t *len, int n, int pad_to_4, float *dst,
                      void *stream) {
  DMP_ROW_CHECK(n >= 0);
  if (n == 0) return DMP_OK;
  DMP_ROW_CHECK(src && dst_off && len && dst);
  if (!ok16(dst)) return DMP_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    DMP_ROW_CHECK(len[i] >= 0 && dst_off[i] >= 0);          // (src[i] == NULL: the segment is cleared)
    if (dst_off[i] % 4) return DMP_ERR_UNSUPPORTED;
  }
  for (int base = 0; base < n; base += DMP_PACK_MAX_SEGMENTS) {
    const int cnt = n - base < DMP_PACK_MAX_SEGMENTS ? n - base : DMP_PACK_MAX_SEGMENTS;
    PackSegs a;
    a.pad = pad_to_4 ? 1 : 0;
    int64_t longest = 0;
    for (int i = 0; i < cnt; ++i) {
      a.src[i] = src[base + i]; a.off[i] = dst_off[base + i]; a.len[i] = len[base + i];
      if (a.len[i] > longest) longest = a.len[i];
    }
    if (longest == 0) continue;
    int64_t nb = (longest + 4 * kBlock - 1) / (4 * kBlock);
    if (nb > 64) nb = 64;
    pack_segments_kernel<<<dim3((unsigned)nb, (unsigned)cnt), kBlock, 0, (hipStream_t)stream>>>(a, dst);
    const int rc = check_launch();
    if (rc != DMP_OK) return rc;
  }
  return DMP_OK;
}

int dmp_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                   int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                   void *stream) {
  return dmp_adamw_step_skip(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step,
                             nullptr, nullptr, 0, stream);
}

static int adamw_launch(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                        int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                        double *state, const int64_t *skip_lo, const int64_t *skip_hi, int nskip, void *stream,
                        int32_t *veto = nullptr, int32_t veto_mask = 0) {
  DMP_ROW_CHECK(n >= 0 && (state || (step >= 1 && lr >= 0)) && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0);
  DMP_ROW_CHECK(nskip >= 0 && nskip <= DMP_ADAMW_MAX_SKIP && (nskip == 0 || (skip_lo && skip_hi)));
  if (n == 0) return DMP_OK;
  DMP_ROW_CHECK(param && grad && exp_avg && exp_avg_sq);
  if (!ok16(param) || !ok16(grad) || !ok16(exp_avg) || !ok16(exp_avg_sq) || !ok16(max_exp_avg_sq)) return DMP_ERR_UNSUPPORTED;
  AdamArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.vmax = max_exp_avg_sq; a.n = n;
  a.beta1_c = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.beta2_c = (float)(1.0 - beta2); a.eps = (float)eps;
  a.state = state; a.lr_wd = weight_decay; a.beta1 = beta1; a.beta2d = beta2; a.veto = veto;
  a.seg_off = nullptr; a.seg_tab = nullptr; a.P = 0;
  a.decay = 1.f; a.step_size = 0.f; a.inv_bc2_sqrt = 1.f;
  if (!state) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.step_size = (float)(lr / bc1); a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  }
  a.nskip = nskip;
  for (int s = 0; s < nskip; ++s) {
    if (skip_lo[s] % 4 || skip_hi[s] % 4 || skip_lo[s] > skip_hi[s]) return DMP_ERR_BAD_ARG;
    a.skip_lo[s] = skip_lo[s]; a.skip_hi[s] = skip_hi[s];
  }
  int64_t nb = (n / 4 + kBlock) / kBlock;
  if (nb > 2048) nb = 2048;
  if (state) adamw_tick<<<1, 1, 0, (hipStream_t)stream>>>(state, veto, veto_mask);   // step += 1, ordered before the update on the stream
  adamw_kernel<<<(unsigned)nb, kBlock, 0, (hipStream_t)stream>>>(a);
  return check_launch();
}

int dmp_adamw_step_skip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                        int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                        const int64_t *skip_lo, const int64_t *skip_hi, int nskip, void *stream) {
  return adamw_launch(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, nullptr,
                      skip_lo, skip_hi, nskip, stream);
}

int dmp_adamw_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                       int64_t n, double *state, double beta1, double beta2, double eps, double weight_decay,
                       const int64_t *skip_lo, const int64_t *skip_hi, int nskip, void *stream) {
  if (!state || (reinterpret_cast<uintptr_t>(state) & 7u)) return DMP_ERR_BAD_ARG;
  return adamw_launch(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, 0.0, beta1, beta2, eps, weight_decay, 0, state,
                      skip_lo, skip_hi, nskip, stream);
}

int dmp_adamw_step_segments(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                            int64_t n, double *state, const int64_t *seg_off, int P, const uint64_t *live, const float *live_dev,
                            float *seg_tab, double beta1, double beta2, double eps, double weight_decay, int32_t *veto,
                            int32_t veto_mask, void *stream) {
  DMP_ROW_CHECK(n >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0);
  if (!state || (reinterpret_cast<uintptr_t>(state) & 7u) || !seg_off || !seg_tab || P < 1 || P > DMP_ADAMW_MAX_SEGMENTS)
    return DMP_ERR_BAD_ARG;
  if (n == 0) return DMP_OK;
  DMP_ROW_CHECK(param && grad && exp_avg && exp_avg_sq);
  if (!ok16(param) || !ok16(grad) || !ok16(exp_avg) || !ok16(exp_avg_sq) || !ok16(max_exp_avg_sq)) return DMP_ERR_UNSUPPORTED;
  SegTick t;
  t.state = state; t.tab = seg_tab; t.P = P; t.beta1 = beta1; t.beta2 = beta2; t.veto = veto; t.mask = veto_mask; t.all = live ? 0 : 1; t.live_dev = live_dev;
  for (int w = 0; w < DMP_ADAMW_MAX_SEGMENTS / 64; ++w) t.live[w] = (live && w < (P + 63) / 64) ? live[w] : 0ull;
  AdamArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.vmax = max_exp_avg_sq; a.n = n;
  a.beta1_c = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.beta2_c = (float)(1.0 - beta2); a.eps = (float)eps;
  a.state = state; a.lr_wd = weight_decay; a.beta1 = beta1; a.beta2d = beta2; a.veto = veto;
  a.decay = 1.f; a.step_size = 0.f; a.inv_bc2_sqrt = 1.f; a.nskip = 0;
  a.seg_off = seg_off; a.seg_tab = seg_tab; a.P = P;
  int64_t nb = (n / 4 + kBlock) / kBlock;
  if (nb > 2048) nb = 2048;
  adamw_tick_segments<<<1, kBlock, 0, (hipStream_t)stream>>>(t);
  adamw_kernel<<<(unsigned)nb, kBlock, 0, (hipStream_t)stream>>>(a);
  return check_launch();
}

int dmp_adamw_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                           int64_t n, double *state, double beta1, double beta2, double eps, double weight_decay,
                           const int64_t *skip_lo, const int64_t *skip_hi, int nskip, int32_t *veto, int32_t veto_mask,
                           void *stream) {
  if (!state || (reinterpret_cast<uintptr_t>(state) & 7u) || !veto) return DMP_ERR_BAD_ARG;
  return adamw_launch(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, 0.0, beta1, beta2, eps, weight_decay, 0, state,
                      skip_lo, skip_hi, nskip, stream, veto, veto_mask);
}

}  // extern "C"
