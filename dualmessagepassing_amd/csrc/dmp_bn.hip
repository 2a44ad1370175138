// BatchNorm1d in TRAINING mode over the rows of an [R, C] matrix, with the activation that follows it in the UNC layers'
// MLPs fused in (UNC model.py:145-157: Linear -> BatchNorm1d -> LeakyReLU -> Linear), gfx950.
//
//   forward    mean_c, var_c (biased) over the rows;  y = act(gamma (x - mean) / sqrt(var + eps) + beta);
//              running_mean / running_var updated with `momentum` (running_var from the UNBIASED variance), as
//              torch.nn.BatchNorm1d does
//   backward   dyb = act'(y) dy;  dbeta = sum dyb;  dgamma = sum dyb xhat;
//              dx = gamma invstd (dyb - dbeta / R - xhat dgamma / R)
//
// Three small launches each way (the library's statistics kernels take 33-34 us at R ~ 3-11 k rows, C = 256, with one
// workgroup per few channels; here the rows are spread over the chip): per-workgroup partial column sums, one workgroup that
// combines them in a fixed order (bit-stable, no atomics) and finalises the statistics, one streaming pass that applies them.
// The sums are taken of x - x[0, c] (the first row as the shift): no cancellation in sum(d^2) - sum(d)^2 / R.
#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kBnMaxBlocks = 256;     // partial rows per launch
constexpr int kBnMaxC = 1024;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }

struct BnArgs {
  const float *x; int64_t ldx;         // [R, C]
  const float *y; int64_t ldy;         // backward: the saved output (the activation's mask)
  const float *dy; int64_t lddy;       // backward: upstream gradient
  int64_t R; int C;
  float *partial;                      // [blocks, 2C]
  float *stats;                        // [4C]: forward mean | invstd ; backward (after finalize) dbeta | dgamma in [2C, 4C)
  float slope;
  const int64_t *rdev;                 // device-side row count or NULL: only the first min(R, *rdev) rows exist (a batch padded to a capacity)
};

__device__ __forceinline__ int64_t live_rows(int64_t R, const int64_t *rdev) {
  if (!rdev) return R;
  const int64_t n = *rdev;
  return n < 1 ? 1 : (n < R ? n : R);
}

// lanes: C / 4 per row (float4 each), kBlock / (C / 4) rows per pass
template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_partial_k(const BnArgs p) {
  __shared__ float4 red[2][kBlock];
  const int G = p.C / 4, rows_per_pass = kBlock / G;
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const bool on = grp < rows_per_pass;
  const int c = lane * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  float4 shift = a, mean = a, invstd = a;
  if (on) {
    if (BWD) { mean = ld4(p.stats + c); invstd = ld4(p.stats + p.C + c); }
    else shift = ld4(p.x + c);
  }
  const int64_t R = live_rows(p.R, p.rdev);
  if (on) {
    const int64_t step = (int64_t)gridDim.x * rows_per_pass;
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_pass + grp; r0 < R; r0 += 4 * step) {
      float4 x[4], y[4], d[4];                               // four rows in flight: the loop is a memory round trip per pass
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * step;
        const bool ok = r < R;
        x[u] = ok ? ld4(p.x + r * p.ldx + c) : (BWD ? mean : shift);
        if (BWD) {
          y[u] = ok ? ld4(p.y + r * p.ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          d[u] = ok ? ld4(p.dy + r * p.lddy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (BWD) {
          float4 g = d[u];
          g.x = act_bwd(y[u].x, g.x, p.slope); g.y = act_bwd(y[u].y, g.y, p.slope); g.z = act_bwd(y[u].z, g.z, p.slope); g.w = act_bwd(y[u].w, g.w, p.slope);
          a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
          b.x += g.x * ((x[u].x - mean.x) * invstd.x); b.y += g.y * ((x[u].y - mean.y) * invstd.y);
          b.z += g.z * ((x[u].z - mean.z) * invstd.z); b.w += g.w * ((x[u].w - mean.w) * invstd.w);
        } else {
          const float4 e = make_float4(x[u].x - shift.x, x[u].y - shift.y, x[u].z - shift.z, x[u].w - shift.w);
          a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
          b.x += e.x * e.x; b.y += e.y * e.y; b.z += e.z * e.z; b.w += e.w * e.w;
        }
      }
    }
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  if (grp == 0) {                                            // fixed-order combine of the row groups
    for (int g = 1; g < rows_per_pass; ++g) {
      const float4 u = red[0][g * G + lane], v = red[1][g * G + lane];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    st4(p.partial + (int64_t)blockIdx.x * 2 * p.C + c, a);
    st4(p.partial + (int64_t)blockIdx.x * 2 * p.C + p.C + c, b);
  }
}

struct BnFinArgs {
  const float *partial; int blocks; int64_t R; int C;
  const float *x0;                     // forward: row 0 of x (the shift)
  float eps, momentum;
  float *running_mean, *running_var;   // may be NULL
  float *stats;                        // forward: writes mean | invstd; backward: writes dbeta | dgamma at [2C, 4C)
  const int64_t *rdev;
};

// a workgroup per 32 columns: 8 lanes own 4 columns each, the 32 row groups of the block split the partial rows among them (a
// fixed split, combined in a fixed order through LDS: bit-stable), sums in double
constexpr int kFinCols = 32;
template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_finalize_k(const BnFinArgs p) {
  __shared__ double red[2][kBlock][4];
  constexpr int G = kFinCols / 4, groups = kBlock / G;
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const int c = blockIdx.x * kFinCols + lane * 4;
  const bool on = c < p.C;                                   // C % 4 == 0: a lane's 4 columns are in or out together
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const int per = (p.blocks + groups - 1) / groups;
  const int s0 = grp * per, s1 = !on ? s0 : ((s0 + per < p.blocks) ? s0 + per : p.blocks);
#pragma unroll 4
  for (int s = s0; s < s1; ++s) {
    const float4 u = ld4(p.partial + (int64_t)s * 2 * p.C + c), v = ld4(p.partial + (int64_t)s * 2 * p.C + p.C + c);
    a[0] += (double)u.x; a[1] += (double)u.y; a[2] += (double)u.z; a[3] += (double)u.w;
    b[0] += (double)v.x; b[1] += (double)v.y; b[2] += (double)v.z; b[3] += (double)v.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = a[j]; red[1][threadIdx.x][j] = b[j]; }
  __syncthreads();
  if (grp != 0 || !on) return;
  for (int g = 1; g < groups; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += red[0][g * G + lane][j]; b[j] += red[1][g * G + lane][j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = c + j;
    if (BWD) {
      p.stats[2 * p.C + cc] = (float)a[j];                   // dbeta
      p.stats[3 * p.C + cc] = (float)b[j];                   // dgamma
    } else {
      const double n = (double)live_rows(p.R, p.rdev), md = a[j] / n;
      double var = b[j] / n - md * md;                       // biased, of the shifted values
      if (var < 0.0) var = 0.0;
      const float mean = (float)(md + (double)p.x0[cc]);
      p.stats[cc] = mean;
      p.stats[p.C + cc] = (float)(1.0 / sqrt(var + (double)p.eps));
      if (p.running_mean) p.running_mean[cc] = (1.f - p.momentum) * p.running_mean[cc] + p.momentum * mean;
      if (p.running_var) {
        const float unbiased = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
        p.running_var[cc] = (1.f - p.momentum) * p.running_var[cc] + p.momentum * unbiased;
      }
    }
  }
}

struct BnApplyArgs {
  const float *x; int64_t ldx; const float *y; int64_t ldy; const float *dy; int64_t lddy;
  const float *gamma, *beta, *stats; int64_t R; int C; float slope; int act;
  float *out; int64_t ldo;
  const int64_t *rdev;                 // rows past the device-side count are written as zeros
};

template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_apply_k(const BnApplyArgs p) {
  const int G = p.C / 4, rows_per_pass = kBlock / G;
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  if (grp >= rows_per_pass) return;
  const int c = lane * 4;
  const float4 mean = ld4(p.stats + c), invstd = ld4(p.stats + p.C + c);
  const float4 gm = p.gamma ? ld4(p.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 bt = (!BWD && p.beta) ? ld4(p.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 db = make_float4(0.f, 0.f, 0.f, 0.f), dg = db;
  const int64_t R = live_rows(p.R, p.rdev);
  const float inv_n = 1.f / (float)R;
  if (BWD) { db = ld4(p.stats + 2 * p.C + c); dg = ld4(p.stats + 3 * p.C + c); }
  for (int64_t r = (int64_t)blockIdx.x * rows_per_pass + grp; r < p.R; r += (int64_t)gridDim.x * rows_per_pass) {
    if (r >= R) { st4(p.out + r * p.ldo + c, make_float4(0.f, 0.f, 0.f, 0.f)); continue; }      // a padding row
    const float4 x = ld4(p.x + r * p.ldx + c);
    const float4 xh = make_float4((x.x - mean.x) * invstd.x, (x.y - mean.y) * invstd.y, (x.z - mean.z) * invstd.z, (x.w - mean.w) * invstd.w);
    float4 o;
    if (BWD) {
      const float4 y = ld4(p.y + r * p.ldy + c);
      float4 d = ld4(p.dy + r * p.lddy + c);
      if (p.act) { d.x = act_bwd(y.x, d.x, p.slope); d.y = act_bwd(y.y, d.y, p.slope); d.z = act_bwd(y.z, d.z, p.slope); d.w = act_bwd(y.w, d.w, p.slope); }
      o.x = gm.x * invstd.x * (d.x - db.x * inv_n - xh.x * (dg.x * inv_n));
      o.y = gm.y * invstd.y * (d.y - db.y * inv_n - xh.y * (dg.y * inv_n));
      o.z = gm.z * invstd.z * (d.z - db.z * inv_n - xh.z * (dg.z * inv_n));
      o.w = gm.w * invstd.w * (d.w - db.w * inv_n - xh.w * (dg.w * inv_n));
    } else {
      o = make_float4(xh.x * gm.x + bt.x, xh.y * gm.y + bt.y, xh.z * gm.z + bt.z, xh.w * gm.w + bt.w);
      if (p.act) { o.x = act_fwd(o.x, p.slope); o.y = act_fwd(o.y, p.slope); o.z = act_fwd(o.z, p.slope); o.w = act_fwd(o.w, p.slope); }
    }
    st4(p.out + r * p.ldo + c, o);
  }
}

inline bool al16(const void *q) { return !q || (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

inline unsigned bn_blocks(int64_t R, int C) {
  // one partial row per ~64 KB of input: few enough for the one-workgroup combine to stay a few microseconds
  const int64_t rows_per_block = 65536 / ((int64_t)C * 4) > 16 ? 65536 / ((int64_t)C * 4) : 16;
  const int64_t nb = (R + rows_per_block - 1) / rows_per_block;
  return (unsigned)(nb < 1 ? 1 : (nb > kBnMaxBlocks ? kBnMaxBlocks : nb));
}

inline bool bn_shape_ok(int C) { return C >= 4 && C % 4 == 0 && C <= kBnMaxC && kBlock % (C / 4) == 0; }

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int64_t dmp_bn_partial_rows(int64_t rows, int C) { return bn_shape_ok(C) ? (int64_t)bn_blocks(rows, C) : 0; }

int dmp_bn_train_fwd_rows(const float *x, int64_t ldx, int64_t rows, const int64_t *rows_dev, int C, const float *gamma, const float *beta,
                          float eps, float momentum, float *running_mean, float *running_var, int act, float slope, float *partial,
                          float *stats, float *out, int64_t ldo, void *stream) {
  if (rows <= 0 || !x || !partial || !stats || !out || ldx < C || ldo < C) return DMP_ERR_BAD_ARG;
  if (!bn_shape_ok(C) || !slope_ok(slope) || ldx % 4 || ldo % 4 || !al16(x) || !al16(out) || !al16(gamma) || !al16(beta) || !al16(partial)
      || !al16(stats))
    return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = bn_blocks(rows, C);
  BnArgs a{x, ldx, nullptr, 0, nullptr, 0, rows, C, partial, stats, slope, rows_dev};
  bn_partial_k<false><<<nb, kBlock, 0, st>>>(a);
  BnFinArgs f{partial, (int)nb, rows, C, x, eps, momentum, running_mean, running_var, stats, rows_dev};
  bn_finalize_k<false><<<(unsigned)((C + kFinCols - 1) / kFinCols), kBlock, 0, st>>>(f);
  BnApplyArgs ap{x, ldx, nullptr, 0, nullptr, 0, gamma, beta, stats, rows, C, slope, act, out, ldo, rows_dev};
  const int rows_per_pass = kBlock / (C / 4);
  const int64_t want = (rows + rows_per_pass - 1) / rows_per_pass;
  bn_apply_k<false><<<(unsigned)(want < 2048 ? want : 2048), kBlock, 0, st>>>(ap);
  return check_launch();
}

int dmp_bn_train_fwd(const float *x, int64_t ldx, int64_t rows, int C, const float *gamma, const float *beta, float eps,
                     float momentum, float *running_mean, float *running_var, int act, float slope, float *partial,
                     float *stats, float *out, int64_t ldo, void *stream) {
  return dmp_bn_train_fwd_rows(x, ldx, rows, nullptr, C, gamma, beta, eps, momentum, running_mean, running_var, act, slope, partial, stats, out,
                               ldo, stream);
}

int dmp_bn_train_bwd_rows(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *dy, int64_t lddy, int64_t rows,
                          const int64_t *rows_dev, int C, const float *gamma, int act, float slope, float *partial, float *stats, float *dx,
                          int64_t ldo, void *stream) {
  if (rows <= 0 || !x || !dy || !partial || !stats || !dx || ldx < C || lddy < C || ldo < C || (act && (!y || ldy < C))) return DMP_ERR_BAD_ARG;
  if (!bn_shape_ok(C) || !slope_ok(slope) || ldx % 4 || ldy % 4 || lddy % 4 || ldo % 4 || !al16(x) || !al16(y) || !al16(dy) || !al16(dx)
      || !al16(gamma) || !al16(partial) || !al16(stats))
    return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = bn_blocks(rows, C);
  // without an activation the mask input is the gradient itself read as "positive everywhere": slope 1 makes act_bwd the identity
  const float sl = act ? slope : 1.f;
  const float *ym = act ? y : dy;
  const int64_t ldm = act ? ldy : lddy;
  BnArgs a{x, ldx, ym, ldm, dy, lddy, rows, C, partial, stats, sl, rows_dev};
  bn_partial_k<true><<<nb, kBlock, 0, st>>>(a);
  BnFinArgs f{partial, (int)nb, rows, C, nullptr, 0.f, 0.f, nullptr, nullptr, stats, rows_dev};
  bn_finalize_k<true><<<(unsigned)((C + kFinCols - 1) / kFinCols), kBlock, 0, st>>>(f);
  BnApplyArgs ap{x, ldx, ym, ldm, dy, lddy, gamma, nullptr, stats, rows, C, sl, 1, dx, ldo, rows_dev};
  const int rows_per_pass = kBlock / (C / 4);
  const int64_t want = (rows + rows_per_pass - 1) / rows_per_pass;
  bn_apply_k<true><<<(unsigned)(want < 2048 ? want : 2048), kBlock, 0, st>>>(ap);
  return check_launch();
}

int dmp_bn_train_bwd(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *dy, int64_t lddy, int64_t rows, int C,
                     const float *gamma, int act, float slope, float *partial, float *stats, float *dx, int64_t ldo,
                     void *stream) {
  return dmp_bn_train_bwd_rows(x, ldx, y, ldy, dy, lddy, rows, nullptr, C, gamma, act, slope, partial, stats, dx, ldo, stream);
}

}  // extern "C"
