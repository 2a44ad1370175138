// Parameter algebra of the fused DMPNN layer for ALL layers of a rep-net in one launch each.
//
// The layer folds the first Linear of its two MLPs into the projections (dmpnn.py:111-156: the projections
// are linear and feed a Linear):  C = M W0^T  with  M = [W_loop; W_in; W_out; bias]  (node side) or
// [W_eloop; W_src - W_dst; W_dst; W_src; ebias]  (edge side).  Per layer that is a dozen [<=513,128]x[128,128]
// products, concatenations and bias additions -- microseconds of arithmetic, but two dozen launches per layer
// and step when done with library calls.  Here:
//   dmp_fold_layers    every block row of every layer's C, written straight into the layouts the layer's
//                      kernels read (Bn [2H,H], bn [H], Wx [H,3H], Wes [H,2H], be [H])
//   dmp_unfold_layers  the backward: dM = dC W0 (row jobs, with the +-d(src-dst) terms folded in) and
//                      dW0 = dC^T M (column jobs over the gathered block rows)
// H = 128 or 64 (one H-wide output panel per job).  Nothing here is bandwidth-relevant; the point is the launch
// count and the length of each workgroup's chain of dependent L2 round trips: the products run on the fp32 MFMA pipe
// with every lane requesting its operands in the MFMA layout in one batch (no staged LDS slices).
#include <initializer_list>

#include "dmp_common.h"

namespace dmp {
namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerWG = 16;

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// out[i, :] = (A[i, :] + s2 * A2[i, :]) op(W) (+ addv), rows x 128, contraction over 128:
//   wT = 1: out[i, j] = sum_k a[i, k] W[j, k]   (C = M W0^T: W0 in nn.Linear layout [out, in])
//   wT = 0: out[i, k] = sum_j a[i, j] W[j, k]   (dM = dC W0)
struct RowJob {
  const float *A; const float *A2; const float *W; const float *Wsub; const float *addv; float *out;
  int lda, lda2, ldo, rows, wT;
  float s2;
};
constexpr int kMaxRowJobs = 13 * DMP_FOLD_MAX_LAYERS;   // fold: 9 + 4 optional transposes per layer; unfold: 8
struct RowJobs { RowJob job[kMaxRowJobs]; int blk0[kMaxRowJobs + 1]; int n; };

template <int kH>
__global__ __launch_bounds__(kThreads) void rowjob_k(const RowJobs t) {
  constexpr int kPad = kH + 4, kQ = kH / 4, kPass = kThreads / kQ;   // LDS row stride, float4 per row, rows per staging pass
  constexpr int NT = kH / 64, NKB = kH / 16, CW = kH / 16;           // 16-column tiles per wave, k blocks, output columns per thread
  __shared__ float As[kRowsPerWG * kPad];
  __shared__ float Ws[kRowsPerWG * kPad];
  int q = 0;
  while (q + 1 < t.n && (int)blockIdx.x >= t.blk0[q + 1]) ++q;
  const RowJob &jb = t.job[q];
  const int i0 = ((int)blockIdx.x - t.blk0[q]) * kRowsPerWG;
  const int tid = threadIdx.x, r = tid >> 4, cg = tid & 15;
  // the 16 x H operand rows (zero past the job's rows)
#pragma unroll
  for (int m = 0; m < kRowsPerWG / kPass; ++m) {
    const int row = tid / kQ + kPass * m, c4 = (tid % kQ) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + row < jb.rows) {
      a = ldg4(jb.A + (int64_t)(i0 + row) * jb.lda + c4);
      if (jb.A2) {
        const float4 b = ldg4(jb.A2 + (int64_t)(i0 + row) * jb.lda2 + c4);
        a.x += jb.s2 * b.x; a.y += jb.s2 * b.y; a.z += jb.s2 * b.z; a.w += jb.s2 * b.w;
      }
    }
    *reinterpret_cast<float4 *>(&As[row * kPad + c4]) = a;
  }
  // 16 x H x H on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32; wave w owns output columns (H/4) w .. (H/4) w + H/4 - 1): every lane
  // requests its share of the whole weight panel in one batch of loads, straight in the operand layout (one L2 round trip
  // instead of one per 32-deep LDS slice); lane (m, kq) contracts k = 16 kb + 4 kq .. + 3 of block kb
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = tid & 63, wave = tid >> 6, mm = lane & 15, kq = lane >> 4;
  float4 bw[NT][NKB];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const int n = wave * (16 * NT) + tt * 16 + mm;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int k = kb * 16 + 4 * kq;
      if (jb.wT) {
        float4 w = ldg4(jb.W + n * kH + k);
        if (jb.Wsub) {
          const float4 u = ldg4(jb.Wsub + n * kH + k);
          w.x -= u.x; w.y -= u.y; w.z -= u.z; w.w -= u.w;
        }
        bw[tt][kb] = w;
      } else {
        const float *c = jb.W + k * kH + n;
        bw[tt][kb] = make_float4(c[0], c[kH], c[2 * kH], c[3 * kH]);
      }
    }
  }
  __syncthreads();
  f32x4 d[NT];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) d[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const float4 a = *reinterpret_cast<const float4 *>(&As[mm * kPad + kb * 16 + 4 * kq]);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      d[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw[tt][kb].x, d[tt], 0, 0, 0);
      d[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw[tt][kb].y, d[tt], 0, 0, 0);
      d[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw[tt][kb].z, d[tt], 0, 0, 0);
      d[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw[tt][kb].w, d[tt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int i = 0; i < 4; ++i) Ws[(4 * kq + i) * kPad + wave * (16 * NT) + tt * 16 + mm] = d[tt][i];
  __syncthreads();
  if (i0 + r < jb.rows) {
#pragma unroll
    for (int v = 0; v < CW / 4; ++v) {                      // thread (r, cg): CW consecutive columns of row r
      float4 o4 = *reinterpret_cast<const float4 *>(&Ws[r * kPad + cg * CW + 4 * v]);
      if (jb.addv) {
        const float4 a4 = ldg4(jb.addv + cg * CW + 4 * v);
        o4.x += a4.x; o4.y += a4.y; o4.z += a4.z; o4.w += a4.w;
      }
      *reinterpret_cast<float4 *>(jb.out + (int64_t)(i0 + r) * jb.ldo + cg * CW + 4 * v) = o4;
    }
  }
}

// out[j, k] = sum over the sources s and their rows i of dC_s[i, j] * (M_s[i, k] - M2_s[i, k]):  dW0 = dC^T M
struct ColSrc { const float *dC; const float *M; const float *M2; int ldc, rows; };
struct ColJob { ColSrc src[5]; float *out; int nsrc; };
constexpr int kMaxColJobs = 6;
struct ColJobs { ColJob job[kMaxColJobs]; int n; };

constexpr int kColW = 16;                                   // one 16 x 16 output tile per workgroup: 64 workgroups per job
constexpr int kColSteps = 8;                                // MFMA steps (x 4 block rows x 4 waves = 128 rows) per batch of loads
template <int kH>
__global__ __launch_bounds__(kThreads) void coljob_k(const ColJobs t) {
  // fp32 MFMA (v_mfma_f32_16x16x4_f32 contracts four block rows per instruction): the four waves take interleaved groups
  // of four rows, every lane requests the operands of a 128-row batch in one go, a batch (or source) ahead of the one
  // being multiplied; the four partial tiles meet in LDS
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[4 * 16 * 17];
  constexpr int kPerJob = (kH / kRowsPerWG) * (kH / kColW);
  const ColJob &jb = t.job[blockIdx.x / kPerJob];
  const int sub = (int)blockIdx.x % kPerJob;
  const int j0 = (sub / (kH / kColW)) * kRowsPerWG, k0 = (sub % (kH / kColW)) * kColW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kq = lane >> 4;
  float an[kColSteps], bn[kColSteps], un[kColSteps];
  auto fetch = [&](int s, int g0) {
    const ColSrc &sr = jb.src[s];
#pragma unroll
    for (int st = 0; st < kColSteps; ++st) {
      const int i = 4 * (g0 + wave + 4 * st) + kq;
      const bool in = i < sr.rows;
      an[st] = in ? sr.dC[(int64_t)i * sr.ldc + j0 + m] : 0.f;
      bn[st] = in ? sr.M[i * kH + k0 + m] : 0.f;
      un[st] = (in && sr.M2) ? sr.M2[i * kH + k0 + m] : 0.f;
    }
  };
  auto advance = [&](int &s, int &g0) {
    g0 += 4 * kColSteps;
    while (s < jb.nsrc && 4 * g0 >= jb.src[s].rows) { ++s; g0 = 0; }
  };
  int s = 0, g0 = 0;
  while (s < jb.nsrc && jb.src[s].rows <= 0) ++s;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  if (s < jb.nsrc) fetch(s, g0);
  while (s < jb.nsrc) {
    float a[kColSteps], b[kColSteps];
#pragma unroll
    for (int st = 0; st < kColSteps; ++st) { a[st] = an[st]; b[st] = bn[st] - un[st]; }
    advance(s, g0);
    if (s < jb.nsrc) fetch(s, g0);
#pragma unroll
    for (int st = 0; st < kColSteps; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], b[st], d, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * kq + i) * 17 + m] = d[i];
  __syncthreads();
  const int jr = tid >> 4, kc = tid & 15;
  jb.out[(j0 + jr) * kH + k0 + kc] =
      (red[(0 * 16 + jr) * 17 + kc] + red[(1 * 16 + jr) * 17 + kc]) + (red[(2 * 16 + jr) * 17 + kc] + red[(3 * 16 + jr) * 17 + kc]);
}

inline void add_row_job(RowJobs &t, int &blocks, const float *A, int lda, const float *A2, int lda2, float s2, const float *W,
                        int wT, float *out, int ldo, const float *addv, int rows, const float *Wsub = nullptr) {
  RowJob &j = t.job[t.n];
  j.Wsub = Wsub;
  j.A = A; j.lda = lda; j.A2 = A2; j.lda2 = lda2; j.s2 = s2; j.W = W; j.wT = wT; j.out = out; j.ldo = ldo; j.addv = addv;
  j.rows = rows;
  t.blk0[t.n] = blocks;
  blocks += (rows + kRowsPerWG - 1) / kRowsPerWG;
  t.n += 1;
  t.blk0[t.n] = blocks;
}

inline bool all16(std::initializer_list<const void *> ps) {
  for (const void *q : ps)
    if (!q || (reinterpret_cast<uintptr_t>(q) & 15u)) return false;
  return true;
}

// ---- several small products in one launch (dmp_small_gemm_jobs): a workgroup = a 16 x 64 tile of one job's output.  These
// products are latency-bound (a handful of workgroups, contractions of 16 .. 512): the contraction runs in chunks of 128
// split over the workgroup's four waves (wave w takes k = k0 + 32 w .. + 31 of each chunk), so a K = 512 product is four
// round trips to memory with 32 independent B loads per lane in flight in each.  The A chunk (16 x 128) goes through LDS
// and is read back as wave-uniform broadcasts.  The four partial tiles are summed in wave order: deterministic.
struct GemmJobs { dmp_gemm_job job[DMP_GEMM_MAX_JOBS]; };
__global__ __launch_bounds__(256) void small_gemm_jobs_k(const GemmJobs t) {
  constexpr int KC = 128, KW = 32;
  __shared__ float4 As[KC][4];                                      // [k][row / 4]
  __shared__ float Red[3][16][64];
  const dmp_gemm_job &j = t.job[blockIdx.y];
  const int tiles_n = (j.N + 63) / 64, tiles_m = (j.M + 15) / 16;
  if ((int)blockIdx.x >= tiles_n * tiles_m) return;                 // whole workgroups only: no barrier is skipped by a part
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (int)(blockIdx.x % tiles_n) * 64 + lane, row0 = (int)(blockIdx.x / tiles_n) * 16;
  const int cc = c < j.N ? c : 0;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float *As_f = reinterpret_cast<float *>(&As[0][0]);
  for (int q = 0; q < j.num_terms; ++q) {
    const dmp_gemm_term &m = j.term[q];
    const float *__restrict__ A = m.A;
    const float *__restrict__ B = m.B;
    const int64_t sa_r = m.transA ? 1 : m.lda, sa_k = m.transA ? m.lda : 1;
    const int64_t sb_k = m.transB ? 1 : m.ldb, sb_c = m.transB ? m.ldb : 1;
    for (int k0 = 0; k0 < m.K; k0 += KC) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {                                 // 2,048 elements of A, read along its unit stride
        const int e = (int)threadIdx.x + 256 * u;
        const int lr = m.transA ? (e & 15) : (e >> 7), lk = m.transA ? (e >> 4) : (e & 127);
        const int r = row0 + lr, k = k0 + lk;
        As_f[lk * 16 + lr] = (r < j.M && k < m.K) ? A[r * sa_r + k * sa_k] : 0.f;
      }
      float bv[KW];
      // op(B) = B^T: a lane's 32 contraction indices are 32 CONSECUTIVE floats of its row of B -- eight 16-byte loads of one whole
      // 128-byte line instead of 32 loads that each touch 64 different lines (the last launch of the first layer's backward: 38 us)
      const bool bvec = m.transB && (m.ldb % 4 == 0) && (m.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);
      if (bvec) {
        const float4 *src = reinterpret_cast<const float4 *>(B + (int64_t)cc * m.ldb + k0 + KW * w);
#pragma unroll
        for (int q4 = 0; q4 < KW / 4; ++q4) {
          const float4 t4 = (k0 + KW * w + 4 * q4 < m.K) ? src[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
          bv[4 * q4 + 0] = t4.x; bv[4 * q4 + 1] = t4.y; bv[4 * q4 + 2] = t4.z; bv[4 * q4 + 3] = t4.w;
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
          const int k = k0 + KW * w + kk;
          bv[kk] = k < m.K ? B[k * sb_k + cc * sb_c] : 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const float4 a = As[KW * w + kk][i4];
          acc[4 * i4 + 0] = fmaf(a.x, bv[kk], acc[4 * i4 + 0]);
          acc[4 * i4 + 1] = fmaf(a.y, bv[kk], acc[4 * i4 + 1]);
          acc[4 * i4 + 2] = fmaf(a.z, bv[kk], acc[4 * i4 + 2]);
          acc[4 * i4 + 3] = fmaf(a.w, bv[kk], acc[4 * i4 + 3]);
        }
      }
    }
  }
  if (w > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Red[w - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (w > 0 || c >= j.N) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = row0 + i;
    const float v = ((acc[i] + Red[0][i][lane]) + Red[1][i][lane]) + Red[2][i][lane];
    if (r < j.M) j.C[(int64_t)r * j.ldc + c] = j.C0 ? v + j.C0[(int64_t)r * j.ldc0 + c] : v;
  }
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_fold_layers(const dmp_layer_weights *w, const dmp_layer_folded *f, int num_layers, int H, void *stream) {
  if (num_layers < 0 || (num_layers > 0 && (!w || !f))) return DMP_ERR_BAD_ARG;
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  const int kH = H;
  for (int l0 = 0; l0 < num_layers; l0 += DMP_FOLD_MAX_LAYERS) {
    RowJobs t;
    t.n = 0;
    int blocks = 0;
    for (int l = l0; l < num_layers && l < l0 + DMP_FOLD_MAX_LAYERS; ++l) {
      const dmp_layer_weights &a = w[l];
      const dmp_layer_folded &o = f[l];
      if (!all16({a.nloop_w, a.in_w, a.out_w, a.nbias, a.eloop_w, a.src_w, a.dst_w, a.ebias, a.nW0, a.nb0, a.eW0, a.eb0, o.Bn,
                  o.bn, o.Wx, o.Wes, o.be}))
        return DMP_ERR_BAD_ARG;
      if (t.n + 13 > kMaxRowJobs) return DMP_ERR_BAD_ARG;      // 9 + 4 jobs per layer below
      for (const void *q : {(const void *)a.nW2, (const void *)a.eW2, (const void *)a.eye, (const void *)o.WesT, (const void *)o.nW2t,
                            (const void *)o.eW2t})
        if (q && (reinterpret_cast<uintptr_t>(q) & 15u)) return DMP_ERR_BAD_ARG;
      // node side: Cn = [W_loop; W_in; W_out; nbias] W0n^T -> Wx[:, 0:H], Bn[0:H], Bn[H:2H], bn (+ nb0)
      add_row_job(t, blocks, a.nloop_w, kH, nullptr, 0, 0.f, a.nW0, 1, o.Wx, 3 * kH, nullptr, kH);
      add_row_job(t, blocks, a.in_w, kH, nullptr, 0, 0.f, a.nW0, 1, o.Bn, kH, nullptr, kH);
      add_row_job(t, blocks, a.out_w, kH, nullptr, 0, 0.f, a.nW0, 1, o.Bn + kH * kH, kH, nullptr, kH);
      add_row_job(t, blocks, a.nbias, kH, nullptr, 0, 0.f, a.nW0, 1, o.bn, kH, a.nb0, 1);
      // edge side: Ce = [W_eloop; W_src - W_dst; W_dst; W_src; ebias] W0e^T -> Wes[:, 0:H], Wes[:, H:2H], Wx[:, H:2H], Wx[:, 2H:3H], be (+ eb0)
      add_row_job(t, blocks, a.eloop_w, kH, nullptr, 0, 0.f, a.eW0, 1, o.Wes, 2 * kH, nullptr, kH);
      add_row_job(t, blocks, a.src_w, kH, a.dst_w, kH, -1.f, a.eW0, 1, o.Wes + kH, 2 * kH, nullptr, kH);
      add_row_job(t, blocks, a.dst_w, kH, nullptr, 0, 0.f, a.eW0, 1, o.Wx + kH, 3 * kH, nullptr, kH);
      add_row_job(t, blocks, a.src_w, kH, nullptr, 0, 0.f, a.eW0, 1, o.Wx + 2 * kH, 3 * kH, nullptr, kH);
      add_row_job(t, blocks, a.ebias, kH, nullptr, 0, 0.f, a.eW0, 1, o.be, kH, a.eb0, 1);
      // the transposed copies the backward / second-Linear kernels read their (coalesced) weight panels from:
      // WesT = [A'^T | B'^T] with A'^T = W0e W_eloop^T, B'^T = W0e (W_src - W_dst)^T;  W2^T = I W2^T
      if (o.WesT) {
        add_row_job(t, blocks, a.eW0, kH, nullptr, 0, 0.f, a.eloop_w, 1, o.WesT, 2 * kH, nullptr, kH);
        add_row_job(t, blocks, a.eW0, kH, nullptr, 0, 0.f, a.src_w, 1, o.WesT + kH, 2 * kH, nullptr, kH, a.dst_w);
      }
      if (o.nW2t && a.nW2 && a.eye) add_row_job(t, blocks, a.eye, kH, nullptr, 0, 0.f, a.nW2, 1, o.nW2t, kH, nullptr, kH);
      if (o.eW2t && a.eW2 && a.eye) add_row_job(t, blocks, a.eye, kH, nullptr, 0, 0.f, a.eW2, 1, o.eW2t, kH, nullptr, kH);
    }
    if (H == 128) rowjob_k<128><<<blocks, kThreads, 0, (hipStream_t)stream>>>(t);
    else rowjob_k<64><<<blocks, kThreads, 0, (hipStream_t)stream>>>(t);
    const int rc = check_launch();
    if (rc != DMP_OK) return rc;
  }
  return DMP_OK;
}

int dmp_unfold_layers(const dmp_layer_weights *w, const dmp_layer_folded_grads *g, const dmp_layer_weight_grads *d,
                      int num_layers, int H, void *stream) {
  if (num_layers < 0 || (num_layers > 0 && (!w || !g || !d))) return DMP_ERR_BAD_ARG;
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  const int kH = H;
  for (int l0 = 0; l0 < num_layers; l0 += DMP_FOLD_MAX_LAYERS) {
    RowJobs t;
    ColJobs c;
    t.n = 0;
    c.n = 0;
    int blocks = 0;
    for (int l = l0; l < num_layers && l < l0 + DMP_FOLD_MAX_LAYERS; ++l) {
      const dmp_layer_weights &a = w[l];
      const dmp_layer_folded_grads &u = g[l];
      const dmp_layer_weight_grads &o = d[l];
      if (!all16({a.nloop_w, a.in_w, a.out_w, a.nbias, a.eloop_w, a.src_w, a.dst_w, a.ebias, a.nW0, a.eW0, u.dBn, u.dbn, u.dWx,
                  u.dWes, u.dbe, o.nloop_w, o.in_w, o.out_w, o.nbias, o.eloop_w, o.src_w, o.dst_w, o.ebias, o.nW0, o.eW0}))
        return DMP_ERR_BAD_ARG;
      if (t.n + 8 > kMaxRowJobs || c.n + 2 > kMaxColJobs) return DMP_ERR_BAD_ARG;
      // dM = dC W0; the rows of d(W_src - W_dst) = dWes[:, H:2H] W0e enter d_dst with -, d_src with +
      add_row_job(t, blocks, u.dWx, 3 * kH, nullptr, 0, 0.f, a.nW0, 0, o.nloop_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dBn, kH, nullptr, 0, 0.f, a.nW0, 0, o.in_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dBn + kH * kH, kH, nullptr, 0, 0.f, a.nW0, 0, o.out_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dbn, kH, nullptr, 0, 0.f, a.nW0, 0, o.nbias, kH, nullptr, 1);
      add_row_job(t, blocks, u.dWes, 2 * kH, nullptr, 0, 0.f, a.eW0, 0, o.eloop_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dWx + kH, 3 * kH, u.dWes + kH, 2 * kH, -1.f, a.eW0, 0, o.dst_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dWx + 2 * kH, 3 * kH, u.dWes + kH, 2 * kH, 1.f, a.eW0, 0, o.src_w, kH, nullptr, kH);
      add_row_job(t, blocks, u.dbe, kH, nullptr, 0, 0.f, a.eW0, 0, o.ebias, kH, nullptr, 1);
      // dW0 = dC^T M over the block rows
      ColJob &n = c.job[c.n++];
      n.out = o.nW0; n.nsrc = 4;
      n.src[0] = ColSrc{u.dWx, a.nloop_w, nullptr, 3 * kH, kH};
      n.src[1] = ColSrc{u.dBn, a.in_w, nullptr, kH, kH};
      n.src[2] = ColSrc{u.dBn + kH * kH, a.out_w, nullptr, kH, kH};
      n.src[3] = ColSrc{u.dbn, a.nbias, nullptr, kH, 1};
      ColJob &e = c.job[c.n++];
      e.out = o.eW0; e.nsrc = 5;
      e.src[0] = ColSrc{u.dWes, a.eloop_w, nullptr, 2 * kH, kH};
      e.src[1] = ColSrc{u.dWes + kH, a.src_w, a.dst_w, 2 * kH, kH};
      e.src[2] = ColSrc{u.dWx + kH, a.dst_w, nullptr, 3 * kH, kH};
      e.src[3] = ColSrc{u.dWx + 2 * kH, a.src_w, nullptr, 3 * kH, kH};
      e.src[4] = ColSrc{u.dbe, a.ebias, nullptr, kH, 1};
    }
    if (H == 128) rowjob_k<128><<<blocks, kThreads, 0, (hipStream_t)stream>>>(t);
    else rowjob_k<64><<<blocks, kThreads, 0, (hipStream_t)stream>>>(t);
    int rc = check_launch();
    if (rc != DMP_OK) return rc;
    if (H == 128) coljob_k<128><<<c.n * (kH / kRowsPerWG) * (kH / kColW), kThreads, 0, (hipStream_t)stream>>>(c);
    else coljob_k<64><<<c.n * (kH / kRowsPerWG) * (kH / kColW), kThreads, 0, (hipStream_t)stream>>>(c);
    rc = check_launch();
    if (rc != DMP_OK) return rc;
  }
  return DMP_OK;
}

int dmp_small_gemm_jobs(const dmp_gemm_job *jobs, int num_jobs, void *stream) {
  if (num_jobs < 0 || num_jobs > DMP_GEMM_MAX_JOBS) return DMP_ERR_BAD_ARG;
  if (num_jobs == 0) return DMP_OK;
  if (!jobs) return DMP_ERR_BAD_ARG;
  GemmJobs t;
  int most = 0;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_gemm_job &j = jobs[i];
    if (j.num_terms < 0 || j.num_terms > DMP_GEMM_MAX_TERMS || j.M < 0 || j.N < 0 || j.ldc < j.N || (j.C0 && j.ldc0 < j.N)) return DMP_ERR_BAD_ARG;
    if (j.M > 0 && j.N > 0 && !j.C) return DMP_ERR_BAD_ARG;
    for (int q = 0; q < j.num_terms; ++q) {
      const dmp_gemm_term &m = j.term[q];
      if (m.K < 0 || (m.K > 0 && j.M > 0 && j.N > 0 && (!m.A || !m.B))) return DMP_ERR_BAD_ARG;
      if (m.lda < (m.transA ? j.M : m.K) || m.ldb < (m.transB ? m.K : j.N)) return DMP_ERR_BAD_ARG;
    }
    t.job[i] = j;
    const int tiles = ((j.M + 15) / 16) * ((j.N + 63) / 64);
    if (tiles > most) most = tiles;
  }
  if (most == 0) return DMP_OK;
  small_gemm_jobs_k<<<dim3((unsigned)most, (unsigned)num_jobs), 256, 0, (hipStream_t)stream>>>(t);
  return check_launch();
}

}  // extern "C"
