// Row-block products of the node side:  C = act([A1 | A2] B + bias + Cadd)  with fp32 operands, on the bf16 matrix
// pipe as six piece products per 16-deep k-group (dmp_mfma_common.h, "bf16x6": fp32-accurate).  gfx950.
//
// The DMPLayer's node-side products (dmpnn.py:113,121,129-140 and their backward: X [W_nl' | W_dst' | W_src'],
// [S | X] [B_n; W_nl'], dP_n B_n^T, dXP W_x^T) have N = nodes rows against E >> N edge rows, K and the output width are
// 1-3 H: their operands live in L2 / the Infinity Cache, and what bounds them is the matrix rate.  The f32-input MFMA
// (what a library fp32 GEMM runs on) has 1/16 of the bf16 MFMA rate; six bf16 piece products per partial product carry
// every partial product to 2^-24 of its magnitude at 6/16 of the f32 form's matrix-pipe cycles.
//
// One 256-thread workgroup (2 x 2 waves) per 128 x BN output tile (BN = 128 or 64), wave tile 64 x BN/2 (2 x BN/64
// MFMA blocks of 32 x 32).  Per 16-deep k-step: every thread loads 8 + 8 floats of the A and B tiles from global memory
// (one k-step ahead, in registers), splits them into bf16 pieces (hi | mid | lo planes) and writes them to the other
// LDS buffer; rows of a plane hold the step's 16 k of one A row / one B column with a stride of 12 dwords (conflict-free
// ds_read_b128 of a lane's 8 consecutive k); one barrier per step.  Both tiles are stored k-contiguous ([row][k] and
// [column][k]): the A fragment of lane (r, h) is A[row r][k = 8h ..], the B fragment B[k = 8h ..][column r].
#include <type_traits>

#include "dmp_mfma_common.h"

namespace dmp {
namespace {

struct Gemm6Args {
  const float *A1; int64_t lda1; int K1;      // A = [A1 | A2]: the first K1 columns from A1, the next K2 from A2
  const float *A2; int64_t lda2; int K2;
  const float *B; int64_t ldb; int transB;    // B[k][n] = B[k * ldb + n]  (transB: B[n * ldb + k])
  const float *bias;                          // [N] or NULL
  const float *Cadd; int64_t ldadd;           // [R, N] added to the product (before the activation) or NULL
  const float *rowscale;                      // [R] or NULL: C = Cadd + rowscale[r] * act(...)  -- gate + residual form
  int act; float slope;                       // act != 0: LeakyReLU(slope) (0 = ReLU)
  float *C; int64_t ldc; int64_t R; int N;
};

constexpr int kBM = 128, kKS = 16;            // rows per tile, k per step
constexpr int kRowD = 12;                     // dwords per plane row: 16 bf16 (8 dwords) + 4 of padding (16-byte aligned rows;
                                              // 12 r mod 64 is a different multiple of 4 for 16 different r mod 16: conflict-free)

template <int BN>
__global__ __launch_bounds__(256, 2) void gemm6_k(Gemm6Args p) {
  constexpr int kRows = kBM + BN;                           // plane rows: the A tile's rows, then the B tile's columns
  constexpr int kPlane = kRows * kRowD;                     // dwords per plane
  constexpr int NJ = BN / 64;                               // MFMA blocks per wave along the columns (wave tile 64 x BN/2)
  __shared__ __attribute__((aligned(16))) uint32_t Ls[2][3 * kPlane];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
  const int ntn = p.N / BN;
  const int64_t row0 = (int64_t)(blockIdx.x / ntn) * kBM;
  const int col0 = (int)(blockIdx.x % ntn) * BN;
  const int K = p.K1 + p.K2, nsteps = K / kKS;

  // ---- loaders: thread t loads 8 consecutive k of A row t / 2 (k half t & 1) and 8 k of B column (BN = 128: t / 2; BN = 64: t / 4
  // with 4 consecutive k ... kept simple: BN = 64 lets half of the threads load B)
  const int a_row = tid >> 1, a_kh = (tid & 1) * 8;
  const bool a_ok = row0 + a_row < p.R;
  const float *a1p = p.A1 + (row0 + a_row) * p.lda1 + a_kh;
  const float *a2p = p.A2 ? p.A2 + (row0 + a_row) * p.lda2 + a_kh : nullptr;
  const bool b_on = BN == 128 || tid < 128;
  const int b_col = tid >> 1, b_kh = (tid & 1) * 8;
  const float *bp = p.transB ? p.B + (int64_t)(col0 + b_col) * p.ldb + b_kh : p.B + (int64_t)b_kh * p.ldb + col0 + b_col;

  // kDepth register sets of prefetched floats: step s lives in set s % kDepth from its request (kDepth - 1 steps before
  // its pieces are written to LDS) -- a 16-deep step is ~800 matrix-pipe cycles, an L2 / Infinity-Cache round trip under
  // load several times that: one step of distance left every step waiting for its loads (measured: 76 TF/s-equivalent)
  constexpr int kDepth = 4;
  float4 ra[kDepth][2], rb[kDepth][2];
  auto load_step = [&](auto set, int s) {                   // global -> register set U, k-step s
    constexpr int U = decltype(set)::value;
    const int k0 = s * kKS;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    ra[U][0] = z; ra[U][1] = z; rb[U][0] = z; rb[U][1] = z;
    if (a_ok) {
      const float *q = k0 < p.K1 ? a1p + k0 : a2p + (k0 - p.K1);      // K1 is a multiple of 16: a step never straddles
      ra[U][0] = *reinterpret_cast<const float4 *>(q);
      ra[U][1] = *reinterpret_cast<const float4 *>(q + 4);
    }
    if (b_on) {
      if (p.transB) {
        rb[U][0] = *reinterpret_cast<const float4 *>(bp + k0);
        rb[U][1] = *reinterpret_cast<const float4 *>(bp + k0 + 4);
      } else {
        const float *q = bp + (int64_t)k0 * p.ldb;
        rb[U][0] = make_float4(q[0], q[p.ldb], q[2 * p.ldb], q[3 * p.ldb]);
        rb[U][1] = make_float4(q[4 * p.ldb], q[5 * p.ldb], q[6 * p.ldb], q[7 * p.ldb]);
      }
    }
  };
  auto store_step = [&](auto set, int buf) {                // register set U -> bf16 piece planes
    constexpr int U = decltype(set)::value;
    Split8 sa, sb;
    split8(ra[U][0], ra[U][1], sa);
    uint32_t *qa = &Ls[buf][a_row * kRowD + a_kh / 2];
    *reinterpret_cast<uint4 *>(qa) = make_uint4(sa.hi.u[0], sa.hi.u[1], sa.hi.u[2], sa.hi.u[3]);
    *reinterpret_cast<uint4 *>(qa + kPlane) = make_uint4(sa.mid.u[0], sa.mid.u[1], sa.mid.u[2], sa.mid.u[3]);
    *reinterpret_cast<uint4 *>(qa + 2 * kPlane) = make_uint4(sa.lo.u[0], sa.lo.u[1], sa.lo.u[2], sa.lo.u[3]);
    if (b_on) {
      split8(rb[U][0], rb[U][1], sb);
      uint32_t *qb = &Ls[buf][(kBM + b_col) * kRowD + b_kh / 2];
      *reinterpret_cast<uint4 *>(qb) = make_uint4(sb.hi.u[0], sb.hi.u[1], sb.hi.u[2], sb.hi.u[3]);
      *reinterpret_cast<uint4 *>(qb + kPlane) = make_uint4(sb.mid.u[0], sb.mid.u[1], sb.mid.u[2], sb.mid.u[3]);
      *reinterpret_cast<uint4 *>(qb + 2 * kPlane) = make_uint4(sb.lo.u[0], sb.lo.u[1], sb.lo.u[2], sb.lo.u[3]);
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  std::integral_constant<int, 0> u0;
  std::integral_constant<int, 1> u1;
  std::integral_constant<int, 2> u2;
  std::integral_constant<int, 3> u3;
  load_step(u0, 0);
  if (nsteps > 1) load_step(u1, 1);
  if (nsteps > 2) load_step(u2, 2);
  if (nsteps > 3) load_step(u3, 3);
  store_step(u0, 0);
  lds_barrier();
  // one k-step: the MFMAs of step s (LDS buffer s & 1); before them the pieces of step s + 1 (register set NEXT) go to the
  // other buffer and step s + kDepth is requested into the set step s has left (THIS)
  auto step = [&](int s, auto self, auto next) {
    const int buf = s & 1;
    const uint32_t *base = &Ls[buf][0];
    Split8 fa[2], fb[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t *q = base + (64 * wr + 32 * i + li) * kRowD + 4 * h;
      fa[i].hi.v = *reinterpret_cast<const bf16x8 *>(q);
      fa[i].mid.v = *reinterpret_cast<const bf16x8 *>(q + kPlane);
      fa[i].lo.v = *reinterpret_cast<const bf16x8 *>(q + 2 * kPlane);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t *q = base + (kBM + (BN / 2) * wc + 32 * j + li) * kRowD + 4 * h;
      fb[j].hi.v = *reinterpret_cast<const bf16x8 *>(q);
      fb[j].mid.v = *reinterpret_cast<const bf16x8 *>(q + kPlane);
      fb[j].lo.v = *reinterpret_cast<const bf16x8 *>(q + 2 * kPlane);
    }
    if (s + 1 < nsteps) store_step(next, buf ^ 1);
    if (s + kDepth < nsteps) load_step(self, s + kDepth);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = mfma_x6(fa[i], fb[j], acc[i][j]);
    lds_barrier();               // LDS only: __syncthreads() would wait for the prefetched global loads too
  };
  for (int s = 0; s < nsteps; s += kDepth) {
    step(s, u0, u1);
    if (s + 1 < nsteps) step(s + 1, u1, u2);
    if (s + 2 < nsteps) step(s + 2, u2, u3);
    if (s + 3 < nsteps) step(s + 3, u3, u0);
  }

  // ---- epilogue: accumulator (i, j, r) of wave (wr, wc): row 64 wr + 32 i + (r&3) + 8 (r>>2) + 4 h, column (BN/2) wc + 32 j + li
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + (BN / 2) * wc + 32 * j + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= p.R) continue;
        float v = acc[i][j][r] + bv;
        if (p.rowscale) {                                   // C = Cadd + rowscale * act(product + bias)
          if (p.act) v = act_fwd(v, p.slope);
          v *= p.rowscale[row];
          if (p.Cadd) v += p.Cadd[row * p.ldadd + col];
        } else {
          if (p.Cadd) v += p.Cadd[row * p.ldadd + col];
          if (p.act) v = act_fwd(v, p.slope);
        }
        p.C[row * p.ldc + col] = v;
      }
  }
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_gemm_x6(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2, const float *B, int64_t ldb,
                int transB, const float *bias, const float *Cadd, int64_t ldadd, const float *rowscale, int act, float slope,
                float *C, int64_t ldc, int64_t R, int N, void *stream) {
  if (R < 0 || N <= 0 || K1 <= 0 || K2 < 0) return DMP_ERR_BAD_ARG;
  if (N % 64 || K1 % kKS || K2 % kKS) return DMP_ERR_UNSUPPORTED;
  if (R == 0) return DMP_OK;
  if (!A1 || !B || !C || lda1 < K1 || (K2 > 0 && (!A2 || lda2 < K2)) || ldc < N || (Cadd && ldadd < N)) return DMP_ERR_BAD_ARG;
  if (ldb < (transB ? K1 + K2 : N)) return DMP_ERR_BAD_ARG;
  if (act && !slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (lda1 % 4 || (K2 > 0 && lda2 % 4) || !aligned16(A1) || (K2 > 0 && !aligned16(A2)) || (transB && (ldb % 4 || !aligned16(B))))
    return DMP_ERR_UNSUPPORTED;
  Gemm6Args p{A1, lda1, K1, K2 > 0 ? A2 : nullptr, lda2, K2, B, ldb, transB, bias, Cadd, ldadd, rowscale, act, slope, C, ldc, R, N};
  const int64_t tiles_m = (R + kBM - 1) / kBM;
  hipStream_t st = (hipStream_t)stream;
  if (N % 128 == 0) {
    const int64_t blocks = tiles_m * (N / 128);
    if (blocks > 0x7fffffff) return DMP_ERR_UNSUPPORTED;
    gemm6_k<128><<<(unsigned)blocks, 256, 0, st>>>(p);
  } else {
    const int64_t blocks = tiles_m * (N / 64);
    if (blocks > 0x7fffffff) return DMP_ERR_UNSUPPORTED;
    gemm6_k<64><<<(unsigned)blocks, 256, 0, st>>>(p);
  }
  return check_launch();
}

}  // extern "C"
