// Fused MFMA kernels of the DMPLayer edge chain (gfx950, fp32 in / fp32 accumulate, exact fp32:
// v_mfma_f32_32x32x2_f32).  K = H = 128 only; other widths take the GEMM + epilogue-kernel path.
//
// Structure (two persistent 256-thread workgroups per CU, out of phase with each other so that one
// streams / stores while the other issues MFMAs; 4 waves each, 2 waves per SIMD in total):
//   * the [128, NC*128] weight panel lives in REGISTERS for the whole kernel: wave w keeps the
//     fragments of its 32-column slice (w&3) of every panel, b[p][s] = B[s + 64h][128p + 32(w&3) + l]
//     (h = lane>>5, l = lane&31): no LDS or cache traffic for the weights inside the loop;
//   * 64-row tiles of the streamed operand go global -> registers (prefetched one tile ahead,
//     in flight under the MFMAs) -> LDS (132-float rows: conflict-free ds_read_b128);
//   * every wave computes both 32-row sub-tiles of a 64-row tile: 64 k-steps x NC panels of
//     32x32x2 MFMAs per sub-tile, k-step s pairs k = s (lanes 0-31) with k = s+64 (lanes 32-63) so that a lane's
//     A operands are 4 consecutive floats of its LDS row (one ds_read_b128 per 4 MFMAs);
//   * the 32x32 accumulators (row = (r&3)+8(r>>2)+4h, col = l) go through a per-wave LDS transpose;
//     the epilogue (gathered node rows, gate, residual, ReLU mask, column sums) runs on the
//     transposed float4s, whose operands were requested before the transpose, and stores 16 B per lane.
//
//   dmp_edge_fwd_fused   H1[e] = relu(Z W'[:, :H] + coef[dst e] Z W'[:, H:] + P[a_e, 0:H] - P[b_e, H:2H] + b)
//                        = the G GEMM + dmp_edge_combine(relu) of the fused layer in one pass
//                        (the [E,2H] product never reaches HBM)
//   dmp_out_fwd_fused    Zn[e] = Z[e] + gate[e] (H1[e] W2^T + b2)   = Linear + dmp_gate_residual
//   dmp_bwd_h1_fused     dG[e] = [dPre | coef[dst e] dPre], dPre = H1[e] > 0 ? dO[e] W2 : 0, + column sums of dPre
//                        = Linear backward + ReLU backward + edge_combine backward in one pass
//   dmp_bwd_z_fused      dZ[e] = base[e] + s(flag) D[dst e, half] + dPre[e] A'^T + coef[dst e] dPre[e] B'^T
//                        = seg_sum2 backward (gather) + K=2H input-gradient GEMM + accumulation in one pass
//   dmp_gemm_k128        plain C = A B (development / tests)
#include "dmp_common.h"

namespace dmp {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTileRows = 64;     // rows per tile: 2 sub-tiles of 32
constexpr int kLdsStride = 132;
constexpr int kThreads = 256;     // 4 waves; two such workgroups share a CU and run out of phase
constexpr int kWaves = kThreads / 64;
constexpr int kScrStride = 36;
constexpr int kPreLoads = kTileRows * 32 / kThreads;  // float4 loads per thread per tile

enum { EPI_NONE = 0, EPI_EDGE = 1, EPI_GATE_RES = 2, EPI_RELU_BWD_G = 3, EPI_DZ = 4 };

struct MfmaArgs {
  const float *A; int64_t lda;      // streamed operand [E,128]
  const float *B; int64_t ldb;      // weights: B[k*ldb + j] (bt == 0) or B[j*ldb + k] (bt == 1)
  int bt;
  float *C; int64_t ldc;            // output [E, 128] (EPI_EDGE / EPI_GATE_RES) or [E, NC*128]
  int64_t E;
  // EPI_EDGE
  const float *P; int64_t ldp;      // node projections [N, >=2H]: P[:, 0:H] (W_dst side), P[:, H:2H] (W_src side)
  const float *coef;                // [N]
  const int32_t *src, *dst; const uint8_t *flag;
  const float *bias;                // [128] or NULL
  // EPI_GATE_RES
  const float *R; int64_t ldr;      // residual rows [E,128] or NULL (EPI_DZ: upstream gradient dZn;
                                    // EPI_RELU_BWD_G: the saved activation H1 for the ReLU mask)
  const float *gate;                // [E] or NULL
  // EPI_RELU_BWD_G
  float *partial;                   // [2*gridDim.x, 128] column-sum partials of dPre
  // EPI_DZ: gathered term  s(flag) * D[dst, flag ? H : 0 + j]
  const float *D; int64_t ldd; float s0, s1;
};

template <int NC, int EPI>
__global__ __launch_bounds__(kThreads, 2) void mfma_k128(MfmaArgs p) {
  __shared__ float As[kTileRows * kLdsStride];
  __shared__ float Cs[kWaves * 32 * kScrStride];
  __shared__ int rowA[kTileRows], rowB[kTileRows];
  __shared__ float rowS[kTileRows];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int cs = wave;  // this wave's 32-column slice; it computes both 32-row sub-tiles of a tile
  const int col = 32 * cs + li;
  float b[NC][64];
#pragma unroll
  for (int q = 0; q < NC; ++q)
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      const int k = s + 64 * h, j = 128 * q + col;
      // bt 0: B[k][j] row-major; 1: transposed storage B^T[j][k]; 2: panel q is the K-slice
      // [128q, 128q+128) of a transposed [128, NC*128] matrix: B_q[k][col] = W[col][128q + k]
      b[q][s] = p.bt == 0 ? p.B[(int64_t)k * p.ldb + j]
              : p.bt == 1 ? p.B[(int64_t)j * p.ldb + k]
                          : p.B[(int64_t)col * p.ldb + 128 * q + k];
    }
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);  // EPI_RELU_BWD_G: this lane's 4 columns

  const int64_t ntiles = (p.E + kTileRows - 1) / kTileRows;
  float4 pre[kPreLoads];
  auto load_tile = [&](int64_t t) {
#pragma unroll
    for (int m = 0; m < kPreLoads; ++m) {
      const int q = threadIdx.x + kThreads * m;
      const int64_t row = t * kTileRows + (q >> 5);
      pre[m] = row < p.E ? *reinterpret_cast<const float4 *>(p.A + row * p.lda + (q & 31) * 4) : make_float4(0, 0, 0, 0);
    }
  };
  int64_t t = blockIdx.x;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile: LDS reads done
#pragma unroll
    for (int m = 0; m < kPreLoads; ++m) {
      const int q = threadIdx.x + kThreads * m;
      *reinterpret_cast<float4 *>(&As[(q >> 5) * kLdsStride + (q & 31) * 4]) = pre[m];
    }
    if (EPI == EPI_EDGE && threadIdx.x < kTileRows) {
      const int64_t e = t * kTileRows + threadIdx.x;
      int a = 0, bb = 0;
      float cf = 0.f;
      if (e < p.E) {
        const int u = p.src[e], v = p.dst[e];
        const bool f = p.flag && p.flag[e];
        a = f ? u : v;
        bb = f ? v : u;
        cf = p.coef[v];
      }
      rowA[threadIdx.x] = a; rowB[threadIdx.x] = bb; rowS[threadIdx.x] = cf;
    }
    if (EPI == EPI_GATE_RES && threadIdx.x < kTileRows) {
      const int64_t e = t * kTileRows + threadIdx.x;
      rowS[threadIdx.x] = (p.gate && e < p.E) ? p.gate[e] : 1.f;
    }
    if (EPI == EPI_RELU_BWD_G && threadIdx.x < kTileRows) {
      const int64_t e = t * kTileRows + threadIdx.x;
      rowS[threadIdx.x] = e < p.E ? p.coef[p.dst[e]] : 0.f;
    }
    if (EPI == EPI_DZ && threadIdx.x < kTileRows) {
      const int64_t e = t * kTileRows + threadIdx.x;
      int v = 0, f = 0;
      float cf = 0.f;
      if (e < p.E) { v = p.dst[e]; f = (p.flag && p.flag[e]) ? 1 : 0; cf = p.coef[v]; }
      rowA[threadIdx.x] = v; rowB[threadIdx.x] = f; rowS[threadIdx.x] = cf;
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);  // next tile in flight under the MFMAs
#pragma unroll 1
    for (int sub = 0; sub < kTileRows / 32; ++sub) {
      const int64_t tile_row = t * kTileRows + 32 * sub;
      f32x16 acc[NC];
#pragma unroll
      for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      const float *arow = &As[(32 * sub + li) * kLdsStride + 64 * h];
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(arow + 4 * s4);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[q][4 * s4 + 0], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[q][4 * s4 + 1], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[q][4 * s4 + 2], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[q][4 * s4 + 3], acc[q], 0, 0, 0);
        }
      }
      float *scr = &Cs[wave * 32 * kScrStride];
      constexpr int NOUT = (EPI == EPI_NONE) ? NC : 1;
      // Row-gathered / row-streamed epilogue operands, fetched as float4 in the layout of the wide
      // store phase (lane -> row 8k + lane/8, 4 consecutive columns); all loads are issued before
      // the accumulators take their trip through the LDS scratch, so their latency overlaps it.
      float4 g0[4], g1[4];
      auto fetch_epilogue_operands = [&]() {
      if (EPI == EPI_EDGE || EPI == EPI_DZ || EPI == EPI_GATE_RES || EPI == EPI_RELU_BWD_G) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = 8 * k + (lane >> 3), c4 = 32 * cs + (lane & 7) * 4;
          const int rl = 32 * sub + rr;
          const int64_t row = tile_row + rr;
          g0[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          g1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < p.E) {
            if (EPI == EPI_EDGE) {
              g0[k] = *reinterpret_cast<const float4 *>(p.P + (int64_t)rowA[rl] * p.ldp + c4);
              g1[k] = *reinterpret_cast<const float4 *>(p.P + (int64_t)rowB[rl] * p.ldp + 128 + c4);
            } else if (EPI == EPI_DZ) {
              g0[k] = *reinterpret_cast<const float4 *>(p.D + (int64_t)rowA[rl] * p.ldd + (rowB[rl] ? 128 : 0) + c4);
              if (p.R) g1[k] = *reinterpret_cast<const float4 *>(p.R + row * p.ldr + c4);
            } else if (p.R) {  // EPI_GATE_RES: residual rows; EPI_RELU_BWD_G: saved activation
              g0[k] = *reinterpret_cast<const float4 *>(p.R + row * p.ldr + c4);
            }
          }
        }
      }
      };
      // NC = 2: the accumulators fill the register file, so the operands are requested after the
      // accumulators have been parked in the scratch; NC = 1: before, overlapping the round trip
      if (NC == 1) fetch_epilogue_operands();
#pragma unroll
      for (int q = 0; q < NOUT; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = acc[q][r];
          if (EPI == EPI_EDGE || EPI == EPI_DZ) v = v + rowS[32 * sub + rr] * acc[NC - 1][r];
          scr[rr * kScrStride + li] = v;
        }
        if (NC != 1) fetch_epilogue_operands();
        // written and read by the same wave: LDS operations of one wave complete in order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = 8 * k + (lane >> 3), c4 = (lane & 7) * 4;
          float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + c4]);
          const int64_t row = tile_row + rr;
          if (row < p.E) {
            if (EPI == EPI_EDGE) {
              // ((G0 + coef G1) + (P[a] - P[b])) + bias, then ReLU: the reference's order (dmpnn.py:147-152)
              const float4 bi = p.bias ? *reinterpret_cast<const float4 *>(p.bias + 32 * cs + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
              v.x = fmaxf((v.x + (g0[k].x - g1[k].x)) + bi.x, 0.f);
              v.y = fmaxf((v.y + (g0[k].y - g1[k].y)) + bi.y, 0.f);
              v.z = fmaxf((v.z + (g0[k].z - g1[k].z)) + bi.z, 0.f);
              v.w = fmaxf((v.w + (g0[k].w - g1[k].w)) + bi.w, 0.f);
            } else if (EPI == EPI_DZ) {
              const float sg = rowB[32 * sub + rr] ? p.s1 : p.s0;
              v.x += g1[k].x + sg * g0[k].x; v.y += g1[k].y + sg * g0[k].y;
              v.z += g1[k].z + sg * g0[k].z; v.w += g1[k].w + sg * g0[k].w;
            } else if (EPI == EPI_GATE_RES) {
              const float gt = rowS[32 * sub + rr];
              const float4 bi = p.bias ? *reinterpret_cast<const float4 *>(p.bias + 32 * cs + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
              v.x = (v.x + bi.x) * gt + g0[k].x; v.y = (v.y + bi.y) * gt + g0[k].y;
              v.z = (v.z + bi.z) * gt + g0[k].z; v.w = (v.w + bi.w) * gt + g0[k].w;
            } else if (EPI == EPI_RELU_BWD_G) {
              // dPre = H1 > 0 ? dH1 : 0;  dG = [dPre | coef[dst] dPre];  column sums of dPre
              v.x = g0[k].x > 0.f ? v.x : 0.f; v.y = g0[k].y > 0.f ? v.y : 0.f;
              v.z = g0[k].z > 0.f ? v.z : 0.f; v.w = g0[k].w > 0.f ? v.w : 0.f;
              colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
              const float cf = rowS[32 * sub + rr];
              *reinterpret_cast<float4 *>(p.C + row * p.ldc + 128 + 32 * cs + c4) =
                  make_float4(v.x * cf, v.y * cf, v.z * cf, v.w * cf);
            }
            *reinterpret_cast<float4 *>(p.C + row * p.ldc + 128 * q + 32 * cs + c4) = v;
          }
        }
      }
    }
  }
  if (EPI == EPI_RELU_BWD_G) {
    // PARTIALS: lanes with equal (lane & 7) hold the same 4 columns for 8 different rows: fixed-order
    // xor-shuffle tree over lane>>3, then one partial row per (workgroup, row half)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      colsum.x += __shfl_xor(colsum.x, off, 64); colsum.y += __shfl_xor(colsum.y, off, 64);
      colsum.z += __shfl_xor(colsum.z, off, 64); colsum.w += __shfl_xor(colsum.w, off, 64);
    }
    if (lane < 8)
      *reinterpret_cast<float4 *>(p.partial + (int64_t)blockIdx.x * 128 + 32 * cs + lane * 4) = colsum;
  }
}

// (the column-sum write-out of EPI_RELU_BWD_G lives at the end of mfma_k128, see PARTIALS below)
inline unsigned grid_blocks(int64_t E, int per_cu = 2) {
  const int64_t ntiles = (E + kTileRows - 1) / kTileRows;
  const int64_t cap = 256 * per_cu;  // resident workgroups: 2 per CU (NC = 2: 256 VGPRs), 3 for NC = 1
  return (unsigned)(ntiles < cap ? (ntiles > 0 ? ntiles : 1) : cap);
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_gemm_k128(const float *A, int64_t lda, const float *B, int64_t ldb, int b_transposed, float *C,
                  int64_t ldc, int64_t E, int ncols, void *stream) {
  if (E < 0 || lda < 128 || ldc < ncols || (ncols != 128 && ncols != 256)) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!A || !B || !C || lda % 4 || ldc % 4 || !aligned16(A) || !aligned16(C)) return DMP_ERR_BAD_ARG;
  MfmaArgs p{};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.bt = b_transposed; p.C = C; p.ldc = ldc; p.E = E;
  hipStream_t st = (hipStream_t)stream;
  if (ncols == 128) mfma_k128<1, EPI_NONE><<<grid_blocks(E, 3), kThreads, 0, st>>>(p);
  else mfma_k128<2, EPI_NONE><<<grid_blocks(E), kThreads, 0, st>>>(p);
  return check_launch();
}

int dmp_edge_fwd_fused(const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *P, int64_t ldp,
                       const float *coef, const float *bias, const int32_t *src, const int32_t *dst,
                       const uint8_t *flag, int64_t E, int H, float *H1, int64_t ldh, void *stream) {
  if (E < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Z || !W || !P || !coef || !src || !dst || !H1 || ldz < H || ldw < 2 * H || ldp < 2 * H || ldh < H)
    return DMP_ERR_BAD_ARG;
  if (ldz % 4 || ldh % 4 || !aligned16(Z) || !aligned16(H1)) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = Z; p.lda = ldz; p.B = W; p.ldb = ldw; p.bt = 0; p.C = H1; p.ldc = ldh; p.E = E;
  p.P = P; p.ldp = ldp; p.coef = coef; p.src = src; p.dst = dst; p.flag = flag; p.bias = bias;
  mfma_k128<2, EPI_EDGE><<<grid_blocks(E), kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int dmp_out_fwd_fused(const float *Hin, int64_t ldh, const float *W2, int64_t ldw, const float *bias,
                      const float *gate, const float *R, int64_t ldr, int64_t E, int H, float *out,
                      int64_t ldo, void *stream) {
  if (E < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Hin || !W2 || !out || ldh < H || ldw < H || ldo < H || (R && ldr < H)) return DMP_ERR_BAD_ARG;
  if (ldh % 4 || ldo % 4 || (R && ldr % 4) || !aligned16(Hin) || !aligned16(out) || (R && !aligned16(R)))
    return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = Hin; p.lda = ldh; p.B = W2; p.ldb = ldw; p.bt = 1;  // nn.Linear weight [out, in]: B[k][j] = W2[j][k]
  p.C = out; p.ldc = ldo; p.E = E; p.bias = bias; p.gate = gate; p.R = R; p.ldr = ldr;
  mfma_k128<1, EPI_GATE_RES><<<grid_blocks(E, 2), kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int64_t dmp_mfma_partial_rows(int64_t E) { return (int64_t)grid_blocks(E, 3); }

int dmp_bwd_h1_fused(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                     const float *coef, const int32_t *dst, int64_t E, int H, float *dG, int64_t ldg,
                     float *partial, void *stream) {
  if (E < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (!partial) return DMP_ERR_BAD_ARG;
  if (E == 0) return hipMemsetAsync(partial, 0, sizeof(float) * 128, (hipStream_t)stream) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!dO || !W2 || !H1 || !coef || !dst || !dG || ldo < H || ldw < H || ldh < H || ldg < 2 * H) return DMP_ERR_BAD_ARG;
  if (ldo % 4 || ldh % 4 || ldg % 4 || !aligned16(dO) || !aligned16(H1) || !aligned16(dG) || !aligned16(partial))
    return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = dO; p.lda = ldo; p.B = W2; p.ldb = ldw; p.bt = 0;  // dH1 = dO @ W2, W2 [out, in] = B[k = out][j = in]
  p.C = dG; p.ldc = ldg; p.E = E; p.R = H1; p.ldr = ldh; p.coef = coef; p.dst = dst; p.partial = partial;
  mfma_k128<1, EPI_RELU_BWD_G><<<grid_blocks(E, 3), kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int dmp_bwd_z_fused(const float *dPre, int64_t ldp, const float *W, int64_t ldw, const float *D, int64_t ldd,
                    const float *base, int64_t ldb, const float *coef, const int32_t *dst, const uint8_t *flag,
                    float s0, float s1, int64_t E, int H, float *dZ, int64_t ldz, void *stream) {
  if (E < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!dPre || !W || !D || !coef || !dst || !dZ || ldp < H || ldw < 2 * H || ldd < 2 * H || ldz < H || (base && ldb < H))
    return DMP_ERR_BAD_ARG;
  if (ldp % 4 || ldz % 4 || (base && ldb % 4) || !aligned16(dPre) || !aligned16(dZ) || (base && !aligned16(base)))
    return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = dPre; p.lda = ldp; p.B = W; p.ldb = ldw; p.bt = 2;  // panels = K-slices of W'^T
  p.C = dZ; p.ldc = ldz; p.E = E; p.R = base; p.ldr = ldb; p.coef = coef; p.dst = dst; p.flag = flag;
  p.D = D; p.ldd = ldd; p.s0 = s0; p.s1 = s1;
  mfma_k128<2, EPI_DZ><<<grid_blocks(E), kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

}  // extern "C"
