// Fused MFMA kernels of the DMPLayer edge chain (gfx950, fp32 in / fp32 accumulate, exact fp32:
// v_mfma_f32_32x32x2_f32).  K = H = 128; the one-panel kernels (out_fwd, bwd_h1, the plain product) also
// K = H = 64 (the reference's shipped hidden_dim) with H / 32 = 2 waves per workgroup; other widths take the
// GEMM + epilogue-kernel path.
//
// Structure: persistent 256-thread workgroups (4 waves, one 32-column slice of every output panel
// each), 2 (NC = 2) or 3 (NC = 1) resident per CU and not synchronised with each other: while one
// issues the MFMAs of a 32-row tile, the others stage / transpose / store theirs.
//   * the [128, NC*128] weight panel lives in REGISTERS for the whole kernel: wave w keeps the
//     fragments of its 32-column slice of every panel, b[p][s] = B[s + 64h][128p + 32w + l]
//     (h = lane>>5, l = lane&31): no LDS or cache traffic for the weights inside the loop;
//   * 32-row tiles of the streamed operand go global -> registers (prefetched two tiles ahead) ->
//     LDS (132-float rows: conflict-free ds_read_b128); 64 k-steps x NC panels of 32x32x2 MFMAs per
//     tile, k-step s pairs k = s (lanes 0-31) with k = s+64 (lanes 32-63) so that a lane's A operands
//     are 4 consecutive floats of its LDS row (one ds_read_b128 per 4 MFMAs, requested one group
//     of MFMAs ahead);
//   * the 32x32 accumulators (row = (r&3)+8(r>>2)+4h, col = l) go through a per-wave LDS transpose;
//     the epilogue (gathered node rows, gate, residual, ReLU mask, column sums) runs on the
//     transposed float4s and stores 16 B per lane.
//   * per tile:  MFMA phase | barrier | stage next tile, prefetch the one after | barrier | epilogue.
//
// What shaped the non-MFMA part (per-phase cycle counters and knob builds of rounds 1-3; the timing-only knob code is no
// longer part of this translation unit -- the findings are in DESIGN.md, appendix):
//   * while another wave of the SIMD streams f32 MFMAs back to back (64 cycles each, they occupy the
//     f32 vector lanes), a wave gets roughly ONE instruction issued per MFMA -- of any kind.  What
//     does not fit under the partner's MFMA phase (64 x NC instructions per tile) is exposed.  So
//     the instruction count of everything else is what matters: wave-uniform values are forced
//     into SGPRs, all global accesses are buffer instructions (SGPR descriptor rebased per tile,
//     per-lane offsets computed once, bounds checks by the descriptor's range: no address
//     arithmetic, no predicates, no branches), the per-edge selectors / scales are precomputed
//     arrays (dmp_edge_select_build) prefetched like the rows instead of dependent index loads;
//   * loads and stores retire through ONE in-order counter (vmcnt): the rows are staged before the
//     epilogue's stores are issued, so the wait for prefetched rows never sits behind younger stores;
//     the barrier orders LDS only (__syncthreads would drain vmcnt);
//   * 128-bit buffer stores with an SGPR offset read their data registers late (see voffC below).
// An experimental second driver (PP = 1, dmp_dev_set_mfma_variant) pairs two wave groups in one
// 512-thread workgroup and forces them half an iteration apart with barriers ("ping-pong"); it wins
// on the bare GEMM but not with the fused epilogues, and is not used by default.
//
//   dmp_edge_fwd_fused   H1[e] = relu(Z W'[:, :H] + coefE[e] Z W'[:, H:] + P[selA e, 0:H] - P[selB e, H:2H] + b)
//                        = the G GEMM + dmp_edge_combine(relu) of the fused layer in one pass
//                        (the [E,2H] product never reaches HBM)
//   dmp_out_fwd_fused    Zn[e] = Z[e] + gate[e] (H1[e] W2^T + b2)   = Linear + dmp_gate_residual
//   dmp_bwd_h1_fused     dG[e] = [dPre | coefE[e] dPre], dPre = H1[e] > 0 ? dO[e] W2 : 0, + column sums of dPre
//                        = Linear backward + ReLU backward + edge_combine backward in one pass
//   dmp_bwd_z_fused      dZ[e] = base[e] + s(flag) D[dst e, half] + dPre[e] A'^T + coefE[e] dPre[e] B'^T
//                        = seg_sum2 backward (gather) + K=2H input-gradient GEMM + accumulation in one pass
//   dmp_gemm_k128        plain C = A B (development / tests)
#include "dmp_mfma_common.h"


namespace dmp {
namespace {


enum { EPI_NONE = 0, EPI_EDGE = 1, EPI_GATE_RES = 2, EPI_RELU_BWD_G = 3, EPI_DZ = 4 };

struct MfmaArgs {
  const float *A; int64_t lda;      // streamed operand [E,128]
  const float *B; int64_t ldb;      // weights: B[k*ldb + j] (bt == 0) or B[j*ldb + k] (bt == 1), bt == 2 see below
  int bt;
  float *C; int64_t ldc;            // output [E, 128] (EPI_EDGE / EPI_GATE_RES / EPI_DZ) or [E, NC*128]
  int64_t E;
  // per-row (per-edge) arrays, prefetched with the rows
  const int32_t *idxA;              // EPI_EDGE: node whose P[:, 0:H] row is added; EPI_DZ: dst node
  const int32_t *idxB;              // EPI_EDGE: node whose P[:, H:2H] row is subtracted
  const uint8_t *flag;              // EPI_DZ: is_reversed (selects the half of D and the sign) or NULL
  const float *rowscale;            // EPI_EDGE / EPI_RELU_BWD_G / EPI_DZ: coef[dst e]; EPI_GATE_RES: gate or NULL (1)
  // gathered table: EPI_EDGE P [N, ldt>=2H]; EPI_DZ D [N, ldt>=2H]
  const float *T; int64_t ldt; int64_t num_nodes;
  const float *bias;                // [128] or NULL
  const float *R; int64_t ldr;      // EPI_GATE_RES: residual rows [E,128] or NULL; EPI_DZ: upstream gradient
                                    // or NULL; EPI_RELU_BWD_G: the saved activation H1 for the ReLU mask
  float *partial;                   // EPI_RELU_BWD_G: [2*gridDim.x, 128] column-sum partials of dPre
  float *partialA;                  // EPI_RELU_BWD_G, pipelined form, optional: [gridDim.x, H] column sums of the FETCHED rows of A -- with
                                    // a row mask of a 0 / 1 gate that is sum_e gate_e A[e] (the bias gradient of the Linear behind the gate)
  float s0, s1;                     // EPI_DZ: sign / scale of the gathered term by flag
  int gated;                        // EPI_RELU_BWD_G, dPre only: rowscale is the edge gate applied to the product's rows
  int both;                         // EPI_RELU_BWD_G: write [dPre | coef dPre] (rowscale = coef[dst e]) instead of dPre alone
  float slope;                      // EPI_EDGE / EPI_RELU_BWD_G: negative slope of the activation (0 = ReLU)
  int dead_rows;                    // pipelined form with a row mask: bit 0 = R's masked-out rows are zeros (not fetched: EPI_GATE_RES),
                                    // bit 1 = the masked-out rows of C are not stored (every reader of C leaves them out)
  // EPI_GATE_RES / EPI_RELU_BWD_G (pipelined form): bit r of rowmask[t] == 0 -> row 32 t + r has gate 0, and the rows of A (and,
  // for EPI_RELU_BWD_G, of R) it would have contributed are not fetched: their products are multiplied by the zero gate anyway
  // (out = R + 0 (...), dPre = act'(.) 0).  NULL: every row is fetched.
  const uint32_t *rowmask;
};

// X6 (the pipelined one-panel form only): the product on the bf16 matrix pipe as six piece products per 16-deep k-group
// (dmp_mfma_common.h, "bf16x6": fp32-accurate) -- the weight panel lives in registers as pieces (96 instead of 64 VGPRs:
// two workgroups per CU instead of three), a staged tile is three bf16 planes (split once, by the staging thread).  At
// 18 GFLOP per E-row launch the f32-input MFMA is as tight a bound as the kernel's bytes (114 us of matrix pipe at its
// peak); on bf16x6 (43 us) the row masks' saved bytes become time.
template <int NC, int EPI, int PP, int H = 128, bool X6 = false>
__global__ __launch_bounds__(PP ? kPPThreads : 2 * H, PP ? 1 : (X6 ? 2 : (NC == 1 ? 3 : 2))) void mfma_pp(MfmaArgs p) {
  static_assert(H == 128 || (H == 64 && !PP), "H = 64: independent workgroups only");
  static_assert(!X6 || (!PP && NC == 1 && (EPI == EPI_GATE_RES || EPI == EPI_RELU_BWD_G)), "bf16x6: the pipelined one-panel form");
  constexpr int NG = PP ? 2 : 1;                            // wave groups per workgroup
  constexpr int kGThreads = 2 * H;                          // threads of a wave group: H / 32 waves, one 32-column slice each
  constexpr int kStride = H + 4, kQ = H / 4, kHalf = H / 2, kSteps4 = H / 8;   // LDS row stride, float4 per row, k per lane half
  constexpr uint32_t kRowBytes = H * 4u;
  // kPipe (one panel, streamed epilogue operand: out_fwd / bwd_h1): one barrier per tile, the staging of the next
  // tile (second As buffer), the row requests of the tile after it and the epilogue operand requests of the NEXT
  // tile (second operand register set: streamed rows need a whole tile of latency) in the MFMA shadow.
  constexpr bool kPipe = !PP && NC == 1 && (EPI == EPI_GATE_RES || EPI == EPI_RELU_BWD_G);
  constexpr int NBUF = kPipe ? 2 : 1, NPAR = kPipe ? 3 : 2;
  // X6: a staged tile = three bf16 planes (hi | mid | lo), rows of kStrideD dwords (H bf16 + 8 of padding: conflict-free
  // ds_read_b128 of 8 consecutive k per lane, as in dmp_typed.hip)
  constexpr int kStrideD = (H + 8) / 2, kPlane = kSub * kStrideD, kGroups = kHalf / 8;
  __shared__ __attribute__((aligned(16))) float As[NG * NBUF][X6 ? 3 * kPlane : kSub * kStride];
  __shared__ float Cs[(H / 32) * NG][32 * kScrStride];
  __shared__ uint32_t rowA[NG][NPAR][kSub], rowB[NG][NPAR][kSub];   // [group][tile parity (kPipe: tile % 3)][row]
  __shared__ float rowS[NG][NPAR][kSub];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int grp = PP ? wave >> 2 : 0, cs = PP ? wave & 3 : wave, gtid = threadIdx.x & (kGThreads - 1);
  const int col = 32 * cs + li;
  float b[NC][X6 ? 1 : kHalf];
  Split8 B6[X6 ? kGroups : 1];                              // X6: fragment g = the pieces of k-steps kHalf h + 8 g .. + 7
  auto wload = [&](int q, int s) {
    const int k = s + kHalf * h, j = H * q + col;
    // bt 0: B[k][j] row-major; 1: transposed storage B^T[j][k]; 2: panel q is the K-slice
    // [Hq, Hq+H) of a transposed [H, NC*H] matrix: B_q[k][col] = W[col][Hq + k]
    return p.bt == 0 ? p.B[(int64_t)k * p.ldb + j]
         : p.bt == 1 ? p.B[(int64_t)j * p.ldb + k]
                     : p.B[(int64_t)col * p.ldb + H * q + k];
  };
  if (X6) {
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = wload(0, 8 * g + j);
      split8(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), B6[X6 ? g : 0]);
    }
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(B6[X6 ? g : 0].hi.u[j]), "v"(B6[X6 ? g : 0].mid.u[j]), "v"(B6[X6 ? g : 0].lo.u[j]));
  } else {
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int s = 0; s < kHalf; ++s) b[q][X6 ? 0 : s] = wload(q, s);
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int s = 0; s < kHalf; ++s) asm volatile("" ::"v"(b[q][X6 ? 0 : s]));  // loads complete here, not inside the loop
  }
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);  // EPI_RELU_BWD_G: this lane's 4 columns
  float *As_g = As[grp * NBUF];
  float *scr = Cs[wave];

  // per-lane constants of the wide (float4) layout: lane -> row 8k + lane/8, 4 columns from c4
  const int lrow = lane >> 3, c4 = 32 * cs + (lane & 7) * 4;
  // Stores take a per-lane VGPR offset for every 8-row group and NO SGPR offset: a 128-bit buffer
  // store with an SGPR offset reads its data registers late (the "store data" hazard the ISA
  // documents for this form); next to a streaming MFMA wave that read was seen arriving after the
  // registers had been reused, two instructions later.
  uint32_t voffC[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) voffC[k] = ((uint32_t)(8 * k + lrow) * (uint32_t)p.ldc + (uint32_t)c4) * 4u;
  const uint32_t voffR = ((uint32_t)lrow * (uint32_t)p.ldr + (uint32_t)c4) * 4u;
  const uint32_t voffA = ((uint32_t)(gtid / kQ) * (uint32_t)p.lda + (uint32_t)(gtid % kQ) * 4u) * 4u;
  const uint32_t voffT = (uint32_t)c4 * 4u;
  // SGPR byte offsets of the 8-row groups of a tile
  const uint32_t grpA = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8 * p.lda * 4));
  const uint32_t grpR = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8 * p.ldr * 4));
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((EPI == EPI_EDGE || EPI == EPI_GATE_RES) && p.bias) bias4 = *reinterpret_cast<const float4 *>(p.bias + c4);
  // EPI_RELU_BWD_G writes [dPre | coef dPre] when asked for both halves (p.both), dPre only otherwise
  const bool both_halves = EPI == EPI_RELU_BWD_G && p.both != 0;
  const float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.slope)));   // SGPR
  const int kOutCols = (EPI == EPI_NONE) ? NC * H : (both_halves ? 2 * H : H);

  // whole-array descriptors: per-row arrays and the gathered table
  const uint32_t rows4 = (uint32_t)(p.E * 4);
  const rsrc_t rs_idxA = make_rsrc(p.idxA, p.idxA ? rows4 : 0u);
  const rsrc_t rs_idxB = make_rsrc(p.idxB, p.idxB ? rows4 : 0u);
  const rsrc_t rs_flag = make_rsrc(p.flag, p.flag ? (uint32_t)p.E : 0u);
  const rsrc_t rs_scale = make_rsrc(p.rowscale, p.rowscale ? rows4 : 0u);
  const rsrc_t rs_T = make_rsrc(p.T, p.T ? (uint32_t)(p.num_nodes * p.ldt * 4) : 0u);  // < 4 GiB (checked by the host)

  // tile indices are 32-bit and wave-uniform (SGPRs)
  const int ntiles = (int)((p.E + kSub - 1) / kSub);
  const int stride = (int)gridDim.x * NG;
  const int first = (int)blockIdx.x * NG;
  const int niter = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;  // group 0's tile count (>= group 1's)
  auto tile_rows = [&](int t) {                            // valid rows of tile t (0 past the end), scalar
    const int64_t left = p.E - (int64_t)t * kSub;
    return (int)(left < 0 ? 0 : (left > kSub ? kSub : left));
  };

  // row mask of tile t as a wave-uniform word (scalar load); tiles past the end: no rows
  constexpr bool kMasked = kPipe;
  auto mask_of = [&](int t) -> uint32_t {
    if (!kMasked || !p.rowmask) return 0xffffffffu;
    return t < ntiles ? p.rowmask[t] : 0u;
  };
  // a lane's byte offset for row r of a tile with mask mk: out of the descriptor's range for a masked row (the load returns
  // zeros and moves nothing)
  // (mk8 = the mask shifted down by the row group's 8 m: a scalar; bit = this lane's row bit inside a group of 8 rows)
  auto voff_if = [&](uint32_t mk8, uint32_t bit, uint32_t voff) -> uint32_t { return voff | ((mk8 & bit) ? 0u : 0x80000000u); };
  const uint32_t abit = 1u << (gtid / kQ);                 // A rows of a thread: gtid / kQ + 8 m
  const uint32_t rbit = 1u << (lane >> 3);                 // operand rows of a lane: 8 k + lane / 8
  // what a mask says about the residual rows (EPI_GATE_RES) / about the stores: everything unless the caller said otherwise
  const uint32_t res_all = (kMasked && (p.dead_rows & 1)) ? 0u : 0xffffffffu;
  const uint32_t store_all = (kMasked && (p.dead_rows & 2)) ? 0u : 0xffffffffu;

  float4 pre[kSubLoads];
  uint32_t pre_a = 0, pre_b = 0;
  float pre_s = 0.f;
  auto load_rows = [&](int t) {                           // global -> registers: rows + per-row scalars of tile t
    const rsrc_t ra = make_rsrc(p.A + (int64_t)t * kSub * p.lda, tile_bytes(tile_rows(t), p.lda, H));
    const uint32_t mk = mask_of(t);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      pre[m] = buf_load4(ra, kMasked ? voff_if(mk >> (8 * m), abit, voffA) : voffA, m * grpA);
    if (EPI != EPI_NONE && gtid < kSub) {
      const uint32_t so = (uint32_t)t * (kSub * 4u);      // E * 4 < 2^32 (checked by the host)
      if (EPI == EPI_EDGE || EPI == EPI_DZ) pre_a = __builtin_amdgcn_raw_buffer_load_b32(rs_idxA, gtid * 4, (int)so, 0);
      if (EPI == EPI_EDGE) pre_b = __builtin_amdgcn_raw_buffer_load_b32(rs_idxB, gtid * 4, (int)so, 0);
      if (EPI == EPI_DZ) pre_b = __builtin_amdgcn_raw_buffer_load_b8(rs_flag, gtid, (int)(so >> 2), 0);
      pre_s = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_scale, gtid * 4, (int)so, 0));
    }
  };
  auto stage = [&](int par) {                             // registers -> LDS
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      if (true || pre[m].x == 123.456f)
      *reinterpret_cast<float4 *>(&As_g[((gtid / kQ) + 8 * m) * kStride + (gtid % kQ) * 4]) = pre[m];
    if (EPI != EPI_NONE && gtid < kSub) {
      uint32_t a = 0, bb = 0;
      if (EPI == EPI_EDGE) {                              // byte offsets of the two gathered P rows
        a = pre_a * (uint32_t)(p.ldt * 4);
        bb = pre_b * (uint32_t)(p.ldt * 4) + kRowBytes;
      } else if (EPI == EPI_DZ) {
        bb = pre_b;
        a = pre_a * (uint32_t)(p.ldt * 4) + (bb ? kRowBytes : 0u);
      }
      rowA[grp][par][gtid] = a; rowB[grp][par][gtid] = bb;
      rowS[grp][par][gtid] = (EPI == EPI_GATE_RES && !p.rowscale) ? 1.f : pre_s;
    }
  };

  float4 csA = make_float4(0.f, 0.f, 0.f, 0.f);            // partialA: this thread's 4 columns of the rows it stages
  auto stage_row_to = [&](int buf, int m) {
    if (EPI == EPI_RELU_BWD_G && kPipe && p.partialA) { csA.x += pre[m].x; csA.y += pre[m].y; csA.z += pre[m].z; csA.w += pre[m].w; }
    if (X6) {                                              // split once, here: three bf16 planes
      uint2 ph, pm, pl;
      split_pair(pre[m].x, pre[m].y, ph.x, pm.x, pl.x);
      split_pair(pre[m].z, pre[m].w, ph.y, pm.y, pl.y);
      uint32_t *q = reinterpret_cast<uint32_t *>(&As[buf][0]) + ((gtid / kQ) + 8 * m) * kStrideD + (gtid % kQ) * 2;
      *reinterpret_cast<uint2 *>(q) = ph;
      *reinterpret_cast<uint2 *>(q + kPlane) = pm;
      *reinterpret_cast<uint2 *>(q + 2 * kPlane) = pl;
    } else {
      *reinterpret_cast<float4 *>(&As[buf][((gtid / kQ) + 8 * m) * kStride + (gtid % kQ) * 4]) = pre[m];
    }
  };
  auto stage_scalars_to = [&](int par) {
    if (gtid < kSub) rowS[grp][par][gtid] = (EPI == EPI_GATE_RES && !p.rowscale) ? 1.f : pre_s;
  };
  auto load_scalars_of = [&](int t) {
    if (gtid < kSub) {
      const uint32_t so = (uint32_t)t * (kSub * 4u);
      pre_s = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_scale, gtid * 4, (int)so, 0));
    }
  };

  f32x16 acc[NC];
  float4 g0[4], g1[4];
  float4 g0n[kPipe ? 4 : 1];                               // kPipe: the next tile's epilogue operand rows
  auto fetch_operands = [&](int t, int par) {             // epilogue operands of tile t, in store layout
    const rsrc_t rr_ = make_rsrc(p.R ? p.R + (int64_t)t * kSub * p.ldr : nullptr,
                                 p.R ? tile_bytes(tile_rows(t), p.ldr, H) : 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = 8 * k + lrow;
      if (EPI == EPI_EDGE) {
        g0[k] = buf_load4(rs_T, rowA[grp][par][rr] + voffT, 0);
        g1[k] = buf_load4(rs_T, rowB[grp][par][rr] + voffT, 0);
      } else if (EPI == EPI_DZ) {
        g0[k] = buf_load4(rs_T, rowA[grp][par][rr] + voffT, 0);
        g1[k] = buf_load4(rr_, voffR, k * grpR);
      } else if (EPI == EPI_RELU_BWD_G && kMasked) {
        g0[k] = buf_load4(rr_, voff_if(mask_of(t) >> (8 * k), rbit, voffR), k * grpR);
      } else if (EPI == EPI_GATE_RES && kMasked) {
        g0[k] = buf_load4(rr_, voff_if((mask_of(t) | res_all) >> (8 * k), rbit, voffR), k * grpR);
      } else if (EPI == EPI_GATE_RES || EPI == EPI_RELU_BWD_G) {
        g0[k] = buf_load4(rr_, voffR, k * grpR);
      }
    }
  };
  auto compute = [&]() {                                  // MFMA phase: 64 k-steps x NC panels on As_g
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float *arow = &As_g[li * kStride + kHalf * h];
    // the A operands of k-steps 4(s4+1).. are requested BEFORE the MFMAs of k-steps 4 s4.. issue
    float4 a4 = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
    for (int s4 = 0; s4 < kSteps4; ++s4) {
      float4 an = a4;
      if (s4 + 1 < kSteps4) an = *reinterpret_cast<const float4 *>(arow + 4 * (s4 + 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[q][X6 ? 0 : 4 * s4 + 0], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[q][X6 ? 0 : 4 * s4 + 1], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[q][X6 ? 0 : 4 * s4 + 2], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[q][X6 ? 0 : 4 * s4 + 3], acc[q], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a4 = an;
    }
  };
  auto tile_step = [&](int k, int par3, int t1, int t2, uint32_t mk1, uint32_t mk2) {   // kPipe only: tile k in As[k & 1]; t1 / t2: the next two tiles (mk1 / mk2: their row masks)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    const int buf = k & 1, nxt3 = par3 == 2 ? 0 : par3 + 1;
    const rsrc_t rr1 = make_rsrc(p.R ? p.R + (int64_t)t1 * kSub * p.ldr : nullptr, p.R ? tile_bytes(tile_rows(t1), p.ldr, H) : 0u);
    const rsrc_t ra2 = make_rsrc(p.A + (int64_t)t2 * kSub * p.lda, tile_bytes(tile_rows(t2), p.lda, H));
    const uint32_t mk1r = EPI == EPI_RELU_BWD_G ? mk1 : (mk1 | res_all);         // the operand rows of tile k+1 to fetch
    // in the shadow of the MFMA groups (H = 64: 8 groups, two of these 14 actions after each)
    auto shadow = [&](int i) {
      if (i < 4) g0n[kPipe ? i : 0] = buf_load4(rr1, voff_if(mk1r >> (8 * i), rbit, voffR), i * grpR);   // tile k+1's operand rows 8 i + lrow
      else if (i < 8) stage_row_to(buf ^ 1, i - 4);                               // tile k+1 -> the other buffer
      else if (i == 8) stage_scalars_to(nxt3);
      else if (i < 13) pre[i - 9] = buf_load4(ra2, voff_if(mk2 >> (8 * (i - 9)), abit, voffA), (i - 9) * grpA);   // tile k+2's rows
      else if (i == 13) load_scalars_of(t2);
    };
    if (X6) {
      // per 16-deep k-group g: six MFMAs on the three piece fragments of A (this lane's 8 consecutive k = kHalf h + 8 g ..
      // of row li: one ds_read_b128 per plane, requested one group ahead) and the panel fragment B6[g]; the 14 shadow
      // actions are spread between them (two per group at H = 128, four at H = 64)
      constexpr int kPer = 16 / kGroups;
      const uint32_t *ar = reinterpret_cast<const uint32_t *>(&As[buf][0]) + li * kStrideD + (kHalf / 2) * h;
      Frag8 ah, am, al;
      ah.v = *reinterpret_cast<const bf16x8 *>(ar);
      am.v = *reinterpret_cast<const bf16x8 *>(ar + kPlane);
      al.v = *reinterpret_cast<const bf16x8 *>(ar + 2 * kPlane);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        Frag8 nh = ah, nm = am, nl = al;
        if (g + 1 < kGroups) {
          nh.v = *reinterpret_cast<const bf16x8 *>(ar + 4 * (g + 1));
          nm.v = *reinterpret_cast<const bf16x8 *>(ar + kPlane + 4 * (g + 1));
          nl.v = *reinterpret_cast<const bf16x8 *>(ar + 2 * kPlane + 4 * (g + 1));
        }
        const Split8 &bb = B6[X6 ? g : 0];
        auto after = [&](int n) {                           // the actions due after the n-th MFMA of the group
#pragma unroll
          for (int q = 0; q < kPer; ++q)
            if (6 * (q + 1) / kPer == n) shadow(kPer * g + q);
          __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bb.hi.v, acc[0], 0, 0, 0); after(1);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.lo.v, acc[0], 0, 0, 0); after(2);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.mid.v, acc[0], 0, 0, 0); after(3);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.hi.v, acc[0], 0, 0, 0); after(4);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.mid.v, acc[0], 0, 0, 0); after(5);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.hi.v, acc[0], 0, 0, 0); after(6);
        ah = nh; am = nm; al = nl;
      }
      return;
    }
    const float *arow = &As[buf][li * kStride + kHalf * h];
    float4 a4 = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
    for (int s4 = 0; s4 < kSteps4; ++s4) {
      float4 an = a4;
      if (s4 + 1 < kSteps4) an = *reinterpret_cast<const float4 *>(arow + 4 * (s4 + 1));
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[0][X6 ? 0 : 4 * s4 + 0], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[0][X6 ? 0 : 4 * s4 + 1], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[0][X6 ? 0 : 4 * s4 + 2], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[0][X6 ? 0 : 4 * s4 + 3], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kSteps4 == 16) shadow(s4);
      else { shadow(2 * s4); shadow(2 * s4 + 1); }
      a4 = an;
    }
  };
  // NC = 1: the epilogue operands are requested before the MFMA phase (registers to spare);
  // NC = 2: after the accumulators have been parked in the scratch -- the other group's MFMA
  // phase covers their latency
  constexpr bool kEarly = (NC == 1);
  auto epilogue = [&](int t, int par, uint32_t mk0 = 0xffffffffu) {   // transpose through LDS, combine, store (mk0: the tile's row mask)
    const uint32_t mks = mk0 | store_all;
    constexpr int NOUT = (EPI == EPI_NONE) ? NC : 1;
    const rsrc_t rc = make_rsrc(p.C + (int64_t)t * kSub * p.ldc, tile_bytes(tile_rows(t), p.ldc, kOutCols));
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[q][r];
        if (EPI == EPI_EDGE || EPI == EPI_DZ) v = v + rowS[grp][par][rr] * acc[NC - 1][r];
        scr[rr * kScrStride + li] = v;
      }
      if (EPI != EPI_NONE && !kEarly) fetch_operands(t, par);
      // written and read by the same wave: LDS operations of one wave complete in order
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rr = 8 * k + lrow;
        float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
        if (EPI == EPI_EDGE) {
          // ((G0 + coef G1) + (P[a] - P[b])) + bias, then ReLU: the reference's order (dmpnn.py:147-152)
          v.x = act_fwd((v.x + (g0[k].x - g1[k].x)) + bias4.x, slope);
          v.y = act_fwd((v.y + (g0[k].y - g1[k].y)) + bias4.y, slope);
          v.z = act_fwd((v.z + (g0[k].z - g1[k].z)) + bias4.z, slope);
          v.w = act_fwd((v.w + (g0[k].w - g1[k].w)) + bias4.w, slope);
        } else if (EPI == EPI_DZ) {
          const float sg = rowB[grp][par][rr] ? p.s1 : p.s0;
          v.x += g1[k].x + sg * g0[k].x; v.y += g1[k].y + sg * g0[k].y;
          v.z += g1[k].z + sg * g0[k].z; v.w += g1[k].w + sg * g0[k].w;
        } else if (EPI == EPI_GATE_RES) {
          const float gt = rowS[grp][par][rr];
          v.x = (v.x + bias4.x) * gt + g0[k].x; v.y = (v.y + bias4.y) * gt + g0[k].y;
          v.z = (v.z + bias4.z) * gt + g0[k].z; v.w = (v.w + bias4.w) * gt + g0[k].w;
        } else if (EPI == EPI_RELU_BWD_G) {
          // dPre = H1 > 0 ? dH1 : 0;  dG = [dPre | coef[dst] dPre];  column sums of dPre
          // (rows past the end: H1 reads as 0 -> dPre = 0, nothing stored)
          if (p.gated) {                                     // dO = gate * dOut: a row scale commutes with the product
            const float gt = rowS[grp][par][rr];
            v.x *= gt; v.y *= gt; v.z *= gt; v.w *= gt;
          }
          v.x = act_bwd(g0[k].x, v.x, slope); v.y = act_bwd(g0[k].y, v.y, slope);
          v.z = act_bwd(g0[k].z, v.z, slope); v.w = act_bwd(g0[k].w, v.w, slope);
          colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
          if (both_halves) {
            const float cf = rowS[grp][par][rr];
            buf_store4(make_float4(v.x * cf, v.y * cf, v.z * cf, v.w * cf), rc, (kMasked ? voff_if(mks >> (8 * k), rbit, voffC[k]) : voffC[k]) + kRowBytes, 0);
          }
        }
        buf_store4(v, rc, (kMasked ? voff_if(mks >> (8 * k), rbit, voffC[k]) : voffC[k]) + kRowBytes * q, 0);
      }
    }
  };

  const int t0 = first + grp;
  const int mine = t0 < ntiles ? (ntiles - t0 + stride - 1) / stride : 0;
  auto tile = [&](int k) { return __builtin_amdgcn_readfirstlane(t0 + k * stride); };
  if (kPipe) {
    // Per tile k:  MFMA phase with everything else in its shadow | barrier | epilogue k.
    load_rows(tile(0));
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row_to(0, m);
    stage_scalars_to(0);
    load_rows(tile(1));
    fetch_operands(tile(0), 0);
    lds_barrier();
    int par3 = 0;
    uint32_t mk0 = mask_of(tile(0)), mk1 = mask_of(tile(1)), mk2 = mask_of(tile(2));
    for (int k = 0; k < mine; ++k) {
      const uint32_t mk3 = mask_of(tile(k + 3));             // requested a whole tile before its first use
      tile_step(k, par3, tile(k + 1), tile(k + 2), mk1, mk2);
      const uint32_t mke = mk0;
      mk0 = mk1; mk1 = mk2; mk2 = mk3;
      lds_barrier();               // tile k+1 is staged for everyone, everyone is done with tile k's rows
      epilogue(tile(k), par3, mke);
#pragma unroll
      for (int q = 0; q < 4; ++q) g0[q] = g0n[kPipe ? q : 0];
      par3 = par3 == 2 ? 0 : par3 + 1;
    }
  } else if (!PP) {
    // Independent 256-thread workgroups (2-3 resident per CU, not synchronised with each other).
    // Per tile k:  MFMA phase | barrier | stage k+1, prefetch k+2 | barrier | epilogue k.
    // The wait for the prefetched rows in `stage` only has YOUNGER traffic behind it (the previous
    // epilogue's stores, the early operand loads), which the in-order counter lets it skip.
    load_rows(tile(0));
    stage(0);
    load_rows(tile(1));
    lds_barrier();
    for (int k = 0; k < mine; ++k) {
      const int par = k & 1;
      if (EPI != EPI_NONE && kEarly) fetch_operands(tile(k), par);
      compute();
      lds_barrier();               // every wave is done reading this tile's rows
      stage(par ^ 1);
      load_rows(tile(k + 2));
      lds_barrier();               // tile k+1 is in LDS for everyone
      epilogue(tile(k), par);
    }
  } else {
    // Both groups run the SAME step sequence, one step apart: in step s a group does its memory phase
    // when (s + grp) is even (staging of its next tile, epilogue of the tile it computed last, prefetch
    // of the tile after) and its MFMA phase when (s + grp) is odd.  Tile k of a group: first + grp + k*stride;
    // tiles past the end are all-zero / dropped by the buffer range checks.
    int staged = 0, computed = 0, stored = 0;
    load_rows(tile(0));
    for (int s = 0; s < 2 * niter + 2; ++s) {
      if (((s + grp) & 1) == 0) {
        // Staging first, the next prefetch last (one in-order vmcnt, see the header); unconditional,
        // on one straight-line path, so that the compiler sees the prefetch registers as free and does
        // not drain the epilogue's stores before reusing them.
        stage(staged & 1);
        ++staged;
        if (stored < computed) {
          epilogue(tile(stored), stored & 1);
          ++stored;
        }
        load_rows(tile(staged));
      } else if (computed < staged && computed < mine) {
        if (EPI != EPI_NONE && kEarly) fetch_operands(tile(computed), computed & 1);
        compute();
        ++computed;
      }
      lds_barrier();
    }

  }

  if (EPI == EPI_RELU_BWD_G) {
    // lanes with equal (lane & 7) hold the same 4 columns for 8 different rows: fixed-order
    // xor-shuffle tree over lane>>3, then one partial row per (workgroup, group)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      colsum.x += __shfl_xor(colsum.x, off, 64); colsum.y += __shfl_xor(colsum.y, off, 64);
      colsum.z += __shfl_xor(colsum.z, off, 64); colsum.w += __shfl_xor(colsum.w, off, 64);
    }
    if (lane < 8)
      *reinterpret_cast<float4 *>(p.partial + ((int64_t)blockIdx.x * NG + grp) * H + 32 * cs + lane * 4) = colsum;
    if (kPipe && p.partialA) {
      // the 8 threads that staged the same 4 columns (gtid % kQ), added in a fixed order through the tile buffer
      lds_barrier();
      float *red = reinterpret_cast<float *>(&As[0][0]);
      *reinterpret_cast<float4 *>(&red[(gtid / kQ) * H + (gtid % kQ) * 4]) = csA;
      lds_barrier();
      if (gtid < kQ) {
        float4 t = *reinterpret_cast<const float4 *>(&red[gtid * 4]);
#pragma unroll
        for (int g = 1; g < kGThreads / kQ; ++g) {
          const float4 u = *reinterpret_cast<const float4 *>(&red[g * H + gtid * 4]);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4 *>(p.partialA + (int64_t)blockIdx.x * H + gtid * 4) = t;
      }
    }
  }
}

inline unsigned pp_blocks(int64_t E) {
  const int64_t pairs = ((E + kSub - 1) / kSub + 1) / 2;
  return (unsigned)(pairs < 256 ? (pairs > 0 ? pairs : 1) : 256);
}

int g_variant = 0;  // development switch: 0 = independent workgroups, 1 = ping-pong pairs

inline unsigned wg_blocks(int64_t E, int per_cu) {
  const int64_t ntiles = (E + kSub - 1) / kSub, cap = 256 * per_cu;
  return (unsigned)(ntiles < cap ? (ntiles > 0 ? ntiles : 1) : cap);
}

// the pipelined one-panel kernels (out_fwd, bwd_h1) at H = 128: on the bf16 pipe (two workgroups per CU) unless the
// development switch asks for the exact f32 MFMA (dmp_dev_set_exact_fp32) or for the ping-pong variant
template <int EPI> constexpr bool x6_kind() { return EPI == EPI_GATE_RES || EPI == EPI_RELU_BWD_G; }
template <int NC, int EPI> inline bool x6_on() { return NC == 1 && x6_kind<EPI>() && g_variant == 0 && !g_exact_fp32; }
template <int NC, int EPI>
inline int launch_mfma(const MfmaArgs &p, hipStream_t st) {
  if (g_variant == 1) mfma_pp<NC, EPI, 1><<<pp_blocks(p.E), kPPThreads, 0, st>>>(p);
  else if constexpr (NC == 1 && x6_kind<EPI>()) {
    if (x6_on<NC, EPI>()) mfma_pp<NC, EPI, 0, 128, true><<<wg_blocks(p.E, 2), kGroupThreads, 0, st>>>(p);
    else mfma_pp<NC, EPI, 0><<<wg_blocks(p.E, 3), kGroupThreads, 0, st>>>(p);
  } else mfma_pp<NC, EPI, 0><<<wg_blocks(p.E, NC == 1 ? 3 : 2), kGroupThreads, 0, st>>>(p);
  return check_launch();
}
// H = 64, one panel: 128-thread workgroups, 5 per CU (27 KB of LDS each)
constexpr int kPerCU64 = 5;
template <int EPI>
inline int launch_mfma64(const MfmaArgs &p, hipStream_t st) {
  mfma_pp<1, EPI, 0, 64><<<wg_blocks(p.E, kPerCU64), 128, 0, st>>>(p);
  return check_launch();
}

// 32-bit byte offsets inside the kernel
inline bool fits32(int64_t rows, int64_t ld) { return rows * ld * 4 < ((int64_t)1 << 32); }

// bit r of mask[t] = (gate[32 t + r] != 0); rows past the end: 0
__global__ __launch_bounds__(256) void row_mask_bits_k(const float *__restrict__ gate, int64_t E, uint32_t *__restrict__ mask) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool on = r < E && gate[r] != 0.f;
  const unsigned long long b = __ballot(on);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0 && r < ((E + 31) / 32) * 32) mask[r >> 5] = (uint32_t)(b >> (lane & 32));
}

// several gates in one launch (dmp_row_mask_bits_jobs: the union's node gate and edge gate of a step): blockIdx.y = the job
struct RowMaskJobs { const float *gate[DMP_ROWMASK_MAX_JOBS]; int64_t R[DMP_ROWMASK_MAX_JOBS]; uint32_t *mask[DMP_ROWMASK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void row_mask_bits_jobs_k(const RowMaskJobs t) {
  const int j = blockIdx.y;
  const int64_t E = t.R[j];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= ((E + 31) / 32) * 32) return;          // (whole blocks past the job's rows)
  const bool on = r < E && t.gate[j][r] != 0.f;
  const unsigned long long b = __ballot(on);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0 && r < ((E + 31) / 32) * 32) t.mask[j][r >> 5] = (uint32_t)(b >> (lane & 32));
}

// bit r of mask[t] = (row 32 t + r of X [R, ldx] has a non-zero among its first K entries)
__global__ __launch_bounds__(256) void row_mask_rows_k(const float *__restrict__ X, int64_t ldx, int K, int64_t R, uint32_t *__restrict__ mask) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool on = false;
  if (r < R)
    for (int k = 0; k < K; ++k) on |= X[r * ldx + k] != 0.f;
  const unsigned long long b = __ballot(on);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0 && r < ((R + 31) / 32) * 32) mask[r >> 5] = (uint32_t)(b >> (lane & 32));
}

__global__ void edge_select_k(const int32_t *src, const int32_t *dst, const uint8_t *flag, const float *coef,
                              int64_t E, int32_t *selA, int32_t *selB, float *coefE) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int u = src[e], v = dst[e];
  const bool f = flag && flag[e];
  selA[e] = f ? u : v;
  selB[e] = f ? v : u;
  coefE[e] = coef[v];
}

// the selectors with the nodes under a zero of a 0 / 1 node gate replaced by -1 (an index no descriptor holds: the tile
// kernels' gathers of such a node's row return zeros without touching memory), and the destination likewise
__global__ void edge_select_nodes_k(const int32_t *src, const int32_t *dst, const uint8_t *flag, const uint32_t *nodemask,
                                    int64_t E, int32_t *selA, int32_t *selB, int32_t *dstM) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int u = src[e], v = dst[e];
  if (!((nodemask[u >> 5] >> (u & 31)) & 1u)) u = -1;
  if (!((nodemask[v >> 5] >> (v & 31)) & 1u)) v = -1;
  const bool f = flag && flag[e];
  selA[e] = f ? u : v;
  selB[e] = f ? v : u;
  dstM[e] = v;
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_edge_select_build(const int32_t *src, const int32_t *dst, const uint8_t *flag, const float *coef,
                          int64_t E, int32_t *selA, int32_t *selB, float *coefE, void *stream) {
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!src || !dst || !coef || !selA || !selB || !coefE) return DMP_ERR_BAD_ARG;
  edge_select_k<<<(unsigned)((E + kBlock - 1) / kBlock), kBlock, 0, (hipStream_t)stream>>>(src, dst, flag, coef, E, selA,
                                                                                         selB, coefE);
  return check_launch();
}

int dmp_edge_select_nodes(const int32_t *src, const int32_t *dst, const uint8_t *flag, const uint32_t *nodemask, int64_t E,
                          int32_t *selA, int32_t *selB, int32_t *dstM, void *stream) {
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!src || !dst || !nodemask || !selA || !selB || !dstM) return DMP_ERR_BAD_ARG;
  edge_select_nodes_k<<<(unsigned)((E + kBlock - 1) / kBlock), kBlock, 0, (hipStream_t)stream>>>(src, dst, flag, nodemask, E, selA,
                                                                                               selB, dstM);
  return check_launch();
}

int dmp_gemm_k128(const float *A, int64_t lda, const float *B, int64_t ldb, int b_transposed, float *C,
                  int64_t ldc, int64_t E, int ncols, void *stream) {
  if (E < 0 || lda < 128 || ldc < ncols || (ncols != 128 && ncols != 256)) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!A || !B || !C || lda % 4 || ldc % 4 || !aligned16(A) || !aligned16(C)) return DMP_ERR_BAD_ARG;
  if (!fits32(kSub, lda) || !fits32(kSub, ldc)) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.bt = b_transposed; p.C = C; p.ldc = ldc; p.E = E;
  p.ldr = 128; p.ldt = 256;
  hipStream_t st = (hipStream_t)stream;
  return ncols == 128 ? launch_mfma<1, EPI_NONE>(p, st) : launch_mfma<2, EPI_NONE>(p, st);
}

int dmp_gemm_k64(const float *A, int64_t lda, const float *B, int64_t ldb, int b_transposed, float *C, int64_t ldc, int64_t E,
                 void *stream) {
  if (E < 0 || lda < 64 || ldc < 64) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!A || !B || !C || lda % 4 || ldc % 4 || !aligned16(A) || !aligned16(C)) return DMP_ERR_BAD_ARG;
  if (!fits32(kSub, lda) || !fits32(kSub, ldc)) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.bt = b_transposed; p.C = C; p.ldc = ldc; p.E = E; p.ldr = 64; p.ldt = 128;
  return launch_mfma64<EPI_NONE>(p, (hipStream_t)stream);
}

int dmp_edge_fwd_fused(const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *P, int64_t ldp,
                       int64_t num_nodes, const float *bias, const int32_t *selA, const int32_t *selB,
                       const float *coefE, int64_t E, int H, float slope, float *H1, int64_t ldh, void *stream) {
  if (E < 0 || num_nodes < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Z || !W || !P || !selA || !selB || !coefE || !H1 || ldz < H || ldw < 2 * H || ldp < 2 * H || ldh < H)
    return DMP_ERR_BAD_ARG;
  if (ldz % 4 || ldh % 4 || ldp % 4 || !aligned16(Z) || !aligned16(H1) || !aligned16(P) || (bias && !aligned16(bias)))
    return DMP_ERR_UNSUPPORTED;
  if (!fits32(num_nodes, ldp) || !fits32(E, 1) || !fits32(kSub, ldz) || !fits32(kSub, ldh)) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = Z; p.lda = ldz; p.B = W; p.ldb = ldw; p.bt = 0; p.C = H1; p.ldc = ldh; p.E = E; p.ldr = 128;
  p.T = P; p.ldt = ldp; p.num_nodes = num_nodes; p.idxA = selA; p.idxB = selB; p.rowscale = coefE; p.bias = bias;
  p.slope = slope;
  return launch_mfma<2, EPI_EDGE>(p, (hipStream_t)stream);
}

int dmp_row_mask_bits(const float *gate, int64_t E, uint32_t *mask, void *stream) {
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!gate || !mask) return DMP_ERR_BAD_ARG;
  row_mask_bits_k<<<(unsigned)((E + 255) / 256), 256, 0, (hipStream_t)stream>>>(gate, E, mask);
  return check_launch();
}

int dmp_row_mask_bits_jobs(const dmp_rowmask_job *jobs, int n, void *stream) {
  if (!jobs || n < 1 || n > DMP_ROWMASK_MAX_JOBS) return DMP_ERR_BAD_ARG;
  RowMaskJobs t{};
  int64_t most = 0;
  for (int j = 0; j < n; ++j) {
    if (jobs[j].R < 0 || (jobs[j].R > 0 && (!jobs[j].gate || !jobs[j].mask))) return DMP_ERR_BAD_ARG;
    t.gate[j] = jobs[j].gate; t.R[j] = jobs[j].R; t.mask[j] = jobs[j].mask;
    if (jobs[j].R > most) most = jobs[j].R;
  }
  if (most == 0) return DMP_OK;
  row_mask_bits_jobs_k<<<dim3((unsigned)((most + 255) / 256), (unsigned)n), 256, 0, (hipStream_t)stream>>>(t);
  return check_launch();
}

int dmp_row_mask_rows(const float *X, int64_t ldx, int K, int64_t R, uint32_t *mask, void *stream) {
  if (R < 0 || K < 0) return DMP_ERR_BAD_ARG;
  if (R == 0) return DMP_OK;
  if (!X || !mask || ldx < K) return DMP_ERR_BAD_ARG;
  row_mask_rows_k<<<(unsigned)((R + 255) / 256), 256, 0, (hipStream_t)stream>>>(X, ldx, K, R, mask);
  return check_launch();
}

int dmp_out_fwd_fused_rows(const float *Hin, int64_t ldh, const float *W2, int64_t ldw, const float *bias,
                           const float *gate, const uint32_t *rowmask, int dead_rows, const float *R, int64_t ldr, int64_t E, int H,
                           int w_in_out, float *out, int64_t ldo, void *stream) {
  if (rowmask && !gate) return DMP_ERR_BAD_ARG;
  if (dead_rows < 0 || dead_rows > 3 || (dead_rows && !rowmask)) return DMP_ERR_BAD_ARG;
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!Hin || !W2 || !out || ldh < H || ldw < H || ldo < H || (R && ldr < H)) return DMP_ERR_BAD_ARG;
  if (ldh % 4 || ldo % 4 || (R && ldr % 4) || !aligned16(Hin) || !aligned16(out) || (R && !aligned16(R)) ||
      (bias && !aligned16(bias)))
    return DMP_ERR_UNSUPPORTED;
  if (!fits32(E, 1) || !fits32(kSub, ldh) || !fits32(kSub, ldo) || (R && !fits32(kSub, ldr))) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = Hin; p.lda = ldh; p.B = W2; p.ldb = ldw;
  p.bt = w_in_out ? 0 : 1;  // nn.Linear weight [out, in]: B[k][j] = W2[j][k]; [in, out] (its transpose): B[k][j] = W2[k][j]
  p.C = out; p.ldc = ldo; p.E = E; p.bias = bias; p.rowscale = gate; p.R = R; p.ldr = R ? ldr : H; p.ldt = 2 * H;
  p.rowmask = g_variant == 1 ? nullptr : rowmask;
  p.dead_rows = dead_rows;
  return H == 128 ? launch_mfma<1, EPI_GATE_RES>(p, (hipStream_t)stream) : launch_mfma64<EPI_GATE_RES>(p, (hipStream_t)stream);
}

static int64_t dmp_mfma_partial_rows(int64_t E) {      // = the grid of the EPI_RELU_BWD_G launch (bwd_h1): one partial row per workgroup
  return g_variant == 1 ? 2 * (int64_t)pp_blocks(E) : (int64_t)wg_blocks(E, x6_on<1, EPI_RELU_BWD_G>() ? 2 : 3);
}
int64_t dmp_mfma_partial_rows_h(int64_t E, int H) { return H == 64 ? (int64_t)wg_blocks(E, kPerCU64) : dmp_mfma_partial_rows(E); }

void dmp_dev_set_mfma_variant(int v) { g_variant = v; }

int dmp_bwd_h1_fused_rows(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                          const float *coefE, const float *gate, const uint32_t *rowmask, int skip_dead_stores, int64_t E, int H,
                          float slope, float *dG, int64_t ldg, float *partial, float *partial_rows, void *stream) {
  if (rowmask && !gate) return DMP_ERR_BAD_ARG;
  if (skip_dead_stores && !rowmask) return DMP_ERR_BAD_ARG;
  if (partial_rows && (g_variant == 1 || !aligned16(partial_rows))) return DMP_ERR_UNSUPPORTED;
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0) return DMP_ERR_BAD_ARG;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  const bool both = coefE != nullptr;                      // with coefE: dG = [dPre | coefE dPre]; without: dPre alone
  if (!partial || (gate && both)) return DMP_ERR_BAD_ARG;
  if (E == 0)
    return hipMemsetAsync(partial, 0, sizeof(float) * H * (size_t)dmp_mfma_partial_rows_h(0, H), (hipStream_t)stream) == hipSuccess
               ? DMP_OK : DMP_ERR_HIP;
  if (!dO || !W2 || !H1 || !dG || ldo < H || ldw < H || ldh < H || ldg < (both ? 2 * H : H)) return DMP_ERR_BAD_ARG;
  if (ldo % 4 || ldh % 4 || ldg % 4 || !aligned16(dO) || !aligned16(H1) || !aligned16(dG) || !aligned16(partial))
    return DMP_ERR_UNSUPPORTED;
  if (!fits32(E, 1) || !fits32(kSub, ldo) || !fits32(kSub, ldh) || !fits32(kSub, ldg)) return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = dO; p.lda = ldo; p.B = W2; p.ldb = ldw; p.bt = 0;  // dH1 = dO @ W2, W2 [out, in] = B[k = out][j = in]
  p.C = dG; p.ldc = ldg; p.E = E; p.R = H1; p.ldr = ldh; p.rowscale = gate ? gate : coefE; p.gated = gate != nullptr;
  p.both = both; p.partial = partial; p.ldt = 2 * H; p.slope = slope; p.partialA = partial_rows;
  p.rowmask = g_variant == 1 ? nullptr : rowmask;
  p.dead_rows = skip_dead_stores ? 2 : 0;
  return H == 128 ? launch_mfma<1, EPI_RELU_BWD_G>(p, (hipStream_t)stream) : launch_mfma64<EPI_RELU_BWD_G>(p, (hipStream_t)stream);
}

int dmp_bwd_z_fused(const float *dPre, int64_t ldp, const float *W, int64_t ldw, const float *D, int64_t ldd,
                    int64_t num_nodes, const float *base, int64_t ldb, const float *coefE, const int32_t *dst,
                    const uint8_t *flag, float s0, float s1, int64_t E, int H, float *dZ, int64_t ldz, void *stream) {
  if (E < 0 || num_nodes < 0 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!dPre || !W || !D || !coefE || !dst || !dZ || ldp < H || ldw < 2 * H || ldd < 2 * H || ldz < H || (base && ldb < H))
    return DMP_ERR_BAD_ARG;
  if (ldp % 4 || ldz % 4 || ldd % 4 || (base && ldb % 4) || !aligned16(dPre) || !aligned16(dZ) || !aligned16(D) ||
      (base && !aligned16(base)))
    return DMP_ERR_UNSUPPORTED;
  if (!fits32(num_nodes, ldd) || !fits32(E, 1) || !fits32(kSub, ldp) || !fits32(kSub, ldz) || (base && !fits32(kSub, ldb)))
    return DMP_ERR_UNSUPPORTED;
  MfmaArgs p{};
  p.A = dPre; p.lda = ldp; p.B = W; p.ldb = ldw; p.bt = 2;  // panels = K-slices of W'^T
  p.C = dZ; p.ldc = ldz; p.E = E; p.R = base; p.ldr = base ? ldb : 128; p.rowscale = coefE; p.idxA = dst; p.flag = flag;
  p.T = D; p.ldt = ldd; p.num_nodes = num_nodes; p.s0 = s0; p.s1 = s1;
  return launch_mfma<2, EPI_DZ>(p, (hipStream_t)stream);
}


}  // extern "C"
