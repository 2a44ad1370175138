// Class-typed variants of the fused edge-chain MFMA kernels (gfx950, exact fp32, H = K = 128 or 64).
//
// The DMPLayer edge pre-activation is  Z[e] A' + c_e Z[e] B' + ...  with c_e = coef[dst e], a function
// of out_deg[dst e] only (dmpnn.py:144-151).  All edges of one degree class therefore share ONE weight
// matrix  W_g = A' + c_g B':  sorted by class and cut into 32-row tiles that never mix classes, the two
// [E,H]x[H,H] products of the reference collapse into one (and likewise dZ = dPre W_g^T in backward).
// Rows stay where they are: a tile is a list of 32 edge ids (`slot_edge`, -1 = padding); the kernel
// gathers its rows of the streamed operand by id and scatters its output rows by id.
//
// H = 64 (the reference's shipped hidden_dim, config.py:298-301): the same kernel with H / 32 = 2 waves per
// workgroup (one 32-column slice each), 32 k-steps per tile, twice the workgroups per CU.
//
// Structure as mfma_pp<1, EPI, 0> (dmp_mfma.hip): persistent 256-thread workgroups, 3 per CU, weight
// panel W_g in registers (rebuilt only when the workgroup's contiguous tile range crosses into the
// next class), buffer addressing with per-lane offsets, LDS-only barriers, three-deep prefetch
// (ids of tile k+3, rows of tile k+2, staging of tile k+1 while tile k is computed).
#include <cstdlib>
#include <type_traits>

#include "dmp_mfma_common.h"

namespace dmp {
namespace {

// TEPI_OUT / TEPI_H1: the second Linear of the edge MLP over the tiles of the KEPT edges of a 0 / 1 edge gate (dmp_class_tiles_gated;
// the classes play no part: one plain panel) -- out[e] = R[e] + A[e] W + bias and dPre[e] = act'(R[e]) (.) (A[e] W) with its column
// sums; what dmp_mfma.hip's one-panel kernels do over ALL rows in eid order (their products for the rows under a zero gate are zeros).
enum { TEPI_EDGE = 1, TEPI_DZ = 4, TEPI_REL = 8, TEPI_OUT = 16, TEPI_H1 = 32 };

struct TypedArgs {
  const float *A; int64_t lda;          // streamed operand [E,128], rows gathered by edge id
  const float *W; int64_t ldw;          // [128, ldw >= 256] = [A' | B'] row-major (the forward weight panel)
  int transposed;                       // 0: B_g[k][j] = W[k][j] + c W[k][128+j]   (edge_fwd)
                                        // 1: B_g[k][j] = W[j][k] + c W[j][128+k]   (dZ = dPre W_g^T)
  float *C; int64_t ldc;                // output [E,128], rows scattered by edge id
  int64_t E;
  const int32_t *slot_edge;             // [num_tiles_bound * 32] edge id per slot, -1 = padding
  const int32_t *slot_arow;             // row of A per slot (TEPI_EDGE / TEPI_DZ: the same array as slot_edge)
  int64_t rowsA;                        // rows of A (TEPI_EDGE / TEPI_DZ: E)
  int num_panels;                       // TEPI_REL: W is [num_panels][128][ldw], tile_scale holds the panel index (int bits)
  const float *tile_scale;              // [num_tiles_bound] c_g of the tile's class
  const int32_t *num_tiles;             // device scalar: tiles actually used
  const int32_t *idxA, *idxB;           // TEPI_EDGE: [E] node ids of the added / subtracted P rows; TEPI_DZ: idxA = dst
  const uint8_t *flag;                  // TEPI_DZ: is_reversed or NULL
  const float *T; int64_t ldt; int64_t num_nodes;   // gathered table P / D [N, >= 256]
  const float *bias;                    // TEPI_EDGE: [128] or NULL
  const float *R; int64_t ldr;          // TEPI_DZ: upstream gradient rows [E,128] or NULL
  const float *R2; int64_t ldr2;        // TEPI_OUT: a second addend row e (out = R + R2 + A W + bias) or NULL
  const int32_t *rmap; int64_t rowsR;   // TEPI_DZ: R is a [rowsR, ldr] table, row rmap[e] (< 0: zero) for edge e; NULL: row e
  float s0, s1;                         // TEPI_DZ: scale of the gathered term by flag
  float slope;                          // TEPI_EDGE / TEPI_H1: negative slope of the activation (0 = ReLU)
  int act;                              // TEPI_OUT: != 0: the activation (slope) is applied to the output row
  float *partial;                       // TEPI_H1: [gridDim.x, H] column sums of dPre per workgroup
  float *partialA;                      // TEPI_H1, optional: [gridDim.x, H] column sums of the fetched rows of A
  // TEPI_OUT with CODES: a K-extension of the product -- out[e] += codes[e] Wc (codes [E, ldcodes], Wc [kcodes <= 16, ldwc >= H]):
  // the residual rows of a FIRST layer are a label embedding z0 = codes W_e (basemodel.py:1393-1420), so they need not exist in HBM
  // (codes -> the code row of edge code_row0: edges below it take no codes term -- rows of ANOTHER table, which come in through R,
  // a [rowsR, ldr] array whose missing rows read as zeros)
  const float *codes; int64_t ldcodes; int kcodes; const float *Wc; int64_t ldwc; int64_t code_row0;
};

template <int H> struct TypedGeom {
  static constexpr int kThreads = 2 * H;                  // H / 32 waves: one 32-column slice each
  static constexpr int kStride = H + 4;                   // LDS row stride (floats): conflict-free ds_read_b128
  static constexpr int kQ = H / 4;                        // float4 per row; kThreads / kQ = 8 rows per load pass
  static constexpr int kPerCU = H == 128 ? 3 : 5;         // workgroups per CU the grid is sized for (H = 64: LDS-bound)
};

// X6: the products on the bf16 matrix pipe as six piece products per 16-deep k-group (dmp_mfma_common.h, "bf16x6":
// fp32-accurate, 2.67 x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32); the weight panel is kept in registers as
// pieces (96 instead of 64 VGPRs at H = 128: two workgroups per CU instead of three), the streamed operand is split as
// it is read from LDS.  !X6: exact fp32 MFMA (development / comparison switch, dmp_dev_set_exact_fp32).
// BIG: the streamed / scattered row arrays (A, C, R) are 4 GiB or larger: rows through 64-bit pointers (dmp_mfma_common.h).
template <int EPI, int H, bool X6, bool BIG = false, bool CODES = false>
__device__ __forceinline__ void typed_body(const TypedArgs &p) {
  static_assert(!CODES || (EPI == TEPI_OUT && X6 && !BIG && H == 128), "the K-extension: TEPI_OUT, bf16x6, H = 128, arrays below 4 GiB");
  constexpr int kStride = TypedGeom<H>::kStride, kQ = TypedGeom<H>::kQ, kHalf = H / 2, kSteps4 = H / 8;
  constexpr uint32_t kRowBytes = H * 4u;                  // second half of a gathered [.., 2H] row, second weight panel
  // TEPI_EDGE: one barrier per tile, the staging / requests of the next tiles in the MFMA shadow (tile k in As[k & 1]).
  // TEPI_DZ keeps the two-barrier order (MFMA phase | stage + requests | epilogue): its epilogue waits for streamed
  // base rows, and the extra phase between their request and their use hides them better (measured: +3 % otherwise).
  // TEPI_REL (relation-typed product, rgcn.py:98-123): plain panel W[type of the tile], rows of A gathered through
  // slot_arow, the output row scaled by idxA-as-float[row] (the edge normaliser) -- no epilogue operands.
  constexpr bool kPipelined = EPI != TEPI_DZ || X6;   // X6: one code path (the deeper prefetch lives in the pipelined form)
  // the staged tile: fp32 rows of kStride floats; X6: three bf16 planes (hi | mid | lo pieces, split once by the staging
  // thread), rows of kStrideD dwords (H bf16 + 8 of padding: 68 dwords at H = 128, = 4 mod 64 as the fp32 stride:
  // conflict-free ds_read_b128 of 8 consecutive k per lane)
  constexpr int kStrideD = (H + 8) / 2, kPlane = kSub * kStrideD;
  constexpr int kTileWords = X6 ? 3 * kPlane : kSub * kStride;
  __shared__ __attribute__((aligned(16))) float As[kPipelined ? 2 : 1][kTileWords];
  __shared__ float Cs[H / 32][32 * kScrStride];
  // CODES: the tile's code rows as bf16 pieces, [buffer][plane][row * 8 dwords] (16 codes per row; no padding: three 16-byte reads per
  // tile and wave, a 4-way bank conflict on them costs nothing)
  __shared__ __attribute__((aligned(16))) uint32_t Cx[CODES ? 2 : 1][3][CODES ? kSub * 8 : 4];
  __shared__ uint32_t rowA[3][kSub], rowB[3][kSub], rowC[3][kSub], rowR[3][kSub];   // [tile % 3][row] row INDICES (-1: none): tile k's are read
                                                                                     // (epilogue) while tile k+2's are written
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, cs = wave, gtid = threadIdx.x;
  const int col = 32 * cs + li;
  float *scr = Cs[wave];
  const int lrow = lane >> 3, c4 = 32 * cs + (lane & 7) * 4;
  const uint32_t colA = (uint32_t)(gtid % kQ) * 16u, col4 = (uint32_t)c4 * 4u;
  constexpr bool kPlainPanel = EPI == TEPI_REL || EPI == TEPI_OUT || EPI == TEPI_H1;   // no class term in the panel
  constexpr bool kRowsOnly = EPI == TEPI_OUT || EPI == TEPI_H1;                         // epilogue operand: row e of R, nothing gathered
  constexpr bool kFixedPanel = kRowsOnly;                                               // one panel for the whole launch (no class term, no type)
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((EPI == TEPI_EDGE || EPI == TEPI_OUT) && p.bias) bias4 = *reinterpret_cast<const float4 *>(p.bias + c4);
  const float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.slope)));   // SGPR
  const bool act_out = EPI == TEPI_OUT && __builtin_amdgcn_readfirstlane(p.act) != 0;
  const bool has_r2 = EPI == TEPI_OUT && p.R2 != nullptr;                                   // wave-uniform (a kernel argument)

  const uint32_t rows4 = (uint32_t)(p.E * 4);
  // rows of the big arrays are addressed by INDEX (structured descriptors: any array size, dmp_mfma_common.h)
  const srsrc_t rs_A = make_srsrc(p.A, p.lda, p.rowsA);
  const srsrc_t rs_C = make_srsrc(p.C, p.ldc, p.E);
  const srsrc_t rs_R = make_srsrc(p.R, p.ldr, (p.rmap || CODES) ? p.rowsR : p.E);     // (CODES: rows >= rowsR read as zeros)
  const srsrc_t rs_R2 = make_srsrc(p.R2, p.R2 ? p.ldr2 : (int64_t)H, p.R2 ? p.E : 0);        // (no second addend: zero records)
  const rsrc_t rs_rmap = make_rsrc(p.rmap, p.rmap ? rows4 : 0u);
  const srsrc_t rs_T = make_srsrc(p.T, p.ldt, p.num_nodes);
  const rsrc_t rs_idxA = make_rsrc(p.idxA, p.idxA ? rows4 : 0u);
  const rsrc_t rs_idxB = make_rsrc(p.idxB, p.idxB ? rows4 : 0u);
  const rsrc_t rs_flag = make_rsrc(p.flag, p.flag ? (uint32_t)p.E : 0u);
  constexpr uint32_t kOOB = 0xFFFFF000u;                                   // byte offset beyond any descriptor range
  const srsrc_t rs_codes = make_srsrc(CODES ? p.codes : nullptr, CODES ? p.ldcodes : (int64_t)4, CODES ? p.E - p.code_row0 : 0);  // (codes: row code_row0)
  const int cq = gtid & 3, crow = gtid >> 2;                               // CODES: threads < 128 stage codes 4 cq .. 4 cq + 3 of row crow
  const bool cstage = CODES && gtid < 4 * kSub && 4 * cq < (int)p.ldcodes && 4 * cq < 16;
  Split8 BX;                                                               // CODES: the extension's panel fragment: BX[j] = Wc[8 h + j][col]

  // this workgroup's contiguous tile range (tiles are sorted by class)
  const int ntiles = __builtin_amdgcn_readfirstlane(*p.num_tiles);
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  const rsrc_t rs_slot = make_rsrc(p.slot_edge, (uint32_t)ntiles * (kSub * 4u));   // past the end: reads 0 (guarded below)
  const rsrc_t rs_slotA = make_rsrc(p.slot_arow, (uint32_t)ntiles * (kSub * 4u));

  // W_g fragments: b[s] = B_g[s + (H/2) h][col]; X6: the same values as bf16 pieces, fragment g = k-steps 8g .. 8g+7
  constexpr int kGroups = kHalf / 8;
  float b[X6 ? 1 : kHalf];
  Split8 B6[X6 ? kGroups : 1];
  const rsrc_t rs_W = make_rsrc(p.W, (uint32_t)((EPI == TEPI_REL ? p.num_panels : 1) * H * p.ldw * 4));
  const uint32_t w_first = (uint32_t)(p.transposed ? (int64_t)col * p.ldw + kHalf * h : (int64_t)kHalf * h * p.ldw + col) * 4u;
  const uint32_t w_step = __builtin_amdgcn_readfirstlane((int)(p.transposed ? 4 : p.ldw * 4));  // bytes from k to k+1
  auto load_panel = [&](float c) {
    if (CODES) {                          // rows 8 h .. 8 h + 7 of Wc (rows >= kcodes: beyond the descriptor's range -> zeros)
      const rsrc_t rs_Wc = make_rsrc(p.Wc, (uint32_t)(p.kcodes * p.ldwc * 4));
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        w[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_Wc, (int)(((int64_t)(8 * h + j) * p.ldwc + col) * 4), 0, 0));
      split8(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), BX);
    }
    // The offsets hang off a value the optimiser cannot see through: otherwise it hoists the H
    // address computations out of the tile loop and keeps them live in registers across it.
    uint32_t off;
    asm volatile("v_mov_b32 %0, %1" : "=v"(off) : "v"(w_first));
    if (EPI == TEPI_REL) off += (uint32_t)__float_as_int(c) * (uint32_t)(H * p.ldw * 4);   // panel of the tile's type
    // kPB values per round trip: a plain panel (one load per value) comes in rounds of 32 -- with a few tiles per workgroup
    // (the node side: ~1 k tiles over the grid) the panel's dependent round trips ARE the launch
    constexpr int kPB = (kPlainPanel && X6 && kHalf % 32 == 0) ? 32 : 8;
#pragma unroll
    for (int s0 = 0; s0 < kHalf; s0 += kPB) {
      float w0[kPB], w1[kPlainPanel ? 1 : kPB];
#pragma unroll
      for (int j = 0; j < kPB; ++j) {
        w0[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off, (int)((s0 + j) * w_step), 0));
        if (!kPlainPanel) w1[kPlainPanel ? 0 : j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off + (int)kRowBytes, (int)((s0 + j) * w_step), 0));
      }
#pragma unroll
      for (int q = 0; q < kPB; q += 8) {
        if (X6) {
          float w[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) w[j] = kPlainPanel ? w0[q + j] : w0[q + j] + c * w1[kPlainPanel ? 0 : q + j];
          split8(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), B6[X6 ? (s0 + q) / 8 : 0]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) b[X6 ? 0 : s0 + q + j] = kPlainPanel ? w0[q + j] : w0[q + j] + c * w1[kPlainPanel ? 0 : q + j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prefetch state.  !X6: three deep (ids of tile k+3, rows of tile k+2 in registers, tile k+1 staged while tile k
  // is computed).  X6: one tile deeper -- the kernel then runs at two workgroups per CU, and what it sustains is set by
  // the bytes it keeps in flight, not by its matrix work: rows of tiles k+2 AND k+3 in two register sets (tile t in set
  // t & 1), ids of tile k+4.
  constexpr bool kDeep = false;                   // the deeper row prefetch (two register sets): measured, no gain
  constexpr int NSET = kDeep ? 2 : 1, kAhead = kDeep ? 3 : 2;
  int id_rows[kSubLoads];       // edge ids of the 4 rows this thread loads (rows gtid/32 + 8m)
  int id_own = -1;              // edge id of row gtid (threads < 32): per-row scalars
  float4 pre[NSET][kSubLoads];
  uint32_t pre_a[NSET] = {}, pre_b[NSET] = {};
  int pre_r[NSET] = {};              // TEPI_DZ: row of R for this thread's row (threads < 32)
  int id_c = -1;                     // CODES: slot id of row gtid / 4 (threads < 128)
  float4 pre_c = make_float4(0.f, 0.f, 0.f, 0.f);
  std::integral_constant<int, 0> set0;
  std::integral_constant<int, NSET - 1> set1;
  auto load_ids = [&](int k) {                            // ids of tile lo + k (-1 past the end)
    const bool ok = k < mine;
    const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      id_rows[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slotA, ((gtid / kQ) + 8 * m) * 4, (int)so, 0) : -1;
    if (gtid < kSub) id_own = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, gtid * 4, (int)so, 0) : -1;
    if (CODES && gtid < 4 * kSub) id_c = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slotA, crow * 4, (int)so, 0) : -1;
  };
  int own_staged[NSET];
#pragma unroll
  for (int q = 0; q < NSET; ++q) own_staged[q] = -1;
  auto load_row = [&](auto set, int m) {                    // into register set S
    constexpr int S = decltype(set)::value;
    pre[S][m] = row_load4<BIG>(rs_A, p.A, p.lda, id_rows[m], colA);   // id -1 (padding, past the end): zeros
  };
  auto load_row_scalars = [&](auto set) {
    constexpr int S = decltype(set)::value;
    if (gtid < kSub) {
      const uint32_t eo = id_own >= 0 ? (uint32_t)id_own * 4u : kOOB;
      if (!kRowsOnly) pre_a[S] = __builtin_amdgcn_raw_buffer_load_b32(rs_idxA, (int)eo, 0, 0);
      if (EPI == TEPI_EDGE) pre_b[S] = __builtin_amdgcn_raw_buffer_load_b32(rs_idxB, (int)eo, 0, 0);
      if (EPI == TEPI_DZ) pre_b[S] = __builtin_amdgcn_raw_buffer_load_b8(rs_flag, id_own >= 0 ? id_own : (int)kOOB, 0, 0);
      if (EPI == TEPI_DZ) pre_r[S] = p.rmap ? (id_own >= 0 ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_rmap, (int)eo, 0, 0) : -1) : id_own;
      own_staged[S] = id_own;
    }
    if (CODES) pre_c = cstage ? sbuf_load4(rs_codes, id_c - (int)p.code_row0, (uint32_t)cq * 16u) : make_float4(0.f, 0.f, 0.f, 0.f);   // (id -1, rows below code_row0: zeros)
  };
  auto load_rows = [&](auto set) {                          // rows + per-row scalars of the tile whose ids are loaded
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(set, m);
    load_row_scalars(set);
  };
  float4 csA = make_float4(0.f, 0.f, 0.f, 0.f);            // TEPI_H1, partialA: this thread's 4 columns of the rows it stages
  auto stage_row = [&](int buf, auto set, int m) {           // register set S -> LDS, one of the thread's four row pieces
    constexpr int S = decltype(set)::value;
    if (EPI == TEPI_H1 && p.partialA) { csA.x += pre[S][m].x; csA.y += pre[S][m].y; csA.z += pre[S][m].z; csA.w += pre[S][m].w; }
    if (X6) {
      uint2 ph, pm, pl;
      split_pair(pre[S][m].x, pre[S][m].y, ph.x, pm.x, pl.x);
      split_pair(pre[S][m].z, pre[S][m].w, ph.y, pm.y, pl.y);
      uint32_t *q = reinterpret_cast<uint32_t *>(&As[buf][0]) + ((gtid / kQ) + 8 * m) * kStrideD + (gtid % kQ) * 2;
      *reinterpret_cast<uint2 *>(q) = ph;
      *reinterpret_cast<uint2 *>(q + kPlane) = pm;
      *reinterpret_cast<uint2 *>(q + 2 * kPlane) = pl;
    } else {
      *reinterpret_cast<float4 *>(&As[buf][((gtid / kQ) + 8 * m) * kStride + (gtid % kQ) * 4]) = pre[S][m];
    }
  };
  auto stage_scalars = [&](int par, auto set) {            // per-row byte offsets of the staged tile (threads < 32)
    constexpr int S = decltype(set)::value;
    if (gtid < kSub) {
      const bool ok = own_staged[S] >= 0;
      constexpr uint32_t kNone = 0xFFFFFFFFu;            // row index beyond any descriptor
      uint32_t a = kNone, bb = kNone;
      if (EPI == TEPI_EDGE) {
        if (ok) { a = pre_a[S]; bb = pre_b[S]; }          // the two gathered nodes
      } else if (EPI == TEPI_REL) {
        a = p.idxA ? pre_a[S] : __float_as_uint(1.f);     // the row's scale (float bits)
      } else if (kRowsOnly) {
      } else {
        bb = pre_b[S];                                    // flag: selects the half of the gathered row and the sign
        if (ok) a = pre_a[S];
      }
      rowA[par][gtid] = a; rowB[par][gtid] = bb;
      rowC[par][gtid] = ok ? (uint32_t)own_staged[S] : kNone;
      rowR[par][gtid] = (ok && p.R && (EPI != TEPI_DZ || pre_r[S] >= 0)) ? (uint32_t)(EPI == TEPI_DZ ? pre_r[S] : own_staged[S]) : kNone;
    }
  };
  auto stage_codes = [&](int buf) {                        // CODES: the tile's code rows as three bf16 planes (threads < 128)
    if (CODES && gtid < 4 * kSub) {
      uint2 ph, pm, pl;
      const int kc = (int)p.kcodes - 4 * cq;                 // (a row's padding columns may hold anything: zeros from here on)
      split_pair(kc > 0 ? pre_c.x : 0.f, kc > 1 ? pre_c.y : 0.f, ph.x, pm.x, pl.x);
      split_pair(kc > 2 ? pre_c.z : 0.f, kc > 3 ? pre_c.w : 0.f, ph.y, pm.y, pl.y);
      *reinterpret_cast<uint2 *>(&Cx[CODES ? buf : 0][0][crow * 8 + cq * 2]) = ph;
      *reinterpret_cast<uint2 *>(&Cx[CODES ? buf : 0][1][crow * 8 + cq * 2]) = pm;
      *reinterpret_cast<uint2 *>(&Cx[CODES ? buf : 0][2][crow * 8 + cq * 2]) = pl;
    }
  };
  auto stage = [&](int buf, int par, auto set) {
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(buf, set, m);
    stage_scalars(par, set);
    stage_codes(buf);
  };

  f32x16 acc;
  float4 g0[4], g1[4];
  auto fetch_operand = [&](int par, int k) {               // rows 8k + lrow of the tile
    if (EPI == TEPI_REL) return;
    const int rr = 8 * k + lrow;
    if (EPI == TEPI_EDGE) {
      g0[k] = sbuf_load4(rs_T, (int)rowA[par][rr], col4);
      g1[k] = sbuf_load4(rs_T, (int)rowB[par][rr], col4 + kRowBytes);
    } else if (kRowsOnly) {
      g1[k] = row_load4<BIG>(rs_R, p.R, p.ldr, (int)rowR[par][rr], col4);
      if (EPI == TEPI_OUT && !BIG && has_r2) g0[k] = sbuf_load4(rs_R2, (int)rowC[par][rr], col4);   // (padding rows: zeros)
    } else {
      g0[k] = sbuf_load4(rs_T, (int)rowA[par][rr], col4 + (rowB[par][rr] ? kRowBytes : 0u));
      g1[k] = row_load4<BIG>(rs_R, p.R, p.ldr, (int)rowR[par][rr], col4);
    }
  };
  // X6: the tile's product on the bf16 pipe.  Per 16-deep k-group g: six MFMAs on the three piece fragments of A (this
  // lane's 8 consecutive k = kHalf h + 8g .. of row li: one ds_read_b128 per plane, requested one group ahead) and the
  // panel fragment B6[g]; the caller's 16 actions (requests / staging of the neighbouring tiles) are spread between them.
  auto x6_tile = [&](const float *tile, auto &&action) {
    constexpr int kPer = 16 / kGroups;                      // actions per group: 2 (H = 128) or 4 (H = 64)
    const uint32_t *ar = reinterpret_cast<const uint32_t *>(tile) + li * kStrideD + (kHalf / 2) * h;
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
      int done = 0;
      auto after = [&](int n) {                             // the actions due after the n-th MFMA of the group
#pragma unroll
        for (int q = 0; q < kPer; ++q)
          if (6 * (q + 1) / kPer == n) { action(kPer * g + q); ++done; }
        __builtin_amdgcn_sched_barrier(0);
      };
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bb.hi.v, acc, 0, 0, 0); after(1);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.lo.v, acc, 0, 0, 0); after(2);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.mid.v, acc, 0, 0, 0); after(3);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.hi.v, acc, 0, 0, 0); after(4);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.mid.v, acc, 0, 0, 0); after(5);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.hi.v, acc, 0, 0, 0); after(6);
      (void)done;
      ah = nh; am = nm; al = nl;
    }
    if (CODES) {                                             // + codes Wc: one more 16-deep k-group (this lane: codes 8 h .. 8 h + 7 of row li)
      const int cb = (tile == &As[0][0]) ? 0 : 1;
      Frag8 ch, cm, cl;
      ch.v = *reinterpret_cast<const bf16x8 *>(&Cx[CODES ? cb : 0][0][li * 8 + 4 * h]);
      cm.v = *reinterpret_cast<const bf16x8 *>(&Cx[CODES ? cb : 0][1][li * 8 + 4 * h]);
      cl.v = *reinterpret_cast<const bf16x8 *>(&Cx[CODES ? cb : 0][2][li * 8 + 4 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl.v, BX.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch.v, BX.lo.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cm.v, BX.mid.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cm.v, BX.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch.v, BX.mid.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch.v, BX.hi.v, acc, 0, 0, 0);
    }
  };
  // The MFMA phase, with the epilogue operand requests of the same tile in its shadow: two of the eight loads
  // (and the LDS reads of their row offsets) after each of the first four MFMA groups.
  auto compute = [&](int par) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *arow = &As[0][li * kStride + kHalf * h];
    if (X6) {
      x6_tile(&As[0][0], [&](int i) { if (i < 4) fetch_operand(par, i); });
      return;
    }
    float4 a4 = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
    for (int s4 = 0; s4 < kSteps4; ++s4) {
      float4 an = a4;
      if (s4 + 1 < kSteps4) an = *reinterpret_cast<const float4 *>(arow + 4 * (s4 + 1));
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[X6 ? 0 : 4 * s4 + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[X6 ? 0 : 4 * s4 + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[X6 ? 0 : 4 * s4 + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[X6 ? 0 : 4 * s4 + 3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s4 < 4) fetch_operand(par, s4);
      a4 = an;
    }
  };
  // One tile: the MFMAs of tile k (As[k & 1]) with, in the shadow of its 16 MFMA groups: the tile's epilogue operand
  // requests (groups 0-3), the staging of tile k+1 into the other buffer (groups 4-8), the row requests of tile
  // k+2 (groups 9-13) and the id requests of tile k+3 (group 14).  par3 = k % 3 indexes the per-row offset arrays.
  // H = 64 has 8 MFMA groups: two of these 15 actions after each.
  // `nset`: the register set of tile k+1 (X6: (k + 1) & 1, a compile-time phase; else 0) -- staged now, then refilled
  // with the rows of tile k + kAhead
  auto shadow = [&](int i, int k, int par3, int buf, int nxt3, auto nset) {
    if (i < 4) { if (!X6) fetch_operand(par3, i); }       // X6: requested by the previous tile's epilogue (below)
    else if (i < 8) stage_row(buf ^ 1, nset, i - 4);
    else if (i == 8) { stage_scalars(nxt3, nset); stage_codes(buf ^ 1); }
    else if (i < 13) load_row(nset, i - 9);
    else if (i == 13) load_row_scalars(nset);
    else if (i == 14) load_ids(k + kAhead + 1);
  };
  auto tile_step = [&](int k, int par3, auto nset) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int buf = k & 1, nxt3 = par3 == 2 ? 0 : par3 + 1;
    const float *arow = &As[buf][li * kStride + kHalf * h];
    if (X6) {
      x6_tile(&As[buf][0], [&](int i) { shadow(i, k, par3, buf, nxt3, nset); });
      return;
    }
    float4 a4 = *reinterpret_cast<const float4 *>(arow);
#pragma unroll
    for (int s4 = 0; s4 < kSteps4; ++s4) {
      float4 an = a4;
      if (s4 + 1 < kSteps4) an = *reinterpret_cast<const float4 *>(arow + 4 * (s4 + 1));
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[X6 ? 0 : 4 * s4 + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[X6 ? 0 : 4 * s4 + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[X6 ? 0 : 4 * s4 + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[X6 ? 0 : 4 * s4 + 3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kSteps4 == 16) shadow(s4, k, par3, buf, nxt3, nset);
      else { shadow(2 * s4, k, par3, buf, nxt3, nset); shadow(2 * s4 + 1, k, par3, buf, nxt3, nset); }
      a4 = an;
    }
  };
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);         // TEPI_H1: this lane's 4 columns of dPre
  auto epilogue = [&](int par) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * h) * kScrStride + li] = acc[r];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = 8 * k + lrow;
      float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
      if (EPI == TEPI_EDGE) {
        v.x = act_fwd((v.x + (g0[k].x - g1[k].x)) + bias4.x, slope);
        v.y = act_fwd((v.y + (g0[k].y - g1[k].y)) + bias4.y, slope);
        v.z = act_fwd((v.z + (g0[k].z - g1[k].z)) + bias4.z, slope);
        v.w = act_fwd((v.w + (g0[k].w - g1[k].w)) + bias4.w, slope);
      } else if (EPI == TEPI_REL) {
        const float sg = __uint_as_float(rowA[par][rr]);
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;
      } else if (EPI == TEPI_OUT) {                          // as dmp_mfma.hip's EPI_GATE_RES with gate 1: (product + bias) + residual
        v.x = (v.x + bias4.x) + g1[k].x; v.y = (v.y + bias4.y) + g1[k].y;
        v.z = (v.z + bias4.z) + g1[k].z; v.w = (v.w + bias4.w) + g1[k].w;
        if (!BIG && has_r2) { v.x += g0[k].x; v.y += g0[k].y; v.z += g0[k].z; v.w += g0[k].w; }       // the second addend
        if (act_out) { v.x = act_fwd(v.x, slope); v.y = act_fwd(v.y, slope); v.z = act_fwd(v.z, slope); v.w = act_fwd(v.w, slope); }
      } else if (EPI == TEPI_H1) {                           // padding rows: the activation reads as 0, the product is 0 -> 0
        v.x = act_bwd(g1[k].x, v.x, slope); v.y = act_bwd(g1[k].y, v.y, slope);
        v.z = act_bwd(g1[k].z, v.z, slope); v.w = act_bwd(g1[k].w, v.w, slope);
        colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
      } else {
        const float sg = rowB[par][rr] ? p.s1 : p.s0;
        v.x += g1[k].x + sg * g0[k].x; v.y += g1[k].y + sg * g0[k].y;
        v.z += g1[k].z + sg * g0[k].z; v.w += g1[k].w + sg * g0[k].w;
      }
      row_store4<BIG>(v, rs_C, p.C, p.ldc, (int)rowC[par][rr], col4);   // padding rows: dropped
      // X6: this chunk's operand registers are free -- request the NEXT tile's rows into them now (its row offsets were
      // staged during this tile's MFMA phase, before the barrier).  With the short bf16 MFMA phase, operands requested at
      // the start of their own tile's phase (the f32 form) arrive after the epilogue wants them: this way they have a
      // whole iteration.  Past the last tile the offsets are out of range: the loads return zeros.
      if (X6) fetch_operand(par == 2 ? 0 : par + 1, k);
    }
  };

  if (mine == 0) {
    if (EPI == TEPI_H1) {                                    // this workgroup's partial rows: zeros
      if (gtid < kQ) {
        *reinterpret_cast<float4 *>(p.partial + (int64_t)blockIdx.x * H + gtid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.partialA) *reinterpret_cast<float4 *>(p.partialA + (int64_t)blockIdx.x * H + gtid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  if (kDeep) {
    load_ids(0);
    load_rows(set0);           // tile 0 -> set 0
    load_ids(1);
    load_rows(set1);           // tile 1 -> set 1
    load_ids(2);
    stage(0, 0, set0);         // waits for tile 0's rows only (the counter is in order)
    load_rows(set0);           // tile 2 -> set 0
    load_ids(3);
  } else {
    load_ids(0);
    load_rows(set0);           // tile 0
    load_ids(1);
    // a panel that does not depend on the tile's class (TEPI_OUT / TEPI_H1) is requested HERE, behind tile 0's rows and ahead
    // of the wait for them: its round trip runs beside theirs instead of after it -- with a few tiles per workgroup (the node
    // side of a layer: ~1 k tiles over the grid) the launch is little more than these two latencies
    if (kFixedPanel) load_panel(0.f);
    stage(0, 0, set0);
    load_rows(set0);           // tile 1
    load_ids(2);
  }
  lds_barrier();
  if (X6) {
#pragma unroll
    for (int c = 0; c < 4; ++c) fetch_operand(0, c);        // tile 0's epilogue operands (later tiles: by the epilogue before)
  }
  // The class structure of the range is read 64 tiles at a time: lane l keeps the coefficient of tile 64c + l,
  // bit l of `starts` says "tile 64c + l begins a new class".  The hot loop then tests a scalar bit -- a
  // per-tile coefficient load in its condition was a vector load the loop had to wait for with vmcnt(0),
  // i.e. behind the row prefetch and the epilogue's stores it had just issued.
  int k = 0, par3 = 0;
  float sv = 0.f, c_have = 0.f;
  bool have_panel = false;
  unsigned long long starts = 0;
  while (k < mine) {
    if ((k & 63) == 0) {
      const float last = __shfl(sv, 63);
      sv = k + lane < mine ? p.tile_scale[lo + k + lane] : 0.f;
      float up = __shfl_up(sv, 1);
      if (lane == 0) up = k > 0 ? last : sv;
      starts = __ballot(k + lane < mine && __float_as_uint(sv) != __float_as_uint(up));
    }
    // tiles of one degree class: W_g is built once per class segment of this workgroup's range
    const float c = __shfl(sv, k & 63);
    if (!kFixedPanel && (!have_panel || __float_as_uint(c) != __float_as_uint(c_have))) {
      load_panel(c);
      c_have = c;
      have_panel = true;
    }
    do {
      if (kPipelined) {
        if (kDeep && (k & 1) == 0) tile_step(k, par3, set1);   // tile k+1 sits in set (k + 1) & 1
        else tile_step(k, par3, set0);
        lds_barrier();           // tile k+1 is staged for everyone, everyone is done with tile k's rows
      } else {
        compute(par3);
        lds_barrier();           // every wave is done reading this tile's rows
        stage(0, par3 == 2 ? 0 : par3 + 1, set0);   // tile k+1 (rows were requested one iteration ago)
        load_rows(set0);         // tile k+2 (ids were requested one iteration ago)
        load_ids(k + 3);
        lds_barrier();           // tile k+1 is in LDS for everyone
      }
      epilogue(par3);
      ++k;
      par3 = par3 == 2 ? 0 : par3 + 1;
    } while (k < mine && (k & 63) != 0 && ((starts >> (k & 63)) & 1ull) == 0);
  }
  if (EPI == TEPI_H1) {
    // lanes with equal (lane & 7) hold the same 4 columns for 8 different rows: fixed-order xor-shuffle tree over lane >> 3,
    // then one partial row per workgroup (dmp_mfma.hip's EPI_RELU_BWD_G does the same)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      colsum.x += __shfl_xor(colsum.x, off, 64); colsum.y += __shfl_xor(colsum.y, off, 64);
      colsum.z += __shfl_xor(colsum.z, off, 64); colsum.w += __shfl_xor(colsum.w, off, 64);
    }
    if (lane < 8) *reinterpret_cast<float4 *>(p.partial + (int64_t)blockIdx.x * H + 32 * cs + lane * 4) = colsum;
    if (p.partialA) {
      // the 8 threads that staged the same 4 columns (gtid % kQ), added in a fixed order through the tile buffer
      lds_barrier();
      float *red = reinterpret_cast<float *>(&As[0][0]);
      *reinterpret_cast<float4 *>(&red[(gtid / kQ) * H + (gtid % kQ) * 4]) = csA;
      lds_barrier();
      if (gtid < kQ) {
        float4 t = *reinterpret_cast<const float4 *>(&red[gtid * 4]);
#pragma unroll
        for (int g = 1; g < TypedGeom<H>::kThreads / kQ; ++g) {
          const float4 u = *reinterpret_cast<const float4 *>(&red[g * H + gtid * 4]);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4 *>(p.partialA + (int64_t)blockIdx.x * H + gtid * 4) = t;
      }
    }
  }
}

template <int EPI, int H, bool X6, bool BIG = false>
__global__ __launch_bounds__(TypedGeom<H>::kThreads, (X6) ? 2 : 3) void mfma_typed(TypedArgs p) {
  typed_body<EPI, H, X6, BIG>(p);
}
// TEPI_OUT with the K-extension (the first layer's residual rows from their label codes)
__global__ __launch_bounds__(TypedGeom<128>::kThreads, 2) void mfma_typed_codes(TypedArgs p) {
  typed_body<TEPI_OUT, 128, true, false, true>(p);
}

// Several TEPI_OUT products over the SAME tile list in one launch (the node side of a layer: up to kTypedJobs independent
// 128-wide block products over the kept nodes' tiles): grid.y = the job, grid.x = the workgroups of one job -- every workgroup
// loads ONE panel and walks a contiguous range of the shared tile list.
constexpr int kTypedJobs = 6;
struct TypedJobs { TypedArgs job[kTypedJobs]; };
template <int H>
__global__ __launch_bounds__(TypedGeom<H>::kThreads, 2) void mfma_typed_jobs(TypedJobs js) {
  typed_body<TEPI_OUT, H, true, false>(js.job[blockIdx.y]);
}

inline unsigned typed_blocks(int64_t tiles_bound, int per_cu = 3) {
  static const int dev_per_cu = [] { const char *e = getenv("DMP_DEV_TYPED_PER_CU"); return e ? atoi(e) : 0; }();   // development probe
  if (dev_per_cu > 0) per_cu = dev_per_cu;
  const int64_t cap = 256 * per_cu;
  return (unsigned)(tiles_bound < cap ? (tiles_bound > 0 ? tiles_bound : 1) : cap);
}
inline bool fits32(int64_t rows, int64_t ld) { return rows * ld * 4 < ((int64_t)1 << 32) - 8192; }


template <int EPI, int H>
inline int launch_typed(const TypedArgs &p, int64_t tiles_bound, hipStream_t st) {
  const bool big = !fits4g(p.rowsA, p.lda) || !fits4g(p.E, p.ldc) || (p.R && !fits4g(p.rmap ? p.rowsR : p.E, p.ldr));
  if (big) {                      // 64-bit row addressing: the bf16x6 form only
    if (g_exact_fp32) return DMP_ERR_UNSUPPORTED;
    mfma_typed<EPI, H, true, true><<<typed_blocks(tiles_bound, H == 128 ? 2 : 4), TypedGeom<H>::kThreads, 0, st>>>(p);
    return check_launch();
  }
  if (g_exact_fp32)
    mfma_typed<EPI, H, false><<<typed_blocks(tiles_bound, H == 128 ? 3 : TypedGeom<64>::kPerCU), TypedGeom<H>::kThreads, 0, st>>>(p);
  else
    mfma_typed<EPI, H, true><<<typed_blocks(tiles_bound, H == 128 ? 2 : 4), TypedGeom<H>::kThreads, 0, st>>>(p);
  return check_launch();
}

// a tile slot list whose gated-out edges read as padding (dmp_mask_slots)
__global__ __launch_bounds__(256) void mask_slots_k(const int32_t *__restrict__ slot, int64_t n, const float *__restrict__ gate,
                                                    int64_t E, int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t e = slot[i];
  out[i] = (e >= 0 && e < E && gate[e] != 0.f) ? e : -1;
}

}  // namespace
}  // namespace dmp

namespace dmp { int g_exact_fp32 = 0; }   // development switch: 1 = the f32-input MFMA (exact fp32) instead of the bf16x6 products

using namespace dmp;

extern "C" {

void dmp_dev_set_exact_fp32(int on) { g_exact_fp32 = on ? 1 : 0; }
int dmp_dev_get_exact_fp32(void) { return g_exact_fp32; }

int dmp_edge_fwd_typed(const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *P, int64_t ldp,
                       int64_t num_nodes, const float *bias, const int32_t *selA, const int32_t *selB,
                       const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles,
                       int64_t tiles_bound, int64_t E, int H, float slope, float *H1, int64_t ldh, void *stream) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || num_nodes < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Z || !W || !P || !selA || !selB || !slot_edge || !tile_scale || !num_tiles || !H1 || ldz < H || ldw < 2 * H ||
      ldp < 2 * H || ldh < H)
    return DMP_ERR_BAD_ARG;
  if (ldz % 4 || ldh % 4 || ldp % 4 || !aligned16(Z) || !aligned16(H1) || !aligned16(P) || (bias && !aligned16(bias)))
    return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldp) || !stride_ok(ldz) || !stride_ok(ldh) || E >= ((int64_t)1 << 30) || num_nodes >= ((int64_t)1 << 31) ||
      !fits32(tiles_bound * kSub, 1))
    return DMP_ERR_UNSUPPORTED;
  TypedArgs p{};
  p.A = Z; p.lda = ldz; p.W = W; p.ldw = ldw; p.transposed = 0; p.C = H1; p.ldc = ldh; p.E = E;
  p.slot_edge = slot_edge; p.slot_arow = slot_edge; p.rowsA = E; p.tile_scale = tile_scale; p.num_tiles = num_tiles;
  p.idxA = selA; p.idxB = selB; p.T = P; p.ldt = ldp; p.num_nodes = num_nodes; p.bias = bias; p.slope = slope;
  return H == 128 ? launch_typed<TEPI_EDGE, 128>(p, tiles_bound, (hipStream_t)stream)
                  : launch_typed<TEPI_EDGE, 64>(p, tiles_bound, (hipStream_t)stream);
}

// rows of the partial arrays of dmp_bwd_h1_typed = the grid of its launch (launch_typed)
int64_t dmp_typed_partial_rows(int64_t tiles_bound, int H) {
  return (int64_t)typed_blocks(tiles_bound, g_exact_fp32 ? (H == 128 ? 3 : TypedGeom<64>::kPerCU) : (H == 128 ? 2 : 4));
}

static int rows_typed_check(const float *A, int64_t lda, const float *W, int64_t ldw, const float *R, int64_t ldr, const float *C,
                            int64_t ldc, const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles,
                            int64_t tiles_bound, int64_t E, int H) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!A || !W || !C || !slot_edge || !tile_scale || !num_tiles || lda < H || ldw < H || ldc < H || (R && ldr < H)) return DMP_ERR_BAD_ARG;
  if (lda % 4 || ldc % 4 || (R && ldr % 4) || !aligned16(A) || !aligned16(C) || (R && !aligned16(R))) return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(lda) || !stride_ok(ldc) || (R && !stride_ok(ldr)) || E >= ((int64_t)1 << 30) || !fits32(tiles_bound * kSub, 1) ||
      !fits32(H, ldw))
    return DMP_ERR_UNSUPPORTED;
  return DMP_OK;
}

int dmp_out_fwd_typed(const dmp_typed_job *jobs, int num_jobs, const int32_t *slot_edge, const float *tile_scale,
                      const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, void *stream) {
  if (!jobs || num_jobs < 1 || num_jobs > kTypedJobs) return DMP_ERR_BAD_ARG;
  if (E == 0) return (E < 0) ? DMP_ERR_BAD_ARG : DMP_OK;
  TypedJobs js{};
  for (int j = 0; j < num_jobs; ++j) {
    const dmp_typed_job &q = jobs[j];
    const int rc = rows_typed_check(q.Hin, q.ldh, q.W2, q.ldw, q.R, q.ldr, q.out, q.ldo, slot_edge, tile_scale, num_tiles, tiles_bound, E, H);
    if (rc != DMP_OK) return rc;
    if (q.bias && !aligned16(q.bias)) return DMP_ERR_UNSUPPORTED;
    if (q.act && !slope_ok(q.slope)) return DMP_ERR_UNSUPPORTED;
    if (q.R2 && (q.ldr2 < H || q.ldr2 % 4 || !aligned16(q.R2) || !stride_ok(q.ldr2) || !fits4g(E, q.ldr2))) return DMP_ERR_UNSUPPORTED;
    TypedArgs &p = js.job[j];
    p.A = q.Hin; p.lda = q.ldh; p.W = q.W2; p.ldw = q.ldw; p.transposed = q.w_in_out ? 0 : 1;   // [in, out]: B[k][j] = W2[k][j]; nn.Linear's [out, in]: W2[j][k]
    p.C = q.out; p.ldc = q.ldo; p.E = E; p.slot_edge = slot_edge; p.slot_arow = slot_edge; p.rowsA = E; p.tile_scale = tile_scale;
    p.num_tiles = num_tiles; p.bias = q.bias; p.R = q.R; p.ldr = q.R ? q.ldr : H; p.R2 = q.R2; p.ldr2 = q.R2 ? q.ldr2 : H;
    p.num_panels = 1; p.act = q.act ? 1 : 0; p.slope = q.slope;
  }
  hipStream_t st = (hipStream_t)stream;
  if (num_jobs == 1) {
    if (js.job[0].R2 && (!fits4g(js.job[0].rowsA, js.job[0].lda) || !fits4g(E, js.job[0].ldc))) return DMP_ERR_UNSUPPORTED;   // (the 64-bit row form has no second addend)
    return H == 128 ? launch_typed<TEPI_OUT, 128>(js.job[0], tiles_bound, st) : launch_typed<TEPI_OUT, 64>(js.job[0], tiles_bound, st);
  }
  if (g_exact_fp32) {          // development switch: one launch per job on the f32-input MFMA
    for (int j = 0; j < num_jobs; ++j) {
      const int rc = H == 128 ? launch_typed<TEPI_OUT, 128>(js.job[j], tiles_bound, st) : launch_typed<TEPI_OUT, 64>(js.job[j], tiles_bound, st);
      if (rc != DMP_OK) return rc;
    }
    return DMP_OK;
  }
  for (int j = 0; j < num_jobs; ++j)
    if (!fits4g(js.job[j].rowsA, js.job[j].lda) || !fits4g(E, js.job[j].ldc) || (js.job[j].R && !fits4g(E, js.job[j].ldr))) return DMP_ERR_UNSUPPORTED;
  // the grid the single launch would get, shared out over the jobs (every workgroup loads one panel)
  const unsigned cap = typed_blocks(tiles_bound, H == 128 ? 2 : 4);
  unsigned gx = (cap + num_jobs - 1) / num_jobs;
  if (gx < 1) gx = 1;
  const dim3 grid(gx, (unsigned)num_jobs);
  if (H == 128) mfma_typed_jobs<128><<<grid, TypedGeom<128>::kThreads, 0, st>>>(js);
  else mfma_typed_jobs<64><<<grid, TypedGeom<64>::kThreads, 0, st>>>(js);
  return check_launch();
}

int dmp_out_fwd_typed_codes(const dmp_typed_job *job, const float *codes, int64_t ldc, int kcodes, const float *Wc, int64_t ldwc,
                            int64_t code_row0, int64_t r_rows, const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E,
                            int H, void *stream) {
  if (H != 128 || g_exact_fp32) return DMP_ERR_UNSUPPORTED;
  if (!job || !codes || !Wc || kcodes < 1 || kcodes > 16 || ldc < kcodes || ldwc < H || code_row0 < 0 || code_row0 > E || r_rows < 0 || r_rows > E)
    return DMP_ERR_BAD_ARG;
  if (E == 0) return (E < 0) ? DMP_ERR_BAD_ARG : DMP_OK;
  const dmp_typed_job &q = *job;
  const int rc = rows_typed_check(q.Hin, q.ldh, q.W2, q.ldw, q.R, q.ldr, q.out, q.ldo, slot_edge, tile_scale, num_tiles, tiles_bound, E, H);
  if (rc != DMP_OK) return rc;
  if (q.R2 || (q.bias && !aligned16(q.bias)) || (q.act && !slope_ok(q.slope))) return DMP_ERR_UNSUPPORTED;
  if (ldc % 4 || !aligned16(codes) || !stride_ok(ldc) || !fits4g(E, ldc) || !fits4g(E, q.ldh) || !fits4g(E, q.ldo) || (q.R && !fits4g(E, q.ldr)) ||
      !fits32(kcodes, ldwc))
    return DMP_ERR_UNSUPPORTED;
  TypedArgs p{};
  p.A = q.Hin; p.lda = q.ldh; p.W = q.W2; p.ldw = q.ldw; p.transposed = q.w_in_out ? 0 : 1;
  p.C = q.out; p.ldc = q.ldo; p.E = E; p.slot_edge = slot_edge; p.slot_arow = slot_edge; p.rowsA = E; p.tile_scale = tile_scale;
  p.num_tiles = num_tiles; p.bias = q.bias; p.R = q.R; p.ldr = q.R ? q.ldr : H; p.R2 = nullptr; p.ldr2 = H;
  p.num_panels = 1; p.act = q.act ? 1 : 0; p.slope = q.slope;
  p.codes = codes + code_row0 * ldc; p.ldcodes = ldc; p.kcodes = kcodes; p.Wc = Wc; p.ldwc = ldwc; p.code_row0 = code_row0; p.rowsR = q.R ? r_rows : 0;
  mfma_typed_codes<<<typed_blocks(tiles_bound, 2), TypedGeom<128>::kThreads, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

int dmp_bwd_h1_typed(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                     const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E,
                     int H, float slope, float *dG, int64_t ldg, float *partial, float *partial_rows, void *stream) {
  if (!partial || !aligned16(partial) || (partial_rows && !aligned16(partial_rows))) return DMP_ERR_BAD_ARG;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) {
    const size_t bytes = sizeof(float) * H * (size_t)dmp_typed_partial_rows(tiles_bound, H);
    if (hipMemsetAsync(partial, 0, bytes, (hipStream_t)stream) != hipSuccess) return DMP_ERR_HIP;
    if (partial_rows && hipMemsetAsync(partial_rows, 0, bytes, (hipStream_t)stream) != hipSuccess) return DMP_ERR_HIP;
    return DMP_OK;
  }
  if (!H1) return DMP_ERR_BAD_ARG;
  const int rc = rows_typed_check(dO, ldo, W2, ldw, H1, ldh, dG, ldg, slot_edge, tile_scale, num_tiles, tiles_bound, E, H);
  if (rc != DMP_OK) return rc;
  TypedArgs p{};
  p.A = dO; p.lda = ldo; p.W = W2; p.ldw = ldw; p.transposed = 0;                    // dH1 = dO @ W2, W2 [out, in] = B[k = out][j = in]
  p.C = dG; p.ldc = ldg; p.E = E; p.slot_edge = slot_edge; p.slot_arow = slot_edge; p.rowsA = E; p.tile_scale = tile_scale;
  p.num_tiles = num_tiles; p.R = H1; p.ldr = ldh; p.slope = slope; p.partial = partial; p.partialA = partial_rows; p.num_panels = 1;
  return H == 128 ? launch_typed<TEPI_H1, 128>(p, tiles_bound, (hipStream_t)stream)
                  : launch_typed<TEPI_H1, 64>(p, tiles_bound, (hipStream_t)stream);
}

int dmp_mask_slots(const int32_t *slot, int64_t n, const float *gate, int64_t E, int32_t *out, void *stream) {
  if (n < 0 || E < 0) return DMP_ERR_BAD_ARG;
  if (n == 0) return DMP_OK;
  if (!slot || !gate || !out) return DMP_ERR_BAD_ARG;
  mask_slots_k<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(slot, n, gate, E, out);
  return check_launch();
}

int dmp_bwd_z_typed_arow(const float *dPre, int64_t ldp, const float *W, int64_t ldw, const float *D, int64_t ldd,
                         int64_t num_nodes, const float *base, int64_t ldb, const int32_t *dst, const uint8_t *flag,
                         float s0, float s1, const int32_t *slot_edge, const int32_t *slot_arow, const float *tile_scale,
                         const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, int w_transposed,
                         const int32_t *base_map, int64_t base_rows, float *dZ, int64_t ldz, void *stream) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || num_nodes < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (E == 0) return DMP_OK;
  if (!dPre || !W || !D || !dst || !slot_edge || !tile_scale || !num_tiles || !dZ || ldp < H || ldw < 2 * H || ldd < 2 * H ||
      ldz < H || (base && ldb < H))
    return DMP_ERR_BAD_ARG;
  if (ldp % 4 || ldz % 4 || ldd % 4 || (base && ldb % 4) || !aligned16(dPre) || !aligned16(dZ) || !aligned16(D) ||
      (base && !aligned16(base)))
    return DMP_ERR_UNSUPPORTED;
  if (base_map && (!base || base_rows < 0)) return DMP_ERR_BAD_ARG;
  if (!stride_ok(ldd) || !stride_ok(ldp) || !stride_ok(ldz) || (base && !stride_ok(ldb)) || E >= ((int64_t)1 << 30) ||
      num_nodes >= ((int64_t)1 << 31) || !fits32(tiles_bound * kSub, 1))
    return DMP_ERR_UNSUPPORTED;
  TypedArgs p{};
  p.A = dPre; p.lda = ldp; p.W = W; p.ldw = ldw; p.transposed = w_transposed ? 0 : 1; p.C = dZ; p.ldc = ldz; p.E = E;
  p.slot_edge = slot_edge; p.slot_arow = slot_arow ? slot_arow : slot_edge; p.rowsA = E; p.tile_scale = tile_scale; p.num_tiles = num_tiles;
  p.idxA = dst; p.flag = flag; p.T = D; p.ldt = ldd; p.num_nodes = num_nodes; p.R = base; p.ldr = base ? ldb : H;
  p.s0 = s0; p.s1 = s1; p.rmap = base_map; p.rowsR = base_rows;
  return H == 128 ? launch_typed<TEPI_DZ, 128>(p, tiles_bound, (hipStream_t)stream)
                  : launch_typed<TEPI_DZ, 64>(p, tiles_bound, (hipStream_t)stream);
}

int dmp_rel_gemm(const float *A, int64_t lda, int64_t rows_a, const float *W, int64_t ldw, int num_rels, int w_transposed,
                 const int32_t *slot_arow, const int32_t *slot_row, const int32_t *tile_type, const int32_t *num_tiles,
                 int64_t tiles_bound, const float *row_scale, int64_t rows_c, int H, float *C, int64_t ldc, void *stream) {
  if (rows_a < 0 || rows_c < 0 || tiles_bound < 0 || num_rels < 1 || H != 128) return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (rows_c == 0 || tiles_bound == 0) return DMP_OK;
  if (!A || !W || !slot_arow || !slot_row || !tile_type || !num_tiles || !C || lda < H || ldw < H || ldc < H) return DMP_ERR_BAD_ARG;
  if (lda % 4 || ldc % 4 || !aligned16(A) || !aligned16(C)) return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(lda) || !stride_ok(ldc) || rows_a >= ((int64_t)1 << 31) || rows_c >= ((int64_t)1 << 30) || !fits32(tiles_bound * kSub, 1) ||
      !fits32((int64_t)num_rels * 128, ldw))
    return DMP_ERR_UNSUPPORTED;
  TypedArgs p{};
  p.A = A; p.lda = lda; p.rowsA = rows_a; p.W = W; p.ldw = ldw; p.num_panels = num_rels; p.transposed = w_transposed ? 1 : 0;
  p.C = C; p.ldc = ldc; p.E = rows_c; p.slot_edge = slot_row; p.slot_arow = slot_arow;
  p.tile_scale = reinterpret_cast<const float *>(tile_type); p.num_tiles = num_tiles;
  p.idxA = reinterpret_cast<const int32_t *>(row_scale);
  return launch_typed<TEPI_REL, 128>(p, tiles_bound, (hipStream_t)stream);
}

}  // extern "C"
