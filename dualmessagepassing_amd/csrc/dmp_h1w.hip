// Backward of the edge MLP's second Linear over the tiles of the kept edges, BOTH of its products in one launch (gfx950, H = 128):
//
//   dPre[e] = act'(H1[e]) (.) (dO[e] W2)        (the input gradient, dmpnn.py:147-152 reversed; what dmp_bwd_h1_typed makes)
//   dW2     = sum_e dO[e]^T H1[e]               (the weight gradient;                            what dmp_atb_typed(plain) makes)
//   db2     = sum_e dO[e],  dbe = sum_e dPre[e] (the two bias gradients)
//
// Both products read the SAME two [E, H] operands (dO, H1).  As two launches every kept row of both arrays crosses the fabric
// twice, and the weight-gradient kernel spends its time splitting fp32 fragments into bf16 pieces in every wave that needs
// them.  Here a 512-thread workgroup (one per CU) owns a contiguous range of 32-row tiles and its eight waves take two ROLES:
//
//   waves 0-3 ("rows")    the product dO W2 as in csrc/dmp_typed.hip (TEPI_H1): the W2 panel in registers as bf16 pieces, one
//                         32-column slice per wave, activation derivative + column sums + 16-byte row stores in the epilogue;
//                         their 256 threads fetch, split and stage the tile's dO rows;
//   waves 4-7 ("columns") the product dO^T H1: every wave a 64 x 64 QUADRANT of the [128, 128] total in 4 accumulators, the
//                         contraction over the tile's 32 rows as two 16-deep k-groups; their 256 threads fetch, split and stage
//                         the tile's H1 rows.
//
// ONE LDS image per operand serves both: bf16 pieces (hi | mid | lo planes) of the fp32 rows, split once by the staging thread,
// rows of 256 bytes with the 16-byte chunks XOR-swizzled (cdna_hip_programming.md T10, image (b)).  The row product reads it by
// rows (ds_read_b128: a lane's 8 consecutive k of its row), the column product through the hardware transpose read
// ds_read_b64_tr_b16 (a lane's 8 consecutive ROWS of its column: two reads per fragment) -- no second copy, no fragment is split
// twice, and the column waves run no VALU work in their loop beyond addresses.  The row waves need only the SIGN of H1 (the
// activation's derivative on the saved output): they read it from the hi plane (bf16 rounding keeps the sign of every fp32 value
// of magnitude >= 2^-133; below that the saved output is treated as not positive).
//
// Products on the bf16 pipe as six piece products (dmp_mfma_common.h, "bf16x6": fp32-accurate).  dPre is bit-identical to
// dmp_bwd_h1_typed's (same MFMA order); dW2 differs from dmp_atb_typed's by the summation order of the workgroup partials.
// Measured for the design: the row kernel alone keeps 92 % of its rate at ONE workgroup per CU instead of two -- a single
// workgroup's prefetch depth already covers the fabric latency, so the second role rides in the other half of the SIMDs' slots.
#include <type_traits>

#include "dmp_mfma_common.h"

namespace dmp {
namespace {

struct H1WArgs {
  const float *dO; int64_t ldo;          // upstream gradient rows [E, 128]
  const float *H1; int64_t ldh;          // saved activation outputs [E, 128]
  const float *W2; int64_t ldw;          // nn.Linear weight [128 out, ldw >= 128 in]: dH1 = dO @ W2, B[k = out][j = in] = W2[k][j]
  float *dPre; int64_t ldg;              // output rows [E, 128], scattered by edge id (rows outside the tiles are not written)
  int64_t E;
  const int32_t *slot_edge;              // [num_tiles_bound * 32] edge id per slot, -1 = padding
  const int32_t *num_tiles;              // device scalar: tiles in use
  float slope;
  float *partial;                        // [gridDim.x, 128] column sums of dPre per workgroup
  float *partialA;                       // [gridDim.x, 128] column sums of the fetched dO rows (db2), or NULL
  float *partialW;                       // [gridDim.x, 128 * 128] dO^T H1 per workgroup
};

constexpr int kHW = 128, kThreadsW = 512;
constexpr int kPlaneB = 32 * 256;                    // bytes of one bf16 plane of a 32-row tile
constexpr int kBufB = 3 * kPlaneB;                   // one tile of one operand (hi | mid | lo): 24576
constexpr int kOpB = 2 * kBufB;                      // both buffers of one operand: 49152
constexpr int kImgB = 2 * kOpB;                      // dO then H1: 98304
constexpr int kScrB = 4 * 32 * kScrStride * 4;       // the row waves' accumulator transposes: 18432
constexpr int kRowCB = 3 * kSub * 4;                 // store rows of three tiles in flight
constexpr int kH1WLdsBytes = kImgB + kScrB + kRowCB; // 117120: dynamic LDS, opted in once per device
constexpr int kMaxDevicesW2 = 64;

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s *lds_v4s_ptr;

// chunk swizzle of the image: 16-byte chunk ch (0..15) of row `row` lives at 256 row + 16 (ch ^ swz(row))
__device__ __forceinline__ uint32_t swz(uint32_t row) { return ((row & 3u) << 2) | ((row >> 2) & 3u); }

__global__ __launch_bounds__(kThreadsW, 1) void h1w_k(const H1WArgs p) {
  extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
  float *const scr_all = reinterpret_cast<float *>(lds + kImgB);
  uint32_t *const rowC = reinterpret_cast<uint32_t *>(lds + kImgB + kScrB);     // [3][32]: store row of every slot (-1: none)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool cols = wave >= 4;                           // role: false = the row product, true = the column product
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int u = threadIdx.x & 255;                       // staging thread of its operand: rows u / 32 + 8 m, float4 piece u % 32
  const int srow = u >> 5, scol = u & 31;
  constexpr uint32_t kNone = 0xFFFFFFFFu;

  const int ntiles = __builtin_amdgcn_readfirstlane(*p.num_tiles);
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  const rsrc_t rs_slot = make_rsrc(p.slot_edge, (uint32_t)ntiles * (kSub * 4u));
  // the operand this thread stages: dO (row waves) or H1 (column waves)
  const srsrc_t rs_X = cols ? make_srsrc(p.H1, p.ldh, p.E) : make_srsrc(p.dO, p.ldo, p.E);
  const uint32_t op_off = cols ? (uint32_t)kOpB : 0u;

  // ---- staging: global -> registers (requested two tiles ahead) -> three bf16 planes in LDS (one tile ahead)
  int id_rows[kSubLoads];
  int id_own = -1, own_staged = -1;
  float4 pre[kSubLoads];
  float4 csA = make_float4(0.f, 0.f, 0.f, 0.f);          // row waves, partialA: this thread's 4 columns of the dO rows it stages
  auto load_ids = [&](int k) {
    const bool ok = k < mine;
    const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      id_rows[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, (srow + 8 * m) * 4, (int)so, 0) : -1;
    if (!cols && u < kSub) id_own = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, u * 4, (int)so, 0) : -1;
  };
  auto load_row = [&](int m) { pre[m] = sbuf_load4(rs_X, id_rows[m], (uint32_t)scol * 16u); };   // id -1: zeros
  auto load_rows = [&]() {
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(m);
    if (!cols && u < kSub) own_staged = id_own;
  };
  auto stage_row = [&](int buf, int m) {
    if (!cols && p.partialA) { csA.x += pre[m].x; csA.y += pre[m].y; csA.z += pre[m].z; csA.w += pre[m].w; }
    uint2 ph, pm, pl;
    split_pair(pre[m].x, pre[m].y, ph.x, pm.x, pl.x);
    split_pair(pre[m].z, pre[m].w, ph.y, pm.y, pl.y);
    const uint32_t r = (uint32_t)(srow + 8 * m);
    const uint32_t off = op_off + (uint32_t)buf * kBufB + 256u * r + 16u * ((uint32_t)(scol >> 1) ^ swz(r)) + 8u * (uint32_t)(scol & 1);
    *reinterpret_cast<uint2 *>(lds + off) = ph;
    *reinterpret_cast<uint2 *>(lds + off + kPlaneB) = pm;
    *reinterpret_cast<uint2 *>(lds + off + 2 * kPlaneB) = pl;
  };
  auto stage_scalars = [&](int par) {                   // store row of every slot of the staged tile (row waves' threads < 32)
    if (!cols && u < kSub) rowC[par * kSub + u] = own_staged >= 0 ? (uint32_t)own_staged : kNone;
  };

  // ======================================================================== the row product (waves 0-3)
  const int cs = wave & 3;
  const int col = 32 * cs + li;
  float *const scr = scr_all + cs * (32 * kScrStride);
  const int lrow = lane >> 3, c4 = 32 * cs + (lane & 7) * 4;
  const uint32_t col4 = (uint32_t)c4 * 4u;
  const float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.slope)));
  const srsrc_t rs_C = make_srsrc(p.dPre, p.ldg, p.E);
  constexpr int kGroups = 8;                             // 16-deep k-groups of the 128-deep contraction: lane half h owns k = 64 h ..
  Split8 B6[kGroups];                                    // the W2 panel's fragments of this wave's column slice, as bf16 pieces
  auto load_panel = [&]() {
    const rsrc_t rs_W = make_rsrc(p.W2, (uint32_t)(kHW * p.ldw * 4));
    uint32_t off;
    const uint32_t w_first = (uint32_t)((int64_t)64 * h * p.ldw + col) * 4u;
    asm volatile("v_mov_b32 %0, %1" : "=v"(off) : "v"(w_first));
    const uint32_t w_step = __builtin_amdgcn_readfirstlane((int)(p.ldw * 4));
#pragma unroll
    for (int s0 = 0; s0 < 64; s0 += 32) {
      float w0[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) w0[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off, (int)((s0 + j) * w_step), 0));
#pragma unroll
      for (int q = 0; q < 32; q += 8)
        split8(make_float4(w0[q], w0[q + 1], w0[q + 2], w0[q + 3]), make_float4(w0[q + 4], w0[q + 5], w0[q + 6], w0[q + 7]), B6[(s0 + q) / 8]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // row read of the A operand: row li, chunk 8 h + g  ->  256 li + 16 ((8 h + g) ^ swz(li)) = a0 ^ (16 g)
  const uint32_t a0 = 256u * (uint32_t)li + 16u * ((uint32_t)(8 * h) ^ swz((uint32_t)li));
  // the sign operand of the epilogue: H1's hi plane, rows 8 k + lrow, this lane's 4 columns
  uint2 hs[4];
  auto fetch_sign = [&](int buf, int k) {
    const uint32_t rr = (uint32_t)(8 * k + lrow);
    const uint32_t ch = (uint32_t)(4 * cs + ((lane & 7) >> 1));
    hs[k] = *reinterpret_cast<const uint2 *>(lds + kOpB + (uint32_t)buf * kBufB + 256u * rr + 16u * (ch ^ swz(rr)) + 8u * (uint32_t)(lane & 1));
  };
  f32x16 acc;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  // one tile of the row product: 8 k-groups x 6 piece MFMAs with, in their shadow, the sign reads of this tile (groups 0-1),
  // the staging of tile k+1 (2-4), the row requests of tile k+2 (4-6) and the id requests of tile k+3 (7)
  auto rows_step = [&](int k, int par3) {
    const int buf = k & 1, nxt3 = par3 == 2 ? 0 : par3 + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char *img = lds + (uint32_t)buf * kBufB;
    Frag8 ah, am, al;
    ah.v = *reinterpret_cast<const bf16x8 *>(img + a0);
    am.v = *reinterpret_cast<const bf16x8 *>(img + kPlaneB + a0);
    al.v = *reinterpret_cast<const bf16x8 *>(img + 2 * kPlaneB + a0);
    auto action = [&](int i) {
      if (i < 4) fetch_sign(buf, i);
      else if (i < 8) stage_row(buf ^ 1, i - 4);
      else if (i == 8) stage_scalars(nxt3);
      else if (i < 13) load_row(i - 9);
      else if (i == 13) { if (u < kSub) own_staged = id_own; }
      else if (i == 14) load_ids(k + 3);
    };
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      Frag8 nh = ah, nm = am, nl = al;
      if (g + 1 < kGroups) {
        const uint32_t an = a0 ^ (uint32_t)(16 * (g + 1));
        nh.v = *reinterpret_cast<const bf16x8 *>(img + an);
        nm.v = *reinterpret_cast<const bf16x8 *>(img + kPlaneB + an);
        nl.v = *reinterpret_cast<const bf16x8 *>(img + 2 * kPlaneB + an);
      }
      const Split8 &bb = B6[g];
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bb.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.lo.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.mid.v, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      action(2 * g);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.mid.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.hi.v, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      action(2 * g + 1);
      ah = nh; am = nm; al = nl;
    }
  };
  auto rows_epilogue = [&](int par) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * h) * kScrStride + li] = acc[r];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = 8 * k + lrow;
      float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
      // the saved output's sign from its hi piece (padding rows: zeros -> slope * 0 = 0, and the store is dropped)
      v.x = act_bwd(__uint_as_float(hs[k].x << 16), v.x, slope); v.y = act_bwd(__uint_as_float(hs[k].x & 0xFFFF0000u), v.y, slope);
      v.z = act_bwd(__uint_as_float(hs[k].y << 16), v.z, slope); v.w = act_bwd(__uint_as_float(hs[k].y & 0xFFFF0000u), v.w, slope);
      colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
      sbuf_store4(v, rs_C, (int)rowC[par * kSub + rr], col4);
    }
  };

  // ======================================================================== the column product (waves 4-7)
  const int pw = wave & 1, qw = (wave >> 1) & 1;         // quadrant: output rows 64 pw .., columns 64 qw ..
  // transposed read (T10): group G = lane / 16 reads the block rows 16 kg + 8 (G >> 1) + 4 t .. + 3, columns 16 (G & 1) .. + 15 of
  // the wave's 32-column block; lane 4 q + pp of the group supplies row + q, columns 4 pp .. 4 pp + 3 and receives column (lane & 15),
  // rows + 0 .. 3 in its four elements: fragment element 4 t + e of lane (li, h) = row 16 kg + 8 h + 4 t + e, column li.
  //   address = 256 row + 16 (chunk ^ swz(row)) + 8 (pp & 1),  chunk = 8 half + 4 blk + 2 (G & 1) + (pp >> 1),
  //   swz(row) = (q << 2) | (2 (G >> 1) + t)   -- blk toggles bit 2 of the chunk, t bit 0:  address = t0 ^ (64 blk) ^ (16 t) + 1024 t + 4096 kg
  const int G16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  auto tr_base = [&](int half) {
    const uint32_t row0 = (uint32_t)(8 * (G16 >> 1) + q4);
    const uint32_t ch0 = (uint32_t)(8 * half + 2 * (G16 & 1) + (pp >> 1));
    const uint32_t sw0 = ((uint32_t)q4 << 2) | (uint32_t)(2 * (G16 >> 1));
    return 256u * row0 + 16u * (ch0 ^ sw0) + 8u * (uint32_t)(pp & 1);
  };
  const uint32_t tA = tr_base(pw), tB = (uint32_t)kOpB + tr_base(qw);
  auto tr_frag = [&](uint32_t base, int buf, int plane, int blk, int kg, Frag8 &f) {
    const uint32_t o0 = (base ^ (uint32_t)(64 * blk)) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg);
    const uint32_t o1 = (base ^ (uint32_t)(64 * blk) ^ 16u) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg + 1024);
    const v4s x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o0));
    const v4s y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o1));
    f.v = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto tr_split = [&](uint32_t base, int buf, int blk, int kg, Split8 &s) {
    tr_frag(base, buf, 0, blk, kg, s.hi);
    tr_frag(base, buf, 1, blk, kg, s.mid);
    tr_frag(base, buf, 2, blk, kg, s.lo);
  };
  // one tile of the column product: 2 k-groups x (2 x 2) blocks x 6 piece MFMAs; in their shadow the staging of tile k+1's H1 rows,
  // the row requests of tile k+2 and the id requests of tile k+3
  auto cols_step = [&](int k, f32x16 (&wacc)[2][2]) {
    const int buf = k & 1;
    int act = 0;
    // eight blocks (kg, jb, ib); the fragments the next block needs are requested BEFORE this block's MFMAs
    Split8 fa[2], fb, nfa[2], nfb;
    tr_split(tA, buf, 0, 0, fa[0]);
    tr_split(tA, buf, 1, 0, fa[1]);
    tr_split(tB, buf, 0, 0, fb);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int kg = b >> 2, jb = (b >> 1) & 1, ib = b & 1;
      if (ib == 1 && b + 1 < 8) {                             // the next block starts a new B fragment (and, at b == 3, new A fragments)
        tr_split(tB, buf, jb ^ 1, jb == 1 ? kg + 1 : kg, nfb);
        if (b == 3) { tr_split(tA, buf, 0, 1, nfa[0]); tr_split(tA, buf, 1, 1, nfa[1]); }
      }
      __builtin_amdgcn_sched_barrier(0);
      wacc[ib][jb] = mfma_x6(fa[ib], fb, wacc[ib][jb]);
      __builtin_amdgcn_sched_barrier(0);
      if (act < 4) stage_row(buf ^ 1, act);
      else load_row(act - 4);
      ++act;
      if (ib == 1 && b + 1 < 8) {
        fb = nfb;
        if (b == 3) { fa[0] = nfa[0]; fa[1] = nfa[1]; }
      }
    }
    load_ids(k + 3);
  };

  // ======================================================================== the loop (both roles: one barrier per tile)
  if (mine == 0) {                                       // nothing to do: this workgroup's partial rows are zeros
    if (threadIdx.x < 32) {
      *reinterpret_cast<float4 *>(p.partial + (int64_t)blockIdx.x * kHW + threadIdx.x * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.partialA) *reinterpret_cast<float4 *>(p.partialA + (int64_t)blockIdx.x * kHW + threadIdx.x * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 *pw4 = reinterpret_cast<float4 *>(p.partialW + (int64_t)blockIdx.x * (kHW * kHW));
    for (int m = threadIdx.x; m < kHW * kHW / 4; m += kThreadsW) pw4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  load_ids(0);
  load_rows();                 // tile 0
  load_ids(1);
  // The two roles run their own loops (same number of barriers): nothing of one role's register state -- the panel's 96
  // registers, the quadrant's 64 accumulators -- is live in the other's code.
  float *tot = reinterpret_cast<float *>(lds);   // after the loop the image is free: the [128, 128] total goes through it for 16-byte stores
  if (cols) {
    f32x16 wacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[i][j][r] = 0.f;
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(0, m);
    load_rows();               // tile 1
    load_ids(2);
    lds_barrier();
    for (int k = 0; k < mine; ++k) {
      cols_step(k, wacc);
      lds_barrier();           // tile k+1 is staged for everyone, everyone is done with tile k's image
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tot[(64 * pw + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * kHW + 64 * qw + 32 * j + li] = wacc[i][j][r];
  } else {
    load_panel();              // (requested behind tile 0's rows: the two round trips run side by side)
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(0, m);
    stage_scalars(0);
    load_rows();               // tile 1
    load_ids(2);
    lds_barrier();
    int par3 = 0;
    for (int k = 0; k < mine; ++k) {
      rows_step(k, par3);
      lds_barrier();
      rows_epilogue(par3);
      par3 = par3 == 2 ? 0 : par3 + 1;
    }
    __syncthreads();
    // column sums of dPre: lanes with equal (lane & 7) hold the same 4 columns for 8 different rows (fixed-order xor tree)
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      colsum.x += __shfl_xor(colsum.x, off, 64); colsum.y += __shfl_xor(colsum.y, off, 64);
      colsum.z += __shfl_xor(colsum.z, off, 64); colsum.w += __shfl_xor(colsum.w, off, 64);
    }
    if (lane < 8) *reinterpret_cast<float4 *>(p.partial + (int64_t)blockIdx.x * kHW + 32 * cs + lane * 4) = colsum;
  }
  __syncthreads();
  {
    float4 *pw4 = reinterpret_cast<float4 *>(p.partialW + (int64_t)blockIdx.x * (kHW * kHW));
#pragma unroll 4
    for (int m = threadIdx.x; m < kHW * kHW / 4; m += kThreadsW) pw4[m] = *reinterpret_cast<const float4 *>(&tot[m * 4]);
  }
  if (p.partialA) {
    // column sums of the staged dO rows: the 8 row-wave threads that staged the same 4 columns, added in a fixed order
    __syncthreads();
    if (!cols) *reinterpret_cast<float4 *>(&tot[srow * kHW + scol * 4]) = csA;
    __syncthreads();
    if (threadIdx.x < 32) {
      float4 t = *reinterpret_cast<const float4 *>(&tot[threadIdx.x * 4]);
#pragma unroll
      for (int g = 1; g < 8; ++g) {
        const float4 w = *reinterpret_cast<const float4 *>(&tot[g * kHW + threadIdx.x * 4]);
        t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
      }
      *reinterpret_cast<float4 *>(p.partialA + (int64_t)blockIdx.x * kHW + threadIdx.x * 4) = t;
    }
  }
}


// ============================================================================================================================
// The tall-skinny weight gradients  T = sum_e Z[e]^T D[e]  (and, class-typed, B = sum_e c_e Z[e]^T D[e]) over a tile list on the
// same image (dmp_atb2_typed / dmp_atb2_jobs; what csrc/dmp_atb.hip's bf16x6 kernels compute).  There every wave reads fp32
// fragments down the columns of an fp32 LDS tile and splits them into bf16 pieces ITSELF -- the splits (5.5 VALU instructions
// per element, every fragment split by every wave that needs it) are what bound those kernels (2.9 TB/s of rows at bench.py's
// shape).  Here the staging thread splits each fetched element ONCE into the three bf16 planes and all eight waves of the
// 512-thread workgroup read their fragments through ds_read_b64_tr_b16: no VALU work in the MFMA loop beyond addresses.
// Waves 0-3's threads stage Z rows, waves 4-7's stage D rows (two tiles of rows in flight per thread); wave w owns the
// 64 x 32 block of the [128, 128] total at rows 64 (w & 1), columns 32 (w >> 1): two accumulators.  A class that ends inside
// the workgroup's tile range emits  T += acc, B += c acc  straight from the accumulators into its own partial (read-modify-
// write by the same lane: a fixed order) and restarts them.  blockIdx.y = the job: several products over the SAME tile list
// (the node side's three weight gradients over the kept nodes' tiles) share one launch of one workgroup per CU.
struct Atb2Job { const float *Z; int64_t ldz; const float *D; int64_t ldd; float *pT, *pB; int64_t pstride; int ldp; };
constexpr int kAtb2MaxJobs = 6;
struct Atb2Args {
  Atb2Job job[kAtb2MaxJobs];
  int64_t E;                        // rows of every operand
  const int32_t *slot_edge;         // [tiles * 32] row of every slot, -1 = padding
  const float *tile_scale;          // [tiles] coefficient of the tile's class
  const int32_t *num_tiles;         // [1] tiles in use (device)
};
constexpr int kAtb2LdsBytes = kImgB;

__global__ __launch_bounds__(kThreadsW, 1) void atb2_k(const Atb2Args p) {
  extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
  const Atb2Job &jb_ = p.job[blockIdx.y];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool dside = wave >= 4;                          // staging role: false = rows of Z, true = rows of D
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int u = threadIdx.x & 255;
  const int srow = u >> 5, scol = u & 31;

  const int ntiles = __builtin_amdgcn_readfirstlane(*p.num_tiles);
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  const rsrc_t rs_slot = make_rsrc(p.slot_edge, (uint32_t)ntiles * (kSub * 4u));
  const srsrc_t rs_X = dside ? make_srsrc(jb_.D, jb_.ldd, p.E) : make_srsrc(jb_.Z, jb_.ldz, p.E);
  const uint32_t op_off = dside ? (uint32_t)kOpB : 0u;
  float *const pt = jb_.pT + (int64_t)blockIdx.x * jb_.pstride;
  float *const pb = jb_.pB ? jb_.pB + (int64_t)blockIdx.x * jb_.pstride : nullptr;
  const int ldp = jb_.ldp;

  const int pw = wave & 1, cw = wave >> 1;               // this wave's block of the total: rows 64 pw .., columns 32 cw ..
  f32x16 acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  };
  // accumulator (i, r): output row 64 pw + 32 i + (r & 3) + 8 (r >> 2) + 4 h, column 32 cw + li -- through buffer descriptors:
  // one per-lane offset, the (i, r) part of the address as a scalar offset (no per-element address registers)
  const rsrc_t rs_pt = make_rsrc(pt, (uint32_t)(kHW * ldp * 4));
  const rsrc_t rs_pb = make_rsrc(pb, pb ? (uint32_t)(kHW * ldp * 4) : 0u);
  const uint32_t e_voff = (uint32_t)((64 * pw + 4 * h) * ldp + 32 * cw + li) * 4u;
  const uint32_t ldp4 = (uint32_t)__builtin_amdgcn_readfirstlane(ldp * 4);
  bool emitted = false;
  auto emit = [&](float c) {
    // (all reads of what this lane stored before are requested together, then one wait: an element at a time the emission
    // was 32 dependent round trips -- ~50 us per class boundary, more than the whole rest of the launch)
    float t0[2][16], b0[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t so = (uint32_t)(32 * i + (r & 3) + 8 * (r >> 2)) * ldp4;
        t0[i][r] = emitted ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_pt, (int)e_voff, (int)so, 0)) : 0.f;
        b0[i][r] = (emitted && pb) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_pb, (int)e_voff, (int)so, 0)) : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t so = (uint32_t)(32 * i + (r & 3) + 8 * (r >> 2)) * ldp4;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(emitted ? acc[i][r] + t0[i][r] : acc[i][r]), rs_pt, (int)e_voff, (int)so, 0);
        if (pb) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(emitted ? c * acc[i][r] + b0[i][r] : c * acc[i][r]), rs_pb, (int)e_voff, (int)so, 0);
      }
    emitted = true;
  };
  if (mine == 0) {                                       // (uniform per workgroup) this workgroup's partial: zeros
    zero_acc();
    emit(0.f);
    return;
  }

  // ---- staging: two tiles of rows in flight per thread (sets alternate by tile parity), one tile staged ahead.  The slot ids of
  // a tile are requested one step BEFORE the rows of the tile ahead of it: loads retire through one in-order counter, so ids
  // requested behind a tile's row requests could only be waited for together with those rows -- one tile in flight, not two.
  int id_rows[2][kSubLoads];
  float4 pre[2][kSubLoads];
  auto load_ids = [&](auto set, int k) {
    constexpr int S = decltype(set)::value;
    const bool ok = k < mine;
    const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      id_rows[S][m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, (srow + 8 * m) * 4, (int)so, 0) : -1;
  };
  auto load_row = [&](auto set, int m) {
    constexpr int S = decltype(set)::value;
    pre[S][m] = sbuf_load4(rs_X, id_rows[S][m], (uint32_t)scol * 16u);
  };
  auto stage_row = [&](auto set, int buf, int m) {
    constexpr int S = decltype(set)::value;
    uint2 ph, pm, pl;
    split_pair(pre[S][m].x, pre[S][m].y, ph.x, pm.x, pl.x);
    split_pair(pre[S][m].z, pre[S][m].w, ph.y, pm.y, pl.y);
    const uint32_t r = (uint32_t)(srow + 8 * m);
    const uint32_t off = op_off + (uint32_t)buf * kBufB + 256u * r + 16u * ((uint32_t)(scol >> 1) ^ swz(r)) + 8u * (uint32_t)(scol & 1);
    *reinterpret_cast<uint2 *>(lds + off) = ph;
    *reinterpret_cast<uint2 *>(lds + off + kPlaneB) = pm;
    *reinterpret_cast<uint2 *>(lds + off + 2 * kPlaneB) = pl;
  };
  std::integral_constant<int, 0> s0;
  std::integral_constant<int, 1> s1;

  // transposed reads (see h1w_k): fragment element 4 t + e of lane (li, h) = row 16 kg + 8 h + 4 t + e of the tile, column li of the block
  const int G16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  auto tr_base = [&](int chunk8) {                       // chunk8: the block's first 16-byte chunk (a multiple of 4 blocks of 32 columns)
    const uint32_t row0 = (uint32_t)(8 * (G16 >> 1) + q4);
    const uint32_t ch0 = (uint32_t)(chunk8 + 2 * (G16 & 1) + (pp >> 1));
    const uint32_t sw0 = ((uint32_t)q4 << 2) | (uint32_t)(2 * (G16 >> 1));
    return 256u * row0 + 16u * (ch0 ^ sw0) + 8u * (uint32_t)(pp & 1);
  };
  const uint32_t tA = tr_base(8 * pw), tB = (uint32_t)kOpB + tr_base(4 * cw);    // A: chunks 8 pw + 4 i ..; B: chunks 4 cw ..
  auto tr_frag = [&](uint32_t base, int buf, int plane, int blk, int kg, Frag8 &f) {
    const uint32_t o0 = (base ^ (uint32_t)(64 * blk)) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg);
    const uint32_t o1 = (base ^ (uint32_t)(64 * blk) ^ 16u) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg + 1024);
    const v4s x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o0));
    const v4s y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o1));
    f.v = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto tr_split = [&](uint32_t base, int buf, int blk, int kg, Split8 &s) {
    tr_frag(base, buf, 0, blk, kg, s.hi);
    tr_frag(base, buf, 1, blk, kg, s.mid);
    tr_frag(base, buf, 2, blk, kg, s.lo);
  };
  // one tile (buffer k & 1): 2 k-groups x 2 row blocks x 6 piece MFMAs; in their shadow: the staging of tile k+1 (its rows sit in
  // set (k + 1) & 1 = nset), the id requests of tile k+4 (into the OTHER id set), then the row requests of tile k+3 into nset
  // (its ids came one step ago, ahead of tile k+2's row requests)
  auto tile_step = [&](int k, auto nset, auto oset) {
    const int buf = k & 1;
    int act = 0;
    // four blocks (kg, ib); the fragments of block b+1 are requested BEFORE the MFMAs of block b (a read that an MFMA waits for
    // exposes its whole LDS latency: 8 times per tile and wave otherwise)
    Split8 fb, fa, nfa, nfb;
    tr_split(tB, buf, 0, 0, fb);
    tr_split(tA, buf, 0, 0, fa);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int kg = b >> 1, ib = b & 1;
      if (b + 1 < 4) {
        tr_split(tA, buf, (b + 1) & 1, (b + 1) >> 1, nfa);
        if (ib == 1) tr_split(tB, buf, 0, kg + 1, nfb);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[ib] = mfma_x6(fa, fb, acc[ib]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q, ++act) {
        if (act < 4) stage_row(nset, buf ^ 1, act);
        else {
          if (act == 4) load_ids(oset, k + 4);
          load_row(nset, act - 4);
        }
      }
      if (b + 1 < 4) {
        fa = nfa;
        if (ib == 1) fb = nfb;
      }
    }
  };
  // start the pipeline: tile 0 staged in buffer 0, the rows of tiles 1 / 2 requested into sets 1 / 0, the ids of tile 3 into set 1
  auto start = [&]() {
    load_ids(s0, 0);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(s0, m);
    load_ids(s1, 1);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(s0, 0, m);
    load_ids(s0, 2);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(s1, m);
    load_ids(s1, 3);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(s0, m);
    lds_barrier();
  };

  zero_acc();
  float cur = 0.f, sv = 0.f;
  unsigned long long starts = 0;
  start();
  for (int k = 0; k < mine; ++k) {
    // the class structure of the range, 64 tiles at a time (lane l: tile 64 c + l; bit l of `starts`: that tile begins a new class)
    if ((k & 63) == 0) {
      const float last = __shfl(sv, 63);
      sv = k + lane < mine ? p.tile_scale[lo + k + lane] : 0.f;
      float up = __shfl_up(sv, 1);
      if (lane == 0) up = k > 0 ? last : sv;
      starts = __ballot(k + lane < mine && __float_as_uint(sv) != __float_as_uint(up));
    }
    if ((starts >> (k & 63)) & 1ull) {                    // (wave-uniform) tile k begins a new class: the finished class's total goes out
      if (pb) { emit(cur); zero_acc(); }                  // (the plain total needs no class: nothing to scale)
    }
    cur = __shfl(sv, k & 63);
    if (k & 1) tile_step(k, s0, s1);                      // tile k+1 is even: its rows sit in set 0
    else tile_step(k, s1, s0);
    lds_barrier();                                        // tile k+1 is staged for everyone, everyone is done with tile k's image
  }
  emit(cur);
}

inline unsigned atb2_blocks(int64_t tiles_bound, int num_jobs) {
  int64_t g = 256 / (num_jobs > 0 ? num_jobs : 1);       // one workgroup per CU over all jobs
  if (g < 1) g = 1;
  if (g > tiles_bound) g = tiles_bound > 0 ? tiles_bound : 1;
  return (unsigned)g;
}

inline bool atb2_lds_ready() {
  static bool done[kMaxDevicesW2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesW2) dev = 0;
  if (done[dev]) return true;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&atb2_k), hipFuncAttributeMaxDynamicSharedMemorySize, kAtb2LdsBytes);
  if (e != hipSuccess) { set_last_hip_error(e); return false; }
  done[dev] = true;
  return true;
}


// ============================================================================================================================
// The edge chain's input gradient AND its class-typed weight gradient in one launch (dmp_bwd_z_w; what dmp_bwd_z_typed_arow and
// dmp_atb_typed make in two):
//     dZ[e]  = base[e] + s(flag e) D[dst e, half(flag e)] + dPre[e] W_g(e)^T           (dmpnn.py:142-156 backward, W_g = A' + c_g B')
//     dWes   = [ sum_e Z[e]^T dPre[e] | sum_e c(e) Z[e]^T dPre[e] ]
// over the SAME class-sorted tile list: both read dPre.  The two roles of h1w_k on the same image: waves 0-3 ("rows") are
// mfma_typed<TEPI_DZ> -- the per-class panel in registers, rebuilt where the workgroup's contiguous tile range crosses into the next
// class, the gathered D rows and the base rows in the epilogue; their threads stage dPre.  Waves 4-7 ("columns") keep a 64 x 64
// quadrant of Z^T dPre each and emit  T += acc, B += c acc  into the workgroup's partial where a class ends (atb2_k's emission);
// their threads stage Z.  dZ is bit-identical to dmp_bwd_z_typed_arow's.
struct DzwArgs {
  const float *dPre; int64_t ldp;        // [E, 128] rows gathered by slot id (a padding slot / a masked-out row: -1 -> zeros)
  const float *Z; int64_t ldz;           // [E, 128]
  const float *W; int64_t ldw;           // [128, ldw >= 256] = [A'^T | B'^T]: B_g[k][j] = W[k][j] + c W[k][128 + j]
  float *dZ; int64_t ldo;                // output rows, scattered by slot id
  int64_t E;
  const int32_t *slot_edge;              // [tiles * 32] edge id per slot, -1 = padding
  const float *tile_scale;               // [tiles] c_g of the tile's class
  const int32_t *num_tiles;
  const int32_t *dst; const uint8_t *flag;          // [E]: the gathered row of D (-1: none) and its half / sign
  const float *D; int64_t ldd; int64_t num_nodes;   // [N, >= 256]
  const float *base; int64_t ldb;                   // upstream rows [rowsR, >= 128] or NULL
  const int32_t *rmap; int64_t rowsR;               // base row of edge e = rmap[e] (< 0: none), or NULL: row e
  float s0, s1;
  float *partialW; int64_t pstride;                 // [gridDim.x] partials of [128, 256] = [T | B]
};

__global__ __launch_bounds__(kThreadsW, 1) void dzw_k(const DzwArgs p) {
  extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
  float *const scr_all = reinterpret_cast<float *>(lds + kImgB);
  uint32_t *const rowA = reinterpret_cast<uint32_t *>(lds + kImgB + kScrB);     // [3][32] each: gathered D row, flag, store row, base row
  uint32_t *const rowB = rowA + 3 * kSub, *const rowC = rowB + 3 * kSub, *const rowR = rowC + 3 * kSub;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool cols = wave >= 4;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int u = threadIdx.x & 255;
  const int srow = u >> 5, scol = u & 31;
  constexpr uint32_t kNone = 0xFFFFFFFFu, kOOB = 0xFFFFF000u;

  const int ntiles = __builtin_amdgcn_readfirstlane(*p.num_tiles);
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  const rsrc_t rs_slot = make_rsrc(p.slot_edge, (uint32_t)ntiles * (kSub * 4u));
  const srsrc_t rs_X = cols ? make_srsrc(p.Z, p.ldz, p.E) : make_srsrc(p.dPre, p.ldp, p.E);
  const uint32_t op_off = cols ? (uint32_t)kOpB : 0u;
  const uint32_t rows4 = (uint32_t)(p.E * 4);
  const rsrc_t rs_dst = make_rsrc(p.dst, rows4);
  const rsrc_t rs_flag = make_rsrc(p.flag, p.flag ? (uint32_t)p.E : 0u);
  const rsrc_t rs_rmap = make_rsrc(p.rmap, p.rmap ? rows4 : 0u);
  float *const pw_ = p.partialW + (int64_t)blockIdx.x * p.pstride;

  if (mine == 0) {                                       // this workgroup's partial: zeros
    float4 *pw4 = reinterpret_cast<float4 *>(pw_);
    for (int m = threadIdx.x; m < kHW * 2 * kHW / 4; m += kThreadsW) pw4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }

  // ---- staging (as h1w_k): rows requested two tiles ahead, staged one tile ahead; the row waves' threads < 32 also carry the
  // per-row scalars of the slot they own
  int id_rows[kSubLoads];
  int id_own = -1, own_staged = -1;
  float4 pre[kSubLoads];
  uint32_t pre_a = 0, pre_b = 0;
  int pre_r = -1;
  auto load_ids = [&](int k) {
    const bool ok = k < mine;
    const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m)
      id_rows[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, (srow + 8 * m) * 4, (int)so, 0) : -1;
    if (!cols && u < kSub) id_own = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, u * 4, (int)so, 0) : -1;
  };
  auto load_row = [&](int m) { pre[m] = sbuf_load4(rs_X, id_rows[m], (uint32_t)scol * 16u); };
  auto load_row_scalars = [&]() {
    if (!cols && u < kSub) {
      const uint32_t eo = id_own >= 0 ? (uint32_t)id_own * 4u : kOOB;
      pre_a = __builtin_amdgcn_raw_buffer_load_b32(rs_dst, (int)eo, 0, 0);
      pre_b = __builtin_amdgcn_raw_buffer_load_b8(rs_flag, id_own >= 0 ? id_own : (int)kOOB, 0, 0);
      pre_r = p.rmap ? (id_own >= 0 ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_rmap, (int)eo, 0, 0) : -1) : id_own;
      own_staged = id_own;
    }
  };
  auto stage_row = [&](int buf, int m) {
    uint2 ph, pm, pl;
    split_pair(pre[m].x, pre[m].y, ph.x, pm.x, pl.x);
    split_pair(pre[m].z, pre[m].w, ph.y, pm.y, pl.y);
    const uint32_t r = (uint32_t)(srow + 8 * m);
    const uint32_t off = op_off + (uint32_t)buf * kBufB + 256u * r + 16u * ((uint32_t)(scol >> 1) ^ swz(r)) + 8u * (uint32_t)(scol & 1);
    *reinterpret_cast<uint2 *>(lds + off) = ph;
    *reinterpret_cast<uint2 *>(lds + off + kPlaneB) = pm;
    *reinterpret_cast<uint2 *>(lds + off + 2 * kPlaneB) = pl;
  };
  auto stage_scalars = [&](int par) {
    if (!cols && u < kSub) {
      const bool ok = own_staged >= 0;
      rowA[par * kSub + u] = ok ? pre_a : kNone;
      rowB[par * kSub + u] = pre_b;
      rowC[par * kSub + u] = ok ? (uint32_t)own_staged : kNone;
      rowR[par * kSub + u] = (ok && p.base && pre_r >= 0) ? (uint32_t)pre_r : kNone;
    }
  };
  // the class structure of the range, 64 tiles at a time (both roles): lane l keeps the coefficient of tile 64 c + l, bit l of
  // `starts` says "tile 64 c + l begins a new class"
  float sv = 0.f;
  unsigned long long starts = 0;
  auto class_chunk = [&](int k) {
    const float last = __shfl(sv, 63);
    sv = k + lane < mine ? p.tile_scale[lo + k + lane] : 0.f;
    float up = __shfl_up(sv, 1);
    if (lane == 0) up = k > 0 ? last : sv;
    starts = __ballot(k + lane < mine && __float_as_uint(sv) != __float_as_uint(up));
  };

  load_ids(0);
#pragma unroll
  for (int m = 0; m < kSubLoads; ++m) load_row(m);
  load_row_scalars();
  load_ids(1);

  if (cols) {
    // ================================================================== the column product Z^T dPre (waves 4-7)
    const int pw = wave & 1, qw = (wave >> 1) & 1;
    const int G16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    auto tr_base = [&](int half) {
      const uint32_t row0 = (uint32_t)(8 * (G16 >> 1) + q4);
      const uint32_t ch0 = (uint32_t)(8 * half + 2 * (G16 & 1) + (pp >> 1));
      const uint32_t sw0 = ((uint32_t)q4 << 2) | (uint32_t)(2 * (G16 >> 1));
      return 256u * row0 + 16u * (ch0 ^ sw0) + 8u * (uint32_t)(pp & 1);
    };
    const uint32_t tA = (uint32_t)kOpB + tr_base(pw), tB = tr_base(qw);      // A: Z (operand 1); B: dPre (operand 0)
    auto tr_frag = [&](uint32_t base, int buf, int plane, int blk, int kg, Frag8 &f) {
      const uint32_t o0 = (base ^ (uint32_t)(64 * blk)) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg);
      const uint32_t o1 = (base ^ (uint32_t)(64 * blk) ^ 16u) + (uint32_t)(buf * kBufB + plane * kPlaneB + 4096 * kg + 1024);
      const v4s x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o0));
      const v4s y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(lds + o1));
      f.v = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto tr_split = [&](uint32_t base, int buf, int blk, int kg, Split8 &sp) {
      tr_frag(base, buf, 0, blk, kg, sp.hi);
      tr_frag(base, buf, 1, blk, kg, sp.mid);
      tr_frag(base, buf, 2, blk, kg, sp.lo);
    };
    f32x16 wacc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) wacc[i][j][r] = 0.f;
    };
    // accumulator (i, j, r): output row 64 pw + 32 i + (r & 3) + 8 (r >> 2) + 4 h, column 64 qw + 32 j + li of T (B: + 128)
    const rsrc_t rs_pw = make_rsrc(pw_, (uint32_t)(kHW * 2 * kHW * 4));
    const uint32_t e_voff = (uint32_t)((64 * pw + 4 * h) * (2 * kHW) + 64 * qw + li) * 4u;
    bool emitted = false;
    auto emit = [&](float c) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {                         // a row block at a time: 32 + 32 reads in flight, then the adds
        float t0[2][16], b0[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int so = ((32 * i + (r & 3) + 8 * (r >> 2)) * (2 * kHW) + 32 * j) * 4;
            t0[j][r] = emitted ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_pw, (int)e_voff, so, 0)) : 0.f;
            b0[j][r] = emitted ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_pw, (int)e_voff, so + kHW * 4, 0)) : 0.f;
          }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int so = ((32 * i + (r & 3) + 8 * (r >> 2)) * (2 * kHW) + 32 * j) * 4;
            const float a = wacc[i][j][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(emitted ? a + t0[j][r] : a), rs_pw, (int)e_voff, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(emitted ? c * a + b0[j][r] : c * a), rs_pw, (int)e_voff, so + kHW * 4, 0);
          }
      }
      emitted = true;
    };
    auto cols_step = [&](int k) {
      const int buf = k & 1;
      int act = 0;
      // eight blocks (kg, jb, ib); the fragments the next block needs are requested BEFORE this block's MFMAs
      Split8 fa[2], fb, nfa[2], nfb;
      tr_split(tA, buf, 0, 0, fa[0]);
      tr_split(tA, buf, 1, 0, fa[1]);
      tr_split(tB, buf, 0, 0, fb);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int kg = b >> 2, jb = (b >> 1) & 1, ib = b & 1;
        if (ib == 1 && b + 1 < 8) {                             // the next block starts a new B fragment (and, at b == 3, new A fragments)
          tr_split(tB, buf, jb ^ 1, jb == 1 ? kg + 1 : kg, nfb);
          if (b == 3) { tr_split(tA, buf, 0, 1, nfa[0]); tr_split(tA, buf, 1, 1, nfa[1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        wacc[ib][jb] = mfma_x6(fa[ib], fb, wacc[ib][jb]);
        __builtin_amdgcn_sched_barrier(0);
        if (act < 4) stage_row(buf ^ 1, act);
        else load_row(act - 4);
        ++act;
        if (ib == 1 && b + 1 < 8) {
          fb = nfb;
          if (b == 3) { fa[0] = nfa[0]; fa[1] = nfa[1]; }
        }
      }
      load_ids(k + 3);
    };
    zero_acc();
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(0, m);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(m);       // tile 1
    load_ids(2);
    lds_barrier();
    float cur = 0.f;
    for (int k = 0; k < mine; ++k) {
      if ((k & 63) == 0) class_chunk(k);
      if ((starts >> (k & 63)) & 1ull) { emit(cur); zero_acc(); }     // tile k begins a new class: the finished class's total goes out
      cur = __shfl(sv, k & 63);
      cols_step(k);
      lds_barrier();
    }
    emit(cur);
    return;
  }

  // ==================================================================== the row product dPre W_g^T (waves 0-3): mfma_typed<TEPI_DZ>
  const int cs = wave & 3;
  const int col = 32 * cs + li;
  float *const scr = scr_all + cs * (32 * kScrStride);
  const int lrow = lane >> 3, c4 = 32 * cs + (lane & 7) * 4;
  const uint32_t col4 = (uint32_t)c4 * 4u;
  constexpr uint32_t kRowBytes = kHW * 4u;
  const srsrc_t rs_C = make_srsrc(p.dZ, p.ldo, p.E);
  const srsrc_t rs_T = make_srsrc(p.D, p.ldd, p.num_nodes);
  const srsrc_t rs_R = make_srsrc(p.base, p.base ? p.ldb : (int64_t)kHW, p.base ? (p.rmap ? p.rowsR : p.E) : 0);
  constexpr int kGroups = 8;
  Split8 B6[kGroups];
  const rsrc_t rs_W = make_rsrc(p.W, (uint32_t)(kHW * p.ldw * 4));
  const uint32_t w_first = (uint32_t)((int64_t)64 * h * p.ldw + col) * 4u;
  const uint32_t w_step = __builtin_amdgcn_readfirstlane((int)(p.ldw * 4));
  auto load_panel = [&](float c) {
    uint32_t off;
    asm volatile("v_mov_b32 %0, %1" : "=v"(off) : "v"(w_first));
#pragma unroll
    for (int s0 = 0; s0 < 64; s0 += 8) {
      float w0[8], w1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w0[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off, (int)((s0 + j) * w_step), 0));
        w1[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off + (int)kRowBytes, (int)((s0 + j) * w_step), 0));
      }
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = w0[j] + c * w1[j];
      split8(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), B6[s0 / 8]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const uint32_t a0 = 256u * (uint32_t)li + 16u * ((uint32_t)(8 * h) ^ swz((uint32_t)li));
  f32x16 acc;
  float4 g0[4], g1[4];
  auto fetch_operand = [&](int par, int k) {               // rows 8 k + lrow of the tile: the gathered D row's half, the base row
    const int rr = 8 * k + lrow;
    g0[k] = sbuf_load4(rs_T, (int)rowA[par * kSub + rr], col4 + (rowB[par * kSub + rr] ? kRowBytes : 0u));
    g1[k] = sbuf_load4(rs_R, (int)rowR[par * kSub + rr], col4);
  };
  auto rows_step = [&](int k, int par3) {
    const int buf = k & 1, nxt3 = par3 == 2 ? 0 : par3 + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char *img = lds + (uint32_t)buf * kBufB;
    Frag8 ah, am, al;
    ah.v = *reinterpret_cast<const bf16x8 *>(img + a0);
    am.v = *reinterpret_cast<const bf16x8 *>(img + kPlaneB + a0);
    al.v = *reinterpret_cast<const bf16x8 *>(img + 2 * kPlaneB + a0);
    auto action = [&](int i) {
      if (i < 4) {}
      else if (i < 8) stage_row(buf ^ 1, i - 4);
      else if (i == 8) stage_scalars(nxt3);
      else if (i < 13) load_row(i - 9);
      else if (i == 13) load_row_scalars();
      else if (i == 14) load_ids(k + 3);
    };
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      Frag8 nh = ah, nm = am, nl = al;
      if (g + 1 < kGroups) {
        const uint32_t an = a0 ^ (uint32_t)(16 * (g + 1));
        nh.v = *reinterpret_cast<const bf16x8 *>(img + an);
        nm.v = *reinterpret_cast<const bf16x8 *>(img + kPlaneB + an);
        nl.v = *reinterpret_cast<const bf16x8 *>(img + 2 * kPlaneB + an);
      }
      const Split8 &bb = B6[g];
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bb.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.lo.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.mid.v, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      action(2 * g);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bb.hi.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.mid.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bb.hi.v, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      action(2 * g + 1);
      ah = nh; am = nm; al = nl;
    }
  };
  auto rows_epilogue = [&](int par) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * h) * kScrStride + li] = acc[r];
    const int nxt = par == 2 ? 0 : par + 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = 8 * k + lrow;
      float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
      const float sg = rowB[par * kSub + rr] ? p.s1 : p.s0;
      v.x += g1[k].x + sg * g0[k].x; v.y += g1[k].y + sg * g0[k].y;
      v.z += g1[k].z + sg * g0[k].z; v.w += g1[k].w + sg * g0[k].w;
      sbuf_store4(v, rs_C, (int)rowC[par * kSub + rr], col4);
      fetch_operand(nxt, k);     // this chunk's operand registers are free: the NEXT tile's rows into them (its scalars were staged before the barrier)
    }
  };
#pragma unroll
  for (int m = 0; m < kSubLoads; ++m) stage_row(0, m);
  stage_scalars(0);
#pragma unroll
  for (int m = 0; m < kSubLoads; ++m) load_row(m);         // tile 1
  load_row_scalars();
  load_ids(2);
  lds_barrier();
#pragma unroll
  for (int c = 0; c < 4; ++c) fetch_operand(0, c);         // tile 0's epilogue operands (later tiles: by the epilogue before)
  int par3 = 0;
  float c_have = 0.f;
  bool have_panel = false;
  for (int k = 0; k < mine; ++k) {
    if ((k & 63) == 0) class_chunk(k);
    const float c = __shfl(sv, k & 63);
    if (!have_panel || __float_as_uint(c) != __float_as_uint(c_have)) {      // W_g is built once per class segment of this range
      load_panel(c);
      c_have = c;
      have_panel = true;
    }
    rows_step(k, par3);
    lds_barrier();
    rows_epilogue(par3);
    par3 = par3 == 2 ? 0 : par3 + 1;
  }
}

constexpr int kDzwLdsBytes = kImgB + kScrB + 4 * 3 * kSub * 4;
inline bool dzw_lds_ready() {
  static bool done[kMaxDevicesW2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesW2) dev = 0;
  if (done[dev]) return true;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dzw_k), hipFuncAttributeMaxDynamicSharedMemorySize, kDzwLdsBytes);
  if (e != hipSuccess) { set_last_hip_error(e); return false; }
  done[dev] = true;
  return true;
}

inline unsigned h1w_blocks(int64_t tiles_bound) {
  const int64_t cap = 256;                               // one workgroup per CU
  return (unsigned)(tiles_bound < cap ? (tiles_bound > 0 ? tiles_bound : 1) : cap);
}

constexpr int kMaxDevicesW = 64;
inline bool h1w_lds_ready() {
  static bool done[kMaxDevicesW] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesW) dev = 0;
  if (done[dev]) return true;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&h1w_k), hipFuncAttributeMaxDynamicSharedMemorySize, kH1WLdsBytes);
  if (e != hipSuccess) { set_last_hip_error(e); return false; }
  done[dev] = true;
  return true;
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int64_t dmp_bwd_h1_w_blocks(int64_t tiles_bound) { return (int64_t)h1w_blocks(tiles_bound); }

int dmp_bwd_h1_w(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                 const int32_t *slot_edge, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, float slope,
                 float *dPre, int64_t ldg, float *partial, float *partial_rows, float *partial_w, void *stream) {
  if (H != 128 || g_exact_fp32) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!partial || !partial_w || !aligned16(partial) || !aligned16(partial_w) || (partial_rows && !aligned16(partial_rows))) return DMP_ERR_BAD_ARG;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (E == 0) {
    const size_t rows = (size_t)h1w_blocks(tiles_bound);
    if (hipMemsetAsync(partial, 0, sizeof(float) * H * rows, st) != hipSuccess) return DMP_ERR_HIP;
    if (partial_rows && hipMemsetAsync(partial_rows, 0, sizeof(float) * H * rows, st) != hipSuccess) return DMP_ERR_HIP;
    if (hipMemsetAsync(partial_w, 0, sizeof(float) * H * H * rows, st) != hipSuccess) return DMP_ERR_HIP;
    return DMP_OK;
  }
  if (!dO || !W2 || !H1 || !dPre || !slot_edge || !num_tiles || ldo < H || ldw < H || ldh < H || ldg < H) return DMP_ERR_BAD_ARG;
  if (ldo % 4 || ldh % 4 || ldg % 4 || !aligned16(dO) || !aligned16(H1) || !aligned16(dPre)) return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldo) || !stride_ok(ldh) || !stride_ok(ldg) || E >= ((int64_t)1 << 30) || tiles_bound * kSub * 4 >= ((int64_t)1 << 32) - 8192 ||
      (int64_t)H * ldw * 4 >= ((int64_t)1 << 32) - 8192 || !fits4g(E, ldo) || !fits4g(E, ldh) || !fits4g(E, ldg))
    return DMP_ERR_UNSUPPORTED;                           // (arrays of 4 GiB and more: the two-launch form has the 64-bit row kernels)
  if (!h1w_lds_ready()) return DMP_ERR_HIP;
  H1WArgs a{};
  a.dO = dO; a.ldo = ldo; a.H1 = H1; a.ldh = ldh; a.W2 = W2; a.ldw = ldw; a.dPre = dPre; a.ldg = ldg; a.E = E;
  a.slot_edge = slot_edge; a.num_tiles = num_tiles; a.slope = slope; a.partial = partial; a.partialA = partial_rows; a.partialW = partial_w;
  h1w_k<<<h1w_blocks(tiles_bound), kThreadsW, kH1WLdsBytes, st>>>(a);
  return check_launch();
}


int dmp_bwd_z_w(const float *dPre, int64_t ldp, const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *D, int64_t ldd,
                int64_t num_nodes, const float *base, int64_t ldb, const int32_t *dst, const uint8_t *flag, float s0, float s1,
                const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H,
                const int32_t *base_map, int64_t base_rows, float *dZ, int64_t ldo, float *partial_w, void *stream) {
  if (H != 128 || g_exact_fp32) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || num_nodes < 0 || tiles_bound < 0 || !partial_w || !aligned16(partial_w)) return DMP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (E == 0) return hipMemsetAsync(partial_w, 0, sizeof(float) * H * 2 * H * (size_t)h1w_blocks(tiles_bound), st) == hipSuccess ? DMP_OK : DMP_ERR_HIP;
  if (!dPre || !Z || !W || !D || !dst || !slot_edge || !tile_scale || !num_tiles || !dZ || ldp < H || ldz < H || ldw < 2 * H || ldd < 2 * H ||
      ldo < H || (base && ldb < H))
    return DMP_ERR_BAD_ARG;
  if (base_map && (!base || base_rows < 0)) return DMP_ERR_BAD_ARG;
  if (ldp % 4 || ldz % 4 || ldo % 4 || ldd % 4 || (base && ldb % 4) || !aligned16(dPre) || !aligned16(Z) || !aligned16(dZ) || !aligned16(D) ||
      (base && !aligned16(base)))
    return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldp) || !stride_ok(ldz) || !stride_ok(ldo) || !stride_ok(ldd) || (base && !stride_ok(ldb)) || E >= ((int64_t)1 << 30) ||
      num_nodes >= ((int64_t)1 << 31) || tiles_bound * kSub * 4 >= ((int64_t)1 << 32) - 8192 || (int64_t)H * ldw * 4 >= ((int64_t)1 << 32) - 8192 ||
      !fits4g(E, ldp) || !fits4g(E, ldz) || !fits4g(E, ldo) || (base && !fits4g(base_map ? base_rows : E, ldb)) || !fits4g(num_nodes, ldd))
    return DMP_ERR_UNSUPPORTED;
  if (!dzw_lds_ready()) return DMP_ERR_HIP;
  DzwArgs a{};
  a.dPre = dPre; a.ldp = ldp; a.Z = Z; a.ldz = ldz; a.W = W; a.ldw = ldw; a.dZ = dZ; a.ldo = ldo; a.E = E; a.slot_edge = slot_edge;
  a.tile_scale = tile_scale; a.num_tiles = num_tiles; a.dst = dst; a.flag = flag; a.D = D; a.ldd = ldd; a.num_nodes = num_nodes;
  a.base = base; a.ldb = base ? ldb : H; a.rmap = base_map; a.rowsR = base_rows; a.s0 = s0; a.s1 = s1;
  a.partialW = partial_w; a.pstride = (int64_t)H * 2 * H;
  dzw_k<<<h1w_blocks(tiles_bound), kThreadsW, kDzwLdsBytes, st>>>(a);
  return check_launch();
}

int64_t dmp_atb2_blocks(int64_t tiles_bound, int num_jobs) { return (int64_t)atb2_blocks(tiles_bound, num_jobs); }

int dmp_atb2_jobs(const dmp_atb2_job *jobs, int num_jobs, const int32_t *slot_row, const float *tile_scale, const int32_t *num_tiles,
                  int64_t tiles_bound, int64_t rows, int H, void *stream) {
  if (H != 128 || g_exact_fp32) return DMP_ERR_UNSUPPORTED;
  if (!jobs || num_jobs < 1 || num_jobs > kAtb2MaxJobs || rows < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!slot_row || !tile_scale || !num_tiles) return DMP_ERR_BAD_ARG;
  if (rows >= ((int64_t)1 << 30) || tiles_bound * kSub * 4 >= ((int64_t)1 << 32) - 8192) return DMP_ERR_UNSUPPORTED;
  Atb2Args a{};
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_atb2_job &j = jobs[i];
    if (!j.partial_T || j.ldp < H || (rows > 0 && (!j.Z || !j.D || j.ldz < H || j.ldd < H))) return DMP_ERR_BAD_ARG;
    if (j.ldz % 4 || j.ldd % 4 || (rows > 0 && (!aligned16(j.Z) || !aligned16(j.D)))) return DMP_ERR_UNSUPPORTED;
    if (!stride_ok(j.ldz) || !stride_ok(j.ldd) || !fits4g(rows, j.ldz) || !fits4g(rows, j.ldd)) return DMP_ERR_UNSUPPORTED;
    a.job[i].Z = j.Z; a.job[i].ldz = j.ldz; a.job[i].D = j.D; a.job[i].ldd = j.ldd; a.job[i].pT = j.partial_T; a.job[i].pB = j.partial_B;
    a.job[i].pstride = j.partial_stride; a.job[i].ldp = j.ldp;
  }
  a.E = rows; a.slot_edge = slot_row; a.tile_scale = tile_scale; a.num_tiles = num_tiles;
  if (!atb2_lds_ready()) return DMP_ERR_HIP;
  atb2_k<<<dim3(atb2_blocks(tiles_bound, num_jobs), (unsigned)num_jobs), kThreadsW, kAtb2LdsBytes, (hipStream_t)stream>>>(a);
  return check_launch();
}

}  // extern "C"
