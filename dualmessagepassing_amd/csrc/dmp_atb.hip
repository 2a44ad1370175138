// Weight gradients of the edge chain as tall-skinny products  T = Z^T D  ([E,H]^T [E,H] -> [H,H] per output block,
// H = 128 or 64; the contraction runs over the edges), fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   dmp_atb_typed   T = sum_e Z[e]^T dPre[e]  and  B = sum_e c_e Z[e]^T dPre[e]  over the class-sorted tile
//                   list (c_e = the coefficient of e's degree class): dA' and dB' of the class-typed edge
//                   chain (dmpnn.py:144-156 backward), one product's worth of MFMAs for both.
//   dmp_rel_atb     T_t = sum over the edges e of relation type t of w_e X[src e]^T D[dst e]  for every type t: the
//                   per-type weight gradients of the relational layers (rgcn.py:98-123 backward); rows gathered through
//                   two slot arrays, blockIdx.y = type, the type's tile range split over blockIdx.x.
//   dmp_atb_rows    T = (g (.) Z)^T D  and the column sums of g (.) Z  over plain row tiles: the gradient of
//                   the second edge Linear with the layer's edge gate and its bias gradient fused in
//                   (dmpnn.py:263-273 backward: dO = gate * dOut, dW2 = dO^T H1, db2 = sum dO).
//
// A workgroup (4 waves) owns a contiguous range of 32-row tiles and keeps the running [H,H] total in
// registers: wave (p, q) holds the H/2 x H block of output rows (H/2)p.. for the tile's rows 16q..16q+15
// (H = 128: 8 accumulators, H = 64: 2; the contraction is split over the two wave pairs so that every LDS operand
// read feeds 8/6 MFMAs instead of 4/5); the two halves are added through LDS and the total goes to the workgroup's
// partial with 16-byte accesses.  Typed: a class that ends inside the range emits  T += acc, B += c * acc
// and restarts the accumulators (and the load pipeline) -- ranges are contiguous in the class-sorted
// order, so this happens at most (classes in use) times per launch, and no coefficient-weighted second
// accumulator set is needed.  The per-workgroup partials are summed by dmp_reduce_partials: every sum
// has a fixed order, bit-stable for a given tile list.
#include <type_traits>

#include "dmp_mfma_common.h"

namespace dmp {
namespace {

struct AtbArgs {
  const float *Z; int64_t ldz;
  const float *D; int64_t ldd;
  int64_t E;                       // rows of Z and D
  const int32_t *slot_edge;        // typed: [tiles * 32] row of every slot, -1 = padding
  const float *tile_scale;         // typed: [tiles] coefficient of the tile's class
  const int32_t *num_tiles;        // typed: [1] tiles in use (device)
  int plain_tiles;                 // rows variant: ceil(E / 32)
  const float *gate;               // rows variant: row scale of Z or NULL
  float *pT, *pB;                  // [gridDim.x] partials, `pstride` floats apart, rows `ldp` floats apart
  int64_t pstride; int ldp;
  float *pCS;                      // rows variant: [gridDim.x, cs_ld] column sums of g (.) Z
  int nb, cs_ld;                   // rows variant: blockIdx.y = a * nb + b picks the 128-column blocks a of Z and b of D
  const int32_t *slot_d;           // relational variant: row of D per slot (slot_edge: row of Z), -1 = padding
  const float *slot_scale;         // relational variant: [tiles * 32] scale of Z's row per slot, or NULL
  const int32_t *type_tile_ptr;    // relational variant: [types + 1] first tile of every type
  const uint32_t *rowmask;         // rows variant with a gate: bit r of rowmask[t] == 0 -> row 32 t + r has gate 0 and is not fetched
  int rows_x6;                     // rows variant: 1 = on the bf16 pipe (dmp_atb_rows_masked picks it)
};

enum { ATB_ROWS = 0, ATB_TYPED = 1, ATB_REL = 2, ATB_PLAIN = 3 };   // PLAIN: the rows variant without a gate and without column sums (compile time: the gated form
                                                                    // spills on the bf16 pipe; a 0 / 1 gate is fully expressed by the row mask)

// X6: the products on the bf16 matrix pipe (dmp_mfma_common.h, "bf16x6": fp32-accurate, 6 x 32 instead of 8 x 64 matrix-pipe
// cycles per 16 contracted rows).  Wave (p, q) contracts exactly one 16-row k-group per tile (rows 16q ..): a lane's
// fragment is 8 consecutive ROWS (8h ..) of one column, read as 8 ds_read_b32 down the fp32 tile and split in registers
// (the same 48 LDS reads per wave and tile as the f32 form's 8 k-steps x 6 operands).
// BIG: an operand array of 4 GiB or more: rows through 64-bit pointers (dmp_mfma_common.h).
template <int MODE, int H, bool X6, bool BIG = false>
__device__ __forceinline__ void atb_body(const AtbArgs &p, const int ya, const int yb) {
  constexpr bool TYPED = MODE == ATB_TYPED, REL = MODE == ATB_REL;   // REL: the control flow of the rows variant over gathered rows
  constexpr bool PLAIN = MODE == ATB_PLAIN, ROWSLIKE = MODE == ATB_ROWS || PLAIN;
  // two tile buffers (Zs | Ds, 2 x 32 x (H+4) floats each) = 67584 bytes at H = 128; emit() reuses the first H*H floats for the total
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kStride = H + 4, kQ = H / 4, kPass = kGroupThreads / kQ, NL = kSub / kPass;   // float4 per row, rows per load pass, passes
  // QUAD: wave (p, c) owns a QUADRANT of the [H, H] total and contracts both 16-row k-groups of a tile (4 accumulators at
  // H = 128); else wave (p, q) owns the half (H/2) p .. of the rows, ALL columns, and contracts k-group q (8 accumulators; the
  // two q's totals meet in LDS at the end).  The class-typed product takes the quadrants: with 8 accumulators its bf16x6
  // form needed 347 registers and spilled 91 of them to scratch (round 5: 100 -> 80 us at bench.py's shape); the row forms
  // (node side: every tile of the rows is walked, masked or not) are bound by the fragment splits, which the quadrants do a
  // third more of (8 instead of 6 fragments per wave and tile: 89 -> 101 us) -- they keep the halves.
  constexpr bool QUAD = MODE == ATB_TYPED;
  constexpr int NI = H / 64, NJ = QUAD ? H / 64 : H / 32;                                       // accumulator blocks per wave: NI x NJ
  constexpr int KG = QUAD ? 2 : 1;                                                              // k-groups a wave contracts per tile
  constexpr int kTile = kSub * kStride, kBuf = 2 * kTile;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pw = wave & 1, qw = wave >> 1;                 // QUAD: rows (H/2) pw .., columns (H/2) qw ..; else rows (H/2) pw .., k-group qw
  const int kq = QUAD ? 0 : qw, cq = QUAD ? qw : 0;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, gtid = threadIdx.x;
  const uint32_t colA = (uint32_t)(gtid % kQ) * 16u;
  constexpr uint32_t kOOB = 0xFFFFF000u;
  // rows by INDEX (structured descriptors, dmp_mfma_common.h): any array size; the column block is in the base
  const srsrc_t rs_Z = make_srsrc(p.Z + H * ya, p.ldz, p.E);
  const srsrc_t rs_D = make_srsrc(p.D + H * yb, p.ldd, p.E);
  const bool gated = REL ? p.slot_scale != nullptr : (!TYPED && !PLAIN && p.gate != nullptr);
  const rsrc_t rs_G = make_rsrc(p.gate, !REL && gated ? (uint32_t)(p.E * 4) : 0u);
  const int first = REL ? __builtin_amdgcn_readfirstlane(p.type_tile_ptr[blockIdx.y]) : 0;
  const int ntiles = REL ? __builtin_amdgcn_readfirstlane(p.type_tile_ptr[blockIdx.y + 1]) - first
                         : TYPED ? __builtin_amdgcn_readfirstlane(*p.num_tiles) : p.plain_tiles;
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = first + (int)blockIdx.x * chunk;
  const int hi = lo + chunk < first + ntiles ? lo + chunk : first + ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  const uint32_t slot_bytes = REL ? (uint32_t)p.plain_tiles * (kSub * 4u) : TYPED ? (uint32_t)ntiles * (kSub * 4u) : 0u;
  const rsrc_t rs_slot = make_rsrc(p.slot_edge, slot_bytes);
  const rsrc_t rs_slotD = make_rsrc(p.slot_d, REL ? slot_bytes : 0u);
  const rsrc_t rs_slotS = make_rsrc(p.slot_scale, REL && gated ? slot_bytes : 0u);
  float *pt = p.pT + (int64_t)(REL ? blockIdx.y * gridDim.x + blockIdx.x : blockIdx.x) * p.pstride + (int64_t)ya * H * p.ldp + yb * H;
  float *const pcs = MODE == ATB_ROWS && p.pCS && yb == 0 ? p.pCS + (int64_t)blockIdx.x * p.cs_ld + ya * H : nullptr;
  float *pb = (TYPED && p.pB) ? p.pB + (int64_t)blockIdx.x * p.pstride : nullptr;   // NULL: the plain total alone

  f32x16 acc[NI][NJ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  int id_rows[NL], id_rowsD[NL];
  float sc_rows[NL];
  float4 preZ[2][NL], preD[2][NL];                         // two sets of prefetched rows: tiles of even / odd pipeline phase
  float preG[2][NL];
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);             // gated: this thread's 4 columns of sum g (.) Z
  auto load_ids = [&](int k) {
    const bool ok = k < mine;
    if (TYPED || REL) {
      const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
      for (int m = 0; m < NL; ++m) {
        id_rows[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, ((gtid / kQ) + kPass * m) * 4, (int)so, 0) : -1;
        if (REL) {
          id_rowsD[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slotD, ((gtid / kQ) + kPass * m) * 4, (int)so, 0) : -1;
          sc_rows[m] = 1.f;
          if (gated) sc_rows[m] = ok ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_slotS, ((gtid / kQ) + kPass * m) * 4, (int)so, 0)) : 0.f;
        }
      }
    } else {
      // rows under a zero gate (dmp_row_mask_bits) are not fetched: they contribute gate * z = 0 to every sum
      const uint32_t mk = (ROWSLIKE && p.rowmask && ok) ? p.rowmask[lo + k] : 0xffffffffu;
#pragma unroll
      for (int m = 0; m < NL; ++m) {
        const int rt = (gtid / kQ) + kPass * m;
        const int64_t r = (int64_t)(lo + k) * kSub + rt;
        id_rows[m] = ok && r < p.E && ((mk >> rt) & 1u) ? (int)r : -1;
      }
    }
  };
  auto load_row = [&](auto set, int m) {                    // rows of slice m of the tile whose ids are in id_rows
    constexpr int S = decltype(set)::value;
    const bool ok = id_rows[m] >= 0;
    preZ[S][m] = row_load4<BIG>(rs_Z, p.Z + H * ya, p.ldz, id_rows[m], colA);          // -1: zeros
    const int idD = REL ? id_rowsD[m] : id_rows[m];
    preD[S][m] = row_load4<BIG>(rs_D, p.D + H * yb, p.ldd, ok ? idD : -1, colA);
    if (REL) preG[S][m] = sc_rows[m];
    if (MODE == ATB_ROWS) {
      preG[S][m] = 1.f;
      if (gated) preG[S][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_G, ok ? (int)((uint32_t)id_rows[m] * 4u) : (int)kOOB, 0, 0));
    }
  };
  auto load_rows = [&](auto set) {
#pragma unroll
    for (int m = 0; m < NL; ++m) load_row(set, m);
  };
  auto stage_row = [&](auto set, int m) {                   // slice m of row set S -> tile buffer S
    constexpr int S = decltype(set)::value;
    const int o = S * kBuf + ((gtid / kQ) + kPass * m) * kStride + (gtid % kQ) * 4;
    float4 z = preZ[S][m];
    if (REL) {
      if (gated) z = make_float4(z.x * preG[S][m], z.y * preG[S][m], z.z * preG[S][m], z.w * preG[S][m]);
    } else if (!TYPED && gated) {
      z = make_float4(z.x * preG[S][m], z.y * preG[S][m], z.z * preG[S][m], z.w * preG[S][m]);
      cs.x += z.x; cs.y += z.y; cs.z += z.z; cs.w += z.w;
    } else if (!TYPED && !PLAIN && p.pCS) {
      cs.x += z.x; cs.y += z.y; cs.z += z.z; cs.w += z.w;
    }
    *reinterpret_cast<float4 *>(&smem[o]) = z;
    *reinterpret_cast<float4 *>(&smem[o + kTile]) = preD[S][m];
  };
  auto stage = [&](auto set) {
#pragma unroll
    for (int m = 0; m < NL; ++m) stage_row(set, m);
  };
  // One tile of pipeline phase PH (tiles alternate between the two LDS buffers / row sets, counted from the last
  // start_at): the MFMAs of tile k (buffer PH) with, in their shadow, the staging of tile k+1 into the other
  // buffer (k-steps 0-3), the row requests of tile k+3 into the set just staged (k-steps 4-7) and the id
  // requests of tile k+4 -- rows are requested two tiles before they are staged.
  // k-step s of wave (p, q): row 16q + s of the tile (lanes 0-31) paired with row 16q + 8 + s (lanes 32-63);
  // A operands: columns 64p + li and 64p + 32 + li of Z, B operands: columns 32j + li of D.
  // The operands of step s+1 are requested before the MFMAs of step s.
  auto tile_step = [&](int k, auto phase) {
    constexpr int PH = decltype(phase)::value;
    std::integral_constant<int, PH ^ 1> other;
    const float *zp = &smem[PH * kBuf + (16 * kq + 8 * h) * kStride + (H / 2) * pw + li];
    const float *dp = &smem[PH * kBuf + kTile + (16 * kq + 8 * h) * kStride + (H / 2) * cq + li];
    if (X6) {
      // fragments: element j of lane (li, h) = row 16q + 8h + j of the operand's column; the staging of tile k+1, the
      // row requests of tile k+3 and the id requests of tile k+4 are spread between the NI x NJ blocks' MFMAs
      auto frag = [&](const float *col, Split8 &f) {            // two rows at a time: no eight-float temporary
#pragma unroll
        for (int t = 0; t < 4; ++t)
          split_pair(col[(2 * t) * kStride], col[(2 * t + 1) * kStride], f.hi.u[t], f.mid.u[t], f.lo.u[t]);
      };
      int act = 0;
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {                        // QUAD: the tile's two 16-row k-groups, one after the other
        Split8 fa[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) frag(zp + 16 * kg * kStride + 32 * i, fa[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          Split8 fb;                                           // one B fragment at a time: NI + 1 fragments live
          frag(dp + 16 * kg * kStride + 32 * j, fb);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            acc[i][j] = mfma_x6(fa[i], fb, acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            // 2 NL actions over KG NI NJ blocks (H = 128: 8 actions, 8 blocks; H = 64: 4 actions, 2 blocks)
            constexpr int kPerBlock = (2 * NL + KG * NI * NJ - 1) / (KG * NI * NJ);
#pragma unroll
            for (int u = 0; u < kPerBlock; ++u, ++act) {
              if (act < NL) stage_row(other, act < NL ? act : 0);
              else if (act < 2 * NL) load_row(other, act < 2 * NL ? act - NL : 0);
            }
          }
        }
      }
      load_ids(k + 4);
      return;
    }
    // k-step s of the wave: row 16 (s / 8) + (s % 8) of the tile (lanes 0-31) paired with that row + 8 (lanes 32-63)
    auto srow = [](int s) { return (16 * (s >> 3) + (s & 7)); };
    float a[NI], b[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = zp[32 * i];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = dp[32 * j];
#pragma unroll
    for (int s = 0; s < 8 * KG; ++s) {
      float x[NI], y[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) x[i] = s + 1 < 8 * KG ? zp[srow(s + 1) * kStride + 32 * i] : a[i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) y[j] = s + 1 < 8 * KG ? dp[srow(s + 1) * kStride + 32 * j] : b[j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s < NL) stage_row(other, s < NL ? s : 0);
      else if (s < 2 * NL) load_row(other, s < 2 * NL ? s - NL : 0);
      if (s == 8 * KG - 1) load_ids(k + 4);
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = x[i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = y[j];
    }
  };
  // accumulator (i, j, r) of wave (p, c): output row (H/2)p + 32i + (r&3) + 8(r>>2) + 4h, column (H/2)c + 32j + li
  auto out_index = [&](int i, int j, int r) { return ((H / 2) * pw + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * H + (H / 2) * cq + 32 * j + li; };

  // Emit the accumulators: the two row halves are added through LDS (which must be free: the staged tile is
  // given up), then every thread adds its 16 float4 of the [128,128] total to the workgroup's partial --
  // T (+)= total, B (+)= c * total; the first emission stores, later ones (typed: one per class that ends
  // inside the range) read back what the SAME thread wrote before.
  bool emitted = false;
  auto emit = [&](float c) {
    __syncthreads();
    if (QUAD || qw == 1) {      // QUAD: every wave holds its own quadrant of the total: straight to LDS (the 16-byte stores below read whole rows)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) smem[out_index(i, j, r)] = acc[i][j][r];
    }
    __syncthreads();
    if (!QUAD) {
      if (qw == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = out_index(i, j, r);
              smem[o] = acc[i][j][r] + smem[o];
            }
      }
      __syncthreads();
    }
#pragma unroll 4
    for (int m = 0; m < H * H / (4 * kGroupThreads); ++m) {
      const int l = (m * kGroupThreads + gtid) * 4;         // element (l / H, l % H) of the total
      const float4 v = *reinterpret_cast<const float4 *>(&smem[l]);
      const int o = (l / H) * p.ldp + (l % H);
      float4 t = v, b = make_float4(c * v.x, c * v.y, c * v.z, c * v.w);
      if (emitted) {
        const float4 t0 = *reinterpret_cast<const float4 *>(pt + o);
        t.x += t0.x; t.y += t0.y; t.z += t0.z; t.w += t0.w;
        if (TYPED && pb) {
          const float4 b0 = *reinterpret_cast<const float4 *>(pb + o);
          b.x += b0.x; b.y += b0.y; b.z += b0.z; b.w += b0.w;
        }
      }
      *reinterpret_cast<float4 *>(pt + o) = t;
      if (TYPED && pb) *reinterpret_cast<float4 *>(pb + o) = b;
    }
    emitted = true;
  };
  // (re)start the pipeline at tile k: tile k staged in LDS, tile k+1's rows and tile k+2's ids requested
  std::integral_constant<int, 0> ph0;
  std::integral_constant<int, 1> ph1;
  // (re)start the pipeline at tile k (phase 0): tile k staged in buffer 0, the rows of tiles k+1 / k+2 requested
  // into sets 1 / 0, the ids of tile k+3 requested
  auto start_at = [&](int k) {
    load_ids(k);
    load_rows(ph0);
    load_ids(k + 1);
    stage(ph0);
    load_rows(ph1);
    load_ids(k + 2);
    load_rows(ph0);
    load_ids(k + 3);
    lds_barrier();
  };

  // typed: the class structure of the range is read 64 tiles at a time -- lane l keeps the coefficient of tile
  // 64c + l, bit l of `starts` says "tile 64c + l begins a new class" -- so that the hot loop tests a scalar
  // bit instead of waiting for a load behind the row prefetch
  float cur = 0.f, sv = 0.f;
  unsigned long long starts = 0;
  if (mine > 0) {
    int k = 0;
    bool restart = true;                                    // the hot loop was left: the pipeline starts over at tile k
    while (k < mine) {
      if (TYPED) {
        bool boundary = true;                               // left the hot loop inside a chunk: tile k begins a new class
        if ((k & 63) == 0) {
          const float last = __shfl(sv, 63);
          sv = k + lane < mine ? p.tile_scale[lo + k + lane] : 0.f;
          float up = __shfl_up(sv, 1);
          if (lane == 0) up = k > 0 ? last : sv;
          starts = __ballot(k + lane < mine && __float_as_uint(sv) != __float_as_uint(up));
          boundary = (starts & 1ull) != 0;
        }
        if (boundary) {                                     // classes are sorted: the finished class's total goes out
          emit(cur);
          zero_acc();
          __syncthreads();                                  // LDS goes back to the tiles
        }
        cur = __shfl(sv, k & 63);
      }
      if (restart) start_at(k);
      // the hot loop: tiles of one class inside one chunk of 64, two pipeline phases per trip
      auto more = [&]() { return k < mine && (!TYPED || ((k & 63) != 0 && ((starts >> (k & 63)) & 1ull) == 0)); };
      for (;;) {
        tile_step(k, ph0);
        lds_barrier();                                      // tile k+1 is staged for everyone, tile k's buffer is free
        ++k;
        if (!more()) break;
        tile_step(k, ph1);
        lds_barrier();
        ++k;
        if (!more()) break;
      }
    }
  }
  emit(cur);
  if (!TYPED && !PLAIN && pcs) {
    // column sums: the 8 threads that staged the same 4 columns, added in a fixed order
    __syncthreads();
    *reinterpret_cast<float4 *>(&smem[(gtid / kQ) * H + (gtid % kQ) * 4]) = cs;
    __syncthreads();
    if (gtid < kQ) {
      float4 t = *reinterpret_cast<const float4 *>(&smem[gtid * 4]);
#pragma unroll
      for (int g = 1; g < kPass; ++g) {
        const float4 u = *reinterpret_cast<const float4 *>(&smem[g * H + gtid * 4]);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4 *>(pcs + gtid * 4) = t;
    }
  }
}

template <int MODE, int H = 128, bool X6 = true, bool BIG = false>
__global__ __launch_bounds__(kGroupThreads, 2) void atb_k(const AtbArgs p) {
  atb_body<MODE, H, X6, BIG>(p, (MODE == ATB_ROWS || MODE == ATB_PLAIN) ? (int)blockIdx.y / p.nb : 0,
                             (MODE == ATB_ROWS || MODE == ATB_PLAIN) ? (int)blockIdx.y % p.nb : 0);
}

// Several products over the SAME rows in one launch (the node side's three weight gradients): blockIdx.y picks a
// job = one 128 x 128 output block with its own operands, gate and partial; the launch's one wave of workgroups
// is shared by all blocks, so every workgroup gets a long tile range and the per-workgroup costs (pipeline
// start, LDS reduction, 64 KB partial) are paid 512 times in total instead of 512 times per product.
constexpr int kMaxAtbJobs = DMP_ATB_MAX_JOBS;
struct AtbJobs { AtbArgs job[kMaxAtbJobs]; };
template <int H, bool X6 = true, int MODE = ATB_ROWS>
__global__ __launch_bounds__(kGroupThreads, 2) void atb_jobs_k(const AtbJobs t) { atb_body<MODE, H, X6>(t.job[blockIdx.y], 0, 0); }

constexpr int kAtbLdsBytes = 2 * 2 * kSub * kLdsStride * 4;   // 67584: above the 64 KB static limit -> dynamic LDS, opted in once
constexpr int kAtbLdsBytes64 = 2 * 2 * kSub * 68 * 4;         // H = 64: 34816
constexpr int atb_lds_bytes(int H) { return H == 128 ? kAtbLdsBytes : kAtbLdsBytes64; }
// The dynamic-LDS opt-in is a per-device function attribute: set once per (kernel, device) -- a process that drives
// several devices passes through here once for each (the writes of the flag race benignly: the call is idempotent).
constexpr int kMaxDevices = 64;
inline bool opt_in_lds(const void *kernel, int bytes, bool (&done)[kMaxDevices]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  if (done[dev]) return true;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { set_last_hip_error(e); return false; }
  done[dev] = true;
  return true;
}
template <int MODE, int H = 128>
bool lds_ready() {
  static bool done[kMaxDevices] = {}, done_exact[kMaxDevices] = {}, done_big[kMaxDevices] = {};
  return opt_in_lds(reinterpret_cast<const void *>(&atb_k<MODE, H, true>), atb_lds_bytes(H), done) &&
         opt_in_lds(reinterpret_cast<const void *>(&atb_k<MODE, H, false>), atb_lds_bytes(H), done_exact) &&
         opt_in_lds(reinterpret_cast<const void *>(&atb_k<MODE, H, MODE == ATB_TYPED, true>), atb_lds_bytes(H), done_big);
}
// The bf16x6 form is used where it was measured faster: the class-typed product (its rows are gathered, the f32 form's
// matrix work is what bounds it: 172 -> 148 us at bench.py's shape).  The plain-row forms stay on the f32-input MFMA: the
// gated rows variant needs 42 more registers than the file has at H = 128 (spills: 168 -> 179 us), the multi-job launch of
// the node side came out equal (138 vs 140 us).
template <int MODE, int H>
void launch_atb(const AtbArgs &a, dim3 grid, hipStream_t st) {
  if (!fits4g(a.E, a.ldz) || !fits4g(a.E, a.ldd)) {           // an operand of 4 GiB or more: 64-bit row addressing
    atb_k<MODE, H, MODE == ATB_TYPED, true><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(a);
    return;
  }
  const bool x6 = MODE == ATB_TYPED || MODE == ATB_PLAIN || (MODE == ATB_ROWS && a.rows_x6);   // PLAIN: no gate / column sums in the kernel -- it fits the file
  if (g_exact_fp32 || !x6) atb_k<MODE, H, false><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(a);
  else atb_k<MODE, H, true><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(a);
}

inline unsigned atb_blocks(int64_t tiles, int H = 128) {
  // (measured, round 5: one workgroup per CU instead of two -- half the [H, H] partials to write and to reduce -- is 45 us per step slower)
  const int64_t cap = 256 * (H == 64 ? 4 : 2);             // resident workgroups per CU: two (H = 128), four (H = 64)
  return (unsigned)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}
inline bool fits32(int64_t rows, int64_t ld) { return rows * ld * 4 < ((int64_t)1 << 32) - 8192; }

// one wave of workgroups over all (a, b) output blocks together
static unsigned rows_blocks(int64_t rows, int M, int N, int H) {
  const int64_t tiles = (rows + kSub - 1) / kSub, nblk = (int64_t)(M / H) * (N / H);
  int64_t g = (H == 64 ? 1024 : 512) / (nblk > 0 ? nblk : 1);
  if (g < 1) g = 1;
  if (g > tiles) g = tiles > 0 ? tiles : 1;
  return (unsigned)g;
}


// plain: no job has a gate or column sums -- the launch runs the ungated form on the bf16 pipe (it fits the register file there,
// the gated form does not: see launch_atb)
template <int H>
static int atb_jobs_launch(const AtbJobs &t, int num_jobs, int64_t rows, hipStream_t st, bool plain) {
  static bool done[kMaxDevices] = {}, done_exact[kMaxDevices] = {}, done_plain[kMaxDevices] = {};
  if (!opt_in_lds(reinterpret_cast<const void *>(&atb_jobs_k<H, true>), atb_lds_bytes(H), done) ||
      !opt_in_lds(reinterpret_cast<const void *>(&atb_jobs_k<H, false>), atb_lds_bytes(H), done_exact) ||
      !opt_in_lds(reinterpret_cast<const void *>(&atb_jobs_k<H, true, ATB_PLAIN>), atb_lds_bytes(H), done_plain))
    return DMP_ERR_HIP;
  const dim3 grid(rows_blocks(rows, H, H * (num_jobs > 0 ? num_jobs : 1), H), (unsigned)num_jobs);
  if (plain && !g_exact_fp32) atb_jobs_k<H, true, ATB_PLAIN><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);
  else atb_jobs_k<H, false><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);   // f32-input MFMA (see launch_atb)
  return check_launch();
}


// the tile-list form of the multi-job launch: blockIdx.y = the job, blockIdx.x splits the SHARED tile list
static unsigned tile_jobs_blocks(int64_t tiles_bound, int num_jobs, int H) {
  int64_t g = (int64_t)atb_blocks(tiles_bound, H) / (num_jobs > 0 ? num_jobs : 1);
  if (g < 1) g = 1;
  if (g > tiles_bound) g = tiles_bound > 0 ? tiles_bound : 1;
  return (unsigned)g;
}
template <int H>
static int atb_tile_jobs_launch(const AtbJobs &t, int num_jobs, int64_t tiles_bound, hipStream_t st) {
  static bool done[kMaxDevices] = {}, done_exact[kMaxDevices] = {};
  if (!opt_in_lds(reinterpret_cast<const void *>(&atb_jobs_k<H, true, ATB_TYPED>), atb_lds_bytes(H), done) ||
      !opt_in_lds(reinterpret_cast<const void *>(&atb_jobs_k<H, false, ATB_TYPED>), atb_lds_bytes(H), done_exact))
    return DMP_ERR_HIP;
  const dim3 grid(tile_jobs_blocks(tiles_bound, num_jobs, H), (unsigned)num_jobs);
  if (g_exact_fp32) atb_jobs_k<H, false, ATB_TYPED><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);
  else atb_jobs_k<H, true, ATB_TYPED><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);
  return check_launch();
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int64_t dmp_atb_typed_blocks_h(int64_t tiles_bound, int H) { return (int64_t)atb_blocks(tiles_bound, H == 64 ? 64 : 128); }

int dmp_atb_typed(const float *Z, int64_t ldz, const float *dPre, int64_t ldp, const int32_t *slot_edge,
                  const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H,
                  float *partial_T, float *partial_B, void *stream) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!partial_T || !num_tiles || !slot_edge || !tile_scale) return DMP_ERR_BAD_ARG;   // partial_B NULL: Z^T dPre alone
  if (E > 0 && (!Z || !dPre || ldz < H || ldp < H)) return DMP_ERR_BAD_ARG;
  if (ldz % 4 || ldp % 4 || (E > 0 && (!aligned16(Z) || !aligned16(dPre))) || !aligned16(partial_T) || (partial_B && !aligned16(partial_B)))
    return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldz) || !stride_ok(ldp) || E >= ((int64_t)1 << 31) || !fits32(tiles_bound * kSub, 1)) return DMP_ERR_UNSUPPORTED;
  AtbArgs a{};
  a.Z = Z; a.ldz = ldz; a.D = dPre; a.ldd = ldp; a.E = E; a.slot_edge = slot_edge; a.tile_scale = tile_scale;
  a.num_tiles = num_tiles; a.pT = partial_T; a.pB = partial_B;
  const bool wide = partial_B && partial_B == partial_T + H;            // one [G][H][2H] buffer ([T | B] side by side) or two [G][H*H]
  a.pstride = wide ? 2 * H * H : H * H;
  a.ldp = wide ? 2 * H : H;
  if (H == 128) {
    if (!lds_ready<ATB_TYPED, 128>()) return DMP_ERR_HIP;
    launch_atb<ATB_TYPED, 128>(a, dim3(atb_blocks(tiles_bound)), (hipStream_t)stream);
  } else {
    if (!lds_ready<ATB_TYPED, 64>()) return DMP_ERR_HIP;
    launch_atb<ATB_TYPED, 64>(a, dim3(atb_blocks(tiles_bound, 64)), (hipStream_t)stream);
  }
  return check_launch();
}

int64_t dmp_atb_rows_blocks_h(int64_t rows, int M, int N, int H) { return (int64_t)rows_blocks(rows, M, N, H == 64 ? 64 : 128); }
int64_t dmp_atb_jobs_blocks_h(int64_t rows, int num_jobs, int H) {
  const int h = H == 64 ? 64 : 128;
  return (int64_t)rows_blocks(rows, h, h * (num_jobs > 0 ? num_jobs : 1), h);
}
/* ... of the tile-list form (rows of every job's partial = the grid's x extent) */
int64_t dmp_atb_tile_jobs_blocks(int64_t tiles_bound, int num_jobs, int H) { return (int64_t)tile_jobs_blocks(tiles_bound, num_jobs, H == 64 ? 64 : 128); }
int dmp_atb_rows_jobs_h(const dmp_atb_job *jobs, int num_jobs, int64_t rows, int H, const int32_t *slot_row,
                        const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, void *stream) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (rows < 0 || num_jobs < 1 || num_jobs > kMaxAtbJobs || !jobs) return DMP_ERR_BAD_ARG;
  AtbJobs t;
  if (slot_row) {
    // over a TILE LIST (rows gathered by slot, the tile count in device memory) instead of the rows 0 .. rows-1: the node side of
    // a layer under a 0 / 1 node gate walks the kept nodes' tiles (dmp_kept_rows(tiles = 1)) -- 41 % of the tiles at bench.py's labels
    if (!tile_scale || !num_tiles || tiles_bound < 0) return DMP_ERR_BAD_ARG;
    if (!fits32(tiles_bound * kSub, 1) || rows >= ((int64_t)1 << 31)) return DMP_ERR_UNSUPPORTED;
    for (int i = 0; i < num_jobs; ++i) {
      const dmp_atb_job &j = jobs[i];
      if (j.gate || j.partial_colsum || j.rowmask) return DMP_ERR_BAD_ARG;      // the list says which rows take part
      if (!j.partial || (rows > 0 && (!j.A || !j.B || j.lda < H || j.ldb < H)) || j.ldp < H) return DMP_ERR_BAD_ARG;
      if (j.lda % 4 || j.ldb % 4 || j.ldp % 4 || (rows > 0 && (!aligned16(j.A) || !aligned16(j.B))) || !aligned16(j.partial))
        return DMP_ERR_UNSUPPORTED;
      if (!stride_ok(j.lda) || !stride_ok(j.ldb) || !fits4g(rows, j.lda) || !fits4g(rows, j.ldb)) return DMP_ERR_UNSUPPORTED;
      AtbArgs a{};
      a.Z = j.A; a.ldz = j.lda; a.D = j.B; a.ldd = j.ldb; a.E = rows; a.slot_edge = slot_row; a.tile_scale = tile_scale;
      a.num_tiles = num_tiles; a.pT = j.partial; a.pB = nullptr; a.pstride = j.partial_stride; a.ldp = j.ldp;
      t.job[i] = a;
    }
    return H == 128 ? atb_tile_jobs_launch<128>(t, num_jobs, tiles_bound, (hipStream_t)stream)
                    : atb_tile_jobs_launch<64>(t, num_jobs, tiles_bound, (hipStream_t)stream);
  }
  bool plain = true;
  for (int i = 0; i < num_jobs; ++i) {
    const dmp_atb_job &j = jobs[i];
    plain = plain && !j.gate && !j.partial_colsum;
    if (!j.partial || (rows > 0 && (!j.A || !j.B || j.lda < H || j.ldb < H)) || j.ldp < H) return DMP_ERR_BAD_ARG;
    if (j.lda % 4 || j.ldb % 4 || j.ldp % 4 || (rows > 0 && (!aligned16(j.A) || !aligned16(j.B))) || !aligned16(j.partial) ||
        (j.partial_colsum && !aligned16(j.partial_colsum)))
      return DMP_ERR_UNSUPPORTED;
    if (!stride_ok(j.lda) || !stride_ok(j.ldb) || rows > 0x7fffffff - kSub || !fits4g(rows, j.lda) || !fits4g(rows, j.ldb))
      return DMP_ERR_UNSUPPORTED;      // the multi-job launch has no 64-bit form (node-side operands)
    AtbArgs a{};
    a.Z = j.A; a.ldz = j.lda; a.D = j.B; a.ldd = j.ldb; a.E = rows; a.plain_tiles = (int)((rows + kSub - 1) / kSub);
    a.gate = j.gate; a.pT = j.partial; a.pstride = j.partial_stride; a.ldp = j.ldp; a.pCS = j.partial_colsum;
    a.nb = 1; a.cs_ld = j.cs_ld; a.rowmask = j.rowmask;
    t.job[i] = a;
  }
  return H == 128 ? atb_jobs_launch<128>(t, num_jobs, rows, (hipStream_t)stream, plain)
                  : atb_jobs_launch<64>(t, num_jobs, rows, (hipStream_t)stream, plain);
}

int64_t dmp_rel_atb_blocks(int num_rels) { return num_rels >= 512 ? 1 : 512 / (num_rels > 0 ? num_rels : 1); }

int dmp_rel_atb(const float *X, int64_t ldx, int64_t rows_x, const float *D, int64_t ldd, int64_t rows_d,
                const int32_t *slot_x, const int32_t *slot_d, const float *slot_scale, const int32_t *type_tile_ptr,
                int num_rels, int64_t tiles, int H, float *partial, void *stream) {
  if (rows_x < 0 || rows_d < 0 || tiles < 0 || num_rels < 1 || num_rels > 65535 || H != 128)
    return H == 128 ? DMP_ERR_BAD_ARG : DMP_ERR_UNSUPPORTED;
  if (!partial || !type_tile_ptr || (tiles > 0 && (!X || !D || !slot_x || !slot_d || ldx < H || ldd < H))) return DMP_ERR_BAD_ARG;
  if (ldx % 4 || ldd % 4 || (tiles > 0 && (!aligned16(X) || !aligned16(D))) || !aligned16(partial)) return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldx) || !stride_ok(ldd) || rows_x >= ((int64_t)1 << 31) || rows_d >= ((int64_t)1 << 31) || !fits32(tiles * kSub, 1))
    return DMP_ERR_UNSUPPORTED;
  AtbArgs a{};
  a.Z = X; a.ldz = ldx; a.D = D; a.ldd = ldd; a.E = rows_x > rows_d ? rows_x : rows_d;
  a.slot_edge = slot_x; a.slot_d = slot_d; a.slot_scale = slot_scale; a.type_tile_ptr = type_tile_ptr;
  a.plain_tiles = (int)tiles; a.pT = partial; a.pstride = 128 * 128; a.ldp = 128;
  if (!lds_ready<ATB_REL, 128>()) return DMP_ERR_HIP;
  const dim3 grid((unsigned)dmp_rel_atb_blocks(num_rels), (unsigned)num_rels);
  launch_atb<ATB_REL, 128>(a, grid, (hipStream_t)stream);
  return check_launch();
}

int dmp_atb_rows_masked(const float *A, int64_t lda, const float *B, int64_t ldb, const float *gate, const uint32_t *rowmask,
                        int x6, int64_t rows, int M, int N, int H, float *partial, float *partial_colsum, void *stream) {
  if (rowmask && !gate) return DMP_ERR_BAD_ARG;
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (rows < 0 || M <= 0 || N <= 0) return DMP_ERR_BAD_ARG;
  if (M % H || N % H || (int64_t)(M / H) * (N / H) > 65535) return DMP_ERR_UNSUPPORTED;
  if (!partial) return DMP_ERR_BAD_ARG;
  if (rows > 0 && (!A || !B || lda < M || ldb < N)) return DMP_ERR_BAD_ARG;
  if (lda % 4 || ldb % 4 || (rows > 0 && (!aligned16(A) || !aligned16(B))) || !aligned16(partial) ||
      (partial_colsum && !aligned16(partial_colsum)))
    return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(lda) || !stride_ok(ldb) || rows > 0x7fffffff - kSub) return DMP_ERR_UNSUPPORTED;
  AtbArgs a{};
  a.Z = A; a.ldz = lda; a.D = B; a.ldd = ldb; a.E = rows; a.plain_tiles = (int)((rows + kSub - 1) / kSub);
  a.gate = gate; a.pT = partial; a.pstride = (int64_t)M * N; a.ldp = N; a.pCS = partial_colsum; a.nb = N / H; a.cs_ld = M;
  a.rowmask = rowmask; a.rows_x6 = x6 ? 1 : 0;
  const dim3 grid(rows_blocks(rows, M, N, H), (unsigned)((M / H) * (N / H)));
  if (H == 128) {
    if (!lds_ready<ATB_ROWS, 128>()) return DMP_ERR_HIP;
    launch_atb<ATB_ROWS, 128>(a, grid, (hipStream_t)stream);
  } else {
    if (!lds_ready<ATB_ROWS, 64>()) return DMP_ERR_HIP;
    launch_atb<ATB_ROWS, 64>(a, grid, (hipStream_t)stream);
  }
  return check_launch();
}

int dmp_atb_rows_plain(const float *A, int64_t lda, const float *B, int64_t ldb, const uint32_t *rowmask, int64_t rows, int M, int N,
                       int H, float *partial, void *stream) {
  if (H != 128 && H != 64) return DMP_ERR_UNSUPPORTED;
  if (rows < 0 || M <= 0 || N <= 0) return DMP_ERR_BAD_ARG;
  if (M % H || N % H || (int64_t)(M / H) * (N / H) > 65535) return DMP_ERR_UNSUPPORTED;
  if (!partial) return DMP_ERR_BAD_ARG;
  if (rows > 0 && (!A || !B || lda < M || ldb < N)) return DMP_ERR_BAD_ARG;
  if (lda % 4 || ldb % 4 || (rows > 0 && (!aligned16(A) || !aligned16(B))) || !aligned16(partial)) return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(lda) || !stride_ok(ldb) || rows > 0x7fffffff - kSub) return DMP_ERR_UNSUPPORTED;
  AtbArgs a{};
  a.Z = A; a.ldz = lda; a.D = B; a.ldd = ldb; a.E = rows; a.plain_tiles = (int)((rows + kSub - 1) / kSub);
  a.pT = partial; a.pstride = (int64_t)M * N; a.ldp = N; a.nb = N / H; a.cs_ld = M; a.rowmask = rowmask;
  const dim3 grid(rows_blocks(rows, M, N, H), (unsigned)((M / H) * (N / H)));
  if (H == 128) {
    if (!lds_ready<ATB_PLAIN, 128>()) return DMP_ERR_HIP;
    launch_atb<ATB_PLAIN, 128>(a, grid, (hipStream_t)stream);
  } else {
    if (!lds_ready<ATB_PLAIN, 64>()) return DMP_ERR_HIP;
    launch_atb<ATB_PLAIN, 64>(a, grid, (hipStream_t)stream);
  }
  return check_launch();
}

}  // extern "C"
