// The DMPLayer edge chain forward as ONE kernel (gfx950, H = 128, bf16x6 products: dmp_mfma_common.h):
//
//     H1[e] = act(Z[e] W_g + P[selA e, 0:H] - P[selB e, H:2H] + b')         (dmp_edge_fwd_typed)
//     Zn[e] = (residual ? Z[e] : 0) + gate[e] (H1[e] W2^T + b2)             (dmp_out_fwd_fused)
//
// (dmpnn.py:142-156 with the first MLP Linear folded into the projections, then emlp's second Linear, the rep-net's gate
// and residual, dmpnn.py:262-275).  As two launches the H1 rows are written and read back and the Z rows are read twice:
// 6 passes over [E, H] arrays; here Z is read once (its second use hits L2), H1 is written once (the backward needs it)
// and Zn once: 3 passes.  Both products are class-typed-tile products over the same tile list.
//
// A 512-thread workgroup per CU, two wave groups with their own register-resident weight pieces (both panels do not fit
// one wave's registers):
//   group A (waves 0-3)  the class-typed pipeline of dmp_typed.hip (TEPI_EDGE, bf16x6): ids of tile k+3, rows of tile k+2,
//                        staging of tile k+1 (split into bf16 piece planes) in the shadow of tile k's MFMAs; its epilogue
//                        stores the H1 rows AND writes them as piece planes into the H1 image of the tile (Hs[k & 1]);
//   group B (waves 4-7)  one tile behind: H1 image of tile k-1 x the W2 pieces, epilogue Z + gate (. + b2), rows scattered
//                        by edge id; the Z rows are requested before its MFMAs.
// One barrier per tile for all eight waves: after it the next tile's Z planes are staged, the H1 image of the tile just
// finished by group A is complete, and group B is done with the image written one tile earlier.
#include <type_traits>

#include "dmp_mfma_common.h"

namespace dmp {
namespace {

struct ChainArgs {
  const float *Z; int64_t ldz;                          // layer input rows [E, H]
  const float *W; int64_t ldw;                          // [H, >= 2H] = [A' | B']: W_g = A' + c_g B'
  const float *P; int64_t ldp; int64_t num_nodes;       // gathered projections [N, >= 2H]
  const float *bias;                                    // [H] or NULL
  const int32_t *selA, *selB;                           // [E] nodes of the added / subtracted P rows
  const int32_t *slot_edge; const float *tile_scale; const int32_t *num_tiles;
  int64_t E; float slope;
  float *H1; int64_t ldh;                               // saved activation [E, H] (output)
  const float *W2t; int64_t ldw2;                       // [H (in), >= H (out)]: the second Linear's weight, transposed
  const float *b2;                                      // [H] or NULL
  const float *gate;                                    // [E] or NULL (1)
  int residual;
  float *Zn; int64_t ldo;                               // output rows [E, H]
};

constexpr int kCH = 128;                                  // hidden width
constexpr int kCStrideD = (kCH + 8) / 2, kCPlane = kSub * kCStrideD;   // bf16 plane rows: 68 dwords (see dmp_typed.hip)
constexpr int kCHalf = kCH / 2, kCGroups = kCHalf / 8, kCQ = kCH / 4;

template <bool BIG>
__global__ __launch_bounds__(512, 2) void edge_chain_k(ChainArgs p) {
  constexpr uint32_t kRowBytes = kCH * 4u;
  __shared__ __attribute__((aligned(16))) uint32_t As[2][3 * kCPlane];     // Z tile pieces (group A's operand)
  __shared__ __attribute__((aligned(16))) uint32_t Hs[2][3 * kCPlane];     // H1 tile pieces (group B's operand)
  __shared__ float Cs[8][32 * kScrStride];
  __shared__ uint32_t rowA[3][kSub], rowB[3][kSub], rowC[3][kSub];         // [tile % 3][row]: gathered nodes, edge id (-1: none)
  __shared__ float rowG[3][kSub];                                          // the row's gate
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool grpB = wave >= 4;
  const int cs = wave & 3;                                                 // column slice of the group's product
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, gtid = threadIdx.x & 255;
  const int col = 32 * cs + li;
  float *scr = Cs[wave];
  const int lrow = lane >> 3, c4 = 32 * cs + (lane & 7) * 4;
  const uint32_t colA = (uint32_t)(gtid % kCQ) * 16u, col4 = (uint32_t)c4 * 4u;
  const float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.slope)));

  const int ntiles = __builtin_amdgcn_readfirstlane(*p.num_tiles);
  const int chunk = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = (int)blockIdx.x * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int mine = hi > lo ? hi - lo : 0;
  if (mine == 0) return;                                                   // uniform over the workgroup

  const srsrc_t rs_Z = make_srsrc(p.Z, p.ldz, p.E);
  const srsrc_t rs_H1 = make_srsrc(p.H1, p.ldh, p.E);
  const srsrc_t rs_Zn = make_srsrc(p.Zn, p.ldo, p.E);
  const srsrc_t rs_P = make_srsrc(p.P, p.ldp, p.num_nodes);

  // the tile's product on the bf16 pipe: fragments of the image's three planes x the wave's panel pieces (dmp_typed.hip)
  f32x16 acc;
  auto x6_tile = [&](const uint32_t *tile, const Split8 *B6, auto &&action) {
    const uint32_t *ar = tile + li * kCStrideD + (kCHalf / 2) * h;
    Frag8 ah, am, al;
    ah.v = *reinterpret_cast<const bf16x8 *>(ar);
    am.v = *reinterpret_cast<const bf16x8 *>(ar + kCPlane);
    al.v = *reinterpret_cast<const bf16x8 *>(ar + 2 * kCPlane);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < kCGroups; ++g) {
      Frag8 nh = ah, nm = am, nl = al;
      if (g + 1 < kCGroups) {
        nh.v = *reinterpret_cast<const bf16x8 *>(ar + 4 * (g + 1));
        nm.v = *reinterpret_cast<const bf16x8 *>(ar + kCPlane + 4 * (g + 1));
        nl.v = *reinterpret_cast<const bf16x8 *>(ar + 2 * kCPlane + 4 * (g + 1));
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
  // accumulators -> the wave's scratch (row-major 32 x 32), read back as 4 float4 per lane (rows 8k + lrow)
  auto park = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * h) * kScrStride + li] = acc[r];
  };

  if (!grpB) {
    // =========================================================================================== group A: Z -> H1
    const uint32_t rows4 = (uint32_t)(p.E * 4);
    const rsrc_t rs_selA = make_rsrc(p.selA, rows4), rs_selB = make_rsrc(p.selB, rows4);
    const rsrc_t rs_gate = make_rsrc(p.gate, p.gate ? rows4 : 0u);
    const rsrc_t rs_slot = make_rsrc(p.slot_edge, (uint32_t)ntiles * (kSub * 4u));
    constexpr uint32_t kOOB = 0xFFFFF000u;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4 *>(p.bias + c4);

    Split8 B6[kCGroups];
    const rsrc_t rs_W = make_rsrc(p.W, (uint32_t)(kCH * p.ldw * 4));
    const uint32_t w_first = (uint32_t)((int64_t)kCHalf * h * p.ldw + col) * 4u;
    const uint32_t w_step = __builtin_amdgcn_readfirstlane((int)(p.ldw * 4));
    auto load_panel = [&](float c) {
      uint32_t off;
      asm volatile("v_mov_b32 %0, %1" : "=v"(off) : "v"(w_first));
#pragma unroll
      for (int s0 = 0; s0 < kCHalf; s0 += 8) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float w0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off, (int)((s0 + j) * w_step), 0));
          const float w1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_W, (int)off + (int)kRowBytes, (int)((s0 + j) * w_step), 0));
          w[j] = w0 + c * w1;
        }
        split8(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), B6[s0 / 8]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    int id_rows[kSubLoads], id_own = -1, own_staged = -1;
    float4 pre[kSubLoads];
    uint32_t pre_a = 0, pre_b = 0;
    float pre_g = 1.f;
    auto load_ids = [&](int k) {
      const bool ok = k < mine;
      const uint32_t so = (uint32_t)(lo + k) * (kSub * 4u);
#pragma unroll
      for (int m = 0; m < kSubLoads; ++m)
        id_rows[m] = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, ((gtid / kCQ) + 8 * m) * 4, (int)so, 0) : -1;
      if (gtid < kSub) id_own = ok ? (int)__builtin_amdgcn_raw_buffer_load_b32(rs_slot, gtid * 4, (int)so, 0) : -1;
    };
    auto load_row = [&](int m) { pre[m] = row_load4<BIG>(rs_Z, p.Z, p.ldz, id_rows[m], colA); };
    auto load_row_scalars = [&]() {
      if (gtid < kSub) {
        const uint32_t eo = id_own >= 0 ? (uint32_t)id_own * 4u : kOOB;
        pre_a = __builtin_amdgcn_raw_buffer_load_b32(rs_selA, (int)eo, 0, 0);
        pre_b = __builtin_amdgcn_raw_buffer_load_b32(rs_selB, (int)eo, 0, 0);
        pre_g = p.gate ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_gate, (int)eo, 0, 0)) : 1.f;
        own_staged = id_own;
      }
    };
    auto stage_row = [&](int buf, int m) {
      uint2 ph, pm, pl;
      split_pair(pre[m].x, pre[m].y, ph.x, pm.x, pl.x);
      split_pair(pre[m].z, pre[m].w, ph.y, pm.y, pl.y);
      uint32_t *q = &As[buf][0] + ((gtid / kCQ) + 8 * m) * kCStrideD + (gtid % kCQ) * 2;
      *reinterpret_cast<uint2 *>(q) = ph;
      *reinterpret_cast<uint2 *>(q + kCPlane) = pm;
      *reinterpret_cast<uint2 *>(q + 2 * kCPlane) = pl;
    };
    auto stage_scalars = [&](int par) {
      if (gtid < kSub) {
        const bool ok = own_staged >= 0;
        rowA[par][gtid] = ok ? pre_a : 0xFFFFFFFFu;
        rowB[par][gtid] = ok ? pre_b : 0xFFFFFFFFu;
        rowC[par][gtid] = ok ? (uint32_t)own_staged : 0xFFFFFFFFu;
        rowG[par][gtid] = pre_g;
      }
    };
    float4 g0[4], g1[4];
    auto fetch_operand = [&](int par, int k) {
      const int rr = 8 * k + lrow;
      g0[k] = sbuf_load4(rs_P, (int)rowA[par][rr], col4);
      g1[k] = sbuf_load4(rs_P, (int)rowB[par][rr], col4 + kRowBytes);
    };
    auto shadow = [&](int i, int k, int buf, int nxt3) {
      if (i >= 4 && i < 8) stage_row(buf ^ 1, i - 4);
      else if (i == 8) stage_scalars(nxt3);
      else if (i >= 9 && i < 13) load_row(i - 9);
      else if (i == 13) load_row_scalars();
      else if (i == 14) load_ids(k + 3);
    };
    // epilogue of tile k: H1 rows to memory and, as bf16 pieces, into the H1 image group B multiplies one tile later
    auto epilogue = [&](int k, int par) {
      park();
      uint32_t *img = &Hs[k & 1][0];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = 8 * q + lrow;
        float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
        v.x = act_fwd((v.x + (g0[q].x - g1[q].x)) + bias4.x, slope);
        v.y = act_fwd((v.y + (g0[q].y - g1[q].y)) + bias4.y, slope);
        v.z = act_fwd((v.z + (g0[q].z - g1[q].z)) + bias4.z, slope);
        v.w = act_fwd((v.w + (g0[q].w - g1[q].w)) + bias4.w, slope);
        const int id = (int)rowC[par][rr];
        if (id < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);              // padding rows: zeros in the image
        row_store4<BIG>(v, rs_H1, p.H1, p.ldh, id, col4);
        uint2 ph, pm, pl;
        split_pair(v.x, v.y, ph.x, pm.x, pl.x);
        split_pair(v.z, v.w, ph.y, pm.y, pl.y);
        uint32_t *w = img + rr * kCStrideD + c4 / 2;
        *reinterpret_cast<uint2 *>(w) = ph;
        *reinterpret_cast<uint2 *>(w + kCPlane) = pm;
        *reinterpret_cast<uint2 *>(w + 2 * kCPlane) = pl;
      }
    };

    // ---- prologue: tile 0 staged, tile 1's rows and tile 2's ids requested
    load_ids(0);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(m);
    load_row_scalars();
    load_ids(1);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) stage_row(0, m);
    stage_scalars(0);
#pragma unroll
    for (int m = 0; m < kSubLoads; ++m) load_row(m);
    load_row_scalars();
    load_ids(2);
    lds_barrier();                                                         // B0: matched by group B

    float sv = 0.f, c_have = 0.f;
    bool have_panel = false;
    int par3 = 0;
    for (int k = 0; k <= mine; ++k) {                                      // iteration `mine`: group B's last tile only
      if (k < mine) {
        if ((k & 63) == 0) sv = k + lane < mine ? p.tile_scale[lo + k + lane] : 0.f;
        const float c = __shfl(sv, k & 63);
        if (!have_panel || __float_as_uint(c) != __float_as_uint(c_have)) {
          load_panel(c);
          c_have = c;
          have_panel = true;
        }
        const int buf = k & 1, nxt3 = par3 == 2 ? 0 : par3 + 1;
        // the tile's gathered projection rows (their node ids were staged one tile ago, before the barrier just passed)
#pragma unroll
        for (int c = 0; c < 4; ++c) fetch_operand(par3, c);
        x6_tile(&As[buf][0], B6, [&](int i) { shadow(i, k, buf, nxt3); });
        epilogue(k, par3);
        par3 = par3 == 2 ? 0 : par3 + 1;
      }
      lds_barrier();                                                       // B(k + 1)
    }
  } else {
    // =========================================================================================== group B: H1 -> Zn
    Split8 B2[kCGroups];
    {
      // W2^T [in, out] row-major: fragment element k = kHalf h + 8 g + j, column `col`
      const float *w = p.W2t + (int64_t)(kCHalf * h) * p.ldw2 + col;
#pragma unroll
      for (int g = 0; g < kCGroups; ++g) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[(int64_t)(8 * g + j) * p.ldw2];
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), B2[g]);
      }
    }
    float4 b24 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.b2) b24 = *reinterpret_cast<const float4 *>(p.b2 + c4);
    lds_barrier();                                                         // B0
    int par3 = 0;                                                          // slot of tile k - 1
    for (int k = 0; k <= mine; ++k) {
      if (k >= 1) {
        const int t = k - 1;
        float4 zr[4];
        // the residual rows of tile t (read by group A two tiles ago: L2), requested before the MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int id = (int)rowC[par3][8 * q + lrow];
          zr[q] = (p.residual) ? row_load4<BIG>(rs_Z, p.Z, p.ldz, id, col4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        x6_tile(&Hs[t & 1][0], B2, [&](int) {});
        park();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = 8 * q + lrow;
          float4 v = *reinterpret_cast<const float4 *>(&scr[rr * kScrStride + (lane & 7) * 4]);
          const float gt = rowG[par3][rr];
          v.x = (v.x + b24.x) * gt + zr[q].x; v.y = (v.y + b24.y) * gt + zr[q].y;
          v.z = (v.z + b24.z) * gt + zr[q].z; v.w = (v.w + b24.w) * gt + zr[q].w;
          row_store4<BIG>(v, rs_Zn, p.Zn, p.ldo, (int)rowC[par3][rr], col4);
        }
        par3 = par3 == 2 ? 0 : par3 + 1;
      }
      lds_barrier();                                                       // B(k + 1)
    }
  }
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_edge_chain_fwd(const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *P, int64_t ldp, int64_t num_nodes,
                       const float *bias, const int32_t *selA, const int32_t *selB, const int32_t *slot_edge,
                       const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, float slope,
                       float *H1, int64_t ldh, const float *W2t, int64_t ldw2, const float *b2, const float *gate, int residual,
                       float *Zn, int64_t ldo, void *stream) {
  if (H != kCH) return DMP_ERR_UNSUPPORTED;
  if (E < 0 || num_nodes < 0 || tiles_bound < 0) return DMP_ERR_BAD_ARG;
  if (!slope_ok(slope)) return DMP_ERR_UNSUPPORTED;
  if (E == 0) return DMP_OK;
  if (!Z || !W || !P || !selA || !selB || !slot_edge || !tile_scale || !num_tiles || !H1 || !W2t || !Zn || ldz < H || ldw < 2 * H ||
      ldp < 2 * H || ldh < H || ldw2 < H || ldo < H)
    return DMP_ERR_BAD_ARG;
  if (ldz % 4 || ldh % 4 || ldp % 4 || ldo % 4 || !aligned16(Z) || !aligned16(H1) || !aligned16(P) || !aligned16(Zn) ||
      (bias && !aligned16(bias)) || (b2 && !aligned16(b2)))
    return DMP_ERR_UNSUPPORTED;
  if (!stride_ok(ldz) || !stride_ok(ldh) || !stride_ok(ldp) || !stride_ok(ldo) || E >= ((int64_t)1 << 30) || num_nodes >= ((int64_t)1 << 31) ||
      tiles_bound * kSub * 4 >= ((int64_t)1 << 32) || !fits4g(num_nodes, ldp) || g_exact_fp32)
    return DMP_ERR_UNSUPPORTED;
  ChainArgs p{Z, ldz, W, ldw, P, ldp, num_nodes, bias, selA, selB, slot_edge, tile_scale, num_tiles, E, slope,
              H1, ldh, W2t, ldw2, b2, gate, residual, Zn, ldo};
  const unsigned blocks = (unsigned)(tiles_bound < 256 ? (tiles_bound > 0 ? tiles_bound : 1) : 256);   // one 512-thread workgroup per CU
  const bool big = !fits4g(E, ldz) || !fits4g(E, ldh) || !fits4g(E, ldo);
  if (big) edge_chain_k<true><<<blocks, 512, 0, (hipStream_t)stream>>>(p);
  else edge_chain_k<false><<<blocks, 512, 0, (hipStream_t)stream>>>(p);
  return check_launch();
}

}  // extern "C"
