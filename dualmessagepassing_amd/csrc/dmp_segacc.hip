// Scatter-add of edge rows into BOTH endpoint rows in ONE pass over the edge rows (gfx950 / MI355X).
//
// The backward of the DMPLayer's gathered node projections (SubgraphCountingMatching/models/dmpnn.py:111-127: the
// message UDF reads P_d[dst] - P_s[src], swapped for reversed edges) sums every row of dPre [E, H] into two node rows:
//     out[a_e, 0:H ] += s0 * M[e]        a_e = is_reversed ? src : dst      (dP_d)
//     out[b_e, H:2H] += s1 * M[e]        b_e = the other endpoint           (dP_s, s1 = -1)
// dmp_seg_sum2 over the incidence CSR does this node by node and reads every edge row twice; the second read -- by the
// other endpoint's workgroup -- misses the XCD's L2 for a third of the rows (PMC: 1.29 x the algorithmic bytes).
//
// Here the EDGE rows are streamed once, in eid order, and the sums live in REGISTERS: in a block-diagonal batch the edges
// of a graph are contiguous and touch only that graph's nodes, so a workgroup owns a TILE of whole graphs with at most
// kAccNodes = 64 nodes, and wave c of it owns the 64-column slice [64c, 64c + 64) of every row: lane l keeps
// acc[node][half][64c + l] for all 64 nodes and both halves in 128 VGPRs (v64..v191, pinned).  A row arrives as one dword
// per lane (a 256-byte piece of the row per wave-instruction; the waves of the workgroup together read whole rows; 32
// rows in flight per wave), its two endpoints reach the scalar unit by v_readlane, and the two adds are
//     t = v[64 + a_e] ; t += x ; v[64 + a_e] = t          (VGPR index mode: v_mov with a relative source / destination)
//     u = v[128 + b_e]; u += x ; v[128 + b_e] = u
// -- no LDS, no atomics, no barrier, six vector instructions per row.  Every accumulator receives its addends in
// ascending eid: one fixed summation order, run-to-run bit-stable, and the bits of dmp_seg_sum2 over
// dmp_incidence_build's CSR (rows merged by eid).  Rows whose endpoint lies outside the tile (never in a block-diagonal
// batch) add into a trash register.
//
// Forms that were built and measured at bench.py's shape (E = 548,864 rows of 512 bytes; dmp_seg_sum2 over the incidence
// CSR: 74-77 us stand-alone, 80.8 us in the step), all bit-identical to it:
//   * sums in LDS, ds_add_f32 one row per instruction: 749 us -- the LDS float atomic serialises the 64 lanes of a
//     wave-instruction (~200 cycles each);
//   * this form: 73-82 us stand-alone, 71 us in the step, 1.0 x its algorithmic bytes; 32, 48 or 60 rows in flight per
//     wave make no difference (82.4 / 81.0 / 81.4 us on one box);
//   * wave h sums half h of ALL columns from whole-row loads (512 contiguous bytes per instruction, every row requested by
//     both waves of the workgroup): 91 us -- the partner's duplicate requests are not free; with workgroups walking two
//     tiles each (an even mix of pattern and target tiles, no workgroup left alone at the end): 95 us.
#include "dmp_mfma_common.h"

namespace dmp {
namespace {

constexpr int kAccNodes = 64;      // node rows of a tile = accumulator registers per half
constexpr int kRing = 32;          // rows in flight per wave (one dword per lane each) = rows per super-group

typedef float f32x32 __attribute__((ext_vector_type(32)));

struct GraphTiles { const int64_t *node_off, *edge_off; int64_t Ba, Bb; int ka, kb; };

// One row: x into v[64 + i] and into v[128 + j], each as  read (v_mov, SRC0 relative) -> v_add -> write (v_mov, DST
// relative) -- the form the compiler itself gives an indexed register update.  Two findings are built into this block
// (scripts/stress_segacc.py reproduces both):
//  * `v_add_f32 v[64 + i], v[64 + i], x` with SRC0 and DST relative in ONE instruction (index mode 0x9) computes the right
//    sums but corrupts OTHER waves on the CU: a GEMM running beside it on a second stream returned wrong values in 12 % of
//    the launches, two processes sharing the GPU died with memory faults -- while every single-stream test passed (inside
//    one stream a kernel never shares a CU with another kernel).  With one relative operand per v_mov: 0 of 2000.
//  * s_set_gpr_idx_* write M0, which the compiler does not know about: saved and put back around the block.
#define DMP_ROW(X, I, J)                                                                                               \
      "s_set_gpr_idx_on " I ", 0x1\n\tv_mov_b32 %[t], v64\n\ts_set_gpr_idx_idx " J "\n\tv_mov_b32 %[u], v128\n\t"       \
      "s_set_gpr_idx_off\n\tv_add_f32 %[t], %[t], " X "\n\tv_add_f32 %[u], %[u], " X "\n\t"                             \
      "s_set_gpr_idx_on " I ", 0x8\n\tv_mov_b32 v64, %[t]\n\ts_set_gpr_idx_idx " J "\n\tv_mov_b32 v128, %[u]\n\t"       \
      "s_set_gpr_idx_off\n\t"
// four rows, in order (consecutive rows may meet in a register)
#define DMP_ACC4(X0, X1, X2, X3, I0, I1, I2, I3, J0, J1, J2, J3)                                                       \
  asm volatile(                                                                                                        \
      "s_mov_b32 %[m0s], m0\n\t"                                                                                       \
      DMP_ROW("%[x0]", "%[i0]", "%[j0]") DMP_ROW("%[x1]", "%[i1]", "%[j1]")                                            \
      DMP_ROW("%[x2]", "%[i2]", "%[j2]") DMP_ROW("%[x3]", "%[i3]", "%[j3]")                                            \
      "s_mov_b32 m0, %[m0s]"                                                                                           \
      : "+{v[64:95]}"(A0), "+{v[96:127]}"(A1), "+{v[128:159]}"(B0), "+{v[160:191]}"(B1), "+{v192}"(trash),            \
        [m0s] "=&s"(m0_saved), [t] "=&v"(tmp_t), [u] "=&v"(tmp_u)                                                      \
      : [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3), [i0] "s"(I0), [i1] "s"(I1), [i2] "s"(I2),              \
        [i3] "s"(I3), [j0] "s"(J0), [j1] "s"(J1), [j2] "s"(J2), [j3] "s"(J3)                                          \
      : "memory")

// one row (the masked form: a row whose mask bit is clear skips the block -- a scalar branch around twelve instructions)
#define DMP_ACC1(X0, I0, J0)                                                                                           \
  asm volatile(                                                                                                        \
      "s_mov_b32 %[m0s], m0\n\t"                                                                                       \
      DMP_ROW("%[x0]", "%[i0]", "%[j0]")                                                                               \
      "s_mov_b32 m0, %[m0s]"                                                                                           \
      : "+{v[64:95]}"(A0), "+{v[96:127]}"(A1), "+{v[128:159]}"(B0), "+{v[160:191]}"(B1), "+{v192}"(trash),            \
        [m0s] "=&s"(m0_saved), [t] "=&v"(tmp_t), [u] "=&v"(tmp_u)                                                      \
      : [x0] "v"(X0), [i0] "s"(I0), [j0] "s"(J0)                                                                       \
      : "memory")

// H: row width (64 or 128) = 64 columns per wave, H / 64 waves per workgroup, one workgroup per tile.
// MASKED: bit (e & 31) of rowmask[e >> 5] clear = row e of M is known to be all zeros (the gradient rows a 0 / 1 edge gate
// wiped, dmp_row_mask_bits): such a row is requested past the end of its descriptor -- the load returns zeros without
// memory traffic, the instruction stream is the same.
// NODES: bit (v & 31) of nodemask[v >> 5] clear = node row v of `out` is DEAD (a node under a zero of a 0 / 1 node gate: its
// gradient row is multiplied by that zero further down): not stored -- the store goes through a descriptor of zero records;
// the caller's selectors route such a node's addends to the trash register (dmp_edge_select_nodes: -1).
// RING: rows in flight per wave = rows per super-group (32 or 64).  Measured (round 5, masked form at bench.py's shape, 42 % of
// the rows kept): a ring of 64 is SLOWER (57.3 against 50.5 us) -- the launch is not short of requests in flight
template <int H, bool MASKED = false, bool NODES = false, int RING = kRing>
__global__ __launch_bounds__(H) void seg_acc_graphs_k(
    const float *__restrict__ M, int64_t ldm, const int32_t *__restrict__ selA, const int32_t *__restrict__ selB,
    const GraphTiles ts, float s0, float s1, float *__restrict__ out, int64_t ldo,
    const uint32_t *__restrict__ rowmask = nullptr, int64_t mask_words = 0,
    const uint32_t *__restrict__ nodemask = nullptr, int64_t node_words = 0) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;

  // the tile: graphs [g0, g1)
  const int64_t tile = blockIdx.x;
  const int64_t tiles_a = ts.Ba > 0 ? (ts.Ba + ts.ka - 1) / ts.ka : 0;
  int64_t g0, g1;
  if (tile < tiles_a) { g0 = tile * ts.ka; g1 = g0 + ts.ka < ts.Ba ? g0 + ts.ka : ts.Ba; }
  else { g0 = ts.Ba + (tile - tiles_a) * ts.kb; g1 = g0 + ts.kb < ts.Ba + ts.Bb ? g0 + ts.kb : ts.Ba + ts.Bb; }
  const int64_t n0 = ts.node_off[g0], e0 = ts.edge_off[g0];
  const int nodes = min((int)(ts.node_off[g1] - n0), kAccNodes);
  const int R = (int)(ts.edge_off[g1] - e0);

  // raw buffer descriptors over the tile's rows / selectors / output rows: per-lane offsets are computed once, the row
  // offset is a scalar, rows past the tile's end read zeros and stores past its last node are dropped (no address
  // arithmetic, predicates or branches in the loops)
  const rsrc_t rsM = make_rsrc(M + e0 * ldm, tile_bytes(R, ldm, H));
  const rsrc_t rsA = make_rsrc(selA + e0, (uint32_t)R * 4u), rsB = make_rsrc(selB + e0, (uint32_t)R * 4u);
  const uint32_t voff = (uint32_t)(64 * c + lane) * 4u;
  const uint32_t rowb = (uint32_t)(ldm * 4);
  // (a masked-out row: the same load through a descriptor of zero records -- one scalar select, no vector instruction)
  const float *const baseM = M + e0 * ldm;
  const uint32_t bytesM = tile_bytes(R, ldm, H);
  auto load_row = [&](int r, uint32_t live = 1u) -> float {     // row r of the tile, this wave's columns (scalar row offset)
    const rsrc_t rs = MASKED ? make_rsrc(baseM, live ? bytesM : 0u) : rsM;
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)((uint32_t)r * rowb), 0));
  };
  // mask bits of the tile's rows 32 sg .. 32 sg + 31 (rows e0 + 32 sg ...: two words of the mask, wave-uniform)
  // (word indices clamped to the mask's last word: what they say about rows past the tile's end is not used -- those rows
  // lie past the end of the descriptor anyway)
  const int msh = (int)(e0 & 31), mw0 = (int)(e0 >> 5), mlast = (int)mask_words - 1;
  auto mask_sg = [&](int sg) -> uint64_t {                      // bit k: row RING sg + k of the tile
    if (!MASKED) return ~0ull;
    const int w = mw0 + sg * (RING / 32);
    const uint32_t q0 = rowmask[min(w, mlast)], q1 = rowmask[min(w + 1, mlast)];
    const uint64_t lo = (((((uint64_t)q1) << 32) | q0) >> msh) & 0xffffffffull;
    if (RING == 32) return lo;
    const uint32_t q2 = rowmask[min(w + 2, mlast)];
    return lo | ((((((uint64_t)q2) << 32) | q1) >> msh) << 32);
  };
  auto load_sel = [&](int sg, int &a, int &b) {                 // endpoints of row RING sg + lane
    a = (int)__builtin_amdgcn_raw_buffer_load_b32(rsA, lane * 4, sg * (RING * 4), 0);
    b = (int)__builtin_amdgcn_raw_buffer_load_b32(rsB, lane * 4, sg * (RING * 4), 0);
  };

  f32x32 A0, A1, B0, B1;
#pragma unroll
  for (int i = 0; i < 32; ++i) { A0[i] = 0.f; A1[i] = 0.f; B0[i] = 0.f; B1[i] = 0.f; }
  float trash = 0.f;

  const int nsg = (R + RING - 1) / RING;                      // super-groups of RING rows (one endpoint dword per lane < RING)
  float v[RING];
  int na, nb, m0_saved;
  float tmp_t, tmp_u;
  load_sel(0, na, nb);
  asm volatile("" ::: "memory");     // keep the endpoint loads OLDER than the ring (the waits count younger operations)
  uint64_t m_next = mask_sg(0);      // the mask words run two super-groups ahead of the adds (scalar loads)
#pragma unroll
  for (int k = 0; k < RING; ++k) v[k] = load_row(k, (uint32_t)((m_next >> k) & 1ull));
  m_next = mask_sg(1);
  uint64_t m_cur = mask_sg(0);          // the rows of the super-group being added
  for (int sg = 0; sg < nsg; ++sg) {
    const uint64_t m_load = m_next;  // rows of super-group sg + 1, requested below
    m_next = mask_sg(sg + 2);
    // register indices of this super-group's rows: node - n0 relative to v64 (half 0) / v128 (half 1); the trash
    // register v192 = index 128 / 64 for rows past the end and endpoints outside the tile
    const bool in = lane < RING && sg * RING + lane < R;
    const uint32_t ua = (uint32_t)(na - (int32_t)n0), ub = (uint32_t)(nb - (int32_t)n0);
    int pa = (in && ua < (uint32_t)nodes) ? (int)ua : 128;
    int pb = (in && ub < (uint32_t)nodes) ? (int)ub : 64;
#pragma unroll
    for (int g = 0; g < RING; g += 4) {
      // the next super-group's endpoints: requested ahead of this iteration's row loads, so that the wait for them at
      // the top of the next iteration (all but the 32 youngest operations) leaves the ring in flight
      if (g == 0) load_sel(sg + 1, na, nb);                     // past the last super-group: zeros, never used
      asm volatile("" : "+v"(pa), "+v"(pb));                    // the block's eight v_readlane stay HERE (hoisted to the loop
                                                                // top, the 64 results of a ring held 64 SGPRs live)
      if (MASKED) {
        // a masked-out row (its load came back as zeros through the zero-record descriptor) adds nothing: its twelve
        // instructions are skipped by a scalar branch -- the kernel is bound by this per-row instruction stream, and under a
        // ScalarFilter gate 58 % of the rows are such rows.  The loads stay unconditional: the wait counts stay static.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((m_cur >> (g + q)) & 1ull) {
            const int iq = __builtin_amdgcn_readlane(pa, g + q), jq = __builtin_amdgcn_readlane(pb, g + q);
            DMP_ACC1(v[g + q], iq, jq);
          }
        }
      } else {
        const int i0 = __builtin_amdgcn_readlane(pa, g), i1 = __builtin_amdgcn_readlane(pa, g + 1);
        const int i2 = __builtin_amdgcn_readlane(pa, g + 2), i3 = __builtin_amdgcn_readlane(pa, g + 3);
        const int j0 = __builtin_amdgcn_readlane(pb, g), j1 = __builtin_amdgcn_readlane(pb, g + 1);
        const int j2 = __builtin_amdgcn_readlane(pb, g + 2), j3 = __builtin_amdgcn_readlane(pb, g + 3);
        DMP_ACC4(v[g], v[g + 1], v[g + 2], v[g + 3], i0, i1, i2, i3, j0, j1, j2, j3);
      }
      // ... and the same four ring slots take the rows RING ahead: AFTER the adds (the block is a memory barrier to the
      // compiler), so the loads land in the registers the adds have just read -- no second register set, no copies
      const int r = (sg + 1) * RING + g;
      v[g] = load_row(r, (uint32_t)((m_load >> g) & 1ull)); v[g + 1] = load_row(r + 1, (uint32_t)((m_load >> (g + 1)) & 1ull));
      v[g + 2] = load_row(r + 2, (uint32_t)((m_load >> (g + 2)) & 1ull)); v[g + 3] = load_row(r + 3, (uint32_t)((m_load >> (g + 3)) & 1ull));
    }
    m_cur = m_load;
  }
  // the sums: 256 contiguous bytes per store instruction (this wave's columns of one node row and half)
  const uint32_t bytesO = tile_bytes(nodes, ldo, 2 * H);
  const rsrc_t rsO = make_rsrc(out + n0 * ldo, bytesO);
  const uint32_t orow = (uint32_t)(ldo * 4);
  // the kept bits of the tile's 64 node rows (wave-uniform: scalar loads; word indices clamped to the mask's last word --
  // what they say about rows past the tile's last node is not used, those stores fall outside the descriptor anyway)
  uint32_t nm0 = 0xffffffffu, nm1 = 0xffffffffu;
  if (NODES) {
    const int w = (int)(n0 >> 5), sh = (int)(n0 & 31), wl = (int)node_words - 1;
    const uint32_t q0 = nodemask[min(w, wl)], q1 = nodemask[min(w + 1, wl)], q2 = nodemask[min(w + 2, wl)];
    nm0 = (uint32_t)(((((uint64_t)q1) << 32) | q0) >> sh);
    nm1 = (uint32_t)(((((uint64_t)q2) << 32) | q1) >> sh);
  }
  auto put = [&](int node, float a, float b) {
    const uint32_t live = NODES ? (((node < 32 ? nm0 : nm1) >> (node & 31)) & 1u) : 1u;
    const rsrc_t rs = NODES ? make_rsrc(out + n0 * ldo, live ? bytesO : 0u) : rsO;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a * s0), rs, (int)voff, (int)((uint32_t)node * orow), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b * s1), rs, (int)(voff + H * 4), (int)((uint32_t)node * orow), 0);
  };
#pragma unroll
  for (int i = 0; i < 32; ++i) put(i, A0[i], B0[i]);
#pragma unroll
  for (int i = 0; i < 32; ++i) put(32 + i, A1[i], B1[i]);
}

}  // namespace
}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_seg_sum2_graphs_max_nodes(void) { return kAccNodes; }

static int seg_sum2_graphs_impl(const float *M, int64_t ldm, const int32_t *sel_a, const int32_t *sel_b, const int64_t *node_off,
                                const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb, int H, float s0, float s1,
                                float *out, int64_t ldo, const uint32_t *rowmask, int64_t mask_words, void *stream,
                                const uint32_t *nodemask = nullptr, int64_t node_words = 0) {
  if (Ba < 0 || Bb < 0 || H <= 0 || ldm < H || ldo < 2 * H || (Ba > 0 && ka < 1) || (Bb > 0 && kb < 1)) return DMP_ERR_BAD_ARG;
  if (Ba + Bb == 0) return DMP_OK;
  if (!M || !sel_a || !sel_b || !node_off || !edge_off || !out) return DMP_ERR_BAD_ARG;
  if ((H != 64 && H != 128) || ldm % 4 || ldo % 4 || !aligned16(M) || !aligned16(out)) return DMP_ERR_UNSUPPORTED;
  const int64_t tiles = (Ba > 0 ? (Ba + ka - 1) / ka : 0) + (Bb > 0 ? (Bb + kb - 1) / kb : 0);
  if (tiles >= ((int64_t)1 << 31)) return DMP_ERR_UNSUPPORTED;
  GraphTiles ts{node_off, edge_off, Ba, Bb, ka > 0 ? ka : 1, kb > 0 ? kb : 1};
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)tiles;
  if (rowmask && nodemask) {
    if (H == 128) seg_acc_graphs_k<128, true, true><<<nb, 128, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo, rowmask, mask_words, nodemask, node_words);
    else seg_acc_graphs_k<64, true, true><<<nb, 64, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo, rowmask, mask_words, nodemask, node_words);
  } else if (rowmask) {
    if (H == 128) seg_acc_graphs_k<128, true><<<nb, 128, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo, rowmask, mask_words);
    else seg_acc_graphs_k<64, true><<<nb, 64, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo, rowmask, mask_words);
  } else if (H == 128) seg_acc_graphs_k<128><<<nb, 128, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo);
  else seg_acc_graphs_k<64><<<nb, 64, 0, st>>>(M, ldm, sel_a, sel_b, ts, s0, s1, out, ldo);
  return check_launch();
}

int dmp_seg_sum2_graphs(const float *M, int64_t ldm, const int32_t *sel_a, const int32_t *sel_b, const int64_t *node_off,
                        const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb, int H, float s0, float s1,
                        float *out, int64_t ldo, void *stream) {
  return seg_sum2_graphs_impl(M, ldm, sel_a, sel_b, node_off, edge_off, Ba, Bb, ka, kb, H, s0, s1, out, ldo, nullptr, 0, stream);
}

int dmp_seg_sum2_graphs_masked(const float *M, int64_t ldm, const int32_t *sel_a, const int32_t *sel_b, const int64_t *node_off,
                               const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb, int H, float s0, float s1,
                               float *out, int64_t ldo, const uint32_t *rowmask, int64_t num_rows, const uint32_t *nodemask,
                               int64_t num_nodes, void *stream) {
  if (!rowmask || num_rows < 0 || (nodemask && num_nodes <= 0)) return DMP_ERR_BAD_ARG;
  return seg_sum2_graphs_impl(M, ldm, sel_a, sel_b, node_off, edge_off, Ba, Bb, ka, kb, H, s0, s1, out, ldo, rowmask,
                              (num_rows + 31) / 32, stream, nodemask, nodemask ? (num_nodes + 31) / 32 : 0);
}

}  // extern "C"
