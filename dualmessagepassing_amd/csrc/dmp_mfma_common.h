// Shared pieces of the fp32 MFMA kernels (dmp_mfma.hip, dmp_typed.hip): tile geometry, the
// LDS-only barrier and the raw-buffer access helpers.  gfx950 only.
#pragma once
#include "dmp_common.h"

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kLdsStride = 132;
constexpr int kScrStride = 36;
constexpr int kSub = 32;                                  // rows per tile / phase
constexpr int kGroupThreads = 256;                        // 4 waves: one column slice each
constexpr int kPPThreads = 2 * kGroupThreads;
constexpr int kSubLoads = kSub * 32 / kGroupThreads;      // float4 loads per thread per tile (4)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence over every address
// space: each wave would drain its outstanding global loads AND stores before the barrier.
// Nothing is exchanged through global memory inside these kernels.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Raw buffer access: SGPR descriptor (base + byte range, out-of-range loads return 0 and stores are
// dropped) + per-lane byte offset + SGPR byte offset.  No VALU address arithmetic.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, uint32_t voff, uint32_t soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store4(float4 v, rsrc_t r, uint32_t voff, uint32_t soff) {
  u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, 0);
}
// ---- fp32 products on the bf16 matrix pipe ("bf16x6").  gfx950's f32-input MFMA runs at the f32 vector rate, 1/16 of
// the bf16 MFMA rate.  An fp32 value is the EXACT sum of three bf16 pieces (8 + 8 + 8 significant bits, round-to-nearest
// splits: x = hi + mid + lo up to the last piece's rounding, 2^-25 relative); a product a b is the sum of nine piece
// products, of which the six with piece indices i + j <= 2 carry everything down to 2^-24 of |a b| (the three dropped
// ones are below 2^-23 together).  Piece products are exact in the MFMA's fp32 accumulation (8 x 8 bits), so six
// v_mfma_f32_32x32x16_bf16 give an fp32-accurate 32x32x16 block in 6 x 32 cycles instead of 8 x 64 for the f32 form
// (2.67 x), at ~5.5 VALU instructions per operand element for the split.
typedef short bf16x8 __attribute__((ext_vector_type(8)));    // 8 bf16 = 4 VGPRs: one MFMA operand fragment
union Frag8 { bf16x8 v; uint32_t u[4]; };
struct Split8 { Frag8 hi, mid, lo; };
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {   // (bf16(a), bf16(b)), round to nearest even; a in the low half
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_pair(float f0, float f1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = cvt_pk_bf16(f0, f1);
  const float r0 = f0 - __uint_as_float(h << 16), r1 = f1 - __uint_as_float(h & 0xFFFF0000u);   // exact
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);   // exact
  l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ void split8(const float4 &x, const float4 &y, Split8 &s) {   // 8 consecutive k of one row / column
  split_pair(x.x, x.y, s.hi.u[0], s.mid.u[0], s.lo.u[0]);
  split_pair(x.z, x.w, s.hi.u[1], s.mid.u[1], s.lo.u[1]);
  split_pair(y.x, y.y, s.hi.u[2], s.mid.u[2], s.lo.u[2]);
  split_pair(y.z, y.w, s.hi.u[3], s.mid.u[3], s.lo.u[3]);
}
// acc += A B for a 32x32x16 block given as pieces; the small terms first
__device__ __forceinline__ f32x16 mfma_x6(const Split8 &a, const Split8 &b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo.v, b.hi.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi.v, b.lo.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid.v, b.mid.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid.v, b.hi.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi.v, b.mid.v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi.v, b.hi.v, acc, 0, 0, 0);
  return acc;
}

// ---- structured buffer access: address = base + index * stride + offset, the descriptor holds the row stride in bytes
// (<= 16383) and the number of rows; a row index beyond it (e.g. -1) reads zeros / drops the store, and no multiply is
// needed.  It does NOT reach beyond 4 GiB: index * stride is formed modulo 2^32 on gfx950 (scripts/sbuf_probe.hip: row
// 8,388,608 of a 512-byte-stride array reads row 0), although the range check is on the index.  Arrays of 4 GiB and
// more (one [E, 128] fp32 array of BASELINE config 4's shard is 4.3 GB) take the kernels' BIG instantiations, which
// address rows through 64-bit pointers (row_load4 / row_store4 below).  clang has no builtin for the indexed form: the
// LLVM intrinsics are declared by name (waitcnt tracking stays with the compiler, unlike inline asm).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ i32x4 llvm_struct_buffer_load_v4i32(i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");
__device__ void llvm_struct_buffer_store_v4i32(i32x4 data, i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4i32");
typedef i32x4 srsrc_t;
__device__ __forceinline__ srsrc_t make_srsrc(const void *base, int64_t ld_floats, int64_t rows) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  srsrc_t r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(((uint32_t)(a >> 32) & 0xFFFFu) | ((uint32_t)(ld_floats * 4) << 16));
  r.z = base ? (int)(uint32_t)rows : 0;
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ float4 sbuf_load4(srsrc_t r, int row, uint32_t col_bytes) {
  const i32x4 v = llvm_struct_buffer_load_v4i32(r, row, (int)col_bytes, 0, 0);
  return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void sbuf_store4(float4 v, srsrc_t r, int row, uint32_t col_bytes) {
  i32x4 u;
  u.x = __float_as_int(v.x); u.y = __float_as_int(v.y); u.z = __float_as_int(v.z); u.w = __float_as_int(v.w);
  llvm_struct_buffer_store_v4i32(u, r, row, (int)col_bytes, 0, 0);
}
extern int g_exact_fp32;      // development switch (dmp_dev_set_exact_fp32): 1 = f32-input MFMA instead of the bf16x6 products

// Row access of the kernels that gather / scatter rows by id.  !BIG: structured descriptor (above).  BIG: 64-bit pointer
// arithmetic; a negative row (padding slot, past the end) reads row 0 and returns zeros / skips the store.
template <bool BIG>
__device__ __forceinline__ float4 row_load4(srsrc_t rs, const float *base, int64_t ld, int row, uint32_t col_bytes) {
  if (!BIG) return sbuf_load4(rs, row, col_bytes);
  const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)(row < 0 ? 0 : row) * ld + (col_bytes >> 2));
  return row < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : v;
}
template <bool BIG>
__device__ __forceinline__ void row_store4(float4 v, srsrc_t rs, float *base, int64_t ld, int row, uint32_t col_bytes) {
  if (!BIG) { sbuf_store4(v, rs, row, col_bytes); return; }
  if (row >= 0) *reinterpret_cast<float4 *>(base + (int64_t)row * ld + (col_bytes >> 2)) = v;
}
inline bool fits4g(int64_t rows, int64_t ld_floats) { return rows * ld_floats * 4 < ((int64_t)1 << 32) - 65536; }

inline bool stride_ok(int64_t ld_floats) { return ld_floats > 0 && ld_floats * 4 <= 16383; }

// bytes of a tile of `rows` rows with leading dimension ld (floats) of which `cols` are touched
__device__ __forceinline__ uint32_t tile_bytes(int rows, int64_t ld, int cols) {
  return rows > 0 ? (uint32_t)(((int64_t)(rows - 1) * ld + cols) * 4) : 0u;
}

}  // namespace dmp
