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
// bytes of a tile of `rows` rows with leading dimension ld (floats) of which `cols` are touched
__device__ __forceinline__ uint32_t tile_bytes(int rows, int64_t ld, int cols) {
  return rows > 0 ? (uint32_t)(((int64_t)(rows - 1) * ld + cols) * 4) : 0u;
}

}  // namespace dmp
