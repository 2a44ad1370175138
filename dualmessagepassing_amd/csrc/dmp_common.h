// Shared host/device helpers for libdmp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmp_hip.h"

namespace dmp {

constexpr int kWave = 64;     // CDNA4 wavefront
constexpr int kBlock = 256;   // 4 waves per workgroup, one per SIMD
constexpr int kXcds = 8;      // MI355X: 8 XCDs, each with a private 4 MiB L2

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).
// Remap so that each XCD walks one contiguous slice of the rows: a batched
// graph is block diagonal, so a graph's node rows (re-read by all of its edges)
// then stay in ONE XCD's L2.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg / kXcds, r = nwg % kXcds, xcd = orig % kXcds;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / kXcds;
}

void set_last_hip_error(hipError_t e);

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_hip_error(e);
    return DMP_ERR_HIP;
  }
  return DMP_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The MLP activations of the layers: ReLU (slope 0) or LeakyReLU with negative slope a (the reference's default
// `leaky_relu`, a = 1/5.5: utils/act.py:27,466; UNC model.py:145-157).  For 0 <= a <= 1:
//   forward   max(x, a x)            == x > 0 ? x : a x   (torch's formula, one rounding of a x)
//   backward  y > 0 ? d : a d        on the SAVED OUTPUT y: sign(y) == sign(x) for a > 0, and a == 0 is torch's
//                                    threshold_backward on the output
// (a = 0 gives -0.0 where ReLU gives +0.0: equal under ==, invisible to the > 0 masks and to sums.)
inline bool slope_ok(float a) { return a >= 0.f && a <= 1.f; }
__device__ __forceinline__ float act_fwd(float x, float a) { return fmaxf(x, a * x); }
__device__ __forceinline__ float act_bwd(float y, float d, float a) { return y > 0.f ? d : a * d; }

}  // namespace dmp
