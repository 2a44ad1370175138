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

}  // namespace dmp
