// Dev aid: does structured buffer addressing (index * stride) reach beyond 4 GiB on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ i32x4 llvm_struct_buffer_load_v4i32(i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");
__device__ void llvm_struct_buffer_store_v4i32(i32x4 data, i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.store.v4i32");
__device__ i32x4 make_srsrc(const void *base, uint32_t stride, uint32_t rows) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  i32x4 r; r.x = (int)(uint32_t)a; r.y = (int)(((uint32_t)(a >> 32) & 0xFFFFu) | (stride << 16)); r.z = (int)rows; r.w = 0x00020000; return r;
}
__global__ void fill(float *a, int64_t rows) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < rows) for (int c = 0; c < 128; ++c) a[r * 128 + c] = (float)(r % 1000003) + 0.001f * c;
}
__global__ void probe(const float *a, int64_t rows, const int *idx, float *out, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  i32x4 rs = make_srsrc(a, 512, (uint32_t)rows);
  i32x4 v = llvm_struct_buffer_load_v4i32(rs, idx[t], 16, 0, 0);
  out[t] = __int_as_float(v.x);
}
int main() {
  const int64_t rows = 9000000;   // 4.6 GB
  float *a; hipMalloc(&a, rows * 512);
  fill<<<(rows + 255) / 256, 256>>>(a, rows);
  int h_idx[8] = {0, 1000, 8388607, 8388608, 8388609, 8999999, 9000000, -1};
  int *idx; float *out; hipMalloc(&idx, 32); hipMalloc(&out, 32);
  hipMemcpy(idx, h_idx, 32, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, rows, idx, out, 8);
  float h[8]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("row %d -> %.3f (expect %.3f)\n", h_idx[i], h[i], h_idx[i] >= 0 && h_idx[i] < rows ? (float)(h_idx[i] % 1000003) + 0.004f : 0.f);
  return 0;
}
