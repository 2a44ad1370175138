#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[32 * 128];
  for (int i = threadIdx.x; i < 32 * 128; i += 64) img[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
  // group g reads block rows 4g..4g+3, cols 0..15
  const uint16_t* addr = &img[(4 * g + q) * 128 + 4 * p];
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
int main() {
  uint16_t h[32 * 128], *d, *o, r[256];
  for (int i = 0; i < 32 * 128; ++i) h[i] = (uint16_t)((i / 128) * 256 + (i % 128));   // row*256 + col
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", r[l*4+e] >> 8, r[l*4+e] & 255); printf("\n"); }
  return 0;
}
