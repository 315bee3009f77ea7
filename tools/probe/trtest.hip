// Probe of ds_read_b64_tr_b16 (via __builtin_amdgcn_ds_read_tr16_b64_v4f16) lane mapping.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
  __shared__ __fp16 s[16 * 32];  // [16 rows][32 cols], value = row*100 + col
  for (int i = threadIdx.x; i < 16 * 32; i += 64) s[i] = (__fp16)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  // group g reads rows 4g+q, columns 4p..4p+3 (block = 4 rows x 16 cols starting at col 0)
  auto ptr = (__attribute__((address_space(3))) fp16x4*)(s + (g * 4 + q) * 32 + 4 * p);
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(ptr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) if (l < 20 || l % 16 == 0) printf("lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
