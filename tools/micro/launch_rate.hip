// How long does the chip take to run N (nearly) empty 256-thread workgroups that each claim
// L bytes of LDS and V VGPRs? (the fixed per-block cost under the conv kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern "C" __global__ __launch_bounds__(256, 2) void k_big(float* out, int n) {
  extern __shared__ float sm[];
  asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if (n < 0) { sm[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = sm[255 - threadIdx.x]; }
}
extern "C" __global__ __launch_bounds__(256) void k_small(float* out, int n) {
  extern __shared__ float sm[];
  if (n < 0) { sm[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = sm[255 - threadIdx.x]; }
}
template <typename K>
static void run(const char* name, K k, int blocks, size_t lds) {
  float* d;
  hipMalloc(&d, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d, 1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s blocks=%d lds=%zu: %.1f us per launch, %.2f us per block-slot (512 slots)\n", name, blocks, lds,
         1e3 * ms / 20, 1e3 * ms / 20 / (blocks / 512.0));
  hipFree(d);
}
int main() {
  run("big(256 vgpr)", k_big, 8192, 57 * 1024);
  run("big(256 vgpr)", k_big, 8192, 1024);
  run("small", k_small, 8192, 57 * 1024);
  run("small", k_small, 8192, 1024);
  run("big(256 vgpr)", k_big, 65536, 57 * 1024);
  return 0;
}
