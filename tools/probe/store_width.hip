// Store-width probe: the same 2 GiB (beyond the 256 MiB Infinity Cache) written (a) one dword per lane in the MFMA C layout (lanes
// 0-31 one 128-byte row, lanes 32-63 the row four further down; 16 rows per wave-tile), (b) one
// dword per lane fully contiguous (256 B per instruction), (c) 16 bytes per lane. Prints GB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void store_c_layout(float* p, long ntiles) {   // tile = 32 rows x 32 floats
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long t = wave; t < ntiles; t += nw) {
    float* q = p + t * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      q[row * 32 + li] = (float)r;
    }
  }
}
__global__ void store_dword(float* p, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long t = wave; t < ntiles; t += nw) {
    float* q = p + t * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r * 64 + lane] = (float)r;
  }
}
__global__ void store_x4(float* p, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long t = wave; t < ntiles; t += nw) {
    float4* q = reinterpret_cast<float4*>(p + t * 1024);
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r * 64 + lane] = make_float4(r, r, r, r);
  }
}
// rows of 128 B at a stride of 256 B (the stride-2 lattice of the fused backward-data kernel)
__global__ void store_c_strided(float* p, long ntiles) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * (blockDim.x >> 6);
  for (long t = wave; t < ntiles; t += nw) {
    float* q = p + (t >> 1) * 2048 + (t & 1) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      q[row * 64 + li] = (float)r;
    }
  }
}

int main() {
  const long bytes = 2048L << 20, ntiles = bytes / 4096;
  float* p;
  hipMalloc(&p, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"dword, MFMA C layout (2 x 128 B rows / instr)", "dword, contiguous (256 B / instr)",
                          "dwordx4 (1 KiB / instr)", "dword, C layout, rows at stride 256 B"};
  for (int k = 0; k < 4; ++k) {
    for (int rep = 0; rep < 13; ++rep) {
      if (rep == 3) hipEventRecord(e0);
      if (k == 0) hipLaunchKernelGGL(store_c_layout, dim3(2048), dim3(256), 0, 0, p, ntiles);
      if (k == 1) hipLaunchKernelGGL(store_dword, dim3(2048), dim3(256), 0, 0, p, ntiles);
      if (k == 2) hipLaunchKernelGGL(store_x4, dim3(2048), dim3(256), 0, 0, p, ntiles);
      if (k == 3) hipLaunchKernelGGL(store_c_strided, dim3(2048), dim3(256), 0, 0, p, ntiles);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.1f us  %6.0f GB/s\n", names[k], ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e9);
  }
  return 0;
}
