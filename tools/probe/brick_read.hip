// Read-pattern probe: the same [N][D][H][W][C] fp32 tensor (2 x 128^3 x 32 ch = 537 MB, four distinct
// copies in rotation: beyond the 256 MiB Infinity Cache) read once per launch by persistent blocks,
//   (a) linear: block b reads consecutive 32 KB chunks,
//   (b) 8 x 8 x 4 bricks: 32 segments of 1 KB (8 voxels x 128 B) per brick,
//   (c) 32 x 2 x 4 bricks: 8 segments of 4 KB,
//   (d) 8 x 8 columns marching along z: 8 segments of 1 KB per step (the z-ring kernels),
// every thread summing its 16-byte loads (two bricks in flight). Prints GB/s per pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int C = 32, S = 128, NB = 2;

__global__ __launch_bounds__(256) void read_linear(const float4* __restrict__ x, float* out, long n4) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long per = 2048;   // float4 per block chunk = 32 KB
  for (long c0 = (long)blockIdx.x * per; c0 < n4; c0 += (long)gridDim.x * per) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = x[c0 + q * 256 + threadIdx.x];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// lanes 8 per voxel (the kernels' own pattern): wave instruction = 8 voxels x 128 B = 1 KB contiguous
template <int BX, int BY, int BZ>
__global__ __launch_bounds__(256) void read_bricks8(const float4* __restrict__ x, float* out, int nbricks) {
  const int ntx = S / BX, nty = S / BY, ntz = S / BZ;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int t = threadIdx.x, q = t & 7, vl = t >> 3;   // 32 voxels per pass
  constexpr int PASSES = BX * BY * BZ / 32;
  static_assert(BX * BY * BZ % 32 == 0 && S % BX == 0 && S % BY == 0 && S % BZ == 0, "brick tiling");
  for (int b = blockIdx.x; b < nbricks; b += gridDim.x) {
    int r = b;
    const int tx = r % ntx; r /= ntx;
    const int ty = r % nty; r /= nty;
    const int tz = r % ntz;
    const int n = r / ntz;
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int v = vl + 32 * u;
      const int vx = v % BX, vy = (v / BX) % BY, vz = v / (BX * BY);
      const size_t vox = (((size_t)n * S + tz * BZ + vz) * S + ty * BY + vy) * S + tx * BX + vx;
      const float4 f = x[vox * (C / 4) + q];
      acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
    }
  }
  out[(size_t)blockIdx.x * 256 + t] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const size_t n = (size_t)NB * S * S * S * C;
  float* x[4];
  float* out;
  for (int i = 0; i < 4; ++i) {
    hipMalloc(&x[i], n * 4);
    hipMemset(x[i], 0, n * 4);
  }
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nbricks = NB * (S / 8) * (S / 8) * (S / 4);
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 4; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 16; ++i) launch(i & 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %6.0f GB/s\n", name, 1e3 * ms / 16, n * 4.0 / (ms / 16 * 1e-3) / 1e9);
  };
  for (int blocks : {512, 1024, 2048}) {
    printf("blocks %d\n", blocks);
    time([&](int i) { hipLaunchKernelGGL(read_linear, dim3(blocks), dim3(256), 0, 0, (const float4*)x[i], out, (long)(n / 4)); }, "linear 32 KB chunks");
    time([&](int i) { hipLaunchKernelGGL((read_bricks8<8, 8, 4>), dim3(blocks), dim3(256), 0, 0, (const float4*)x[i], out, nbricks); }, "8x8x4 bricks (1 KB runs)");
    time([&](int i) { hipLaunchKernelGGL((read_bricks8<32, 2, 4>), dim3(blocks), dim3(256), 0, 0, (const float4*)x[i], out, nbricks); }, "32x2x4 bricks (4 KB runs)");
    time([&](int i) { hipLaunchKernelGGL((read_bricks8<128, 2, 1>), dim3(blocks), dim3(256), 0, 0, (const float4*)x[i], out, nbricks); }, "128x2x1 bricks (32 KB runs)");
    time([&](int i) { hipLaunchKernelGGL((read_bricks8<8, 8, 1>), dim3(blocks), dim3(256), 0, 0, (const float4*)x[i], out, nbricks * 4); }, "8x8x1 bricks, 4x as many (z-ring step)");
  }
  return 0;
}
