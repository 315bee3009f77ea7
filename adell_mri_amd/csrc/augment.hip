// Device-side batch augmentation: the arithmetic of the transforms the
// reference composes in transform_factory/augmentations.py:19-178 (get_augmentations_unet), for
// volumes that already live in HBM -- no worker processes, no host round trip.
//
//   intensity  RandAdjustContrastd(gamma) : ((x - min) / (max - min + 1e-7))^gamma * (max - min) + min
//              RandStdShiftIntensityd     : x + factor * std(x)
//   noise      RandRicianNoised           : sqrt((x + n1)^2 + n2^2), n1, n2 ~ N(0, std)
//   affine /   RandAffined                : resample at  src = A (dst - centre) + centre  (voxel
//   shear                                   coordinates, trilinear or nearest, zero / border /
//                                           reflection padding)
//
// The random draws (whether an item gets a transform, gamma, factor, std, the matrix) are made on
// the host by adell_mri_amd/utils/augment.py and arrive as per-item parameter rows; the per-element
// noise is Philox-4x32-10 + Box-Muller, a function of (seed, offset, element index). All three
// kernels are HBM-bound single passes.
#include "common.h"

// ---- per-item (min, max, sum, sum of squares) ------------------------------------------------
// grid (blocks per item, N); partial [N][blocks][4]; folded in fixed order by the second kernel
__global__ __launch_bounds__(256) void adell_item_stats_partial_kernel(const float* __restrict__ x,
                                                                       long per_item,
                                                                       float* __restrict__ part) {
  __shared__ float sh[4][4];
  const int n = blockIdx.y;
  const float* xi = x + (size_t)n * per_item;
  float mn = INFINITY, mx = -INFINITY, s1 = 0.f, s2 = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_item; i += (long)gridDim.x * 256) {
    const float v = xi[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    s1 += v;
    s2 += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w][0] = mn; sh[w][1] = mx; sh[w][2] = s1; sh[w][3] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = part + ((size_t)n * gridDim.x + blockIdx.x) * 4;
    p[0] = fminf(fminf(sh[0][0], sh[1][0]), fminf(sh[2][0], sh[3][0]));
    p[1] = fmaxf(fmaxf(sh[0][1], sh[1][1]), fmaxf(sh[2][1], sh[3][1]));
    p[2] = (sh[0][2] + sh[1][2]) + (sh[2][2] + sh[3][2]);
    p[3] = (sh[0][3] + sh[1][3]) + (sh[2][3] + sh[3][3]);
  }
}

// out [N][4] = (min, max, mean, population standard deviation)
__global__ __launch_bounds__(64) void adell_item_stats_final_kernel(const float* __restrict__ part,
                                                                    int blocks, long per_item,
                                                                    float* __restrict__ out) {
  const int n = blockIdx.x;
  if (threadIdx.x != 0) return;
  float mn = INFINITY, mx = -INFINITY;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < blocks; ++b) {
    const float* p = part + ((size_t)n * blocks + b) * 4;
    mn = fminf(mn, p[0]);
    mx = fmaxf(mx, p[1]);
    s1 += (double)p[2];
    s2 += (double)p[3];
  }
  const double mean = s1 / (double)per_item;
  double var = s2 / (double)per_item - mean * mean;
  if (var < 0.0) var = 0.0;
  out[n * 4 + 0] = mn;
  out[n * 4 + 1] = mx;
  out[n * 4 + 2] = (float)mean;
  out[n * 4 + 3] = (float)sqrt(var);
}

static int adell_item_stats_blocks(long per_item) {
  long b = (per_item + 256 * 16 - 1) / (256 * 16);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" long adell_item_stats_workspace(int N, long per_item) {
  if (N <= 0 || per_item <= 0) return ADELL_E_BADARG;
  return (long)sizeof(float) * N * adell_item_stats_blocks(per_item) * 4;
}

extern "C" int adell_item_stats(const float* x, int N, long per_item, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(x && out && workspace, "item_stats: null pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && per_item > 0, "item_stats: bad dims");
  ADELL_REQUIRE((long)workspace_bytes >= adell_item_stats_workspace(N, per_item),
                "item_stats: workspace too small");
  const int blocks = adell_item_stats_blocks(per_item);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_item_stats_partial_kernel, dim3(blocks, N), dim3(256), 0, st, x, per_item,
                     (float*)workspace);
  hipLaunchKernelGGL(adell_item_stats_final_kernel, dim3(N), dim3(64), 0, st,
                     (const float*)workspace, blocks, per_item, out);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---- intensity + noise in one pass ---------------------------------------------------------------
// params [N][8]: {min, range, gamma (<= 0: no contrast change), shift, noise std (<= 0: none), -, -, -}
__global__ __launch_bounds__(256) void adell_aug_intensity_kernel(const float* __restrict__ x,
                                                                  float* __restrict__ out,
                                                                  long per_item,
                                                                  const float* __restrict__ params,
                                                                  uint32_t seed_lo, uint32_t seed_hi,
                                                                  uint32_t rng_offset) {
  const int n = blockIdx.y;
  const float* p = params + (size_t)n * 8;
  const float mn = p[0], range = p[1], gamma = p[2], shift = p[3], nstd = p[4];
  const float inv = 1.0f / (range + 1e-7f);
  const float* xi = x + (size_t)n * per_item;
  float* oi = out + (size_t)n * per_item;
  const long pairs = (per_item + 1) >> 1;     // one Philox call -> 4 normals -> 2 elements
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < pairs; j += (long)gridDim.x * 256) {
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (nstd > 0.f) {
      const long gi = (long)n * pairs + j;
      const uint4 r = adell_philox4((uint32_t)gi, (uint32_t)(gi >> 32), rng_offset, 1u, seed_lo,
                                    seed_hi);
      // Box-Muller: (u1, u2) in (0, 1] x [0, 1) -> two independent standard normals
      const float u1 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f);
      const float u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
      const float u3 = ((float)(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f);
      const float u4 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
      const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
      nz[0] = ra * cosf(6.28318530717958648f * u2) * nstd;
      nz[1] = ra * sinf(6.28318530717958648f * u2) * nstd;
      nz[2] = rb * cosf(6.28318530717958648f * u4) * nstd;
      nz[3] = rb * sinf(6.28318530717958648f * u4) * nstd;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long i = 2 * j + h;
      if (i >= per_item) break;
      float v = xi[i];
      if (gamma > 0.f) {
        const float t = (v - mn) * inv;
        v = (t > 0.f ? exp2f(gamma * log2f(t)) : 0.f) * range + mn;
      }
      v += shift;
      if (nstd > 0.f) {
        const float a = v + nz[2 * h], b = nz[2 * h + 1];
        v = sqrtf(a * a + b * b);
      }
      oi[i] = v;
    }
  }
}

extern "C" int adell_aug_intensity(const float* x, float* out, int N, long per_item,
                                   const float* params, uint64_t seed, uint32_t rng_offset,
                                   void* stream) {
  ADELL_REQUIRE(x && out && params, "aug_intensity: null pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && per_item > 0, "aug_intensity: bad dims");
  long b = ((per_item + 1) / 2 + 256 * 8 - 1) / (256 * 8);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(adell_aug_intensity_kernel, dim3((unsigned)b, N), dim3(256), 0,
                     (hipStream_t)stream, x, out, per_item, params, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), rng_offset);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---- affine resampling ---------------------------------------------------------------------------
// x, out: [N][D][H][W][C] (NDHWC). theta [N][12]: rows of the 3 x 4 matrix that maps centred output
// voxel coordinates (z, y, x) - (size - 1) / 2 to centred input coordinates.
__device__ __forceinline__ float adell_reflect(float c, int size) {
  // reflection about -0.5 and size - 0.5 (torch grid_sample, align_corners = False), then clamp
  const float span = (float)size;
  float t = fabsf(c + 0.5f);
  const float flips = floorf(t / span);
  const float extra = t - flips * span;
  t = ((int)flips & 1) ? span - extra : extra;
  t -= 0.5f;
  return fminf(fmaxf(t, 0.f), (float)(size - 1));
}

__global__ __launch_bounds__(256) void adell_affine_sample_kernel(
    const float* __restrict__ x, float* __restrict__ out, int D, int H, int W, int C,
    const float* __restrict__ theta, int linear, int pad_mode) {
  const int n = blockIdx.y;
  const long V = (long)D * H * W;
  const float* t = theta + (size_t)n * 12;
  const float cz = 0.5f * (D - 1), cy = 0.5f * (H - 1), cx = 0.5f * (W - 1);
  const float* xi = x + (size_t)n * V * C;
  float* oi = out + (size_t)n * V * C;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int ox = (int)(v % W), oy = (int)((v / W) % H), oz = (int)(v / ((long)W * H));
    const float dz = oz - cz, dy = oy - cy, dx = ox - cx;
    float sz = t[0] * dz + t[1] * dy + t[2] * dx + t[3] + cz;
    float sy = t[4] * dz + t[5] * dy + t[6] * dx + t[7] + cy;
    float sx = t[8] * dz + t[9] * dy + t[10] * dx + t[11] + cx;
    if (pad_mode == 1) {          // border: clamp the coordinate
      sz = fminf(fmaxf(sz, 0.f), (float)(D - 1));
      sy = fminf(fmaxf(sy, 0.f), (float)(H - 1));
      sx = fminf(fmaxf(sx, 0.f), (float)(W - 1));
    } else if (pad_mode == 2) {   // reflection
      sz = adell_reflect(sz, D);
      sy = adell_reflect(sy, H);
      sx = adell_reflect(sx, W);
    }
    float* o = oi + v * C;
    if (!linear) {
      const int iz = (int)nearbyintf(sz), iy = (int)nearbyintf(sy), ix = (int)nearbyintf(sx);
      const bool ok = iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float* s = xi + (((long)iz * H + iy) * W + ix) * C;
      for (int c = 0; c < C; ++c) o[c] = ok ? s[c] : 0.f;
      continue;
    }
    const float fz = floorf(sz), fy = floorf(sy), fx = floorf(sx);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const float az = sz - fz, ay = sy - fy, ax = sx - fx;
    for (int c = 0; c < C; ++c) o[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = z0 + (k >> 2), yy = y0 + ((k >> 1) & 1), xx = x0 + (k & 1);
      const float w = ((k >> 2) ? az : 1.f - az) * (((k >> 1) & 1) ? ay : 1.f - ay) *
                      ((k & 1) ? ax : 1.f - ax);
      if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W || w == 0.f) continue;
      const float* s = xi + (((long)zz * H + yy) * W + xx) * C;
      for (int c = 0; c < C; ++c) o[c] += w * s[c];
    }
  }
}

extern "C" int adell_affine_sample(const float* x, float* out, int N, int D, int H, int W, int C,
                                   const float* theta, int linear, int pad_mode, void* stream) {
  ADELL_REQUIRE(x && out && theta && x != out, "affine_sample: null or aliased pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && C > 0, "affine_sample: bad dims");
  ADELL_REQUIRE(pad_mode >= 0 && pad_mode <= 2, "affine_sample: pad_mode 0 zeros / 1 border / 2 reflection");
  long b = ((long)D * H * W + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adell_affine_sample_kernel, dim3((unsigned)b, N), dim3(256), 0,
                     (hipStream_t)stream, x, out, D, H, W, C, theta, linear, pad_mode);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
