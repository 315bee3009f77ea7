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

// =====================================================================================================
// Round 4: the remaining members of get_augmentations_unet (transform_factory/augmentations.py:52-127):
// RandGaussianSmoothd ("blur"), RandBiasFieldd ("rbf"), RandGridDistortiond ("distort"),
// RandGibbsNoised (the k-space half of "noise"); RandSimulateLowResolutiond ("lowres") is composed on
// the host side from the resize kernels of layout.hip. All HBM-bound single passes over [N][D][H][W][C].
// =====================================================================================================

// ---- 1-D filter along one spatial axis, zero padding (one pass per axis of a separable Gaussian) ----
// taps [N][2 R + 1]: the item's filter (an item that is not smoothed gets the unit impulse).
__global__ __launch_bounds__(256) void adell_axis_filter_kernel(
    const float* __restrict__ x, float* __restrict__ out, int D, int H, int W, int C, int axis,
    const float* __restrict__ taps, int R) {
  const int n = blockIdx.y;
  const long V = (long)D * H * W, E = V * C;
  const float* t = taps + (size_t)n * (2 * R + 1);
  const float* xi = x + (size_t)n * E;
  float* oi = out + (size_t)n * E;
  const int L = axis == 0 ? D : (axis == 1 ? H : W);
  const long stride = (axis == 0 ? (long)H * W : (axis == 1 ? (long)W : 1L)) * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long)gridDim.x * 256) {
    const long v = e / C;
    const int pos = axis == 0 ? (int)(v / ((long)H * W)) : (axis == 1 ? (int)((v / W) % H) : (int)(v % W));
    float acc = 0.f;
    for (int k = -R; k <= R; ++k) {
      const int q = pos + k;
      if (q >= 0 && q < L) acc = fmaf(t[k + R], xi[e + (long)k * stride], acc);
    }
    oi[e] = acc;
  }
}

extern "C" int adell_axis_filter(const float* x, float* out, int N, int D, int H, int W, int C,
                                 int axis, const float* taps, int radius, void* stream) {
  ADELL_REQUIRE(x && out && taps && x != out, "axis_filter: null or aliased pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && C > 0, "axis_filter: bad dims");
  ADELL_REQUIRE(axis >= 0 && axis <= 2 && radius >= 0 && radius <= 64, "axis_filter: bad axis / radius");
  long b = ((long)D * H * W * C + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adell_axis_filter_kernel, dim3((unsigned)b, N), dim3(256), 0,
                     (hipStream_t)stream, x, out, D, H, W, C, axis, taps, radius);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---- multiplicative polynomial bias field: out = x * exp(sum_ijk c[i][j][k] P_i(z) P_j(y) P_k(x)) ----
// coef [N][64] = the dense 4 x 4 x 4 Legendre coefficient cube of the item (degree <= 3; entries
// with i + j + k > 3 are zero in MONAI's field), coordinates linspace(-1, 1, size) per axis.
__device__ __forceinline__ void adell_legendre4(float u, float* p) {
  p[0] = 1.f;
  p[1] = u;
  p[2] = 0.5f * (3.f * u * u - 1.f);
  p[3] = 0.5f * (5.f * u * u * u - 3.f * u);
}

__global__ __launch_bounds__(256) void adell_bias_field_kernel(
    const float* __restrict__ x, float* __restrict__ out, int D, int H, int W, int C,
    const float* __restrict__ coef) {
  const int n = blockIdx.y;
  const long V = (long)D * H * W;
  __shared__ float c[64];
  if (threadIdx.x < 64) c[threadIdx.x] = coef[(size_t)n * 64 + threadIdx.x];
  __syncthreads();
  const float* xi = x + (size_t)n * V * C;
  float* oi = out + (size_t)n * V * C;
  const float sz = D > 1 ? 2.f / (D - 1) : 0.f, sy = H > 1 ? 2.f / (H - 1) : 0.f,
              sx = W > 1 ? 2.f / (W - 1) : 0.f;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int ox = (int)(v % W), oy = (int)((v / W) % H), oz = (int)(v / ((long)W * H));
    float pz[4], py[4], px[4];
    adell_legendre4(-1.f + sz * oz, pz);
    adell_legendre4(-1.f + sy * oy, py);
    adell_legendre4(-1.f + sx * ox, px);
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(c[(i * 4 + j) * 4 + k], px[k], s);
        f = fmaf(pz[i] * py[j], s, f);
      }
    const float g = expf(f);
    for (int ch = 0; ch < C; ++ch) oi[v * C + ch] = xi[v * C + ch] * g;
  }
}

extern "C" int adell_bias_field(const float* x, float* out, int N, int D, int H, int W, int C,
                                const float* coef, void* stream) {
  ADELL_REQUIRE(x && out && coef, "bias_field: null pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && C > 0, "bias_field: bad dims");
  long b = ((long)D * H * W + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adell_bias_field_kernel, dim3((unsigned)b, N), dim3(256), 0,
                     (hipStream_t)stream, x, out, D, H, W, C, coef);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---- resampling through per-axis coordinate tables (grid distortion): border padding ----------------
// lut [N][D + H + W]: for output index o along an axis, the input coordinate in voxels.
__global__ __launch_bounds__(256) void adell_axis_lut_sample_kernel(
    const float* __restrict__ x, float* __restrict__ out, int D, int H, int W, int C,
    const float* __restrict__ lut, int linear) {
  const int n = blockIdx.y;
  const long V = (long)D * H * W;
  const float* lz = lut + (size_t)n * (D + H + W);
  const float* ly = lz + D;
  const float* lx = ly + H;
  const float* xi = x + (size_t)n * V * C;
  float* oi = out + (size_t)n * V * C;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const int ox = (int)(v % W), oy = (int)((v / W) % H), oz = (int)(v / ((long)W * H));
    const float sz = fminf(fmaxf(lz[oz], 0.f), (float)(D - 1));
    const float sy = fminf(fmaxf(ly[oy], 0.f), (float)(H - 1));
    const float sx = fminf(fmaxf(lx[ox], 0.f), (float)(W - 1));
    float* o = oi + v * C;
    if (!linear) {
      const int iz = (int)nearbyintf(sz), iy = (int)nearbyintf(sy), ix = (int)nearbyintf(sx);
      const float* s = xi + (((long)iz * H + iy) * W + ix) * C;
      for (int c = 0; c < C; ++c) o[c] = s[c];
      continue;
    }
    const float fz = floorf(sz), fy = floorf(sy), fx = floorf(sx);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const float az = sz - fz, ay = sy - fy, ax = sx - fx;
    for (int c = 0; c < C; ++c) o[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = min(z0 + (k >> 2), D - 1), yy = min(y0 + ((k >> 1) & 1), H - 1),
                xx = min(x0 + (k & 1), W - 1);
      const float w = ((k >> 2) ? az : 1.f - az) * (((k >> 1) & 1) ? ay : 1.f - ay) *
                      ((k & 1) ? ax : 1.f - ax);
      if (w == 0.f) continue;
      const float* s = xi + (((long)zz * H + yy) * W + xx) * C;
      for (int c = 0; c < C; ++c) o[c] += w * s[c];
    }
  }
}

extern "C" int adell_axis_lut_sample(const float* x, float* out, int N, int D, int H, int W, int C,
                                     const float* lut, int linear, void* stream) {
  ADELL_REQUIRE(x && out && lut && x != out, "axis_lut_sample: null or aliased pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && C > 0, "axis_lut_sample: bad dims");
  long b = ((long)D * H * W + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(adell_axis_lut_sample_kernel, dim3((unsigned)b, N), dim3(256), 0,
                     (hipStream_t)stream, x, out, D, H, W, C, lut, linear);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---- k-space low-pass (Gibbs ringing): DFT along each spatial axis of a complex copy, a spherical
// mask about the centre of the SHIFTED spectrum, inverse DFTs, real part --------------------------------
// The transform is a direct O(L^2) DFT per line out of LDS (any length up to 1024, e.g. the 96 of
// BASELINE config 3: no power-of-two restriction); G adjacent lines per block so that the strided
// axes still read whole 64-byte runs. buf: [N][D][H][W][C] float2.
constexpr int kDftMaxL = 1024;
__global__ __launch_bounds__(256) void adell_dft_axis_kernel(float2* __restrict__ buf, long lines,
                                                             int L, long S, int G, int inverse) {
  extern __shared__ float2 sdft[];            // [G][L] data, then [L] twiddles
  float2* tw = sdft + (size_t)G * L;
  const int tid = threadIdx.x;
  for (int k = tid; k < L; k += 256) {
    const float a = (inverse ? 6.283185307179586f : -6.283185307179586f) * (float)k / (float)L;
    float sn, cs;
    sincosf(a, &sn, &cs);
    tw[k] = make_float2(cs, sn);
  }
  const long groups = (lines + G - 1) / G;
  for (long g = blockIdx.x; g < groups; g += gridDim.x) {
    __syncthreads();
    // line id -> (outer, inner): element (outer * L + k) * S + inner; a group = G consecutive inner
    const long line0 = g * G;
    for (int i = tid; i < G * L; i += 256) {
      const int k = i / G, j = i - k * G;
      const long line = line0 + j;
      float2 v = make_float2(0.f, 0.f);
      if (line < lines) {
        const long outer = line / S, inner = line - outer * S;
        v = buf[(outer * L + k) * S + inner];
      }
      sdft[(size_t)j * L + k] = v;
    }
    __syncthreads();
    for (int i = tid; i < G * L; i += 256) {
      const int k = i / G, j = i - k * G;
      const long line = line0 + j;
      if (line >= lines) continue;
      const float2* xl = sdft + (size_t)j * L;
      float re = 0.f, im = 0.f;
      int idx = 0;
      for (int m = 0; m < L; ++m) {
        const float2 w = tw[idx], v = xl[m];
        re = fmaf(v.x, w.x, fmaf(-v.y, w.y, re));
        im = fmaf(v.x, w.y, fmaf(v.y, w.x, im));
        idx += k;
        if (idx >= L) idx -= L;
      }
      const long outer = line / S, inner = line - outer * S;
      // (written after every read of the group: the second barrier above orders the reads of the
      // NEXT group; this group's inputs are all in LDS)
      buf[(outer * L + k) * S + inner] = make_float2(re, im);
    }
  }
}

__global__ __launch_bounds__(256) void adell_kspace_kernel(const float* __restrict__ x,
                                                           float2* __restrict__ buf,
                                                           float* __restrict__ out, int D, int H,
                                                           int W, int C,
                                                           const float* __restrict__ radius,
                                                           int phase) {
  // phase 0: buf = x (complex); phase 1: zero the coefficients outside the item's sphere;
  // phase 2: out = Re(buf) / (D H W)
  const int n = blockIdx.y;
  const long V = (long)D * H * W;
  const float r2 = phase == 1 ? radius[n] * radius[n] : 0.f;
  const float cz = 0.5f * (D - 1), cy = 0.5f * (H - 1), cx = 0.5f * (W - 1);
  const float inv = 1.0f / (float)V;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const size_t e0 = ((size_t)n * V + v) * C;
    if (phase == 0) {
      for (int c = 0; c < C; ++c) buf[e0 + c] = make_float2(x[e0 + c], 0.f);
    } else if (phase == 2) {
      for (int c = 0; c < C; ++c) out[e0 + c] = buf[e0 + c].x * inv;
    } else {
      const int kx = (int)(v % W), ky = (int)((v / W) % H), kz = (int)(v / ((long)W * H));
      // fftshift: coefficient k sits at position (k + L / 2) mod L of the shifted spectrum
      const float dz = (float)((kz + D / 2) % D) - cz, dy = (float)((ky + H / 2) % H) - cy,
                  dx = (float)((kx + W / 2) % W) - cx;
      if (dz * dz + dy * dy + dx * dx > r2)
        for (int c = 0; c < C; ++c) buf[e0 + c] = make_float2(0.f, 0.f);
    }
  }
}

extern "C" long adell_gibbs_workspace(int N, int D, int H, int W, int C) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return ADELL_E_BADARG;
  return (long)N * D * H * W * C * 8;
}

// out = Re(ifftn(ifftshift(fftshift(fftn(x)) * [dist from centre <= radius[n]]))) per item and
// channel over the three spatial axes (an item with radius >= the spectrum's half diagonal is unchanged
// up to rounding). workspace: adell_gibbs_workspace bytes.
extern "C" int adell_gibbs_lowpass(const float* x, float* out, int N, int D, int H, int W, int C,
                                   const float* radius, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  ADELL_REQUIRE(x && out && radius && workspace, "gibbs_lowpass: null pointer");
  ADELL_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && C > 0, "gibbs_lowpass: bad dims");
  ADELL_REQUIRE(D <= kDftMaxL && H <= kDftMaxL && W <= kDftMaxL, "gibbs_lowpass: axis longer than 1024");
  ADELL_REQUIRE(workspace_bytes >= (size_t)adell_gibbs_workspace(N, D, H, W, C),
                "gibbs_lowpass: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float2* buf = (float2*)workspace;
  long b = ((long)D * H * W + 255) / 256;
  if (b > 8192) b = 8192;
  auto pass = [&](int phase) {
    hipLaunchKernelGGL(adell_kspace_kernel, dim3((unsigned)b, N), dim3(256), 0, st, x, buf, out, D, H,
                       W, C, radius, phase);
  };
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dft_axis_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  auto dft = [&](int axis, int inverse) {
    const int L = axis == 0 ? D : (axis == 1 ? H : W);
    if (L == 1) return;
    const long S = (axis == 0 ? (long)H * W : (axis == 1 ? (long)W : 1L)) * C;
    const long total = (long)N * D * H * W * C;
    const long lines = total / L;        // outer * S with outer = N * (dims before the axis)
    int G = 8;
    while (G > 1 && (S % G != 0 || (size_t)(G + 1) * L * 8 > 96 * 1024)) G >>= 1;
    long groups = (lines + G - 1) / G;
    if (groups > 65535 * 4) groups = 65535 * 4;
    hipLaunchKernelGGL(adell_dft_axis_kernel, dim3((unsigned)groups), dim3(256),
                       (size_t)(G + 1) * L * sizeof(float2), st, buf, lines, L, S, G, inverse);
  };
  pass(0);
  for (int a = 0; a < 3; ++a) dft(a, 0);
  pass(1);
  for (int a = 0; a < 3; ++a) dft(a, 1);
  pass(2);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
