// Data-movement kernels (HBM-bound, no arithmetic besides index math and, in the
// backward of the resampler, a fixed-order sum):
//   * channel concat / split of NDHWC tensors -- the N-way torch.cat of DenseBlock
//     (adell_mri/modules/layers/standard_blocks.py:365-371); the 2-way decoder
//     concat never comes here (the conv kernel reads two sources directly);
//   * nearest-neighbour resampling, F.interpolate(x, size) with its default mode
//     (standard_blocks.py:368, unet.py:796-799), forward and backward.
#include "common.h"

// dst[v][coff + c] = src[v][c]   (dir 0)   |   src[v][c] = dst[v][coff + c]  (dir 1)
__global__ __launch_bounds__(256) void adell_copy_channels_kernel(float* __restrict__ dst,
                                                                  float* __restrict__ src,
                                                                  long V, int Cdst, int Csrc,
                                                                  int coff, int dir, int vec) {
  if (vec) {
    const int c4 = Csrc >> 2;
    const long n = V * c4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
      const long v = i / c4;
      const int c = (int)(i - v * c4) << 2;
      float4* d = reinterpret_cast<float4*>(dst + v * Cdst + coff + c);
      float4* s = reinterpret_cast<float4*>(src + v * Csrc + c);
      if (dir == 0) *d = *s; else *s = *d;
    }
  } else {
    const long n = V * Csrc;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
      const long v = i / Csrc;
      const int c = (int)(i - v * Csrc);
      if (dir == 0) dst[v * Cdst + coff + c] = src[v * Csrc + c];
      else src[v * Csrc + c] = dst[v * Cdst + coff + c];
    }
  }
}

// direction 0: write `part` [V][Cpart] into channels [coff, coff+Cpart) of `full` [V][Cfull];
// direction 1: read them back out of `full` into `part`.
extern "C" int adell_copy_channels(float* full, float* part, long V, int Cfull, int Cpart,
                                   int coff, int direction, void* stream) {
  ADELL_REQUIRE(full && part, "copy_channels: null pointer");
  ADELL_REQUIRE(V > 0 && Cpart > 0 && coff >= 0 && coff + Cpart <= Cfull,
                "copy_channels: bad channel range");
  ADELL_REQUIRE(direction == 0 || direction == 1, "copy_channels: direction must be 0/1");
  const int vec = (Cfull % 4 == 0) && (Cpart % 4 == 0) && (coff % 4 == 0) &&
                  (((uintptr_t)full | (uintptr_t)part) & 15) == 0;
  long n = vec ? V * (Cpart / 4) : V * (long)Cpart;
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adell_copy_channels_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, full, part, V, Cfull, Cpart, coff, direction, vec);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// dst[off_r + i] = src_r[i] for every row r of a device-resident table of (source pointer,
// destination offset, element count) triples; one block per row (rows are <= 16384 elements).
// Gathers the per-parameter gradient tensors autograd produced into the flat gradient buffer
// the fused optimiser and the gradient all-reduce work on.
__global__ __launch_bounds__(256) void adell_multi_copy_kernel(const long* __restrict__ table,
                                                               float* __restrict__ dst) {
  const long* row = table + (size_t)blockIdx.x * 3;
  const float* src = reinterpret_cast<const float*>(row[0]);
  float* d = dst + row[1];
  const long n = row[2];
  if (((((uintptr_t)src) | ((uintptr_t)d)) & 15) == 0) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256)
      reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(src)[i];
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = src[i];
  } else {
    for (long i = threadIdx.x; i < n; i += 256) d[i] = src[i];
  }
}

extern "C" int adell_multi_copy(const long* table, int rows, float* dst, void* stream) {
  ADELL_REQUIRE(table && dst && rows > 0, "multi_copy: bad arguments");
  hipLaunchKernelGGL(adell_multi_copy_kernel, dim3((unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, table, dst);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

struct ResampleArgs {
  const float* in;
  float* out;
  int N, C, Di, Hi, Wi, Do, Ho, Wo;
};

// torch's nearest: src = min(floor(dst * in / out), in - 1) (scale computed in float)
__device__ __forceinline__ int adell_nn_src(int o, int in, int out) {
  const float scale = (float)in / (float)out;
  int s = (int)floorf((float)o * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ __launch_bounds__(256) void adell_nearest_fwd_kernel(ResampleArgs a) {
  const long nvox = (long)a.N * a.Do * a.Ho * a.Wo;
  const long n = nvox * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int x = (int)(v % a.Wo); v /= a.Wo;
    const int y = (int)(v % a.Ho); v /= a.Ho;
    const int z = (int)(v % a.Do);
    const int nb = (int)(v / a.Do);
    const int sx = adell_nn_src(x, a.Wi, a.Wo), sy = adell_nn_src(y, a.Hi, a.Ho),
              sz = adell_nn_src(z, a.Di, a.Do);
    a.out[i] = a.in[((((size_t)nb * a.Di + sz) * a.Hi + sy) * a.Wi + sx) * a.C + c];
  }
}

// backward: din[i] = sum of dout over the output voxels whose source is i. The set
// is an axis-aligned box found by scanning a small candidate range per axis.
__device__ __forceinline__ void adell_nn_range(int s, int in, int out, int* lo, int* hi) {
  // candidates around s*out/in; widen by one on both sides and test exactly
  int a = (int)floorf((float)s * (float)out / (float)in) - 1;
  int b = (int)ceilf((float)(s + 1) * (float)out / (float)in) + 1;
  if (a < 0) a = 0;
  if (b > out) b = out;
  while (a < b && adell_nn_src(a, in, out) != s) ++a;
  while (b > a && adell_nn_src(b - 1, in, out) != s) --b;
  *lo = a;
  *hi = b;
}

__global__ __launch_bounds__(256) void adell_nearest_bwd_kernel(ResampleArgs a) {
  // here a.in = dout [N,Do,Ho,Wo,C], a.out = din [N,Di,Hi,Wi,C]
  const long n = (long)a.N * a.Di * a.Hi * a.Wi * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int x = (int)(v % a.Wi); v /= a.Wi;
    const int y = (int)(v % a.Hi); v /= a.Hi;
    const int z = (int)(v % a.Di);
    const int nb = (int)(v / a.Di);
    int x0, x1, y0, y1, z0, z1;
    adell_nn_range(x, a.Wi, a.Wo, &x0, &x1);
    adell_nn_range(y, a.Hi, a.Ho, &y0, &y1);
    adell_nn_range(z, a.Di, a.Do, &z0, &z1);
    float s = 0.f;
    for (int oz = z0; oz < z1; ++oz)
      for (int oy = y0; oy < y1; ++oy)
        for (int ox = x0; ox < x1; ++ox)
          s += a.in[((((size_t)nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.C + c];
    a.out[i] = s;
  }
}

static int adell_resample(const float* in, float* out, int N, int C, int Di, int Hi, int Wi,
                          int Do, int Ho, int Wo, int bwd, hipStream_t st) {
  ADELL_REQUIRE(in && out, "resample: null pointer");
  ADELL_REQUIRE(N > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0,
                "resample: bad dims");
  ResampleArgs a = {in, out, N, C, Di, Hi, Wi, Do, Ho, Wo};
  const long n = (long)N * C * (bwd ? (long)Di * Hi * Wi : (long)Do * Ho * Wo);
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (bwd)
    hipLaunchKernelGGL(adell_nearest_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(adell_nearest_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_interp_nearest_fwd(const float* x, float* y, int N, int C, int Di, int Hi,
                                        int Wi, int Do, int Ho, int Wo, void* stream) {
  return adell_resample(x, y, N, C, Di, Hi, Wi, Do, Ho, Wo, 0, (hipStream_t)stream);
}
extern "C" int adell_interp_nearest_bwd(const float* dy, float* dx, int N, int C, int Di, int Hi,
                                        int Wi, int Do, int Ho, int Wo, void* stream) {
  return adell_resample(dy, dx, N, C, Di, Hi, Wi, Do, Ho, Wo, 1, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// MaxPool3d (torch.nn.MaxPool3d, ceil_mode=False, dilation 1; padding with -inf):
// unet.py:335,368,595-603 (backbone / "resnet" encoders and, through
// init_encoder_backbone, the U-Net++ encoder), res_net.py:180,209.
// idx keeps the argmax as the flat (z*H + y)*W + x position inside the item.
// ---------------------------------------------------------------------------
struct PoolArgs {
  const float* x;
  float* y;
  int* idx;
  const float* dy;
  float* dx;
  int N, C, D, H, W, Do, Ho, Wo, KD, KH, KW, SD, SH, SW, PD, PH, PW;
};

__global__ __launch_bounds__(256) void adell_maxpool3d_fwd_kernel(PoolArgs a) {
  const long n = (long)a.N * a.Do * a.Ho * a.Wo * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int ox = (int)(v % a.Wo); v /= a.Wo;
    const int oy = (int)(v % a.Ho); v /= a.Ho;
    const int oz = (int)(v % a.Do);
    const int nb = (int)(v / a.Do);
    float best = -INFINITY;
    int bi = -1;
    for (int kz = 0; kz < a.KD; ++kz) {
      const int z = oz * a.SD - a.PD + kz;
      if (z < 0 || z >= a.D) continue;
      for (int ky = 0; ky < a.KH; ++ky) {
        const int y = oy * a.SH - a.PH + ky;
        if (y < 0 || y >= a.H) continue;
        for (int kx = 0; kx < a.KW; ++kx) {
          const int x = ox * a.SW - a.PW + kx;
          if (x < 0 || x >= a.W) continue;
          const int flat = (z * a.H + y) * a.W + x;
          const float t = a.x[((size_t)nb * a.D * a.H * a.W + flat) * a.C + c];
          if (t > best || bi < 0) {
            best = t;
            bi = flat;
          }
        }
      }
    }
    a.y[i] = best;
    a.idx[i] = bi;
  }
}

__global__ __launch_bounds__(256) void adell_maxpool3d_bwd_kernel(PoolArgs a) {
  const long n = (long)a.N * a.D * a.H * a.W * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int x = (int)(v % a.W); v /= a.W;
    const int y = (int)(v % a.H); v /= a.H;
    const int z = (int)(v % a.D);
    const int nb = (int)(v / a.D);
    const int flat = (z * a.H + y) * a.W + x;
    // windows o with o*S - P <= pos <= o*S - P + K - 1
    int z0 = (z + a.PD - a.KD + a.SD) / a.SD, z1 = (z + a.PD) / a.SD;
    int y0 = (y + a.PH - a.KH + a.SH) / a.SH, y1 = (y + a.PH) / a.SH;
    int x0 = (x + a.PW - a.KW + a.SW) / a.SW, x1 = (x + a.PW) / a.SW;
    if (z + a.PD - a.KD + 1 < 0) z0 = 0;
    if (y + a.PH - a.KH + 1 < 0) y0 = 0;
    if (x + a.PW - a.KW + 1 < 0) x0 = 0;
    if (z0 < 0) z0 = 0;
    if (y0 < 0) y0 = 0;
    if (x0 < 0) x0 = 0;
    if (z1 >= a.Do) z1 = a.Do - 1;
    if (y1 >= a.Ho) y1 = a.Ho - 1;
    if (x1 >= a.Wo) x1 = a.Wo - 1;
    float s = 0.f;
    for (int oz = z0; oz <= z1; ++oz)
      for (int oy = y0; oy <= y1; ++oy)
        for (int ox = x0; ox <= x1; ++ox) {
          const size_t o = ((((size_t)nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.C + c;
          if (a.idx[o] == flat) s += a.dy[o];
        }
    a.dx[i] = s;
  }
}

// The same for channels in fours and windows that do not overlap (kernel == stride on every axis:
// every pooling layer of the BASELINE configs). One block per output row (n, oz, oy) forward and per
// input row (n, z, y) backward, 16 bytes of channels per thread, no integer division per element
// beyond one by C / 4: the generic kernels above ran the 64 x 128^3 -> 65^3 layer of the backbone
// U-Net at 2.1 (forward) and 0.7 TB/s (backward).
__global__ __launch_bounds__(256) void adell_maxpool3d_fwd_rows_kernel(PoolArgs a) {
  const int row = blockIdx.x;                       // (n * Do + oz) * Ho + oy
  const int oy = row % a.Ho, t = row / a.Ho;
  const int oz = t % a.Do, nb = t / a.Do;
  const int per = a.C >> 2, items = a.Wo * per;
  const float* xb = a.x + (size_t)nb * a.D * a.H * a.W * a.C;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int ox = it / per, c = (it - ox * per) << 2;
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {-1, -1, -1, -1};
    for (int kz = 0; kz < a.KD; ++kz) {
      const int z = oz * a.SD - a.PD + kz;
      if (z < 0 || z >= a.D) continue;
      for (int ky = 0; ky < a.KH; ++ky) {
        const int y = oy * a.SH - a.PH + ky;
        if (y < 0 || y >= a.H) continue;
        for (int kx = 0; kx < a.KW; ++kx) {
          const int x = ox * a.SW - a.PW + kx;
          if (x < 0 || x >= a.W) continue;
          const int flat = (z * a.H + y) * a.W + x;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)flat * a.C + c);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (v[j] > best[j] || bi[j] < 0) {
              best[j] = v[j];
              bi[j] = flat;
            }
        }
      }
    }
    const size_t o = ((size_t)row * a.Wo + ox) * a.C + c;
    *reinterpret_cast<f32x4*>(a.y + o) = best;
    *reinterpret_cast<int4*>(a.idx + o) = make_int4(bi[0], bi[1], bi[2], bi[3]);
  }
}

__global__ __launch_bounds__(256) void adell_maxpool3d_bwd_rows_kernel(PoolArgs a) {
  const int row = blockIdx.x;                       // (n * D + z) * H + y
  const int y = row % a.H, t = row / a.H;
  const int z = t % a.D, nb = t / a.D;
  const int per = a.C >> 2, items = a.W * per;
  // the one window that holds this row (kernel == stride), if any
  const int oz = (z + a.PD) / a.SD, oy = (y + a.PH) / a.SH;
  const bool rowok = oz < a.Do && oy < a.Ho;
  const size_t obase = (((size_t)nb * a.Do + (rowok ? oz : 0)) * a.Ho + (rowok ? oy : 0)) * a.Wo;
  float* dxr = a.dx + (size_t)row * a.W * a.C;
  const int flat0 = (z * a.H + y) * a.W;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int x = it / per, c = (it - x * per) << 2;
    const int ox = (x + a.PW) / a.SW;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (rowok && ox < a.Wo) {
      const size_t o = (obase + ox) * a.C + c;
      const int4 id = *reinterpret_cast<const int4*>(a.idx + o);
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.dy + o);
      const int flat = flat0 + x;
      v[0] = id.x == flat ? g[0] : 0.f;
      v[1] = id.y == flat ? g[1] : 0.f;
      v[2] = id.z == flat ? g[2] : 0.f;
      v[3] = id.w == flat ? g[3] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dxr + (size_t)x * a.C + c) = v;
  }
}

static bool adell_pool_rows_ok(const PoolArgs& a, const void* p0, const void* p1, const void* p2) {
  return a.C % 4 == 0 && a.KD == a.SD && a.KH == a.SH && a.KW == a.SW &&
         (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0 &&
         (long)a.N * a.D * a.H < (1L << 31) && (long)a.D * a.H * a.W < (1L << 31);
}

static int adell_pool_fill(PoolArgs* a, const adell_conv3d_desc* d) {
  ADELL_REQUIRE(d != nullptr, "maxpool: null descriptor");
  ADELL_REQUIRE(d->N > 0 && d->C0 > 0 && d->D > 0 && d->H > 0 && d->W > 0, "maxpool: bad dims");
  ADELL_REQUIRE(d->KD >= 1 && d->KH >= 1 && d->KW >= 1 && d->SD >= 1 && d->SH >= 1 && d->SW >= 1,
                "maxpool: bad kernel/stride");
  ADELL_REQUIRE(2 * d->PD <= d->KD && 2 * d->PH <= d->KH && 2 * d->PW <= d->KW,
                "maxpool: padding must be at most half the kernel");
  ADELL_REQUIRE(d->Do == (d->D + 2 * d->PD - d->KD) / d->SD + 1 &&
                    d->Ho == (d->H + 2 * d->PH - d->KH) / d->SH + 1 &&
                    d->Wo == (d->W + 2 * d->PW - d->KW) / d->SW + 1,
                "maxpool: output dims do not match");
  a->N = d->N; a->C = d->C0; a->D = d->D; a->H = d->H; a->W = d->W;
  a->Do = d->Do; a->Ho = d->Ho; a->Wo = d->Wo;
  a->KD = d->KD; a->KH = d->KH; a->KW = d->KW; a->SD = d->SD; a->SH = d->SH; a->SW = d->SW;
  a->PD = d->PD; a->PH = d->PH; a->PW = d->PW;
  return ADELL_OK;
}

// The pooling geometry reuses adell_conv3d_desc (C0 = channels; C1, Cout ignored).
extern "C" int adell_maxpool3d_fwd(const adell_conv3d_desc* d, const float* x, float* y,
                                   int32_t* argmax, void* stream) {
  PoolArgs a = {};
  int rc = adell_pool_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && y && argmax, "maxpool_fwd: null pointer");
  a.x = x; a.y = y; a.idx = argmax;
  if (adell_pool_rows_ok(a, x, y, argmax)) {
    hipLaunchKernelGGL(adell_maxpool3d_fwd_rows_kernel, dim3((unsigned)(a.N * a.Do * a.Ho)), dim3(256),
                       0, (hipStream_t)stream, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  long blocks = ((long)a.N * a.Do * a.Ho * a.Wo * a.C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adell_maxpool3d_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_maxpool3d_bwd(const adell_conv3d_desc* d, const float* dy,
                                   const int32_t* argmax, float* dx, void* stream) {
  PoolArgs a = {};
  int rc = adell_pool_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(dy && argmax && dx, "maxpool_bwd: null pointer");
  a.dy = dy; a.idx = const_cast<int32_t*>(argmax); a.dx = dx;
  if (adell_pool_rows_ok(a, dy, argmax, dx)) {
    hipLaunchKernelGGL(adell_maxpool3d_bwd_rows_kernel, dim3((unsigned)(a.N * a.D * a.H)), dim3(256), 0,
                       (hipStream_t)stream, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  long blocks = ((long)a.N * a.D * a.H * a.W * a.C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adell_maxpool3d_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Linear upsampling (torch.nn.Upsample(scale_factor=s, mode="bilinear" | "trilinear"),
// align_corners=False) of the "upsample" upscaling path, unet.py:419-443, on NDHWC tensors.
// Source coordinate of output index o along an axis: max(0, (o + 0.5) / scale - 0.5); the two
// taps are floor(src) and min(floor(src) + 1, size - 1) with weights (1 - t, t). A 2-D tensor is
// the D = 1, scale_z = 1 case (t = 0 along z). Backward is a gather over the output positions
// that can touch an input voxel (deterministic, no atomics).
// ---------------------------------------------------------------------------
struct LinArgs {
  const float* x;   // fwd: input;  bwd: dY
  float* y;         // fwd: output; bwd: dX
  int N, C, Di, Hi, Wi, Do, Ho, Wo;
  float rz, ry, rx; // 1 / scale_factor per axis (align_corners: (in - 1) / (out - 1))
  int align;        // align_corners=True: src = o * r
};

__device__ __forceinline__ void adell_lin_taps(int o, float r, int size, int* i0, int* i1, float* t,
                                               int align = 0) {
  float src = align ? (float)o * r : ((float)o + 0.5f) * r - 0.5f;
  src = src < 0.f ? 0.f : src;
  const int f = (int)src;
  *i0 = f < size - 1 ? f : size - 1;
  *i1 = f + 1 < size ? f + 1 : size - 1;
  *t = src - (float)f;
}

__global__ __launch_bounds__(256) void adell_interp_linear_fwd_kernel(LinArgs a) {
  const long n = (long)a.N * a.Do * a.Ho * a.Wo * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int ox = (int)(v % a.Wo); v /= a.Wo;
    const int oy = (int)(v % a.Ho); v /= a.Ho;
    const int oz = (int)(v % a.Do);
    const int nb = (int)(v / a.Do);
    int z0, z1, y0, y1, x0, x1;
    float tz, ty, tx;
    adell_lin_taps(oz, a.rz, a.Di, &z0, &z1, &tz, a.align);
    adell_lin_taps(oy, a.ry, a.Hi, &y0, &y1, &ty, a.align);
    adell_lin_taps(ox, a.rx, a.Wi, &x0, &x1, &tx, a.align);
    const float* xb = a.x + (size_t)nb * a.Di * a.Hi * a.Wi * a.C + c;
    auto at = [&](int z, int y, int x) { return xb[((size_t)(z * a.Hi + y) * a.Wi + x) * a.C]; };
    const float c00 = at(z0, y0, x0) * (1.f - tx) + at(z0, y0, x1) * tx;
    const float c01 = at(z0, y1, x0) * (1.f - tx) + at(z0, y1, x1) * tx;
    const float c10 = at(z1, y0, x0) * (1.f - tx) + at(z1, y0, x1) * tx;
    const float c11 = at(z1, y1, x0) * (1.f - tx) + at(z1, y1, x1) * tx;
    const float c0 = c00 * (1.f - ty) + c01 * ty, c1 = c10 * (1.f - ty) + c11 * ty;
    a.y[i] = c0 * (1.f - tz) + c1 * tz;
  }
}

// weight with which input index `in` enters output index o along one axis
__device__ __forceinline__ float adell_lin_weight(int o, float r, int size, int in, int align) {
  int i0, i1;
  float t;
  adell_lin_taps(o, r, size, &i0, &i1, &t, align);
  return (i0 == in ? 1.f - t : 0.f) + (i1 == in ? t : 0.f);
}

__global__ __launch_bounds__(256) void adell_interp_linear_bwd_kernel(LinArgs a) {
  const long n = (long)a.N * a.Di * a.Hi * a.Wi * a.C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % a.C);
    long v = i / a.C;
    const int ix = (int)(v % a.Wi); v /= a.Wi;
    const int iy = (int)(v % a.Hi); v /= a.Hi;
    const int iz = (int)(v % a.Di);
    const int nb = (int)(v / a.Di);
    // output indices whose taps can reach this input index: src in (in - 1, in + 1)
    // (a conservative range: every candidate's weight is recomputed, most are zero)
    auto lo = [&](int in, float r) {
      if (r <= 0.f) return 0;
      int o = (int)floorf(((float)in - 1.0f) / r - 1.0f);
      return o < 0 ? 0 : o;
    };
    auto hi = [&](int in, float r, int osz) {
      if (r <= 0.f) return osz - 1;
      int o = (int)ceilf(((float)in + 1.5f) / r + 1.0f);
      return o > osz - 1 ? osz - 1 : o;
    };
    const int zl = lo(iz, a.rz), zh = hi(iz, a.rz, a.Do);
    const int yl = lo(iy, a.ry), yh = hi(iy, a.ry, a.Ho);
    const int xl = lo(ix, a.rx), xh = hi(ix, a.rx, a.Wo);
    const float* gb = a.x + (size_t)nb * a.Do * a.Ho * a.Wo * a.C + c;
    float acc = 0.f;
    for (int oz = zl; oz <= zh; ++oz) {
      const float wz = adell_lin_weight(oz, a.rz, a.Di, iz, a.align);
      if (wz == 0.f) continue;
      for (int oy = yl; oy <= yh; ++oy) {
        const float wy = adell_lin_weight(oy, a.ry, a.Hi, iy, a.align);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = xl; ox <= xh; ++ox) {
          const float wx = adell_lin_weight(ox, a.rx, a.Wi, ix, a.align);
          if (wx != 0.f) row += wx * gb[((size_t)(oz * a.Ho + oy) * a.Wo + ox) * a.C];
        }
        acc += wz * wy * row;
      }
    }
    a.y[i] = acc;
  }
}

static int adell_interp_linear(const float* x, float* y, int N, int C, int Di, int Hi, int Wi,
                               int Do, int Ho, int Wo, float sz, float sy, float sx, int align,
                               int bwd, hipStream_t st) {
  ADELL_REQUIRE(x && y, "interp_linear: null pointer");
  ADELL_REQUIRE(N > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0,
                "interp_linear: bad dims");
  ADELL_REQUIRE(sz > 0.f && sy > 0.f && sx > 0.f, "interp_linear: bad scale factors");
  LinArgs a = {x, y, N, C, Di, Hi, Wi, Do, Ho, Wo, 1.f / sz, 1.f / sy, 1.f / sx, align};
  if (align) {   // align_corners=True: the corner voxels map onto each other
    a.rz = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f;
    a.ry = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    a.rx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  }
  const long n = (long)N * C * (bwd ? (long)Di * Hi * Wi : (long)Do * Ho * Wo);
  long blocks = (n + 255) / 256;
  if (blocks > 65535 * 4) blocks = 65535 * 4;
  if (bwd)
    hipLaunchKernelGGL(adell_interp_linear_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(adell_interp_linear_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_interp_linear_fwd(const float* x, float* y, int N, int C, int Di, int Hi,
                                       int Wi, int Do, int Ho, int Wo, float scale_d,
                                       float scale_h, float scale_w, int align_corners,
                                       void* stream) {
  return adell_interp_linear(x, y, N, C, Di, Hi, Wi, Do, Ho, Wo, scale_d, scale_h, scale_w,
                             align_corners, 0, (hipStream_t)stream);
}
extern "C" int adell_interp_linear_bwd(const float* dy, float* dx, int N, int C, int Di, int Hi,
                                       int Wi, int Do, int Ho, int Wo, float scale_d,
                                       float scale_h, float scale_w, int align_corners,
                                       void* stream) {
  return adell_interp_linear(dy, dx, N, C, Di, Hi, Wi, Do, Ho, Wo, scale_d, scale_h, scale_w,
                             align_corners, 1, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// Per-(item, channel) scaling of an NDHWC activation: y[n][v][c] = x[n][v][c] * s[n][c].
// Two call sites of the reference: the tabular feature gates of the U-Net decoder
// (torch.multiply(encoded, transformed_features), unet.py:803-810) and U-out
// (X + X * r, r ~ U(-beta, beta) per item and channel, regularization.py:48-55).
// Backward: dx = dy * s (this kernel again), ds[n][c] = sum_v dy * x (partials + fixed-order fold).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_scale_bc_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ s,
                                                             float* __restrict__ y, long VC, int C) {
  const int nb = blockIdx.y;
  const float* xb = x + (size_t)nb * VC;
  float* yb = y + (size_t)nb * VC;
  const float* sb = s + (size_t)nb * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < VC; i += (long)gridDim.x * 256L)
    yb[i] = xb[i] * sb[i % C];
}

// partial[n][tile][c] = sum over the tile's voxels of dy * x
__global__ __launch_bounds__(256) void adell_scale_bc_dscale_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, long V,
    int C, int vpt) {
  const int nb = blockIdx.y, tile = blockIdx.x;
  const long v0 = (long)tile * vpt;
  const long v1 = (v0 + vpt) < V ? (v0 + vpt) : V;
  const size_t base = (size_t)nb * V * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (long v = v0; v < v1; ++v) acc += dy[base + v * C + c] * x[base + v * C + c];
    part[((size_t)nb * gridDim.x + tile) * C + c] = acc;
  }
}

// block = 64 channels x 16 lanes, each lane a fixed share of the tiles (fp64, fixed order)
__global__ __launch_bounds__(1024) void adell_scale_bc_fold_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ ds, int tiles,
                                                                   int C) {
  __shared__ double sh[16][64];
  const int nb = blockIdx.y;
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double acc = 0.0;
  if (c < C)
    for (int t = vl; t < tiles; t += 16) acc += (double)part[((size_t)nb * tiles + t) * C + c];
  sh[vl][cl] = acc;
  __syncthreads();
  if (vl != 0 || c >= C) return;
  acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += sh[k][cl];
  ds[(size_t)nb * C + c] = (float)acc;
}

extern "C" int adell_scale_bc(const float* x, const float* s, float* y, int N, long V, int C,
                              void* stream) {
  ADELL_REQUIRE(x && s && y && N > 0 && V > 0 && C > 0 && N <= 65535, "scale_bc: bad arguments");
  long blocks = (V * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_scale_bc_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0,
                     (hipStream_t)stream, x, s, y, V * C, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_scale_bc_tiles(long V, int C) {
  long t = V / 64;           // >= 64 voxels per tile
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  (void)C;
  return (int)t;
}

extern "C" long adell_scale_bc_dscale_workspace_floats(int N, long V, int C) {
  if (N <= 0 || V <= 0 || C <= 0) return ADELL_E_BADARG;
  return (long)N * adell_scale_bc_tiles(V, C) * C;
}

extern "C" int adell_scale_bc_dscale(const float* x, const float* dy, float* ds, int N, long V,
                                     int C, float* workspace, void* stream) {
  ADELL_REQUIRE(x && dy && ds && workspace && N > 0 && V > 0 && C > 0 && N <= 65535,
                "scale_bc_dscale: bad arguments");
  const int tiles = adell_scale_bc_tiles(V, C);
  const int vpt = (int)((V + tiles - 1) / tiles);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_scale_bc_dscale_kernel, dim3((unsigned)tiles, (unsigned)N), dim3(256), 0,
                     st, x, dy, workspace, V, C, vpt);
  hipLaunchKernelGGL(adell_scale_bc_fold_kernel, dim3((unsigned)adell_cdiv(C, 64), (unsigned)N),
                     dim3(1024), 0, st, (const float*)workspace, ds, tiles, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Concurrent squeeze-and-excite gate of the multi-branch U-Net (self_attention.py:21-150,
// unet.py:1186-1207): y = acc + x * (s[n][v] + c[n][ch]) * inv[n] on NDHWC tensors. s is the
// spatial gate (sigmoid of a C -> 1 pointwise conv), c the channel gate (sigmoid of the MLP of the
// per-channel means), inv[n] = 1 / (sum of the branch weights of item n), acc the running sum over
// the branches merged so far (or null). One pass instead of two scales, an add and a division.
// Backward: dx = dy * (s + c) * inv, ds[n][v] = inv * sum_ch dy * x, dc[n][ch] = inv * sum_v dy * x
// (per-tile partials + the fixed-order fold of adell_scale_bc); d acc = dy.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_cse_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ c,
    const float* __restrict__ inv, const float* __restrict__ acc, float* __restrict__ y, long V,
    int C) {
  const int nb = blockIdx.y;
  const size_t base = (size_t)nb * V * C;
  const float* sb = s + (size_t)nb * V;
  const float* cb = c + (size_t)nb * C;
  const float iv = inv ? inv[nb] : 1.f;
  const long VC = V * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < VC; i += (long)gridDim.x * 256L) {
    const long v = i / C;
    const int ch = (int)(i - v * C);
    const float g = (sb[v] + cb[ch]) * iv;
    y[base + i] = (acc ? acc[base + i] : 0.f) + x[base + i] * g;
  }
}

// block = 4 waves; a wave walks voxels of the tile, LPV lanes per voxel (64 / LPV voxels at a
// time), lane l of a voxel owns channels l, l + LPV, ... (C <= 8 * LPV)
__global__ __launch_bounds__(256) void adell_cse_apply_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ s,
    const float* __restrict__ c, const float* __restrict__ inv, float* __restrict__ dx,
    float* __restrict__ ds, float* __restrict__ part, long V, int C, int vpt, int lpv) {
  __shared__ float sred[256][8];
  const int nb = blockIdx.y, tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = lane & (lpv - 1), sub = lane / lpv, vpw = 64 / lpv;
  const long v0 = (long)tile * vpt;
  const long v1 = (v0 + vpt) < V ? (v0 + vpt) : V;
  const size_t base = (size_t)nb * V * C;
  const float iv = inv ? inv[nb] : 1.f;
  float cg[8], col[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = cl + lpv * k;
    cg[k] = ch < C ? c[(size_t)nb * C + ch] : 0.f;
    col[k] = 0.f;
  }
  for (long vb = v0 + (long)wave * vpw; vb < v1; vb += 4L * vpw) {
    const long v = vb + sub;
    const bool live = v < v1;
    const float sv = live ? s[(size_t)nb * V + v] : 0.f;
    float row = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ch = cl + lpv * k;
      if (live && ch < C) {
        const size_t i = base + (size_t)v * C + ch;
        const float g = dy[i], xv = x[i];
        dx[i] = g * (sv + cg[k]) * iv;
        const float p = g * xv;
        col[k] += p;
        row += p;
      }
    }
    for (int o = lpv >> 1; o > 0; o >>= 1) row += __shfl_xor(row, o, 64);
    if (live && cl == 0) ds[(size_t)nb * V + v] = row * iv;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sred[tid][k] = col[k];
  __syncthreads();
  // column sums over the 4 waves x (64 / lpv) voxel slots that share a channel (fixed order)
  for (int ch = tid; ch < C; ch += 256) {
    const int l = ch & (lpv - 1), k = ch / lpv;
    float acc = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int sb = 0; sb < vpw; ++sb) acc += sred[w * 64 + sb * lpv + l][k];
    part[((size_t)nb * gridDim.x + tile) * C + ch] = acc * iv;
  }
}

extern "C" int adell_cse_apply(const float* x, const float* s, const float* c, const float* inv,
                               const float* acc, float* y, int N, long V, int C, void* stream) {
  ADELL_REQUIRE(x && s && c && y && N > 0 && V > 0 && C > 0 && N <= 65535,
                "cse_apply: bad arguments");
  long blocks = (V * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_cse_apply_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0,
                     (hipStream_t)stream, x, s, c, inv, acc, y, V, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" long adell_cse_apply_bwd_workspace_floats(int N, long V, int C) {
  return adell_scale_bc_dscale_workspace_floats(N, V, C);
}

extern "C" int adell_cse_apply_bwd(const float* x, const float* dy, const float* s, const float* c,
                                   const float* inv, float* dx, float* ds, float* dc, int N, long V,
                                   int C, float* workspace, void* stream) {
  ADELL_REQUIRE(x && dy && s && c && dx && ds && dc && workspace && N > 0 && V > 0 && C > 0 &&
                    N <= 65535,
                "cse_apply_bwd: bad arguments");
  ADELL_REQUIRE(C <= 512, "cse_apply_bwd: at most 512 channels");
  int lpv = 16;
  while (lpv < 64 && lpv < C) lpv <<= 1;
  const int tiles = adell_scale_bc_tiles(V, C);
  const int vpt = (int)((V + tiles - 1) / tiles);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_cse_apply_bwd_kernel, dim3((unsigned)tiles, (unsigned)N), dim3(256), 0,
                     st, x, dy, s, c, inv, dx, ds, workspace, V, C, vpt, lpv);
  hipLaunchKernelGGL(adell_scale_bc_fold_kernel, dim3((unsigned)adell_cdiv(C, 64), (unsigned)N),
                     dim3(1024), 0, st, (const float*)workspace, dc, tiles, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// out[n][v][ch] = g[n][ch] * scale (the gradient of a per-channel spatial mean, scale = 1 / V)
__global__ __launch_bounds__(256) void adell_bcast_nc_kernel(const float* __restrict__ g,
                                                             float* __restrict__ out, long VC,
                                                             int C, float scale) {
  const int nb = blockIdx.y;
  const float* gb = g + (size_t)nb * C;
  float* ob = out + (size_t)nb * VC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < VC; i += (long)gridDim.x * 256L)
    ob[i] = gb[i % C] * scale;
}

extern "C" int adell_bcast_nc(const float* g, float* out, int N, long V, int C, float scale,
                              void* stream) {
  ADELL_REQUIRE(g && out && N > 0 && V > 0 && C > 0 && N <= 65535, "bcast_nc: bad arguments");
  long blocks = (V * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_bcast_nc_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0,
                     (hipStream_t)stream, g, out, V * C, C, scale);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Fold the x taps of a small-Cin convolution into channels: out[n][z][y][ox][kx * Cin + ci] =
// x[n][z][y][ox - P + kx][ci] (zero outside the row and in the slots beyond K * Cin). A
// K_d x K_h x K conv over Cin <= 4 channels is then a K_d x K_h x 1 conv over Cp = 16 channels:
// one 16-channel MFMA chunk carries K * Cin <= 16 useful values instead of Cin (the 7^3 stem of the
// ResNet backbones on 2 channels: 343 k-steps of 2/16 -> 49 of 14/16).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_fold_x_taps_kernel(const float* __restrict__ x,
                                                                float* __restrict__ out, long rows,
                                                                int W, int Wo, int Cin, int K,
                                                                int P, int Cp) {
  // rows = N * D * H input rows; one thread per (row, ox, quad of output slots)
  const int cq = Cp >> 2;
  const long total = rows * Wo * cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int q = (int)(i % cq);
    const long v = i / cq;
    const int ox = (int)(v % Wo);
    const long row = v / Wo;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int slot = 4 * q + j;
      const int kx = slot / Cin, ci = slot - kx * Cin;
      const int ix = ox - P + kx;
      o[j] = (kx < K && ix >= 0 && ix < W) ? x[(row * W + ix) * Cin + ci] : 0.f;
    }
    *reinterpret_cast<float4*>(out + (v * cq + q) * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int adell_fold_x_taps(const float* x, float* out, int N, int D, int H, int W, int Cin,
                                 int K, int P, int Cp, void* stream) {
  ADELL_REQUIRE(x && out, "fold_x_taps: null pointer");
  ADELL_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin >= 1 && K >= 1 && P >= 0 && Cp % 4 == 0 &&
                    K * Cin <= Cp && W + 2 * P - K + 1 > 0,
                "fold_x_taps: bad arguments");
  const int Wo = W + 2 * P - K + 1;
  const long rows = (long)N * D * H;
  long blocks = (rows * Wo * (Cp / 4) + 255) / 256;
  if (blocks > 65535 * 8) blocks = 65535 * 8;
  hipLaunchKernelGGL(adell_fold_x_taps_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, out, rows, W, Wo, Cin, K, P, Cp);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------------------------
// Softmax over the channel axis of an NDHWC tensor: rows = N * voxels, C contiguous values per row
// (the n_classes > 2 head of the U-Net family, torch.nn.Softmax(dim=1), unet.py:641-655).
// One thread per row (C <= 32: the class count), row in registers, exp via v_exp_f32.
// backward: dx = y * (dy - sum_c dy * y).
// ---------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void adell_channel_softmax_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, long rows, int C) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256L) {
    const float* xr = x + r * C;
    float v[CMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = c < C ? xr[c] : -INFINITY;
      mx = fmaxf(mx, v[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = c < C ? __expf(v[c] - mx) : 0.f;
      sum += v[c];
    }
    const float inv = 1.0f / sum;
    float* yr = y + r * C;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) yr[c] = v[c] * inv;
  }
}

template <int CMAX>
__global__ __launch_bounds__(256) void adell_channel_softmax_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long rows,
    int C) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256L) {
    float p[CMAX], g[CMAX];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      p[c] = c < C ? y[r * C + c] : 0.f;
      g[c] = c < C ? dy[r * C + c] : 0.f;
      dot += p[c] * g[c];
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) dx[r * C + c] = p[c] * (g[c] - dot);
  }
}

static unsigned adell_rows_grid(long rows) {
  long b = (rows + 255) / 256;
  return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

extern "C" int adell_channel_softmax_fwd(const float* x, float* y, long rows, int C, void* stream) {
  ADELL_REQUIRE(x && y && rows > 0 && C > 0 && C <= 32, "channel_softmax: 1 <= C <= 32 classes");
  const dim3 g(adell_rows_grid(rows)), b(256);
  if (C <= 4)
    hipLaunchKernelGGL(adell_channel_softmax_fwd_kernel<4>, g, b, 0, (hipStream_t)stream, x, y, rows, C);
  else if (C <= 8)
    hipLaunchKernelGGL(adell_channel_softmax_fwd_kernel<8>, g, b, 0, (hipStream_t)stream, x, y, rows, C);
  else
    hipLaunchKernelGGL(adell_channel_softmax_fwd_kernel<32>, g, b, 0, (hipStream_t)stream, x, y, rows, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_channel_softmax_bwd(const float* y, const float* dy, float* dx, long rows,
                                         int C, void* stream) {
  ADELL_REQUIRE(y && dy && dx && rows > 0 && C > 0 && C <= 32,
                "channel_softmax_bwd: 1 <= C <= 32 classes");
  const dim3 g(adell_rows_grid(rows)), b(256);
  if (C <= 4)
    hipLaunchKernelGGL(adell_channel_softmax_bwd_kernel<4>, g, b, 0, (hipStream_t)stream, y, dy, dx, rows, C);
  else if (C <= 8)
    hipLaunchKernelGGL(adell_channel_softmax_bwd_kernel<8>, g, b, 0, (hipStream_t)stream, y, dy, dx, rows, C);
  else
    hipLaunchKernelGGL(adell_channel_softmax_bwd_kernel<32>, g, b, 0, (hipStream_t)stream, y, dy, dx, rows, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------------------------
// Per-(item, channel) maximum over the voxels of an NDHWC tensor with its argmax (first maximum
// in voxel order, as torch.max): the pooling in front of the bottleneck classifier
// (X.flatten(2).max(-1).values, unet.py:826-828). One block per item, thread t owns channels
// t, t + 256, ...; consecutive threads read consecutive channels of one voxel (coalesced).
// backward: dx = 0 except dx[n, arg[n, c], c] = dout[n, c].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_channel_max_kernel(const float* __restrict__ x,
                                                                float* __restrict__ out,
                                                                int* __restrict__ arg, long V, int C) {
  const float* xb = x + (size_t)blockIdx.x * V * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float best = xb[c];
    long at = 0;
    for (long v = 1; v < V; ++v) {
      const float t = xb[v * C + c];
      // NaN propagates like torch.max: the first NaN wins
      if (t > best || (t != t && best == best)) {
        best = t;
        at = v;
      }
    }
    out[(size_t)blockIdx.x * C + c] = best;
    arg[(size_t)blockIdx.x * C + c] = (int)at;
  }
}

__global__ __launch_bounds__(256) void adell_channel_max_bwd_kernel(const float* __restrict__ dout,
                                                                    const int* __restrict__ arg,
                                                                    float* __restrict__ dx, long V,
                                                                    int C) {
  const size_t nb = blockIdx.y;
  float* db = dx + nb * V * C;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < V * C; i += (long)gridDim.x * 256L) {
    const int c = (int)(i % C);
    const long v = i / C;
    db[i] = arg[nb * C + c] == v ? dout[nb * C + c] : 0.f;
  }
}

extern "C" int adell_channel_max_fwd(const float* x, float* out, int* arg, int N, long V, int C,
                                     void* stream) {
  ADELL_REQUIRE(x && out && arg && N > 0 && V > 0 && V < (1L << 31) && C > 0,
                "channel_max: bad arguments");
  hipLaunchKernelGGL(adell_channel_max_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream,
                     x, out, arg, V, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_channel_max_bwd(const float* dout, const int* arg, float* dx, int N, long V,
                                     int C, void* stream) {
  ADELL_REQUIRE(dout && arg && dx && N > 0 && N <= 65535 && V > 0 && C > 0,
                "channel_max_bwd: bad arguments");
  long blocks = (V * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_channel_max_bwd_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256),
                     0, (hipStream_t)stream, dout, arg, dx, V, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Layer scale folded into a Linear layer (ConvNeXtBlock3d, res_blocks.py:588-604: gamma * pwconv2(h)
// = h (gamma W)^T + gamma b): W2[c][k] = gamma[c] W[c][k], b2[c] = gamma[c] b[c] and the backward of
// that parameter algebra -- one launch each way instead of ~10 element-wise / reduction launches of
// a tensor library per block (15 blocks per ConvNeXt step). One block per output row c.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_rowscale_fwd_kernel(const float* __restrict__ gamma,
                                                                 const float* __restrict__ W,
                                                                 const float* __restrict__ b,
                                                                 float* __restrict__ W2,
                                                                 float* __restrict__ b2, int K) {
  const int c = blockIdx.x;
  const float g = gamma[c];
  for (int k = threadIdx.x; k < K; k += 256) W2[(size_t)c * K + k] = g * W[(size_t)c * K + k];
  if (threadIdx.x == 0 && b != nullptr) b2[c] = g * b[c];
}

__global__ __launch_bounds__(256) void adell_rowscale_bwd_kernel(
    const float* __restrict__ gamma, const float* __restrict__ W, const float* __restrict__ b,
    const float* __restrict__ dW2, const float* __restrict__ db2, float* __restrict__ dgamma,
    float* __restrict__ dW, float* __restrict__ db, int K) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  const float g = gamma[c];
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {      // fixed order per thread, fixed tree below
    const float d = dW2[(size_t)c * K + k];
    s = fmaf(d, W[(size_t)c * K + k], s);
    dW[(size_t)c * K + k] = g * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (b != nullptr) {
      t = fmaf(db2[c], b[c], t);
      db[c] = g * db2[c];
    }
    dgamma[c] = t;
  }
}

extern "C" int adell_rowscale_fwd(const float* gamma, const float* W, const float* b, float* W2,
                                  float* b2, int C, int K, void* stream) {
  ADELL_REQUIRE(gamma && W && W2 && C > 0 && K > 0 && (b == nullptr || b2 != nullptr),
                "rowscale_fwd: bad arguments");
  hipLaunchKernelGGL(adell_rowscale_fwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     gamma, W, b, W2, b2, K);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_rowscale_bwd(const float* gamma, const float* W, const float* b, const float* dW2,
                                  const float* db2, float* dgamma, float* dW, float* db, int C, int K,
                                  void* stream) {
  ADELL_REQUIRE(gamma && W && dW2 && dgamma && dW && C > 0 && K > 0 &&
                    (b == nullptr || (db2 != nullptr && db != nullptr)),
                "rowscale_bwd: bad arguments");
  hipLaunchKernelGGL(adell_rowscale_bwd_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     gamma, W, b, dW2, db2, dgamma, dW, db, K);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Spatial window of a channels-last volume, either way: out[n][d][h][w][:] = in[n][d + od][h + oh]
// [w + ow][:] where that voxel exists, zeros elsewhere. Positive offsets with a smaller output are
// crop_to_size (layers/utils.py:30-52: the decoder of a backbone U-Net crops the 130^3 output of a
// transposed conv to the 128^3 skip, unet.py:813-816); negative offsets with a larger output are its
// gradient (a zero frame around dY). The tensor library ran the pair as a strided copy, three
// zero-fills, three strided copies into slices of an NCDHW buffer and a transposing copy back
// (4.8 ms per step of config 2b). One block per output row (n, d, h): W * C contiguous floats.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void adell_window_ndhwc_kernel(
    const float* __restrict__ in, float* __restrict__ out, int C, int Di, int Hi, int Wi, int Do,
    int Ho, int Wo, int od, int oh, int ow) {
  const int row = blockIdx.x;                       // (n * Do + d) * Ho + h
  const int h = row % Ho, nd = row / Ho;
  const int d = nd % Do, n = nd / Do;
  const int sd = d + od, sh = h + oh;
  const bool rowok = sd >= 0 && sd < Di && sh >= 0 && sh < Hi;
  const float* src = in + (((size_t)n * Di + (rowok ? sd : 0)) * Hi + (rowok ? sh : 0)) * Wi * C;
  float* dst = out + (size_t)row * Wo * C;
  const int per = C / VEC, items = Wo * per;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int w = it / per, c = (it - w * per) * VEC;
    const int sw = w + ow;
    const bool ok = rowok && sw >= 0 && sw < Wi;
    if constexpr (VEC == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(src + (size_t)sw * C + c);
      *reinterpret_cast<f32x4*>(dst + (size_t)w * C + c) = v;
    } else {
      dst[(size_t)w * C + c] = ok ? src[(size_t)sw * C + c] : 0.f;
    }
  }
}

extern "C" int adell_window_ndhwc(const float* in, float* out, int N, int C, int Di, int Hi, int Wi,
                                  int Do, int Ho, int Wo, int od, int oh, int ow, void* stream) {
  ADELL_REQUIRE(in && out && N > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0,
                "window_ndhwc: bad arguments");
  const long rows = (long)N * Do * Ho;
  ADELL_REQUIRE(rows < (1L << 31) && (long)Wo * C < (1L << 31), "window_ndhwc: too many rows");
  const bool vec = C % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (vec)
    hipLaunchKernelGGL(adell_window_ndhwc_kernel<4>, dim3((unsigned)rows), dim3(256), 0,
                       (hipStream_t)stream, in, out, C, Di, Hi, Wi, Do, Ho, Wo, od, oh, ow);
  else
    hipLaunchKernelGGL(adell_window_ndhwc_kernel<1>, dim3((unsigned)rows), dim3(256), 0,
                       (hipStream_t)stream, in, out, C, Di, Hi, Wi, Do, Ho, Wo, od, oh, ow);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
