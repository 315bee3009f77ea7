// Normalisation statistics + fused Norm -> Dropout -> Activation ("NDA", the
// ordering UNet.adn_fn uses: reference adell_mri/modules/segmentation/unet.py:697-714,
// adell_mri/modules/layers/adn_fn.py:140-152). HBM-bound elementwise work on
// NDHWC fp32; reductions use wavefront shuffles and fixed-order fp64 combines.
#include "common.h"

ADELL_RNG_STEP_DEFINE(norm_act)

// ---------------------------------------------------------------------------
// Per-(n,c) statistics from per-block partials [N][ntiles][C][2] (sum, sumsq)
// written by the conv epilogue or by adell_channel_partials_kernel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_stats_finalize_kernel(
    const float* __restrict__ part, int N, int ntiles, int C, double count, float eps,
    int per_item, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ double sh[8][32][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int nbeg = per_item ? blockIdx.y : 0, nend = per_item ? blockIdx.y + 1 : N;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int n = nbeg; n < nend; ++n) {
      // eight rows in flight, added in row order (the launch is latency-bound: 2-16 blocks)
      const float* p = part + ((size_t)n * ntiles * C + c) * 2;
      int t = sl;
      for (; t + 56 < ntiles; t += 64) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = *reinterpret_cast<const float2*>(p + (size_t)(t + 8 * u) * C * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s1 += (double)v[u].x;
          s2 += (double)v[u].y;
        }
      }
      for (; t < ntiles; t += 8) {
        const float2 v = *reinterpret_cast<const float2*>(p + (size_t)t * C * 2);
        s1 += (double)v.x;
        s2 += (double)v.y;
      }
    }
  }
  sh[sl][cl][0] = s1;
  sh[sl][cl][1] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a += sh[k][cl][0];
      b += sh[k][cl][1];
    }
    const double cnt = per_item ? count : count * N;
    const double m = a / cnt;
    double var = b / cnt - m * m;
    if (var < 0.0) var = 0.0;
    const size_t o = per_item ? (size_t)blockIdx.y * C + c : (size_t)c;
    mean[o] = (float)m;
    rstd[o] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// First level of the two-level reduction used when there are many tiles: block
// (cgroup, n, z) folds tiles [256 z, 256 z + 256) into one row of `out`
// ([N][Z][C][2] floats, summed in fp64).
// (pstride / poff: the channels may be columns [poff, poff + C) of rows pstride wide)
__global__ __launch_bounds__(256) void adell_stats_fold_kernel(const float* __restrict__ part,
                                                               int ntiles, int C, int Z,
                                                               float* __restrict__ out,
                                                               int pstride, int poff) {
  __shared__ double sh[8][32][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, n = blockIdx.y, z = blockIdx.z;
  const int t0 = z * 256;
  const int t1 = t0 + 256 < ntiles ? t0 + 256 : ntiles;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // eight rows in flight, added in row order
    const float* p = part + ((size_t)n * ntiles * pstride + poff + c) * 2;
    int t = t0 + sl;
    for (; t + 56 < t1; t += 64) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float2*>(p + (size_t)(t + 8 * u) * pstride * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s1 += (double)v[u].x;
        s2 += (double)v[u].y;
      }
    }
    for (; t < t1; t += 8) {
      const float2 v = *reinterpret_cast<const float2*>(p + (size_t)t * pstride * 2);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  }
  sh[sl][cl][0] = s1;
  sh[sl][cl][1] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a += sh[k][cl][0];
      b += sh[k][cl][1];
    }
    float* o = out + (((size_t)n * Z + z) * C + c) * 2;
    o[0] = (float)a;
    o[1] = (float)b;
  }
}

extern "C" long adell_stats_finalize_workspace(int N, int ntiles, int C) {
  if (ntiles <= 512) return 0;
  const long Z = (ntiles + 255) / 256;
  return (long)sizeof(float) * N * Z * C * 2;
}

extern "C" int adell_stats_finalize(const float* partials, int N, int ntiles, int C,
                                    long count, float eps, int per_item, float* mean,
                                    float* rstd, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  ADELL_REQUIRE(partials && mean && rstd, "stats_finalize: null pointer");
  ADELL_REQUIRE(N > 0 && ntiles > 0 && C > 0 && count > 0, "stats_finalize: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const long need = adell_stats_finalize_workspace(N, ntiles, C);
  if (need > 0) {
    ADELL_REQUIRE(workspace && (long)workspace_bytes >= need,
                  "stats_finalize: workspace too small");
    const int Z = (ntiles + 255) / 256;
    hipLaunchKernelGGL(adell_stats_fold_kernel, dim3(adell_cdiv(C, 32), N, Z), dim3(256), 0, st,
                       partials, ntiles, C, Z, (float*)workspace, C, 0);
    partials = (const float*)workspace;
    ntiles = Z;
  }
  hipLaunchKernelGGL(adell_stats_finalize_kernel, dim3(adell_cdiv(C, 32), per_item ? N : 1),
                     dim3(256), 0, st, partials, N, ntiles, C, (double)count, eps, per_item, mean,
                     rstd);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Running statistics of a BatchNorm site in training (torch.nn.BatchNorm3d semantics: unbiased
// variance, num_batches_tracked += 1, momentum or the cumulative average when momentum is None),
// from the batch (mean, rstd) adell_stats_finalize wrote: one launch where the tensor library ran
// nine element-wise launches per site (26 sites in the ResNet-backbone U-Net: ~1 ms per step).
__global__ __launch_bounds__(256) void adell_bn_running_update_kernel(
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ rmean,
    float* __restrict__ rvar, long long* __restrict__ nbt, int C, float unbias, float eps,
    float momentum) {
  // (one block: the counter is read by every thread before thread 0 writes it back)
  long long t = nbt != nullptr ? nbt[0] + 1 : 0;
  const float mom = momentum >= 0.f ? momentum : 1.f / (float)t;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float r = rstd[c];
    const float var = (1.f / (r * r) - eps) * unbias;
    rmean[c] = rmean[c] * (1.f - mom) + mom * mean[c];
    rvar[c] = rvar[c] * (1.f - mom) + mom * var;
  }
  if (threadIdx.x == 0 && nbt != nullptr) nbt[0] = t;
}

extern "C" int adell_bn_running_update(const float* mean, const float* rstd, float* running_mean,
                                       float* running_var, long long* num_batches_tracked, int C,
                                       long count, float eps, float momentum, void* stream) {
  ADELL_REQUIRE(mean && rstd && running_mean && running_var && C > 0 && count > 0,
                "bn_running_update: bad arguments");
  ADELL_REQUIRE(momentum >= 0.f || num_batches_tracked != nullptr,
                "bn_running_update: the cumulative average needs num_batches_tracked");
  const float unbias = (float)((double)count / (double)(count > 1 ? count - 1 : 1));
  hipLaunchKernelGGL(adell_bn_running_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean,
                     rstd, running_mean, running_var, num_batches_tracked, C, unbias, eps, momentum);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Partials of a tensor that did not come out of the conv epilogue. Each block
// reduces a slab of ADELL_STATS_SLAB voxels; thread (vl, c) walks voxels
// vl, vl+VL, ... of the slab.
#define ADELL_STATS_SLAB 1024
__global__ __launch_bounds__(256) void adell_channel_partials_kernel(
    const float* __restrict__ x, long V, int C, float* __restrict__ part,
    int ntiles) {
  __shared__ float sh[256][2];
  const int n = blockIdx.y, tile = blockIdx.x;
  const long v0 = (long)tile * ADELL_STATS_SLAB;
  long v1 = v0 + ADELL_STATS_SLAB;
  if (v1 > V) v1 = V;
  const int CG = C < 256 ? C : 256;
  const int VL = 256 / CG;
  const int cl = threadIdx.x % CG, vl = threadIdx.x / CG;
  for (int cb = 0; cb < C; cb += CG) {
    const int c = cb + cl;
    float s1 = 0.f, s2 = 0.f;
    if (vl < VL && c < C) {
      const float* p = x + ((size_t)n * V) * C + c;
      for (long v = v0 + vl; v < v1; v += VL) {
        const float t = p[v * C];
        s1 += t;
        s2 += t * t;
      }
    }
    sh[threadIdx.x][0] = s1;
    sh[threadIdx.x][1] = s2;
    __syncthreads();
    if (vl == 0 && c < C) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < VL; ++k) {
        a += sh[k * CG + cl][0];
        b += sh[k * CG + cl][1];
      }
      float* o = part + (((size_t)n * ntiles + tile) * C + c) * 2;
      o[0] = a;
      o[1] = b;
    }
    __syncthreads();
  }
}

// The same with 16-byte loads (C % 4 == 0): thread = (channel quad, voxel lane), four voxels in
// flight per thread. The scalar form above walked a slab with one dependent 4-byte load per
// iteration: 226 MB (16 channels at 4 x 96^3) took 532 us.
__global__ __launch_bounds__(256) void adell_channel_partials_vec_kernel(
    const float* __restrict__ x, long V, int C, float* __restrict__ part, int ntiles) {
  __shared__ f32x4 sh[256][2];
  const int n = blockIdx.y, tile = blockIdx.x;
  const long v0 = (long)tile * ADELL_STATS_SLAB;
  long v1 = v0 + ADELL_STATS_SLAB;
  if (v1 > V) v1 = V;
  const int CQ = C >> 2;
  const int QG = CQ < 256 ? CQ : 256;          // quads handled per pass
  const int VL = 256 / QG;
  const int ql = threadIdx.x % QG, vl = threadIdx.x / QG;
  for (int qb = 0; qb < CQ; qb += QG) {
    const int q = qb + ql;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (vl < VL && q < CQ) {
      const f32x4* p = reinterpret_cast<const f32x4*>(x + ((size_t)n * V) * C) + q;
      long v = v0 + vl;
      for (; v + 3L * VL < v1; v += 4L * VL) {
        const f32x4 a = p[v * CQ], b = p[(v + VL) * CQ], c = p[(v + 2L * VL) * CQ],
                    d = p[(v + 3L * VL) * CQ];
        s1 += (a + b) + (c + d);
        s2 += (a * a + b * b) + (c * c + d * d);
      }
      for (; v < v1; v += VL) {
        const f32x4 a = p[v * CQ];
        s1 += a;
        s2 += a * a;
      }
    }
    sh[threadIdx.x][0] = s1;
    sh[threadIdx.x][1] = s2;
    __syncthreads();
    if (vl == 0 && q < CQ) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < VL; ++k) {
        a += sh[k * QG + ql][0];
        b += sh[k * QG + ql][1];
      }
      float* o = part + (((size_t)n * ntiles + tile) * C + 4 * q) * 2;
      o[0] = a.x; o[1] = b.x; o[2] = a.y; o[3] = b.y; o[4] = a.z; o[5] = b.z; o[6] = a.w; o[7] = b.w;
    }
    __syncthreads();
  }
}

extern "C" int adell_channel_partials_ntiles(long V) {
  return (int)((V + ADELL_STATS_SLAB - 1) / ADELL_STATS_SLAB);
}

extern "C" int adell_channel_partials(const float* x, int N, long V, int C,
                                      float* partials, void* stream) {
  ADELL_REQUIRE(x && partials, "channel_partials: null pointer");
  ADELL_REQUIRE(N > 0 && V > 0 && C > 0, "channel_partials: bad dims");
  const int nt = adell_channel_partials_ntiles(V);
  if ((C & 3) == 0 && (((uintptr_t)x) & 15) == 0)
    hipLaunchKernelGGL(adell_channel_partials_vec_kernel, dim3(nt, N), dim3(256), 0,
                       (hipStream_t)stream, x, V, C, partials, nt);
  else
    hipLaunchKernelGGL(adell_channel_partials_kernel, dim3(nt, N), dim3(256), 0,
                       (hipStream_t)stream, x, V, C, partials, nt);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// (activations: adell_act_fwd / adell_act_grad in common.h)

struct NormActArgs {
  const float* x;
  const float* mean;   // [N*C] (instance) or [C] (batch) or null (no norm)
  const float* rstd;
  const float* gamma;  // [C] or null
  const float* beta;   // [C] or null
  const float* act_w;  // PReLU weight: [1] or [C]; null otherwise
  float* out;
  long VC;             // voxels-per-item * C
  long total;          // N * VC
  int C;
  int stat_stride_n;   // C for instance statistics, 0 for batch statistics
  int act;
  int act_w_n;         // number of PReLU weights (1 or C)
  float act_p;         // leaky slope / elu alpha
  float drop_p;        // 0 disables dropout
  uint32_t seed_lo, seed_hi;
  uint32_t rng_offset;
  int vec;             // C % 4 == 0: float4 path covers everything
  unsigned long long* mask;  // fast kernel: keep bits of the dropout (adell_norm_act_fwd_mask) or null
  long groups;               // 256-element groups per batch item in `mask`
  int rev;                   // experiment: reversed block order (adell_ew_block)
  // fast kernel, split-row output (adell_norm_act_fwd_split): `out` receives, per voxel and
  // 16-channel chunk, the 64-byte row [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] of fp16 values of
  // out * 2^split_exp -- the LDS row image of the f16x3 convolution kernels (conv_igemm_f16.h),
  // same bytes per element as fp32 -- instead of fp32 values
  int split;
  float split_scale;
};

// hat = (x-mean)*rstd*gamma+beta ; u = dropout(hat) ; out = act(u)
__global__ __launch_bounds__(256) void adell_norm_act_fwd_kernel(NormActArgs a) {
  const long n4 = a.vec ? (a.total >> 2) : 0;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const long e = i << 2;
    const int c = (int)(e % a.C);
    const long nidx = e / a.VC;
    float4 v = reinterpret_cast<const float4*>(a.x)[i];
    float h[4] = {v.x, v.y, v.z, v.w};
    uint4 r = make_uint4(0, 0, 0, 0);
    if (a.drop_p > 0.f)
      r = adell_philox4((uint32_t)i, (uint32_t)(i >> 32), a.rng_offset + g_adell_rng_step, 0u, a.seed_lo,
                        a.seed_hi);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cj = c + j;
      float t = h[j];
      if (a.mean) {
        const long si = nidx * a.stat_stride_n + cj;
        t = (t - a.mean[si]) * a.rstd[si];
      }
      if (a.gamma) t = t * a.gamma[cj];
      if (a.beta) t = t + a.beta[cj];
      if (a.drop_p > 0.f) {
        const float u = (float)(rr[j] >> 8) * (1.0f / 16777216.0f);
        t = (u >= a.drop_p) ? t * keep_scale : 0.f;
      }
      float p = a.act_p;
      if (a.act_w) p = a.act_w[a.act_w_n > 1 ? cj : 0];
      h[j] = adell_act_fwd(a.act, t, p);
    }
    reinterpret_cast<float4*>(a.out)[i] = make_float4(h[0], h[1], h[2], h[3]);
  }
  // scalar tail (total not a multiple of 4 only when C is not)
  const long tail0 = n4 << 2;
  for (long e = tail0 + blockIdx.x * (long)blockDim.x + threadIdx.x; e < a.total;
       e += (long)gridDim.x * blockDim.x) {
    const int cj = (int)(e % a.C);
    const long nidx = e / a.VC;
    float t = a.x[e];
    if (a.mean) {
      const long si = nidx * a.stat_stride_n + cj;
      t = (t - a.mean[si]) * a.rstd[si];
    }
    if (a.gamma) t = t * a.gamma[cj];
    if (a.beta) t = t + a.beta[cj];
    if (a.drop_p > 0.f) {
      const uint4 r = adell_philox4((uint32_t)(e >> 2), (uint32_t)(e >> 34),
                                    a.rng_offset + g_adell_rng_step, 0u, a.seed_lo, a.seed_hi);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
      const float u = (float)(rr[e & 3] >> 8) * (1.0f / 16777216.0f);
      t = (u >= a.drop_p) ? t * keep_scale : 0.f;
    }
    float p = a.act_p;
    if (a.act_w) p = a.act_w[a.act_w_n > 1 ? cj : 0];
    a.out[e] = adell_act_fwd(a.act, t, p);
  }
}

#ifndef ADELL_EW_UNROLL
#define ADELL_EW_UNROLL 4
#endif
// the bandwidth-tuned kernels are compiled once per activation (the runtime switch inside
// the element loop made one 40 KB kernel out of nine small ones)
#define ADELL_ACT_DISPATCH(KERN, act, grid, st, args)                                         \
  switch (act) {                                                                              \
    case 0: hipLaunchKernelGGL(KERN<0>, grid, dim3(256), 0, st, args); break;                 \
    case 1: hipLaunchKernelGGL(KERN<1>, grid, dim3(256), 0, st, args); break;                 \
    case 2: hipLaunchKernelGGL(KERN<2>, grid, dim3(256), 0, st, args); break;                 \
    case 3: hipLaunchKernelGGL(KERN<3>, grid, dim3(256), 0, st, args); break;                 \
    case 4: hipLaunchKernelGGL(KERN<4>, grid, dim3(256), 0, st, args); break;                 \
    case 5: hipLaunchKernelGGL(KERN<5>, grid, dim3(256), 0, st, args); break;                 \
    case 6: hipLaunchKernelGGL(KERN<6>, grid, dim3(256), 0, st, args); break;                 \
    case 7: hipLaunchKernelGGL(KERN<7>, grid, dim3(256), 0, st, args); break;                 \
    default: hipLaunchKernelGGL(KERN<8>, grid, dim3(256), 0, st, args); break;                \
  }
// the same for the <ACT, true> instances (low-rank upstream gradient)
#define ADELL_ACT_DISPATCH_LR(KERN, act, grid, st, args)                                      \
  switch (act) {                                                                              \
    case 0: hipLaunchKernelGGL((KERN<0, true>), grid, dim3(256), 0, st, args); break;         \
    case 1: hipLaunchKernelGGL((KERN<1, true>), grid, dim3(256), 0, st, args); break;         \
    case 2: hipLaunchKernelGGL((KERN<2, true>), grid, dim3(256), 0, st, args); break;         \
    case 3: hipLaunchKernelGGL((KERN<3, true>), grid, dim3(256), 0, st, args); break;         \
    case 4: hipLaunchKernelGGL((KERN<4, true>), grid, dim3(256), 0, st, args); break;         \
    case 5: hipLaunchKernelGGL((KERN<5, true>), grid, dim3(256), 0, st, args); break;         \
    case 6: hipLaunchKernelGGL((KERN<6, true>), grid, dim3(256), 0, st, args); break;         \
    case 7: hipLaunchKernelGGL((KERN<7, true>), grid, dim3(256), 0, st, args); break;         \
    default: hipLaunchKernelGGL((KERN<8, true>), grid, dim3(256), 0, st, args); break;        \
  }
#ifndef ADELL_EW_MAXBLOCKS
#define ADELL_EW_MAXBLOCKS 65535
#endif
// Element range of a block of the bandwidth-tuned kernels. ADELL_EW_CONTIG: each block walks
// ONE contiguous chunk (a multiple of 1024 float4, so a thread keeps its channel quad) instead
// of grid-striding over the whole tensor.
#ifndef ADELL_EW_CONTIG
#define ADELL_EW_CONTIG 1
#endif
// EXPERIMENT ("ew_reverse" switch): visit the tensor in the reverse of the order the producing conv
// wrote it (its blocks advance through eight XCD ranges at once, later batch items last), so that
// the first reads find the most recently written lines still in the 256 MB Infinity Cache.
__device__ __forceinline__ unsigned adell_ew_block(int rev) {
  if (!rev) return blockIdx.x;
  const unsigned nb = gridDim.x, per = nb >> 3;
  if (per == 0 || (nb & 7)) return nb - 1 - blockIdx.x;
  return (blockIdx.x & 7) * per + (per - 1 - (blockIdx.x >> 3));
}
#if ADELL_EW_CONTIG
#define ADELL_EW_RANGE(n4)                                                                   \
  const long chunk_ = (((n4) + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;            \
  const long jbeg = (long)adell_ew_block(a.rev) * chunk_;                         \
  const long jend = jbeg + chunk_ < (n4) ? jbeg + chunk_ : (n4);                             \
  const long j0 = jbeg + threadIdx.x;                                                        \
  const long stride = 256
#else
#define ADELL_EW_RANGE(n4)                                                                   \
  const long jbeg = (long)blockIdx.x * 256;                                                  \
  const long jend = (n4);                                                                    \
  const long j0 = jbeg + threadIdx.x;                                                        \
  const long stride = (long)gridDim.x * 256
#endif
template <int ACT> __global__ void adell_norm_act_fwd_fast_kernel(NormActArgs a);

static int adell_na_fill(NormActArgs* a, const adell_norm_act_desc* d) {
  ADELL_REQUIRE(d != nullptr, "norm_act: null descriptor");
  ADELL_REQUIRE(d->N > 0 && d->V > 0 && d->C > 0, "norm_act: bad dims");
  ADELL_REQUIRE(d->act >= 0 && d->act <= ADELL_ACT_ELU, "norm_act: unknown activation");
  ADELL_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "norm_act: dropout p must be in [0,1)");
  ADELL_REQUIRE(d->stats_per_item == 0 || d->stats_per_item == 1,
                "norm_act: stats_per_item must be 0/1");
  a->VC = d->V * d->C;
  a->total = d->N * a->VC;
  a->C = d->C;
  a->stat_stride_n = d->stats_per_item ? d->C : 0;
  a->act = d->act;
  a->act_w_n = d->act_w_n;
  a->act_p = d->act_p;
  a->drop_p = d->drop_p;
  a->seed_lo = (uint32_t)(d->seed & 0xffffffffu);
  a->seed_hi = (uint32_t)(d->seed >> 32);
  a->rng_offset = d->rng_offset;
  a->rev = 0;
  return ADELL_OK;
}

static int adell_ew_blocks(long n4) {
  long b = (n4 + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" long adell_norm_act_mask_bytes(const adell_norm_act_desc* d) {
  if (!d || d->N <= 0 || d->V <= 0 || d->C <= 0) return ADELL_E_BADARG;
  return d->N * ((d->V * d->C + 255) / 256) * 32;
}

static int adell_norm_act_fwd_impl(const adell_norm_act_desc* d, const float* x,
                                   const float* mean, const float* rstd,
                                   const float* gamma, const float* beta,
                                   const float* act_w, float* out, void* keep_mask,
                                   void* stream, int split = 0, int split_exp = 0) {
  NormActArgs a = {};
  int rc = adell_na_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  a.split = split;
  a.split_scale = __builtin_ldexpf(1.0f, split_exp);
  a.mask = (unsigned long long*)keep_mask;
  a.groups = (a.VC + 255) / 256;
  ADELL_REQUIRE(x && out, "norm_act_fwd: null pointer");
  ADELL_REQUIRE((mean == nullptr) == (rstd == nullptr), "norm_act_fwd: mean/rstd mismatch");
  ADELL_REQUIRE(d->C % 4 == 0 || a.total < (1L << 31),
                "norm_act_fwd: C %% 4 != 0 needs < 2^31 elements");
  a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.act_w = act_w; a.out = out;
  a.vec = (d->C % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)out & 15) == 0);
  // (1 / 2 channels -- the 2-channel input block, the sigmoid head -- take the bandwidth-tuned
  // kernels too: a float4 is then 4 / C voxels)
  const bool narrow = d->C < 4 && a.VC % 4 == 0 && keep_mask == nullptr &&
                      (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  if ((a.vec || narrow) && adell_is_pow2(d->C) && d->C <= 1024 && (a.VC >> 2) < (1L << 40)) {
    long bx = ((a.VC >> 2) + 256 * ADELL_EW_UNROLL - 1) / (256 * ADELL_EW_UNROLL);
    if (bx > ADELL_EW_MAXBLOCKS) bx = ADELL_EW_MAXBLOCKS;
    if (bx < 1) bx = 1;
    ADELL_ACT_DISPATCH(adell_norm_act_fwd_fast_kernel, d->act, dim3((unsigned)bx, (unsigned)d->N),
                       (hipStream_t)stream, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  if (keep_mask != nullptr || split) {
    adell_set_error("norm_act_fwd_mask / _split: needs a power-of-two C <= 1024 and 16-byte aligned tensors");
    return ADELL_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(adell_norm_act_fwd_kernel,
                     dim3(adell_ew_blocks(a.vec ? (a.total >> 2) : a.total)), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_norm_act_fwd(const adell_norm_act_desc* d, const float* x,
                                  const float* mean, const float* rstd,
                                  const float* gamma, const float* beta,
                                  const float* act_w, float* out, void* stream) {
  return adell_norm_act_fwd_impl(d, x, mean, rstd, gamma, beta, act_w, out, nullptr, stream);
}

extern "C" int adell_norm_act_fwd_mask(const adell_norm_act_desc* d, const float* x,
                                       const float* mean, const float* rstd,
                                       const float* gamma, const float* beta,
                                       const float* act_w, float* out, void* keep_mask,
                                       void* stream) {
  ADELL_REQUIRE(d && (d->drop_p == 0.f || keep_mask), "norm_act_fwd_mask: null mask");
  return adell_norm_act_fwd_impl(d, x, mean, rstd, gamma, beta, act_w, out,
                                 d->drop_p > 0.f ? keep_mask : nullptr, stream);
}

// Split-row output (see NormActArgs::split): `out_rows` has the byte size of the fp32 output and holds
// [N][V][C / 16] rows of 64 bytes, every value scaled by 2^split_exp. C: a power of two, 16..1024.
// keep_mask as in adell_norm_act_fwd_mask (may be null when drop_p == 0 or no mask is wanted).
extern "C" int adell_norm_act_fwd_split(const adell_norm_act_desc* d, const float* x,
                                        const float* mean, const float* rstd,
                                        const float* gamma, const float* beta,
                                        const float* act_w, void* out_rows, int split_exp,
                                        void* keep_mask, void* stream) {
  ADELL_REQUIRE(d && d->C >= 16 && d->C % 16 == 0 && adell_is_pow2(d->C) && d->C <= 1024,
                "norm_act_fwd_split: C must be a power of two in 16..1024");
  ADELL_REQUIRE(split_exp >= -100 && split_exp <= 100, "norm_act_fwd_split: bad exponent");
  ADELL_REQUIRE((((uintptr_t)x | (uintptr_t)out_rows) & 15) == 0,
                "norm_act_fwd_split: 16-byte aligned tensors");
  return adell_norm_act_fwd_impl(d, x, mean, rstd, gamma, beta, act_w, (float*)out_rows,
                                 d->drop_p > 0.f ? keep_mask : nullptr, stream, 1, split_exp);
}

// fp32 [N][V][C] -> split rows with per-(item, chunk) exponents xk[N][C / 16] (any tensor whose
// range the caller knows), and back: x = (hi + lo) * 2^-xk. The producer-side fusion above is the
// product path; these two serve consumers that cannot read rows and the tests.
typedef _Float16 na_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 na_half4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void adell_rows_from_f32_kernel(const float* __restrict__ x,
                                                                  const int* __restrict__ xk,
                                                                  char* __restrict__ rows, long V,
                                                                  int C, long total) {
  const int nch = C >> 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int ch = (int)(i % nch);
    const long v = i / nch;                    // voxel over the whole batch
    const int n = (int)(v / V);
    const float scale = __int_as_float((xk[n * nch + ch] + 127) << 23);
    const float4* p = reinterpret_cast<const float4*>(x + v * C + ch * 16);
    na_half8 o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t4 = p[q];
      const float t[4] = {t4.x * scale, t4.y * scale, t4.z * scale, t4.w * scale};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)t[j];
        o[q >> 1][(q & 1) * 4 + j] = h;
        o[2 + (q >> 1)][(q & 1) * 4 + j] = (_Float16)(t[j] - (float)h);
      }
    }
    na_half8* dst = reinterpret_cast<na_half8*>(rows + i * 64);
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
  }
}
__global__ __launch_bounds__(256) void adell_rows_to_f32_kernel(const char* __restrict__ rows,
                                                                const int* __restrict__ xk,
                                                                float* __restrict__ x, long V, int C,
                                                                long total) {
  const int nch = C >> 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int ch = (int)(i % nch);
    const long v = i / nch;
    const int n = (int)(v / V);
    const float inv = __int_as_float((127 - xk[n * nch + ch]) << 23);
    const na_half8* src = reinterpret_cast<const na_half8*>(rows + i * 64);
    const na_half8 h0 = src[0], h1 = src[1], l0 = src[2], l1 = src[3];
    float4* p = reinterpret_cast<float4*>(x + v * C + ch * 16);
    p[0] = make_float4(((float)h0[0] + (float)l0[0]) * inv, ((float)h0[1] + (float)l0[1]) * inv,
                       ((float)h0[2] + (float)l0[2]) * inv, ((float)h0[3] + (float)l0[3]) * inv);
    p[1] = make_float4(((float)h0[4] + (float)l0[4]) * inv, ((float)h0[5] + (float)l0[5]) * inv,
                       ((float)h0[6] + (float)l0[6]) * inv, ((float)h0[7] + (float)l0[7]) * inv);
    p[2] = make_float4(((float)h1[0] + (float)l1[0]) * inv, ((float)h1[1] + (float)l1[1]) * inv,
                       ((float)h1[2] + (float)l1[2]) * inv, ((float)h1[3] + (float)l1[3]) * inv);
    p[3] = make_float4(((float)h1[4] + (float)l1[4]) * inv, ((float)h1[5] + (float)l1[5]) * inv,
                       ((float)h1[6] + (float)l1[6]) * inv, ((float)h1[7] + (float)l1[7]) * inv);
  }
}
static int adell_rows_convert(const void* src, void* dst, int N, long V, int C, const int* xk,
                              int to_rows, void* stream) {
  ADELL_REQUIRE(src && dst && xk && N > 0 && V > 0 && C >= 16 && C % 16 == 0,
                "split rows: bad arguments (C must be a multiple of 16)");
  ADELL_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "split rows: 16-byte aligned tensors");
  const long total = (long)N * V * (C >> 4);
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (to_rows)
    hipLaunchKernelGGL(adell_rows_from_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (const float*)src, xk, (char*)dst, V, C, total);
  else
    hipLaunchKernelGGL(adell_rows_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, (const char*)src, xk, (float*)dst, V, C, total);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
extern "C" int adell_split_rows_from_f32(const float* x, int N, long V, int C, const int* xk,
                                         void* rows, void* stream) {
  return adell_rows_convert(x, rows, N, V, C, xk, 1, stream);
}
extern "C" int adell_split_rows_to_f32(const void* rows, int N, long V, int C, const int* xk,
                                       float* x, void* stream) {
  return adell_rows_convert(rows, x, N, V, C, xk, 0, stream);
}

// ---------------------------------------------------------------------------
// Backward of out = act(dropout(hn * gamma + beta)), hn = (x - mean) * rstd.
//   dt  = dout * act'(u) * mask / (1-p)
//   dx  = rstd * (gamma*dt - c1 - hn*c2),  c1 = mean(gamma*dt), c2 = mean(gamma*dt*hn)
// Pass 1 writes per-slab partials (sum dt, sum dt*hn); the finalize kernel
// turns them into c1/c2 (and dgamma/dbeta); pass 2 is elementwise.
// ---------------------------------------------------------------------------
struct NormActBwdArgs {
  const float* x;
  const float* dout;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  const float* act_w;
  const float* c1;   // [N*C] or [C]
  const float* c2;
  float* dx;
  float* part;       // [N][ntiles][C][2]
  long V;            // voxels per item
  long VC, total;
  int C, stat_stride_n, act, act_w_n, ntiles;
  float act_p, drop_p;
  uint32_t seed_lo, seed_hi, rng_offset;
  int rev;             // experiment: reversed block order (adell_ew_block)
  // low-rank upstream gradient (fast kernels): dout[v][c] = sum_o lr_g[v][o] lr_w[o][c], o < lr_co
  // <= 4 -- the backward-data of a 1x1x1 conv with <= 4 output channels (the logits head,
  // unet.py:626-655) folded into the site's backward: dout is never materialised
  const float* lr_g;   // [N][V][lr_co] or null
  const float* lr_w;   // [lr_co][C]
  int lr_co, lr_shift; // lr_shift = log2(C) - 2: float4 index -> voxel
};

__device__ __forceinline__ void adell_na_bwd_elem(const NormActBwdArgs& a, float x, float dout,
                                                  long nidx, int c, bool keep,
                                                  float keep_scale, float* dt, float* hn) {
  float h = x;
  if (a.mean) {
    const long si = nidx * a.stat_stride_n + c;
    h = (x - a.mean[si]) * a.rstd[si];
  }
  float t = h;
  if (a.gamma) t *= a.gamma[c];
  if (a.beta) t += a.beta[c];
  const float u = keep ? t * keep_scale : 0.f;
  float p = a.act_p;
  if (a.act_w) p = a.act_w[a.act_w_n > 1 ? c : 0];
  const float du = dout * adell_act_grad(a.act, u, p);
  *dt = keep ? du * keep_scale : 0.f;
  *hn = h;
}

__device__ __forceinline__ void adell_na_keep4(const NormActBwdArgs& a, long e4, bool keep[4]) {
  keep[0] = keep[1] = keep[2] = keep[3] = true;
  if (a.drop_p > 0.f) {
    const uint4 r = adell_philox4((uint32_t)e4, (uint32_t)(e4 >> 32), a.rng_offset + g_adell_rng_step, 0u,
                                  a.seed_lo, a.seed_hi);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      keep[j] = (float)(rr[j] >> 8) * (1.0f / 16777216.0f) >= a.drop_p;
  }
}

// grid (ntiles, N); each block reduces ADELL_STATS_SLAB voxels. VEC: C % 4 == 0.
template <bool VEC>
__global__ __launch_bounds__(256) void adell_na_bwd_partials_kernel(NormActBwdArgs a) {
  __shared__ float sh[256][2][VEC ? 4 : 1];
  constexpr int W = VEC ? 4 : 1;
  const int n = blockIdx.y, tile = blockIdx.x;
  const long v0 = (long)tile * ADELL_STATS_SLAB;
  long v1 = v0 + ADELL_STATS_SLAB;
  if (v1 > a.V) v1 = a.V;
  const int CW = a.C / W;  // thread columns
  const int CG = CW < 256 ? CW : 256;
  const int VL = 256 / CG;
  const int cl = threadIdx.x % CG, vl = threadIdx.x / CG;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (int cb = 0; cb < CW; cb += CG) {
    const int cw = cb + cl;
    float A[W], B[W];
#pragma unroll
    for (int j = 0; j < W; ++j) A[j] = B[j] = 0.f;
    if (vl < VL && cw < CW) {
      for (long v = v0 + vl; v < v1; v += VL) {
        const long e = ((long)n * a.V + v) * a.C + (long)cw * W;
        if constexpr (VEC) {
          const float4 xv = *reinterpret_cast<const float4*>(a.x + e);
          const float4 gv = *reinterpret_cast<const float4*>(a.dout + e);
          const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
          bool keep[4];
          adell_na_keep4(a, e >> 2, keep);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float dt, hn;
            adell_na_bwd_elem(a, xs[j], gs[j], n, cw * 4 + j, keep[j], keep_scale, &dt, &hn);
            A[j] += dt;
            B[j] += dt * hn;
          }
        } else {
          bool keep[4];
          adell_na_keep4(a, e >> 2, keep);
          float dt, hn;
          adell_na_bwd_elem(a, a.x[e], a.dout[e], n, cw, keep[e & 3], keep_scale, &dt, &hn);
          A[0] += dt;
          B[0] += dt * hn;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      sh[threadIdx.x][0][j] = A[j];
      sh[threadIdx.x][1][j] = B[j];
    }
    __syncthreads();
    if (vl == 0 && cw < CW) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < VL; ++k) {
          s1 += sh[k * CG + cl][0][j];
          s2 += sh[k * CG + cl][1][j];
        }
        float* o = a.part + (((size_t)n * a.ntiles + tile) * a.C + cw * W + j) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
    __syncthreads();
  }
}

// c1/c2 [N][C] (per_item) or [C]; dgamma/dbeta [C] (optional).
__global__ __launch_bounds__(1024) void adell_na_bwd_finalize_kernel(
    const float* __restrict__ part, int N, int ntiles, int C, double count, int per_item,
    const float* __restrict__ gamma, float* __restrict__ c1, float* __restrict__ c2,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sh[32][32][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double accA = 0.0, accB = 0.0;  // over all items (batch statistics / dgamma)
  for (int n = 0; n < N; ++n) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
      for (int t = sl; t < ntiles; t += 32) {
        const float2 v =
            *reinterpret_cast<const float2*>(part + (((size_t)n * ntiles + t) * C + c) * 2);
        s1 += (double)v.x;
        s2 += (double)v.y;
      }
    sh[sl][cl][0] = s1;
    sh[sl][cl][1] = s2;
    __syncthreads();
    if (sl == 0 && c < C) {
      double A = 0.0, B = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        A += sh[k][cl][0];
        B += sh[k][cl][1];
      }
      accA += A;
      accB += B;
      if (per_item) {
        const double g = gamma ? (double)gamma[c] : 1.0;
        c1[(size_t)n * C + c] = (float)(g * A / count);
        c2[(size_t)n * C + c] = (float)(g * B / count);
      }
    }
    __syncthreads();
  }
  if (sl == 0 && c < C) {
    if (!per_item) {
      const double g = gamma ? (double)gamma[c] : 1.0;
      c1[c] = (float)(g * accA / (count * N));
      c2[c] = (float)(g * accB / (count * N));
    }
    if (dgamma) dgamma[c] = (float)accB;
    if (dbeta) dbeta[c] = (float)accA;
  }
}

// The per-item case without affine-parameter gradients (every InstanceNorm of the U-Nets): one
// block per (8 channels, item) instead of one block walking all items (15 us -> a few us, 35 times
// per training step). Same fixed-order fp64 fold: tile lanes, then a tree over the 32 lanes.
// (pstride / poff: the partials of a site may be columns [poff, poff + C) of rows pstride wide --
// what the fused backward-data epilogue of a two-destination conv writes)
__global__ __launch_bounds__(256) void adell_na_bwd_finalize_item_kernel(
    const float* __restrict__ part, int ntiles, int C, double count,
    const float* __restrict__ gamma, float* __restrict__ c1, float* __restrict__ c2,
    int pstride, int poff) {
  __shared__ double sh[32][8][2];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, n = blockIdx.y;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (c < C) {
    int t = sl;
    for (; t + 32 < ntiles; t += 64) {
      const float2 u =
          *reinterpret_cast<const float2*>(part + (((size_t)n * ntiles + t) * pstride + poff + c) * 2);
      const float2 v = *reinterpret_cast<const float2*>(
          part + (((size_t)n * ntiles + t + 32) * pstride + poff + c) * 2);
      a0 += (double)u.x; b0 += (double)u.y;
      a1 += (double)v.x; b1 += (double)v.y;
    }
    for (; t < ntiles; t += 32) {
      const float2 u =
          *reinterpret_cast<const float2*>(part + (((size_t)n * ntiles + t) * pstride + poff + c) * 2);
      a0 += (double)u.x; b0 += (double)u.y;
    }
  }
  sh[sl][cl][0] = a0 + a1;
  sh[sl][cl][1] = b0 + b1;
  __syncthreads();
  if (sl == 0 && c < C) {
    double A = 0.0, B = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      A += sh[k][cl][0];
      B += sh[k][cl][1];
    }
    const double g = gamma ? (double)gamma[c] : 1.0;
    c1[(size_t)n * C + c] = (float)(g * A / count);
    c2[(size_t)n * C + c] = (float)(g * B / count);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void adell_na_bwd_apply_kernel(NormActBwdArgs a) {
  constexpr int W = VEC ? 4 : 1;
  const long nw = a.total / W;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nw;
       i += (long)gridDim.x * blockDim.x) {
    const long e = i * W;
    const int c = (int)(e % a.C);
    const long nidx = e / a.VC;
    bool keep[4];
    adell_na_keep4(a, e >> 2, keep);
    float xs[W], gs[W], out[W];
    if constexpr (VEC) {
      const float4 xv = *reinterpret_cast<const float4*>(a.x + e);
      const float4 gv = *reinterpret_cast<const float4*>(a.dout + e);
      xs[0] = xv.x; xs[1] = xv.y; xs[2] = xv.z; xs[3] = xv.w;
      gs[0] = gv.x; gs[1] = gv.y; gs[2] = gv.z; gs[3] = gv.w;
    } else {
      xs[0] = a.x[e];
      gs[0] = a.dout[e];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int cj = c + j;
      float dt, hn;
      adell_na_bwd_elem(a, xs[j], gs[j], nidx, cj, VEC ? keep[j] : keep[e & 3], keep_scale,
                        &dt, &hn);
      float r = dt;
      if (a.gamma) r *= a.gamma[cj];
      if (a.mean) {
        const long si = nidx * a.stat_stride_n + cj;
        r = a.rstd[si] * (r - a.c1[si] - hn * a.c2[si]);
      }
      out[j] = r;
    }
    if constexpr (VEC)
      *reinterpret_cast<float4*>(a.dx + e) = make_float4(out[0], out[1], out[2], out[3]);
    else
      a.dx[e] = out[0];
  }
}

template <int ACT, bool LR = false> __global__ void adell_na_bwd_apply_fast_kernel(NormActBwdArgs a);
template <int ACT, bool LR = false> __global__ void adell_na_bwd_partials_fast_kernel(NormActBwdArgs a);
// blocks per item of the grid-strided partials kernel (each covers >= 1024 float4)
static long adell_na_fast_blocks(long V, int C) {
  long bx = ((V * C >> 2) + 256 * 4 - 1) / (256 * 4);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  return bx;
}

static int adell_nab_fill(NormActBwdArgs* a, const adell_norm_act_desc* d) {
  ADELL_REQUIRE(d != nullptr, "norm_act_bwd: null descriptor");
  ADELL_REQUIRE(d->N > 0 && d->V > 0 && d->C > 0, "norm_act_bwd: bad dims");
  ADELL_REQUIRE(d->act >= 0 && d->act <= ADELL_ACT_ELU, "norm_act_bwd: unknown activation");
  ADELL_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "norm_act_bwd: dropout p must be in [0,1)");
  a->V = d->V;
  a->VC = d->V * d->C;
  a->total = d->N * a->VC;
  a->C = d->C;
  a->stat_stride_n = d->stats_per_item ? d->C : 0;
  a->act = d->act;
  a->act_w_n = d->act_w_n;
  a->act_p = d->act_p;
  a->drop_p = d->drop_p;
  a->seed_lo = (uint32_t)(d->seed & 0xffffffffu);
  a->seed_hi = (uint32_t)(d->seed >> 32);
  a->rng_offset = d->rng_offset;
  a->ntiles = adell_channel_partials_ntiles(d->V);
  return ADELL_OK;
}

// workspace floats: partials [N][ntiles][C][2] + c1 [N][C] + c2 [N][C]
extern "C" long adell_norm_act_bwd_workspace(const adell_norm_act_desc* d) {
  if (!d || d->N <= 0 || d->V <= 0 || d->C <= 0) return ADELL_E_BADARG;
  long nt = adell_channel_partials_ntiles(d->V);
  const long ntf = adell_na_fast_blocks(d->V, d->C);
  if (ntf > nt) nt = ntf;
  return (long)sizeof(float) * (d->N * nt * d->C * 2 + 2 * d->N * d->C);
}

static int adell_norm_act_bwd_impl(const adell_norm_act_desc* d, const float* x,
                                   const float* dout, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const float* act_w,
                                   float* dx, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, void* stream, const float* lr_g,
                                   const float* lr_w, int lr_co);

extern "C" int adell_norm_act_bwd(const adell_norm_act_desc* d, const float* x,
                                  const float* dout, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, const float* act_w,
                                  float* dx, float* dgamma, float* dbeta, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(dout, "norm_act_bwd: null pointer");
  return adell_norm_act_bwd_impl(d, x, dout, mean, rstd, gamma, beta, act_w, dx, dgamma, dbeta,
                                 workspace, workspace_bytes, stream, nullptr, nullptr, 0);
}

// The same with a LOW-RANK upstream gradient dout[v][c] = sum_o g[v][o] w[o][c] (g: [N][V][co],
// w: [co][C], co <= 4): the site in front of a 1x1x1 conv with <= 4 output channels (the logits
// head: Conv3d -> ADN -> Conv3d(C -> 1, k = 1), unet.py:626-655) takes that conv's dY and weight
// and never sees -- nor does anyone write -- its full-size backward-data result. Needs the
// bandwidth-tuned kernels (power-of-two C in 4..1024, aligned tensors).
extern "C" int adell_norm_act_bwd_lowrank(const adell_norm_act_desc* d, const float* x,
                                          const float* g, const float* w, int co,
                                          const float* mean, const float* rstd, float* dx,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && g && w && co >= 1 && co <= 4, "norm_act_bwd_lowrank: 1..4 factors expected");
  ADELL_REQUIRE(adell_is_pow2(d->C) && d->C >= 4 && d->C <= 1024 &&
                    (((uintptr_t)x | (uintptr_t)dx) & 15) == 0,
                "norm_act_bwd_lowrank: needs a power-of-two C in 4..1024 and aligned tensors");
  return adell_norm_act_bwd_impl(d, x, x, mean, rstd, nullptr, nullptr, nullptr, dx, nullptr,
                                 nullptr, workspace, workspace_bytes, stream, g, w, co);
}

static int adell_norm_act_bwd_impl(const adell_norm_act_desc* d, const float* x,
                                   const float* dout, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const float* act_w,
                                   float* dx, float* dgamma, float* dbeta, void* workspace,
                                   size_t workspace_bytes, void* stream, const float* lr_g,
                                   const float* lr_w, int lr_co) {
  NormActBwdArgs a = {};
  int rc = adell_nab_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  a.lr_g = lr_g; a.lr_w = lr_w; a.lr_co = lr_co;
  a.lr_shift = adell_ilog2(d->C) - 2;
  ADELL_REQUIRE(x && dout && dx, "norm_act_bwd: null pointer");
  ADELL_REQUIRE((mean == nullptr) == (rstd == nullptr), "norm_act_bwd: mean/rstd mismatch");
  hipStream_t st = (hipStream_t)stream;
  a.x = x; a.dout = dout; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.act_w = act_w; a.dx = dx;
  const bool vec = (d->C % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                   (((uintptr_t)dout & 15) == 0) && (((uintptr_t)dx & 15) == 0);
  const bool narrow = d->C < 4 && a.VC % 4 == 0 && !lr_g && (((uintptr_t)x & 15) == 0) &&
                      (((uintptr_t)dout & 15) == 0) && (((uintptr_t)dx & 15) == 0);
  const bool fast = (vec || narrow) && adell_is_pow2(d->C) && d->C <= 1024;
  if (mean || dgamma || dbeta) {
    ADELL_REQUIRE(workspace && (long)workspace_bytes >= adell_norm_act_bwd_workspace(d),
                  "norm_act_bwd: workspace too small");
    float* part = (float*)workspace;
    long ntmax = a.ntiles;
    if (adell_na_fast_blocks(d->V, d->C) > ntmax) ntmax = adell_na_fast_blocks(d->V, d->C);
    float* c1 = part + (size_t)d->N * ntmax * d->C * 2;
    float* c2 = c1 + (size_t)d->N * d->C;
    a.part = part; a.c1 = c1; a.c2 = c2;
    dim3 grid(a.ntiles, (unsigned)d->N);
    if (fast) {
      a.ntiles = (int)adell_na_fast_blocks(d->V, d->C);
      if (lr_g) {
        ADELL_ACT_DISPATCH_LR(adell_na_bwd_partials_fast_kernel, d->act,
                              dim3(a.ntiles, (unsigned)d->N), st, a);
      } else {
        ADELL_ACT_DISPATCH(adell_na_bwd_partials_fast_kernel, d->act,
                           dim3(a.ntiles, (unsigned)d->N), st, a);
      }
    } else if (vec)
      hipLaunchKernelGGL(adell_na_bwd_partials_kernel<true>, grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(adell_na_bwd_partials_kernel<false>, grid, dim3(256), 0, st, a);
    if (d->stats_per_item && !dgamma && !dbeta && d->N <= 65535)
      hipLaunchKernelGGL(adell_na_bwd_finalize_item_kernel,
                         dim3(adell_cdiv(d->C, 8), (unsigned)d->N), dim3(256), 0, st,
                         (const float*)part, a.ntiles, d->C, (double)d->V, gamma, c1, c2, d->C, 0);
    else
      hipLaunchKernelGGL(adell_na_bwd_finalize_kernel, dim3(adell_cdiv(d->C, 32)), dim3(1024), 0,
                         st, (const float*)part, (int)d->N, a.ntiles, d->C, (double)d->V,
                         d->stats_per_item, gamma, c1, c2, dgamma, dbeta);
  }
  const long nw = vec ? a.total / 4 : a.total;
  if (fast) {
    long bx = ((a.VC >> 2) + 256 * ADELL_EW_UNROLL - 1) / (256 * ADELL_EW_UNROLL);
    if (bx > ADELL_EW_MAXBLOCKS) bx = ADELL_EW_MAXBLOCKS;
    if (bx < 1) bx = 1;
    if (lr_g) {
      ADELL_ACT_DISPATCH_LR(adell_na_bwd_apply_fast_kernel, d->act,
                            dim3((unsigned)bx, (unsigned)d->N), st, a);
    } else {
      ADELL_ACT_DISPATCH(adell_na_bwd_apply_fast_kernel, d->act,
                         dim3((unsigned)bx, (unsigned)d->N), st, a);
    }
  } else if (vec)
    hipLaunchKernelGGL(adell_na_bwd_apply_kernel<true>, dim3(adell_ew_blocks(nw)), dim3(256), 0,
                       st, a);
  else
    hipLaunchKernelGGL(adell_na_bwd_apply_kernel<false>, dim3(adell_ew_blocks(nw)), dim3(256),
                       0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Bandwidth-tuned variants used when C is a power of two <= 1024 (every U-Net
// layer): the grid stride is a multiple of 1024 elements, so a thread keeps ONE
// channel quad for its whole life and hoists mean / rstd / gamma / beta / c1 / c2
// into registers; no 64-bit div/mod per element; 4 independent 16-byte loads are
// in flight per thread before any arithmetic. One grid row per batch item.
// ---------------------------------------------------------------------------
struct NaConst {
  float m[4], r[4], g[4], b[4], p[4], c1[4], c2[4];
};

__device__ __forceinline__ void adell_na_consts(NaConst& k, const float* mean, const float* rstd,
                                                const float* gamma, const float* beta,
                                                const float* act_w, int act_w_n, float act_p,
                                                const float* c1, const float* c2, long sbase,
                                                int c, int cmask) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // (1 or 2 channels: the four lanes of a float4 are 4 / C voxels, channels (c + j) mod C)
    const int cj = (c + j) & cmask;
    k.m[j] = mean ? mean[sbase + cj] : 0.f;
    k.r[j] = rstd ? rstd[sbase + cj] : 1.f;
    k.g[j] = gamma ? gamma[cj] : 1.f;
    k.b[j] = beta ? beta[cj] : 0.f;
    k.p[j] = act_w ? act_w[act_w_n > 1 ? cj : 0] : act_p;
    k.c1[j] = c1 ? c1[sbase + cj] : 0.f;
    k.c2[j] = c2 ? c2[sbase + cj] : 0.f;
  }
}

// this thread's rows of the low-rank weight: w[o][c .. c + 3]
struct NaLowRank {
  float w[4][4];
};
__device__ __forceinline__ void adell_na_lr_load(NaLowRank& l, const NormActBwdArgs& a, int c) {
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      l.w[o][q] = o < a.lr_co ? a.lr_w[o * a.C + c + q] : 0.f;
}
// dout quad jj of item n from the low-rank factors (voxel = jj >> lr_shift)
__device__ __forceinline__ float4 adell_na_lr_dout(const NaLowRank& l, const NormActBwdArgs& a,
                                                   int n, long jj) {
  const float* g = a.lr_g + ((long)n * a.V + (jj >> a.lr_shift)) * a.lr_co;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int o = 0; o < 4; ++o)
    if (o < a.lr_co) {
      const float gv = g[o];
      r.x = fmaf(gv, l.w[o][0], r.x); r.y = fmaf(gv, l.w[o][1], r.y);
      r.z = fmaf(gv, l.w[o][2], r.z); r.w = fmaf(gv, l.w[o][3], r.w);
    }
  return r;
}

template <int ACT>
__global__ __launch_bounds__(256) void adell_norm_act_fwd_fast_kernel(NormActArgs a) {
  const int n = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const long n4 = a.VC >> 2;  // float4 per item
  ADELL_EW_RANGE(n4);
  const int c = (int)((j0 << 2) & (a.C - 1));
  NaConst k;
  adell_na_consts(k, a.mean, a.rstd, a.gamma, a.beta, a.act_w, a.act_w_n, a.act_p, nullptr,
                  nullptr, (long)n * a.stat_stride_n, c, a.C - 1);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const float4* xin = reinterpret_cast<const float4*>(a.x) + (long)n * n4;
  float4* yout = reinterpret_cast<float4*>(a.out) + (long)n * n4;
  for (long j = j0; j < jend; j += stride * ADELL_EW_UNROLL) {
    float4 v[ADELL_EW_UNROLL];
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u)
      if (j + u * stride < jend) v[u] = xin[j + u * stride];
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u) {
      const long jj = j + u * stride;
      if (jj >= jend) break;
      float h[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      uint32_t rr[4] = {0, 0, 0, 0};
      if (a.drop_p > 0.f) {
        const long gi = (long)n * n4 + jj;  // same counter as the generic kernel
        const uint4 r = adell_philox4((uint32_t)gi, (uint32_t)(gi >> 32), a.rng_offset + g_adell_rng_step, 0u,
                                      a.seed_lo, a.seed_hi);
        rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
      }
      bool kept[4] = {true, true, true, true};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = ((h[q] - k.m[q]) * k.r[q]) * k.g[q] + k.b[q];
        if (a.drop_p > 0.f) {
          const float uu = (float)(rr[q] >> 8) * (1.0f / 16777216.0f);
          kept[q] = uu >= a.drop_p;
          t = kept[q] ? t * keep_scale : 0.f;
        }
        h[q] = adell_act_fwd(ACT, t, k.p[q]);
      }
      if (a.split) {
        // this thread's four channels of a 16-channel chunk: 8 bytes of the row's hi half and 8 of
        // its lo half; the four lanes of a chunk fill the 64-byte row between them
        na_half4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // saturate at fp16's largest finite value: the exponent is a host-side bound (functional.
          // _rows_exponent) that assumes exact statistics; an overshoot must not become inf -> NaN
          const float t = __builtin_amdgcn_fmed3f(h[q] * a.split_scale, -65504.f, 65504.f);
          const _Float16 hh = (_Float16)t;
          hi[q] = hh;
          lo[q] = (_Float16)(t - (float)hh);
        }
        const int within = (int)((jj << 2) & 15);
        char* row = reinterpret_cast<char*>(a.out) + 4 * ((long)n * a.VC + (jj << 2) - within);
        *reinterpret_cast<na_half4*>(row + within * 2) = hi;
        *reinterpret_cast<na_half4*>(row + 32 + within * 2) = lo;
      } else {
        yout[jj] = make_float4(h[0], h[1], h[2], h[3]);
      }
      if (a.mask != nullptr) {
        // the 64 lanes of a wave hold 64 consecutive float4 = one 256-element group: word q of
        // the group = the keep bits of sub-element q, bit = lane (lanes past the end: 0)
        unsigned long long b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = __ballot(kept[q]);
        const int ln = threadIdx.x & 63;
        if (ln < 4)
          a.mask[((long)n * a.groups + (jj >> 6)) * 4 + ln] =
              ln == 0 ? b[0] : (ln == 1 ? b[1] : (ln == 2 ? b[2] : b[3]));
      }
    }
  }
}

template <int ACT, bool LR>
__global__ __launch_bounds__(256) void adell_na_bwd_apply_fast_kernel(NormActBwdArgs a) {
  const int n = blockIdx.y;
  const long n4 = a.VC >> 2;
  ADELL_EW_RANGE(n4);
  const int c = (int)((j0 << 2) & (a.C - 1));
  NaConst k;
  adell_na_consts(k, a.mean, a.rstd, a.gamma, a.beta, a.act_w, a.act_w_n, a.act_p, a.c1, a.c2,
                  (long)n * a.stat_stride_n, c, a.C - 1);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const float4* xin = reinterpret_cast<const float4*>(a.x) + (long)n * n4;
  const float4* gin = reinterpret_cast<const float4*>(a.dout) + (long)n * n4;
  float4* dxo = reinterpret_cast<float4*>(a.dx) + (long)n * n4;
  const bool norm = a.mean != nullptr;
  NaLowRank lr;
  if (LR) adell_na_lr_load(lr, a, c);
  for (long j = j0; j < jend; j += stride * ADELL_EW_UNROLL) {
    float4 xv[ADELL_EW_UNROLL], gv[ADELL_EW_UNROLL];
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u)
      if (j + u * stride < jend) {
        xv[u] = xin[j + u * stride];
        gv[u] = LR ? adell_na_lr_dout(lr, a, n, j + u * stride) : gin[j + u * stride];
      }
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u) {
      const long jj = j + u * stride;
      if (jj >= jend) break;
      const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
      const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      bool keep[4];
      adell_na_keep4(a, (long)n * n4 + jj, keep);
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float hn = (xs[q] - k.m[q]) * k.r[q];
        const float t = hn * k.g[q] + k.b[q];
        const float uu = keep[q] ? t * keep_scale : 0.f;
        const float du = gs[q] * adell_act_grad(ACT, uu, k.p[q]);
        const float dt = keep[q] ? du * keep_scale : 0.f;
        float r = dt * k.g[q];
        if (norm) r = k.r[q] * (r - k.c1[q] - hn * k.c2[q]);
        o[q] = r;
      }
      dxo[jj] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// grid (blocks per item, N), grid-strided like the apply kernel: a thread keeps one channel
// quad, accumulates (sum dt, sum dt*xhat) over its whole share in registers, and the block
// folds its 256 / (C/4) threads per quad once at the end (fixed order).
template <int ACT, bool LR>
__global__ __launch_bounds__(256) void adell_na_bwd_partials_fast_kernel(NormActBwdArgs a) {
  __shared__ float sh[8][256];
  const int n = blockIdx.y;
  const long n4 = a.VC >> 2;
  ADELL_EW_RANGE(n4);
  const int c = (int)((j0 << 2) & (a.C - 1));
  NaConst k;
  adell_na_consts(k, a.mean, a.rstd, a.gamma, a.beta, a.act_w, a.act_w_n, a.act_p, nullptr,
                  nullptr, (long)n * a.stat_stride_n, c, a.C - 1);
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const float4* xin = reinterpret_cast<const float4*>(a.x) + (long)n * n4;
  const float4* gin = reinterpret_cast<const float4*>(a.dout) + (long)n * n4;
  float A[4] = {0.f, 0.f, 0.f, 0.f}, B[4] = {0.f, 0.f, 0.f, 0.f};
  NaLowRank lr;
  if (LR) adell_na_lr_load(lr, a, c);
  for (long j = j0; j < jend; j += stride * ADELL_EW_UNROLL) {
    float4 xv[ADELL_EW_UNROLL], gv[ADELL_EW_UNROLL];
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u)
      if (j + u * stride < jend) {
        xv[u] = xin[j + u * stride];
        gv[u] = LR ? adell_na_lr_dout(lr, a, n, j + u * stride) : gin[j + u * stride];
      }
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u) {
      const long jj = j + u * stride;
      if (jj >= jend) break;
      const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
      const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      bool keep[4];
      adell_na_keep4(a, (long)n * n4 + jj, keep);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float hn = (xs[q] - k.m[q]) * k.r[q];
        const float t = hn * k.g[q] + k.b[q];
        const float uu = keep[q] ? t * keep_scale : 0.f;
        const float du = gs[q] * adell_act_grad(ACT, uu, k.p[q]);
        const float dt = keep[q] ? du * keep_scale : 0.f;
        A[q] += dt;
        B[q] += dt * hn;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sh[q][threadIdx.x] = A[q];
    sh[4 + q][threadIdx.x] = B[q];
  }
  __syncthreads();
  if (a.C < 4) {
    // 1 or 2 channels: every thread holds the same "quad" (channels q mod C): fold the lanes of
    // a channel, then the 256 threads in index order (wave shuffles would do; this runs once)
    if (threadIdx.x < 2 * a.C) {
      const int ch = threadIdx.x % a.C, which = threadIdx.x / a.C;
      float s1 = 0.f;
      for (int kk = 0; kk < 256; ++kk)
        for (int q = ch; q < 4; q += a.C) s1 += sh[which * 4 + q][kk];
      a.part[(((size_t)n * a.ntiles + blockIdx.x) * a.C + ch) * 2 + which] = s1;
    }
    return;
  }
  // threads of one block that share a channel quad are CG = C/4 apart
  const int CG = a.C >> 2;
  const int VL = CG >= 256 ? 1 : 256 / CG;
  const int nout = (CG < 256 ? CG : 256) * 8;
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int cl = o % (nout / 8), q8 = o / (nout / 8);
    float s1 = 0.f;
    for (int kk = 0; kk < VL; ++kk) s1 += sh[q8][kk * CG + cl];
    // channel of this quad slot: the block's first thread has quad (blockIdx.x*256) % CG
    const int quad = (int)(((jbeg + cl) << 2) & (a.C - 1));
    const int ch = quad + (q8 & 3);
    a.part[(((size_t)n * a.ntiles + blockIdx.x) * a.C + ch) * 2 + (q8 >> 2)] = s1;
  }
}


// Second half of a site's backward when dout already carries the activation / dropout derivative
// (dt, written by the fused backward-data epilogue of conv_igemm_f16.h):
//   dx = rstd * (dt - c1 - xhat * c2),  xhat = (x - mean) * rstd
// -- no transcendental, no Philox: reads x and dt once, writes dx once (dx may alias dt).
__global__ __launch_bounds__(256) void adell_na_bwd_apply_dt_kernel(NormActBwdArgs a) {
  const int n = blockIdx.y;
  const long n4 = a.VC >> 2;
  ADELL_EW_RANGE(n4);
  const int c = (int)((j0 << 2) & (a.C - 1));
  NaConst k;
  adell_na_consts(k, a.mean, a.rstd, nullptr, nullptr, nullptr, 0, 0.f, a.c1, a.c2,
                  (long)n * a.stat_stride_n, c, a.C - 1);
  const float4* xin = reinterpret_cast<const float4*>(a.x) + (long)n * n4;
  const float4* gin = reinterpret_cast<const float4*>(a.dout) + (long)n * n4;
  float4* dxo = reinterpret_cast<float4*>(a.dx) + (long)n * n4;
  for (long j = j0; j < jend; j += stride * ADELL_EW_UNROLL) {
    float4 xv[ADELL_EW_UNROLL], gv[ADELL_EW_UNROLL];
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u)
      if (j + u * stride < jend) {
        xv[u] = xin[j + u * stride];
        gv[u] = gin[j + u * stride];
      }
#pragma unroll
    for (int u = 0; u < ADELL_EW_UNROLL; ++u) {
      const long jj = j + u * stride;
      if (jj >= jend) break;
      const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
      const float gs[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float hn = (xs[q] - k.m[q]) * k.r[q];
        o[q] = k.r[q] * (gs[q] - k.c1[q] - hn * k.c2[q]);
      }
      dxo[jj] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

extern "C" int adell_norm_act_bwd_from_dt(const adell_norm_act_desc* d, const float* x,
                                          const float* dt, const float* mean, const float* rstd,
                                          const float* partials, int ntiles, int pstride, int poff,
                                          float* dx, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  NormActBwdArgs a = {};
  int rc = adell_nab_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && dt && dx && mean && rstd && partials && workspace,
                "norm_act_bwd_from_dt: null pointer");
  ADELL_REQUIRE(d->stats_per_item == 1 && d->N <= 65535,
                "norm_act_bwd_from_dt: instance statistics only");
  ADELL_REQUIRE(ntiles > 0 && pstride >= d->C && poff >= 0 && poff + d->C <= pstride,
                "norm_act_bwd_from_dt: bad partials layout");
  ADELL_REQUIRE(adell_is_pow2(d->C) && d->C % 4 == 0 && d->C <= 1024 &&
                    (((uintptr_t)x | (uintptr_t)dt | (uintptr_t)dx) & 15) == 0,
                "norm_act_bwd_from_dt: needs a power-of-two C in 4..1024 and aligned tensors");
  // the fused epilogue leaves one row per BRICK (thousands per item): fold them 256 at a time
  // first, on a grid that fills the chip (a single-level fold runs on C / 8 x N blocks: 80 us)
  const int Z = ntiles > 512 ? (ntiles + 255) / 256 : 0;
  ADELL_REQUIRE(workspace_bytes >= sizeof(float) * (2 + 2 * (size_t)Z) * d->N * d->C,
                "norm_act_bwd_from_dt: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* c1 = (float*)workspace;
  float* c2 = c1 + (size_t)d->N * d->C;
  if (Z > 0) {
    float* folded = c2 + (size_t)d->N * d->C;
    hipLaunchKernelGGL(adell_stats_fold_kernel, dim3(adell_cdiv(d->C, 32), (unsigned)d->N, Z),
                       dim3(256), 0, st, partials, ntiles, d->C, Z, folded, pstride, poff);
    partials = folded;
    ntiles = Z;
    pstride = d->C;
    poff = 0;
  }
  hipLaunchKernelGGL(adell_na_bwd_finalize_item_kernel, dim3(adell_cdiv(d->C, 8), (unsigned)d->N),
                     dim3(256), 0, st, partials, ntiles, d->C, (double)d->V, (const float*)nullptr,
                     c1, c2, pstride, poff);
  a.x = x; a.dout = dt; a.mean = mean; a.rstd = rstd; a.c1 = c1; a.c2 = c2; a.dx = dx;
  long bx = ((a.VC >> 2) + 256 * ADELL_EW_UNROLL - 1) / (256 * ADELL_EW_UNROLL);
  if (bx > ADELL_EW_MAXBLOCKS) bx = ADELL_EW_MAXBLOCKS;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(adell_na_bwd_apply_dt_kernel, dim3((unsigned)bx, (unsigned)d->N), dim3(256), 0,
                     st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// PReLU weight gradient (torch.nn.PReLU, the reference's default activation_fn:
// unet.py:56, adn_fn.py:56-152): with u = dropout(norm(x)), y = u > 0 ? u : a*u, so
// dL/da[c] = sum over the elements of channel c with u < 0 of dout * u. Column-sum layout:
// grid (row chunks, groups of 64 channels), block = 64 channels x 4 row lanes, fixed order.
// part: [chunks][C]; the caller folds it (adell_bias_grad-style) to [C] (and to one value
// for a single-parameter PReLU).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_prelu_wgrad_partial_kernel(NormActBwdArgs a, int chunk) {
  __shared__ float sh[4][64];
  const long rows = a.total / a.C;
  const long r0 = (long)blockIdx.x * chunk;
  long r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const float keep_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  float s = 0.f;
  if (c < a.C) {
    for (long r = r0 + vl; r < r1; r += 4) {
      const long n = r / a.V;
      const long e = r * a.C + c;
      float h = a.x[e];
      if (a.mean) {
        const long si = n * a.stat_stride_n + c;
        h = (h - a.mean[si]) * a.rstd[si];
      }
      float t = h;
      if (a.gamma) t *= a.gamma[c];
      if (a.beta) t += a.beta[c];
      bool keep[4];
      adell_na_keep4(a, e >> 2, keep);
      const float u = keep[e & 3] ? t * keep_scale : 0.f;
      if (u < 0.f) s += a.dout[e] * u;
    }
  }
  sh[vl][cl] = s;
  __syncthreads();
  if (vl == 0 && c < a.C)
    a.part[(size_t)blockIdx.x * a.C + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

__global__ __launch_bounds__(1024) void adell_prelu_wgrad_final_kernel(
    const float* __restrict__ part, int nb, int C, int single, float* __restrict__ out) {
  __shared__ double sh[16][64];
  __shared__ double tot[64];
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  double grand = 0.0;
  for (int cb = 0; cb < C; cb += 64) {   // one block walks every channel group (C <= ~1k)
    const int c = cb + cl;
    double s = 0.0;
    if (c < C)
      for (int b = vl; b < nb; b += 16) s += (double)part[(size_t)b * C + c];
    sh[vl][cl] = s;
    __syncthreads();
    if (vl == 0) {
      s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += sh[k][cl];
      if (c < C && !single) out[c] = (float)s;
      tot[cl] = c < C ? s : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < 64; ++k) grand += tot[k];
    __syncthreads();
  }
  if (single && threadIdx.x == 0) out[0] = (float)grand;
}

static int adell_prelu_chunk(long rows) {
  long chunk = (rows + 1023) / 1024;
  if (chunk < 16) chunk = 16;
  return (int)((chunk + 3) / 4 * 4);
}

extern "C" long adell_prelu_wgrad_workspace(const adell_norm_act_desc* d) {
  if (!d || d->N <= 0 || d->V <= 0 || d->C <= 0) return ADELL_E_BADARG;
  const long rows = d->N * d->V;
  const int chunk = adell_prelu_chunk(rows);
  return (long)sizeof(float) * ((rows + chunk - 1) / chunk) * d->C;
}

// dact_w: [act_w_n] (1 or C). Same descriptor / operands as adell_norm_act_bwd.
extern "C" int adell_prelu_wgrad(const adell_norm_act_desc* d, const float* x, const float* dout,
                                 const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, float* dact_w, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  NormActBwdArgs a = {};
  int rc = adell_nab_fill(&a, d);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && dout && dact_w && workspace, "prelu_wgrad: null pointer");
  ADELL_REQUIRE(d->act == ADELL_ACT_PRELU && (d->act_w_n == 1 || d->act_w_n == d->C),
                "prelu_wgrad: needs a PReLU with 1 or C parameters");
  ADELL_REQUIRE((long)workspace_bytes >= adell_prelu_wgrad_workspace(d),
                "prelu_wgrad: workspace too small");
  a.x = x; a.dout = dout; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.part = (float*)workspace;
  const long rows = d->N * d->V;
  const int chunk = adell_prelu_chunk(rows);
  const int nb = (int)((rows + chunk - 1) / chunk);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_prelu_wgrad_partial_kernel, dim3(nb, adell_cdiv(d->C, 64)), dim3(256), 0,
                     st, a, chunk);
  hipLaunchKernelGGL(adell_prelu_wgrad_final_kernel, dim3(1), dim3(1024), 0, st,
                     (const float*)workspace, nb, d->C, d->act_w_n == 1 ? 1 : 0, dact_w);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
