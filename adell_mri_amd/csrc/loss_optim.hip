// Fused segmentation loss (binary generalised dice + binary focal on
// probabilities; reference adell_mri/modules/segmentation/losses.py:14-54,
// 112-164, 251-292 combined as in segmentation/pl.py:218-222) and the fused
// optimiser / EMA updates over flat fp32 parameter buffers (reference:
// torch.optim.SGD(momentum, nesterov, weight_decay) at segmentation/pl.py:563-569,
// torch.optim.AdamW at self_supervised/pl.py:245-250). All HBM-bound.
#include "common.h"

#define ADELL_LOSS_SLAB 8192  // elements per block

struct LossArgs {
  const float* p;  // probabilities [B][S]
  const float* t;  // targets [B][S]
  float* part;     // [B][nblk][3]
  const float* sums;  // [B][3] (num, den, fl) for the backward
  float falpha;       // weight of the positive-class term of the focal loss (losses.py:112-164 `alpha`)
  float* dp;
  long S;
  int nblk;
  float smooth, dice_eps, gamma, focal_eps;
  float gdice, gfocal;  // d(total)/d(dice_b), d(total)/d(focal_b)
  const float* gdice_dev;   // optional per-item device arrays [B] (override the scalars)
  const float* gfocal_dev;
};

__device__ __forceinline__ float adell_powg(float x, float g) {
  return g == 1.0f ? x : (g == 2.0f ? x * x : powf(x, g));
}

__global__ __launch_bounds__(256) void adell_dice_focal_partials_kernel(LossArgs a) {
  __shared__ float sh[4][3];
  const int b = blockIdx.y, blk = blockIdx.x;
  const long i0 = (long)blk * ADELL_LOSS_SLAB;
  long i1 = i0 + ADELL_LOSS_SLAB;
  if (i1 > a.S) i1 = a.S;
  const float* p = a.p + (size_t)b * a.S;
  const float* t = a.t + (size_t)b * a.S;
  float num = 0.f, den = 0.f, fl = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float pi = p[i], ti = t[i];
    num += fmaxf(ti * pi, 0.f);
    den += fmaxf(ti + pi + a.smooth, a.dice_eps);
    const float pc = fmaxf(pi, a.focal_eps);
    const float qc = fmaxf(1.0f - pc, a.focal_eps);
    const float tb = ti > 0.5f ? 1.f : 0.f;
    fl += a.falpha * adell_powg(pc, a.gamma) * logf(pc) * tb +
          adell_powg(qc, a.gamma) * logf(qc) * (1.f - tb);
  }
  num = adell_wave_sum(num);
  den = adell_wave_sum(den);
  fl = adell_wave_sum(fl);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w][0] = num;
    sh[w][1] = den;
    sh[w][2] = fl;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float s = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] +
                    sh[3][threadIdx.x];
    a.part[((size_t)b * a.nblk + blk) * 3 + threadIdx.x] = s;
  }
}

__global__ void adell_dice_focal_finalize_kernel(const float* __restrict__ part, int nblk,
                                                 long S, float* __restrict__ sums,
                                                 float* __restrict__ dice,
                                                 float* __restrict__ focal) {
  const int b = blockIdx.x;
  __shared__ double sh[64][3];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int k = threadIdx.x; k < nblk; k += 64)
    for (int j = 0; j < 3; ++j) acc[j] += (double)part[((size_t)b * nblk + k) * 3 + j];
  for (int j = 0; j < 3; ++j) sh[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (threadIdx.x == 0) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 64; ++k)
      for (int j = 0; j < 3; ++j) s[j] += sh[k][j];
    sums[b * 3 + 0] = (float)s[0];
    sums[b * 3 + 1] = (float)s[1];
    sums[b * 3 + 2] = (float)s[2];
    dice[b] = (float)(1.0 - 2.0 * s[0] / s[1]);
    focal[b] = (float)(-s[2] / (double)S);
  }
}

__global__ __launch_bounds__(256) void adell_dice_focal_bwd_kernel(LossArgs a) {
  const int b = blockIdx.y;
  const float* p = a.p + (size_t)b * a.S;
  const float* t = a.t + (size_t)b * a.S;
  float* dp = a.dp + (size_t)b * a.S;
  const float num = a.sums[b * 3 + 0], den = a.sums[b * 3 + 1];
  const float inv_den2 = 1.0f / (den * den);
  const float invS = 1.0f / (float)a.S;
  const float wd = a.gdice_dev ? a.gdice_dev[b] : a.gdice;
  const float wf = a.gfocal_dev ? a.gfocal_dev[b] : a.gfocal;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < a.S; i += (long)gridDim.x * 256L) {
    const float pi = p[i], ti = t[i];
    const float dnum = (ti * pi > 0.f) ? ti : 0.f;
    const float dden = (ti + pi + a.smooth > a.dice_eps) ? 1.f : 0.f;
    const float gd = -2.0f * (dnum * den - num * dden) * inv_den2;
    const float tb = ti > 0.5f ? 1.f : 0.f;
    float gf = 0.f;
    if (pi > a.focal_eps) {
      const float g1 = a.gamma - 1.0f;
      const float pg = g1 == 0.f ? 1.f : adell_powg(pi, g1);
      gf += a.falpha * tb * (a.gamma * pg * logf(pi) + pg);
      const float q = 1.0f - pi;
      if (q > a.focal_eps) {
        const float qg = g1 == 0.f ? 1.f : adell_powg(q, g1);
        gf -= (1.f - tb) * (a.gamma * qg * logf(q) + qg);
      }
    }
    dp[i] = wd * gd + wf * (-gf * invS);
  }
}

extern "C" long adell_dice_focal_workspace(int B, long S) {
  const long nblk = (S + ADELL_LOSS_SLAB - 1) / ADELL_LOSS_SLAB;
  return (long)sizeof(float) * (B * nblk * 3);
}

// dice[B], focal[B] per-item losses; sums[B][3] is kept for the backward.
extern "C" int adell_dice_focal_fwd(const float* prob, const float* target, int B, long S,
                                    float smooth, float dice_eps, float gamma, float focal_alpha,
                                    float focal_eps, float* dice, float* focal, float* sums,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(prob && target && dice && focal && sums && workspace,
                "dice_focal_fwd: null pointer");
  ADELL_REQUIRE(B > 0 && S > 0, "dice_focal_fwd: bad dims");
  ADELL_REQUIRE((long)workspace_bytes >= adell_dice_focal_workspace(B, S),
                "dice_focal_fwd: workspace too small");
  LossArgs a = {};
  a.p = prob; a.t = target; a.part = (float*)workspace; a.S = S;
  a.nblk = (int)((S + ADELL_LOSS_SLAB - 1) / ADELL_LOSS_SLAB);
  a.smooth = smooth; a.dice_eps = dice_eps; a.gamma = gamma; a.focal_eps = focal_eps;
  a.falpha = focal_alpha;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_dice_focal_partials_kernel, dim3(a.nblk, B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(adell_dice_focal_finalize_kernel, dim3(B), dim3(64), 0, st,
                     (const float*)workspace, a.nblk, S, sums, dice, focal);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// dprob = gdice * d dice_b/dp + gfocal * d focal_b/dp  (per item b).
extern "C" int adell_dice_focal_bwd(const float* prob, const float* target, int B, long S,
                                    float smooth, float dice_eps, float gamma, float focal_alpha,
                                    float focal_eps, const float* sums, float gdice,
                                    float gfocal, float* dprob, void* stream) {
  ADELL_REQUIRE(prob && target && sums && dprob, "dice_focal_bwd: null pointer");
  ADELL_REQUIRE(B > 0 && S > 0, "dice_focal_bwd: bad dims");
  LossArgs a = {};
  a.p = prob; a.t = target; a.sums = sums; a.dp = dprob; a.S = S;
  a.smooth = smooth; a.dice_eps = dice_eps; a.gamma = gamma; a.focal_eps = focal_eps;
  a.falpha = focal_alpha;
  a.gdice = gdice; a.gfocal = gfocal;
  long blocks = (S + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adell_dice_focal_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Same with the upstream gradients of every item on the device (gdice[B], gfocal[B]; either
// may be NULL = 0): no host read-back of the autograd inputs.
extern "C" int adell_dice_focal_bwd_dev(const float* prob, const float* target, int B, long S,
                                        float smooth, float dice_eps, float gamma,
                                        float focal_alpha, float focal_eps, const float* sums,
                                        const float* gdice, const float* gfocal, float* dprob,
                                        void* stream) {
  ADELL_REQUIRE(prob && target && sums && dprob, "dice_focal_bwd: null pointer");
  ADELL_REQUIRE(B > 0 && S > 0, "dice_focal_bwd: bad dims");
  LossArgs a = {};
  a.p = prob; a.t = target; a.sums = sums; a.dp = dprob; a.S = S;
  a.smooth = smooth; a.dice_eps = dice_eps; a.gamma = gamma; a.focal_eps = focal_eps;
  a.falpha = focal_alpha;
  a.gdice = 0.f; a.gfocal = 0.f; a.gdice_dev = gdice; a.gfocal_dev = gfocal;
  long blocks = (S + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adell_dice_focal_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Per-(item, class) sums over the voxels of probabilities p and targets t, [B][V][C]:
//   sums[b][c] = (sum p t, sum p, sum t)
// -- what the Tversky-type losses are made of (losses.py:295-337, 656-698: tp = sum p t,
// "fn" = sum p (1 - t) = sum p - sum p t, "fp" = sum (1 - p) t = sum t - sum p t). Deterministic:
// per-slab partials, fixed-order fp64 fold. Backward: dp[b][v][c] = g[b][c][0] t + g[b][c][1].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_class_sums_partials_kernel(
    const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ part, long V,
    int C, int nblk) {
  __shared__ float sh[4][3];
  const int blk = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const long i0 = (long)blk * ADELL_LOSS_SLAB;
  long i1 = i0 + ADELL_LOSS_SLAB;
  if (i1 > V) i1 = V;
  const float* pb = p + (size_t)b * V * C + c;
  const float* tb = t + (size_t)b * V * C + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float pi = pb[i * C], ti = tb[i * C];
    s0 += pi * ti;
    s1 += pi;
    s2 += ti;
  }
  s0 = adell_wave_sum(s0);
  s1 = adell_wave_sum(s1);
  s2 = adell_wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[w][0] = s0; sh[w][1] = s1; sh[w][2] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    part[(((size_t)b * C + c) * nblk + blk) * 3 + threadIdx.x] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void adell_class_sums_finalize_kernel(const float* __restrict__ part, int nblk,
                                                 float* __restrict__ sums) {
  const int bc = blockIdx.x;
  __shared__ double sh[64][3];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int k = threadIdx.x; k < nblk; k += 64)
    for (int j = 0; j < 3; ++j) acc[j] += (double)part[((size_t)bc * nblk + k) * 3 + j];
  for (int j = 0; j < 3; ++j) sh[threadIdx.x][j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int k = 0; k < 64; ++k) s += sh[k][threadIdx.x];
    sums[bc * 3 + threadIdx.x] = (float)s;
  }
}

__global__ __launch_bounds__(256) void adell_class_sums_bwd_kernel(
    const float* __restrict__ t, const float* __restrict__ g, float* __restrict__ dp, long VC,
    int C) {
  const int b = blockIdx.y;
  const float* tb = t + (size_t)b * VC;
  float* db = dp + (size_t)b * VC;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < VC; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % C);
    const float* gc = g + ((size_t)b * C + c) * 3;
    db[e] = gc[0] * tb[e] + gc[1];
  }
}

extern "C" long adell_class_sums_workspace(int B, long V, int C) {
  const long nblk = (V + ADELL_LOSS_SLAB - 1) / ADELL_LOSS_SLAB;
  return (long)sizeof(float) * B * C * nblk * 3;
}

extern "C" int adell_class_sums_fwd(const float* p, const float* t, int B, long V, int C,
                                    float* sums, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  ADELL_REQUIRE(p && t && sums && workspace, "class_sums_fwd: null pointer");
  ADELL_REQUIRE(B > 0 && B <= 65535 && V > 0 && C > 0 && C <= 65535, "class_sums_fwd: bad dims");
  ADELL_REQUIRE((long)workspace_bytes >= adell_class_sums_workspace(B, V, C),
                "class_sums_fwd: workspace too small");
  const int nblk = (int)((V + ADELL_LOSS_SLAB - 1) / ADELL_LOSS_SLAB);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_class_sums_partials_kernel, dim3(nblk, B, C), dim3(256), 0, st, p, t,
                     (float*)workspace, V, C, nblk);
  hipLaunchKernelGGL(adell_class_sums_finalize_kernel, dim3(B * C), dim3(64), 0, st,
                     (const float*)workspace, nblk, sums);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_class_sums_bwd(const float* t, const float* gsums, int B, long V, int C,
                                    float* dp, void* stream) {
  ADELL_REQUIRE(t && gsums && dp, "class_sums_bwd: null pointer");
  ADELL_REQUIRE(B > 0 && B <= 65535 && V > 0 && C > 0, "class_sums_bwd: bad dims");
  long blocks = (V * C + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_class_sums_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0,
                     (hipStream_t)stream, t, gsums, dp, V * C, C);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Optimisers over flat buffers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_sgd_kernel(float* __restrict__ p,
                                                        const float* __restrict__ g,
                                                        float* __restrict__ buf, long n,
                                                        float lr, float momentum, float wd,
                                                        int nesterov, int first,
                                                        float grad_scale) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (momentum != 0.f && !first) bv = reinterpret_cast<float4*>(buf)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w};
    const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = gg[j] * grad_scale + wd * pp[j];
      if (momentum != 0.f) {
        bb[j] = first ? d : momentum * bb[j] + d;
        d = nesterov ? d + momentum * bb[j] : bb[j];
      }
      pp[j] -= lr * d;
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    if (momentum != 0.f) reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n;
       i += (long)gridDim.x * 256L) {
    float d = g[i] * grad_scale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= lr * d;
  }
}

// torch.optim.SGD semantics (dampening 0). grad_scale multiplies the gradient
// first (1/world_size after a sum all-reduce, or 1/accumulate_grad_batches).
extern "C" int adell_sgd_step(float* param, const float* grad, float* momentum_buf, long n,
                              float lr, float momentum, float weight_decay, int nesterov,
                              int first_step, float grad_scale, void* stream) {
  ADELL_REQUIRE(param && grad, "sgd_step: null pointer");
  ADELL_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step: momentum needs a buffer");
  ADELL_REQUIRE(n > 0, "sgd_step: empty");
  ADELL_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0,
                "sgd_step: buffers must be 16-byte aligned");
  long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adell_sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     param, grad, momentum_buf, n, lr, momentum, weight_decay, nesterov,
                     first_step, grad_scale);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// DECOUPLED: AdamW (p *= 1 - lr * wd); otherwise Adam (the decay enters the gradient: g += wd * p)
template <bool DECOUPLED>
__global__ __launch_bounds__(256) void adell_adamw_kernel(float* __restrict__ p,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ m,
                                                          float* __restrict__ v, long n, float lr,
                                                          float b1, float b2, float eps, float wd,
                                                          float bc1, float bc2_sqrt,
                                                          float grad_scale) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    float gi = g[i] * grad_scale;
    float pi = p[i];
    if (DECOUPLED)
      pi *= 1.0f - lr * wd;
    else
      gi += wd * pi;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

static int adell_adam_launch(bool decoupled, float* param, const float* grad, float* exp_avg,
                             float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, long step, float grad_scale,
                             void* stream) {
  ADELL_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam(w)_step: null pointer");
  ADELL_REQUIRE(n > 0 && step >= 1, "adam(w)_step: bad n / step");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (decoupled)
    hipLaunchKernelGGL(adell_adamw_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2s, grad_scale);
  else
    hipLaunchKernelGGL(adell_adamw_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2s, grad_scale);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// torch.optim.AdamW (amsgrad off), step = 1-based step count.
extern "C" int adell_adamw_step(float* param, const float* grad, float* exp_avg,
                                float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, long step, float grad_scale,
                                void* stream) {
  return adell_adam_launch(true, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                           weight_decay, step, grad_scale, stream);
}

// torch.optim.Adam (amsgrad off): L2 weight decay added to the gradient.
extern "C" int adell_adam_step(float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, long step, float grad_scale,
                               void* stream) {
  return adell_adam_launch(false, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                           weight_decay, step, grad_scale, stream);
}

__global__ __launch_bounds__(256) void adell_ema_kernel(float* __restrict__ shadow,
                                                        const float* __restrict__ p, long n,
                                                        float one_minus_decay) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
    shadow[i] -= one_minus_decay * (shadow[i] - p[i]);
}

// shadow.sub_((1 - decay) * (shadow - param))  (adell_mri/utils/utils.py:447-493)
extern "C" int adell_ema_update(float* shadow, const float* param, long n, float decay,
                                void* stream) {
  ADELL_REQUIRE(shadow && param && n > 0, "ema_update: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     shadow, param, n, 1.0f - decay);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------------------------
// Element-wise segmentation losses on probabilities p[B][V][C] (NDHWC memory; C = 1 for the
// binary family) against targets of the same layout -- the members of the reference's
// loss_factory (adell_mri/utils/utils.py:39-59) beyond the fused binary dice + focal pair:
//   kind 0  binary_cross_entropy   (losses.py:79-109)
//   kind 1  cat_cross_entropy      (losses.py:528-562)   target' = t (1 - ls) + 1 / C   [sic]
//   kind 2  mc_focal_loss          (losses.py:565-607)
//   kind 3  mc_generalized_dice_loss (losses.py:610-653, generalised_dice_score :14-54)
// Forward: per-(item, block, class) partial sums (S0, S1) -> fixed-order fp64 fold -> loss[B] and
// sums[B][C][2] (kept for the backward). cw[C] = class weights / alpha (1-vectors are broadcast
// by the caller). HBM-bound: one read of (p, t); the backward reads (p, t) and writes dp.
// ---------------------------------------------------------------------------------------------
struct SegLossArgs {
  const float* p;
  const float* t;
  const float* cw;     // [C]
  float* part;         // [B][nblk][C][2]
  const float* sums;   // [B][C][2]
  const float* gout;   // [B] upstream gradient (backward)
  float* dp;
  long V;
  int C, nblk, kind;
  float eps, scale, ls, gamma, smooth, w_pos;
};

#define ADELL_SEGLOSS_ROWS 2048   // voxels per block

__device__ __forceinline__ void adell_segloss_terms(const SegLossArgs& a, float p, float t, int c,
                                                    float* s0, float* s1) {
  switch (a.kind) {
    case 0: {
      const float tt = t * (1.f - a.ls) + 0.5f * a.ls;
      *s0 = (a.w_pos * tt * logf(p + a.eps) + (1.f - tt) * logf(1.f - p + a.eps)) * a.scale;
      *s1 = 0.f;
      break;
    }
    case 1: {
      const float tt = t * (1.f - a.ls) + 1.f / (float)a.C;
      *s0 = -tt * logf(p + a.eps) * a.cw[c] * a.scale;
      *s1 = 0.f;
      break;
    }
    case 2: {
      const float pt = t > 0.5f ? p : 1.f - p;
      const float tt = t * (1.f - a.ls) + 1.f / (float)a.C;
      const float ce = -tt * logf(p + a.eps);
      *s0 = a.cw[c] * powf(1.f - pt + a.eps, a.gamma) * ce * a.scale;
      *s1 = 0.f;
      break;
    }
    default: {
      *s0 = fmaxf(t * p * a.scale, 0.f);
      *s1 = fmaxf((t + p + a.smooth) * a.scale, a.eps);
      break;
    }
  }
}

__global__ __launch_bounds__(256) void adell_segloss_partials_kernel(SegLossArgs a) {
  extern __shared__ float sh[];   // [256][2] staging for the per-class fold
  const int b = blockIdx.y, blk = blockIdx.x;
  const long v0 = (long)blk * ADELL_SEGLOSS_ROWS;
  long v1 = v0 + ADELL_SEGLOSS_ROWS;
  if (v1 > a.V) v1 = a.V;
  const float* p = a.p + (size_t)b * a.V * a.C;
  const float* t = a.t + (size_t)b * a.V * a.C;
  const long e0 = v0 * a.C, e1 = v1 * a.C;
  // thread tid visits elements tid, tid + 256, ...: its class is constant iff 256 % C == 0;
  // otherwise accumulate per class in a short loop
  for (int c = 0; c < a.C; ++c) {
    float s0 = 0.f, s1 = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      if ((int)(e % a.C) != c) continue;
      float u0, u1;
      adell_segloss_terms(a, p[e], t[e], c, &u0, &u1);
      s0 += u0;
      s1 += u1;
    }
    s0 = adell_wave_sum(s0);
    s1 = adell_wave_sum(s1);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      sh[(threadIdx.x >> 6) * 2 + 0] = s0;
      sh[(threadIdx.x >> 6) * 2 + 1] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2)
      a.part[(((size_t)b * a.nblk + blk) * a.C + c) * 2 + threadIdx.x] =
          (sh[threadIdx.x] + sh[2 + threadIdx.x]) + (sh[4 + threadIdx.x] + sh[6 + threadIdx.x]);
  }
}

__global__ void adell_segloss_finalize_kernel(SegLossArgs a, float* __restrict__ sums,
                                              float* __restrict__ loss) {
  const int b = blockIdx.x;
  __shared__ double acc[32][2];
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < a.nblk; ++k) {
      s0 += (double)a.part[(((size_t)b * a.nblk + k) * a.C + c) * 2 + 0];
      s1 += (double)a.part[(((size_t)b * a.nblk + k) * a.C + c) * 2 + 1];
    }
    sums[((size_t)b * a.C + c) * 2 + 0] = (float)s0;
    sums[((size_t)b * a.C + c) * 2 + 1] = (float)s1;
    acc[c][0] = s0;
    acc[c][1] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)a.V * a.C;
    double out;
    if (a.kind == 3) {
      double num = 0.0, den = 0.0;
      for (int c = 0; c < a.C; ++c) {
        num += (double)a.cw[c] * acc[c][0];
        den += (double)a.cw[c] * acc[c][1];
      }
      out = 1.0 - 2.0 * num / den;
    } else {
      double s = 0.0;
      for (int c = 0; c < a.C; ++c) s += acc[c][0];
      out = (a.kind == 0 ? -s : s) / n;
    }
    loss[b] = (float)out;
  }
}

__global__ __launch_bounds__(256) void adell_segloss_bwd_kernel(SegLossArgs a) {
  const int b = blockIdx.y;
  const float* p = a.p + (size_t)b * a.V * a.C;
  const float* t = a.t + (size_t)b * a.V * a.C;
  float* dp = a.dp + (size_t)b * a.V * a.C;
  const long n = a.V * a.C;
  const float g = a.gout[b];
  float num = 0.f, den = 1.f;
  if (a.kind == 3) {
    num = den = 0.f;
    for (int c = 0; c < a.C; ++c) {
      num += a.cw[c] * a.sums[((size_t)b * a.C + c) * 2 + 0];
      den += a.cw[c] * a.sums[((size_t)b * a.C + c) * 2 + 1];
    }
  }
  const float inv_n = 1.0f / (float)n, inv_den2 = 1.0f / (den * den);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256L) {
    const int c = (int)(e % a.C);
    const float pe = p[e], te = t[e];
    float d;
    switch (a.kind) {
      case 0: {
        const float tt = te * (1.f - a.ls) + 0.5f * a.ls;
        d = -(a.w_pos * tt / (pe + a.eps) - (1.f - tt) / (1.f - pe + a.eps)) * a.scale * inv_n;
        break;
      }
      case 1: {
        const float tt = te * (1.f - a.ls) + 1.f / (float)a.C;
        d = -tt / (pe + a.eps) * a.cw[c] * a.scale * inv_n;
        break;
      }
      case 2: {
        const bool pos = te > 0.5f;
        const float q = 1.f - (pos ? pe : 1.f - pe) + a.eps;      // 1 - pt + eps
        const float dq = pos ? -1.f : 1.f;
        const float tt = te * (1.f - a.ls) + 1.f / (float)a.C;
        const float lg = logf(pe + a.eps);
        const float qg1 = powf(q, a.gamma - 1.f);
        // d/dp [ q^g * (-tt log(p + eps)) ]
        d = a.cw[c] * (-tt) * (a.gamma * qg1 * dq * lg + qg1 * q / (pe + a.eps)) * a.scale * inv_n;
        break;
      }
      default: {
        const float dnum = (te * pe * a.scale > 0.f) ? a.cw[c] * te * a.scale : 0.f;
        const float dden = ((te + pe + a.smooth) * a.scale > a.eps) ? a.cw[c] * a.scale : 0.f;
        d = -2.0f * (dnum * den - num * dden) * inv_den2;
        break;
      }
    }
    dp[e] = g * d;
  }
}

static int adell_segloss_fill(SegLossArgs* a, int kind, const float* p, const float* t,
                              const float* cw, int B, long V, int C, float eps, float scale,
                              float ls, float gamma, float smooth, float w_pos) {
  ADELL_REQUIRE(kind >= 0 && kind <= 3, "seg_loss: kind must be 0..3");
  ADELL_REQUIRE(p && t && B > 0 && B <= 65535 && V > 0 && C > 0 && C <= 32, "seg_loss: bad arguments");
  ADELL_REQUIRE(kind == 0 || cw, "seg_loss: class weights missing");
  *a = SegLossArgs{};
  a->p = p; a->t = t; a->cw = cw; a->V = V; a->C = C; a->kind = kind;
  a->nblk = (int)((V + ADELL_SEGLOSS_ROWS - 1) / ADELL_SEGLOSS_ROWS);
  a->eps = eps; a->scale = scale; a->ls = ls; a->gamma = gamma; a->smooth = smooth; a->w_pos = w_pos;
  return ADELL_OK;
}

extern "C" long adell_seg_loss_workspace(int B, long V, int C) {
  if (B <= 0 || V <= 0 || C <= 0) return ADELL_E_BADARG;
  return (long)sizeof(float) * B * ((V + ADELL_SEGLOSS_ROWS - 1) / ADELL_SEGLOSS_ROWS) * C * 2;
}

extern "C" int adell_seg_loss_fwd(int kind, const float* p, const float* t, const float* cw, int B,
                                  long V, int C, float eps, float scale, float label_smoothing,
                                  float gamma, float smooth, float w_pos, float* loss,
                                  float* sums, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  SegLossArgs a;
  int rc = adell_segloss_fill(&a, kind, p, t, cw, B, V, C, eps, scale, label_smoothing, gamma,
                              smooth, w_pos);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(loss && sums && workspace &&
                    (long)workspace_bytes >= adell_seg_loss_workspace(B, V, C),
                "seg_loss_fwd: null output or workspace too small");
  a.part = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_segloss_partials_kernel, dim3(a.nblk, B), dim3(256), 8 * sizeof(float),
                     st, a);
  hipLaunchKernelGGL(adell_segloss_finalize_kernel, dim3(B), dim3(32), 0, st, a, sums, loss);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_seg_loss_bwd(int kind, const float* p, const float* t, const float* cw, int B,
                                  long V, int C, float eps, float scale, float label_smoothing,
                                  float gamma, float smooth, float w_pos, const float* sums,
                                  const float* gout, float* dp, void* stream) {
  SegLossArgs a;
  int rc = adell_segloss_fill(&a, kind, p, t, cw, B, V, C, eps, scale, label_smoothing, gamma,
                              smooth, w_pos);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(sums && gout && dp, "seg_loss_bwd: null pointer");
  a.sums = sums; a.gout = gout; a.dp = dp;
  long blocks = (V * C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adell_segloss_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------------------------
// The other members of the reference's optimizer factory (adell_mri/utils/optimizer_factory.py:
// 5-14) over flat fp32 buffers, torch.optim single-tensor semantics with L2 weight decay added to
// the gradient; the per-step scalars (bias corrections, NAdam's mu products, RAdam's rectifier)
// are computed on the host and passed in c[]:
//   kind 0 Adamax   s1 = exp_avg, s2 = exp_inf;   c = {beta1, beta2, lr / (1 - beta1^t)}
//   kind 1 Adagrad  s1 = sum;                     c = {lr / (1 + (t - 1) lr_decay)}
//   kind 2 NAdam    s1 = exp_avg, s2 = exp_avg_sq; c = {beta1, beta2, 1 - beta2^t,
//                                                      lr (1 - mu) / (1 - mu_prod),
//                                                      lr mu_next / (1 - mu_prod mu_next)}
//   kind 3 RAdam    s1 = exp_avg, s2 = exp_avg_sq; c = {beta1, beta2, lr / (1 - beta1^t),
//                                                      rect sqrt(1 - beta2^t) or < 0: not rectified}
//   kind 4 RMSprop  s1 = square_avg;              c = {alpha, lr}      (momentum 0, not centered)
// ---------------------------------------------------------------------------------------------
struct OptimArgs {
  float* p;
  const float* g;
  float* s1;
  float* s2;
  long n;
  int kind;
  float wd, eps, grad_scale;
  float c[5];
};

__global__ __launch_bounds__(256) void adell_optim_kernel(OptimArgs a) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < a.n; i += (long)gridDim.x * 256L) {
    float pi = a.p[i];
    const float gi = a.g[i] * a.grad_scale + a.wd * pi;
    switch (a.kind) {
      case 0: {
        const float m = a.c[0] * a.s1[i] + (1.f - a.c[0]) * gi;
        const float u = fmaxf(a.c[1] * a.s2[i], fabsf(gi) + a.eps);
        a.s1[i] = m;
        a.s2[i] = u;
        pi -= a.c[2] * m / u;
        break;
      }
      case 1: {
        const float s = a.s1[i] + gi * gi;
        a.s1[i] = s;
        pi -= a.c[0] * gi / (sqrtf(s) + a.eps);
        break;
      }
      case 2: {
        const float m = a.c[0] * a.s1[i] + (1.f - a.c[0]) * gi;
        const float v = a.c[1] * a.s2[i] + (1.f - a.c[1]) * gi * gi;
        a.s1[i] = m;
        a.s2[i] = v;
        const float denom = sqrtf(v / a.c[2]) + a.eps;
        pi -= a.c[3] * gi / denom;
        pi -= a.c[4] * m / denom;
        break;
      }
      case 3: {
        const float m = a.c[0] * a.s1[i] + (1.f - a.c[0]) * gi;
        const float v = a.c[1] * a.s2[i] + (1.f - a.c[1]) * gi * gi;
        a.s1[i] = m;
        a.s2[i] = v;
        if (a.c[3] >= 0.f)
          pi -= a.c[2] * m * a.c[3] / (sqrtf(v) + a.eps);
        else
          pi -= a.c[2] * m;
        break;
      }
      default: {
        const float s = a.c[0] * a.s1[i] + (1.f - a.c[0]) * gi * gi;
        a.s1[i] = s;
        pi -= a.c[1] * gi / (sqrtf(s) + a.eps);
        break;
      }
    }
    a.p[i] = pi;
  }
}

extern "C" int adell_optim_step(int kind, float* param, const float* grad, float* state1,
                                float* state2, long n, float weight_decay, float eps,
                                float grad_scale, const float* c5, void* stream) {
  ADELL_REQUIRE(kind >= 0 && kind <= 4, "optim_step: kind must be 0..4");
  ADELL_REQUIRE(param && grad && state1 && c5 && n > 0, "optim_step: bad arguments");
  ADELL_REQUIRE(state2 || kind == 1 || kind == 4, "optim_step: this optimiser needs two state buffers");
  OptimArgs a;
  a.p = param; a.g = grad; a.s1 = state1; a.s2 = state2; a.n = n; a.kind = kind;
  a.wd = weight_decay; a.eps = eps; a.grad_scale = grad_scale;
  for (int i = 0; i < 5; ++i) a.c[i] = c5[i];   // host array
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_optim_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
