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
    fl += adell_powg(pc, a.gamma) * logf(pc) * tb +
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
      gf += tb * (a.gamma * pg * logf(pi) + pg);
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
                                    float smooth, float dice_eps, float gamma,
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
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_dice_focal_partials_kernel, dim3(a.nblk, B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(adell_dice_focal_finalize_kernel, dim3(B), dim3(64), 0, st,
                     (const float*)workspace, a.nblk, S, sums, dice, focal);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// dprob = gdice * d dice_b/dp + gfocal * d focal_b/dp  (per item b).
extern "C" int adell_dice_focal_bwd(const float* prob, const float* target, int B, long S,
                                    float smooth, float dice_eps, float gamma,
                                    float focal_eps, const float* sums, float gdice,
                                    float gfocal, float* dprob, void* stream) {
  ADELL_REQUIRE(prob && target && sums && dprob, "dice_focal_bwd: null pointer");
  ADELL_REQUIRE(B > 0 && S > 0, "dice_focal_bwd: bad dims");
  LossArgs a = {};
  a.p = prob; a.t = target; a.sums = sums; a.dp = dprob; a.S = S;
  a.smooth = smooth; a.dice_eps = dice_eps; a.gamma = gamma; a.focal_eps = focal_eps;
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
                                        float focal_eps, const float* sums, const float* gdice,
                                        const float* gfocal, float* dprob, void* stream) {
  ADELL_REQUIRE(prob && target && sums && dprob, "dice_focal_bwd: null pointer");
  ADELL_REQUIRE(B > 0 && S > 0, "dice_focal_bwd: bad dims");
  LossArgs a = {};
  a.p = prob; a.t = target; a.sums = sums; a.dp = dprob; a.S = S;
  a.smooth = smooth; a.dice_eps = dice_eps; a.gamma = gamma; a.focal_eps = focal_eps;
  a.gdice = 0.f; a.gfocal = 0.f; a.gdice_dev = gdice; a.gfocal_dev = gfocal;
  long blocks = (S + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adell_dice_focal_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0,
                     (hipStream_t)stream, a);
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
