// Kernels of the ConvNeXt / VICReg self-supervised path (BASELINE config 4):
//   * depthwise 3-D convolution (torch.nn.Conv3d(groups=C), adell_mri/modules/layers/
//     res_blocks.py:552-558) forward, backward-data, backward-weight -- an HBM/L2-bound
//     stencil, channels innermost (NDHWC) so every tap is one coalesced float4;
//   * per-channel scale (the layer-scale gamma of the ConvNeXt block) lives on the
//     norm_act kernel; the residual add on add_bcast;
//   * VICReg loss (adell_mri/modules/self_supervised/losses/vicreg.py:30-165) on two
//     [B, D] embeddings and its gradient. The covariance term never forms the D x D
//     matrix: sum_{i!=j} cov_ij^2 = ||Xc Xc^T||_F^2 / (B-1)^2 - sum_i var_i^2 with the
//     B x B Gram matrix, and its gradient is 4 G Xc / (B-1)^2 - 4 var_j xc / (B-1).
#include "common.h"

struct DwArgs {
  const float* x;
  const float* w;   // canonical [C][taps]
  const float* b;   // [C] or null
  float* y;
  int N, C, D, H, W, KD, KH, KW, PD, PH, PW, flip;
};

// y[v][c] = sum_tap x[v + tap - p][c] * w[c][tap (flipped when a.flip)] (+ b[c])
__global__ __launch_bounds__(256) void adell_dwconv3d_kernel(DwArgs a) {
  const int taps = a.KD * a.KH * a.KW;
  const int C4 = a.C >> 2;
  const bool vec = (a.C & 3) == 0;
  const int CW = vec ? C4 : a.C;
  const long total = (long)a.N * a.D * a.H * a.W * CW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int cw = (int)(i % CW);
    long v = i / CW;
    const int x0 = (int)(v % a.W); v /= a.W;
    const int y0 = (int)(v % a.H); v /= a.H;
    const int z0 = (int)(v % a.D);
    const int nb = (int)(v / a.D);
    const int c = vec ? cw * 4 : cw;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.b) {
      acc[0] = a.b[c];
      if (vec) { acc[1] = a.b[c + 1]; acc[2] = a.b[c + 2]; acc[3] = a.b[c + 3]; }
    }
    int tap = 0;
    for (int kz = 0; kz < a.KD; ++kz) {
      const int z = z0 + kz - a.PD;
      for (int ky = 0; ky < a.KH; ++ky) {
        const int y = y0 + ky - a.PH;
        for (int kx = 0; kx < a.KW; ++kx, ++tap) {
          const int x = x0 + kx - a.PW;
          if (z < 0 || z >= a.D || y < 0 || y >= a.H || x < 0 || x >= a.W) continue;
          const int wt = a.flip ? taps - 1 - tap : tap;
          const size_t gv = ((((size_t)nb * a.D + z) * a.H + y) * a.W + x) * a.C + c;
          if (vec) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + gv);
            acc[0] += xv.x * a.w[(size_t)(c + 0) * taps + wt];
            acc[1] += xv.y * a.w[(size_t)(c + 1) * taps + wt];
            acc[2] += xv.z * a.w[(size_t)(c + 2) * taps + wt];
            acc[3] += xv.w * a.w[(size_t)(c + 3) * taps + wt];
          } else {
            acc[0] += a.x[gv] * a.w[(size_t)c * taps + wt];
          }
        }
      }
    }
    const size_t ov = ((((size_t)nb * a.D + z0) * a.H + y0) * a.W + x0) * a.C + c;
    if (vec)
      *reinterpret_cast<float4*>(a.y + ov) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else
      a.y[ov] = acc[0];
  }
}

static int adell_dw_check(int N, int C, int D, int H, int W, int KD, int KH, int KW) {
  ADELL_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, "dwconv: bad dims");
  ADELL_REQUIRE(KD >= 1 && KH >= 1 && KW >= 1 && (KD & 1) && (KH & 1) && (KW & 1),
                "dwconv: odd kernel sizes ('same' padding) only");
  return ADELL_OK;
}

static int adell_dw_launch(DwArgs a, hipStream_t st) {
  const long total = (long)a.N * a.D * a.H * a.W * (((a.C & 3) == 0) ? a.C / 4 : a.C);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_dwconv3d_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Depthwise Conv3d, stride 1, padding k//2 ("same"). w: torch layout [C][1][KD][KH][KW].
extern "C" int adell_dwconv3d_fwd(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                  const float* x, const float* w, const float* bias, float* y,
                                  void* stream) {
  int rc = adell_dw_check(N, C, D, H, W, KD, KH, KW);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && w && y, "dwconv_fwd: null pointer");
  DwArgs a = {x, w, bias, y, N, C, D, H, W, KD, KH, KW, KD / 2, KH / 2, KW / 2, 0};
  return adell_dw_launch(a, (hipStream_t)stream);
}

extern "C" int adell_dwconv3d_bwd_data(int N, int C, int D, int H, int W, int KD, int KH,
                                       int KW, const float* dy, const float* w, float* dx,
                                       void* stream) {
  int rc = adell_dw_check(N, C, D, H, W, KD, KH, KW);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(dy && w && dx, "dwconv_bwd_data: null pointer");
  DwArgs a = {dy, w, nullptr, dx, N, C, D, H, W, KD, KH, KW, KD / 2, KH / 2, KW / 2, 1};
  return adell_dw_launch(a, (hipStream_t)stream);
}

// dw[c][tap] = sum_v x[v + tap - p][c] * dy[v][c];  db[c] = sum_v dy[v][c] (tap == centre
// block also reduces db). grid (taps, channel groups of 64); block = 64 channels x 4 lanes.
__global__ __launch_bounds__(256) void adell_dwconv3d_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
    float* __restrict__ db, int N, int C, int D, int H, int W, int KD, int KH, int KW) {
  __shared__ float sh[4][64][2];
  const int taps = KD * KH * KW;
  const int tap = blockIdx.x;
  const int kx = tap % KW, ky = (tap / KW) % KH, kz = tap / (KW * KH);
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const int dz = kz - KD / 2, dyy = ky - KH / 2, dx = kx - KW / 2;
  float s = 0.f, sb = 0.f;
  if (c < C) {
    const long V = (long)N * D * H * W;
    for (long v = vl; v < V; v += 4) {
      long t = v;
      const int x0 = (int)(t % W); t /= W;
      const int y0 = (int)(t % H); t /= H;
      const int z0 = (int)(t % D);
      const int nb = (int)(t / D);
      const float g = dy[v * C + c];
      sb += g;
      const int xx = x0 + dx, yy = y0 + dyy, zz = z0 + dz;
      if (xx < 0 || xx >= W || yy < 0 || yy >= H || zz < 0 || zz >= D) continue;
      s += g * x[((((size_t)nb * D + zz) * H + yy) * W + xx) * C + c];
    }
  }
  sh[vl][cl][0] = s;
  sh[vl][cl][1] = sb;
  __syncthreads();
  if (vl == 0 && c < C) {
    dw[(size_t)c * taps + tap] = (sh[0][cl][0] + sh[1][cl][0]) + (sh[2][cl][0] + sh[3][cl][0]);
    if (db && tap == 0) db[c] = (sh[0][cl][1] + sh[1][cl][1]) + (sh[2][cl][1] + sh[3][cl][1]);
  }
}

extern "C" int adell_dwconv3d_bwd_weight(int N, int C, int D, int H, int W, int KD, int KH,
                                         int KW, const float* x, const float* dy, float* dw,
                                         float* db, void* stream) {
  int rc = adell_dw_check(N, C, D, H, W, KD, KH, KW);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && dy && dw, "dwconv_bwd_weight: null pointer");
  hipLaunchKernelGGL(adell_dwconv3d_wgrad_kernel, dim3(KD * KH * KW, adell_cdiv(C, 64)), dim3(256),
                     0, (hipStream_t)stream, x, dy, dw, db, N, C, D, H, W, KD, KH, KW);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// VICReg. scratch layout (floats): mean1[D] var1[D] mean2[D] var2[D] G1[B*B] G2[B*B].
// out[3] = (inv, var, cov) unweighted: inv = sum (x1-x2)^2 / (B*D);
// var = (hinge(X1) + hinge(X2)) / 2; cov = (cov(X1) + cov(X2)) / 2.
// One block; fixed-order reductions.
// ---------------------------------------------------------------------------
__device__ float adell_block_sum(float v, float* sh) {
  v = adell_wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(1024) void adell_vicreg_fwd_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, int B, int D, float min_var,
    float eps, float* __restrict__ scratch, float* __restrict__ out) {
  __shared__ float sh[16];
  const float* xs[2] = {x1, x2};
  float hinge = 0.f, var2 = 0.f, g2 = 0.f, inv = 0.f;
  for (int view = 0; view < 2; ++view) {
    const float* x = xs[view];
    float* mean = scratch + view * 2 * D;
    float* var = mean + D;
    float* G = scratch + 4 * D + (size_t)view * B * B;
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
      float m = 0.f;
      for (int b = 0; b < B; ++b) m += x[(size_t)b * D + j];
      m /= (float)B;
      float v = 0.f;
      for (int b = 0; b < B; ++b) {
        const float d = x[(size_t)b * D + j] - m;
        v += d * d;
      }
      v /= (float)(B - 1);
      mean[j] = m;
      var[j] = v;
      const float s = sqrtf(v + eps);
      hinge += fmaxf(min_var - s, 0.f);
      var2 += v * v;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < B * B; p += blockDim.x) {
      const int a = p / B, b = p - a * B;
      float g = 0.f;
      for (int j = 0; j < D; ++j)
        g += (x[(size_t)a * D + j] - mean[j]) * (x[(size_t)b * D + j] - mean[j]);
      G[p] = g;
      g2 += g * g;
    }
    __syncthreads();
  }
  for (long i = threadIdx.x; i < (long)B * D; i += blockDim.x) {
    const float d = x1[i] - x2[i];
    inv += d * d;
  }
  hinge = adell_block_sum(hinge, sh);
  var2 = adell_block_sum(var2, sh);
  g2 = adell_block_sum(g2, sh);
  inv = adell_block_sum(inv, sh);
  if (threadIdx.x == 0) {
    const float bm1 = (float)(B - 1);
    out[0] = inv / ((float)B * (float)D);
    out[1] = 0.5f * hinge / (float)D;
    out[2] = 0.5f * (g2 / (bm1 * bm1) - var2) / (float)D;
  }
}

// dX_v = g_inv * d inv/dX_v + g_var * d var/dX_v + g_cov * d cov/dX_v
__global__ __launch_bounds__(256) void adell_vicreg_bwd_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, int B, int D, float min_var,
    float eps, const float* __restrict__ scratch, const float* __restrict__ g3,
    float* __restrict__ dx1, float* __restrict__ dx2) {
  const float g_inv = g3[0], g_var = g3[1], g_cov = g3[2];
  const long total = (long)B * D;
  const float bm1 = (float)(B - 1);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int b = (int)(i / D), j = (int)(i - (long)b * D);
    const float dinv = 2.f * (x1[i] - x2[i]) / ((float)B * (float)D);
    for (int view = 0; view < 2; ++view) {
      const float* x = view == 0 ? x1 : x2;
      float* dx = view == 0 ? dx1 : dx2;
      if (!dx) continue;
      const float* mean = scratch + view * 2 * D;
      const float* var = mean + D;
      const float* G = scratch + 4 * D + (size_t)view * B * B;
      const float xc = x[i] - mean[j];
      const float s = sqrtf(var[j] + eps);
      float gv = 0.f;
      if (min_var > s) gv = -0.5f * xc / (s * bm1) / (float)D;  // 0.5: mean over the two views
      float gx = 0.f;
      for (int a = 0; a < B; ++a) gx += G[(size_t)b * B + a] * (x[(size_t)a * D + j] - mean[j]);
      const float gc = 0.5f * 4.f * (gx / (bm1 * bm1) - var[j] * xc / bm1) / (float)D;
      dx[i] = g_inv * (view == 0 ? dinv : -dinv) + g_var * gv + g_cov * gc;
    }
  }
}

extern "C" long adell_vicreg_scratch_floats(int B, int D) { return 4L * D + 2L * B * B; }

extern "C" int adell_vicreg_fwd(const float* x1, const float* x2, int B, int D, float min_var,
                                float eps, float* scratch, float* out3, void* stream) {
  ADELL_REQUIRE(x1 && x2 && scratch && out3, "vicreg_fwd: null pointer");
  ADELL_REQUIRE(B > 1 && D > 0, "vicreg_fwd: need B > 1, D > 0");
  hipLaunchKernelGGL(adell_vicreg_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x1, x2, B,
                     D, min_var, eps, scratch, out3);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_vicreg_bwd(const float* x1, const float* x2, int B, int D, float min_var,
                                float eps, const float* scratch, const float* g3,
                                float* dx1, float* dx2, void* stream) {
  ADELL_REQUIRE(x1 && x2 && scratch && g3 && (dx1 || dx2), "vicreg_bwd: null pointer");
  ADELL_REQUIRE(B > 1 && D > 0, "vicreg_bwd: need B > 1, D > 0");
  long blocks = ((long)B * D + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_vicreg_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x1, x2, B, D, min_var, eps, scratch, g3, dx1, dx2);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
