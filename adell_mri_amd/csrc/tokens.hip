// Token-sequence kernels of the ViT encoder in UNETR (reference:
// adell_mri/modules/layers/vit.py:844-1002, linear_blocks.py:358-417):
//   * LayerNorm over the last dimension (torch.nn.LayerNorm) forward/backward,
//   * broadcast add (X + positional_embedding) and its reduction backward,
//   * scaled-dot-product attention forward/backward with an optional additive
//     bias (F.scaled_dot_product_attention as called at linear_blocks.py:407-414).
// The Linear layers run on the conv kernel (a Linear is a 1x1x1 convolution over
// the token axis). Sequences are short (216 tokens at config 3): these kernels
// are latency/HBM-bound fp32 VALU code, one wave per row, wavefront-shuffle
// reductions, fixed summation order.
#include "common.h"

ADELL_RNG_STEP_DEFINE(tokens)

// ---------------------------------------------------------------------------
// LayerNorm: y = (x - mean) * rstd * gamma + beta over rows of length C.
// One wave per row. mean/rstd are saved for the backward.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adell_layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float m = adell_wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - m;
    v += d * d;
  }
  const float r = rsqrtf(adell_wave_sum(v) / (float)C + eps);
  float* yr = y + row * C;
  for (int c = lane; c < C; c += 64) {
    float t = (xr[c] - m) * r;
    if (gamma) t *= gamma[c];
    if (beta) t += beta[c];
    yr[c] = t;
  }
  if (lane == 0) {
    mean[row] = m;
    rstd[row] = r;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma. One wave per row, four rows
// per block: rows are independent, so the grid is rows / 4 blocks whatever C is (the first
// version walked 64 rows per block and accumulated dgamma / dbeta by LDS read-modify-write:
// 14 blocks and 4.9 ms for the [864, 4096] rescaler tensors of UNETR).
__global__ __launch_bounds__(256) void adell_layernorm_bwd_dx_kernel(
    const float* __restrict__ x, const float* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  const float* gr = dy + row * C;
  const float m = mean[row], r = rstd[row];
  float a = 0.f, b = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - m) * r;
    const float g = gr[c] * (gamma ? gamma[c] : 1.f);
    a += g;
    b += g * xh;
  }
  a = adell_wave_sum(a) / (float)C;
  b = adell_wave_sum(b) / (float)C;
  float* dr = dx + row * C;
  for (int c = lane; c < C; c += 64) {   // second read of the row comes from L1 / L2
    const float xh = (xr[c] - m) * r;
    const float g = gr[c] * (gamma ? gamma[c] : 1.f);
    dr[c] = r * (g - a - xh * b);
  }
}

// dgamma / dbeta partials: grid (column tiles of 256, row chunks); thread = one column, walks the
// rows of its chunk (consecutive threads read consecutive columns: coalesced), fixed order.
// part[chunk][2][C] = (sum dy * xhat, sum dy) over the chunk's rows.
__global__ __launch_bounds__(256) void adell_layernorm_bwd_affine_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ part, long rows, int C,
    int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  if (c >= C) return;
  float sg = 0.f, sb = 0.f;
  for (long row = r0; row < r1; ++row) {
    const float d = dy[row * C + c];
    sg += d * (x[row * C + c] - mean[row]) * rstd[row];
    sb += d;
  }
  part[((size_t)blockIdx.y * 2 + 0) * C + c] = sg;
  part[((size_t)blockIdx.y * 2 + 1) * C + c] = sb;
}

__global__ __launch_bounds__(256) void adell_rowsum_final_kernel(
    const float* __restrict__ part, int nb, int n, float* __restrict__ out0,
    float* __restrict__ out1, int half) {
  __shared__ double sh[4][64];
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < n)
    for (int b = vl; b < nb; b += 4) s += (double)part[(size_t)b * n + c];
  sh[vl][cl] = s;
  __syncthreads();
  if (vl != 0 || c >= n) return;
  s = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
  if (c < half) {
    if (out0) out0[c] = (float)s;
  } else if (out1) {
    out1[c - half] = (float)s;
  }
}

extern "C" int adell_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                   float* y, float* mean, float* rstd, long rows, int C,
                                   float eps, void* stream) {
  ADELL_REQUIRE(x && y && mean && rstd, "layernorm_fwd: null pointer");
  ADELL_REQUIRE(rows > 0 && C > 0, "layernorm_fwd: bad dims");
  hipLaunchKernelGGL(adell_layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, gamma, beta, y, mean, rstd, rows, C, eps);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// rows per chunk of the affine-gradient partials: enough chunks for ~512 blocks, >= 8 rows each
static int adell_ln_rows_per_chunk(long rows, int C) {
  const long tiles = (C + 255) / 256;
  long chunks = 512 / tiles;
  if (chunks < 1) chunks = 1;
  long rpc = (rows + chunks - 1) / chunks;
  if (rpc < 8) rpc = 8;
  return (int)rpc;
}
extern "C" long adell_layernorm_bwd_workspace(long rows, int C) {
  const int rpc = adell_ln_rows_per_chunk(rows, C);
  const long nb = (rows + rpc - 1) / rpc;
  return nb * 2 * C * (long)sizeof(float);
}

extern "C" int adell_layernorm_bwd(const float* x, const float* dy, const float* gamma,
                                   const float* mean, const float* rstd, float* dx,
                                   float* dgamma, float* dbeta, long rows, int C,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(x && dy && mean && rstd && dx, "layernorm_bwd: null pointer");
  ADELL_REQUIRE(rows > 0 && C > 0, "layernorm_bwd: bad dims");
  const bool want = dgamma || dbeta;
  if (want)
    ADELL_REQUIRE(workspace && (long)workspace_bytes >= adell_layernorm_bwd_workspace(rows, C),
                  "layernorm_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_layernorm_bwd_dx_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     st, x, dy, gamma, mean, rstd, dx, rows, C);
  if (want) {
    const int rpc = adell_ln_rows_per_chunk(rows, C);
    const int nb = (int)((rows + rpc - 1) / rpc);
    hipLaunchKernelGGL(adell_layernorm_bwd_affine_kernel, dim3(adell_cdiv(C, 256), nb), dim3(256),
                       0, st, x, dy, mean, rstd, (float*)workspace, rows, C, rpc);
    hipLaunchKernelGGL(adell_rowsum_final_kernel, dim3(adell_cdiv(2 * C, 64)), dim3(256), 0, st,
                       (const float*)workspace, nb, 2 * C, dgamma, dbeta, C);
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// out[i] = a[i] + b[i % period]  and its backward db[j] = sum_k dout[k*period + j]
// ---------------------------------------------------------------------------
__global__ void adell_add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       float* __restrict__ out, long n, long period) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i % period];
}
__global__ void adell_sum_bcast_kernel(const float* __restrict__ g, float* __restrict__ db,
                                       long n, long period) {
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < period;
       j += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (long i = j; i < n; i += period) s += g[i];
    db[j] = s;
  }
}
extern "C" int adell_add_bcast(const float* a, const float* b, float* out, long n, long period,
                               void* stream) {
  ADELL_REQUIRE(a && b && out && n > 0 && period > 0 && n % period == 0, "add_bcast: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_add_bcast_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a, b, out, n, period);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
extern "C" int adell_sum_bcast(const float* g, float* db, long n, long period, void* stream) {
  ADELL_REQUIRE(g && db && n > 0 && period > 0 && n % period == 0, "sum_bcast: bad arguments");
  long blocks = (period + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_sum_bcast_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, g, db, n, period);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Attention. Q,K: [BH][T][A]; V,O: [BH][T][Dv]; bias: [nbias][T][T] or null
// (bias row for sequence bh is bh % nbias). scale = 1/sqrt(A) unless given.
// One wave per query row; keys streamed through LDS in tiles of 64 with an
// online softmax; lane j owns key j of the tile for the scores, lane d owns
// output column d (and d+64, ...) for the PV product.
// ---------------------------------------------------------------------------
#define ATT_TK 64
#define ATT_ROWS 16  // query rows per block (4 waves x 4 rows)

struct AttStride { long b, h, r; };

struct AttArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* bias;
  const float* o;    // bwd
  const float* dout; // bwd
  const float* lse;  // bwd: [BH][T] log-sum-exp of the scaled scores
  float* out;        // fwd: O ; bwd: unused
  float* lse_out;    // fwd
  float* dq;
  float* dk;
  float* dv;
  int T, A, Dv, nbias;
  float scale;
  float drop_p;      // attention-probability dropout (0 = off)
  uint32_t seed_lo, seed_hi, rng_offset;
  // MFMA kernels: element strides of sequence bh = b * H + h (item b, head h) and of its token
  // rows, so that Q / K / V can be read inside a packed [B][T][H][q | k | v] projection and O,
  // dQ, dK, dV land in token-major buffers (contiguous [BH][T][*]: H = 1, {T * C, 0, C})
  int H;
  AttStride sq, sk, sv, so, sg, sdq, sdk, sdv;
};

__device__ __forceinline__ size_t att_base(const AttArgs& a, const AttStride& s, int bh) {
  return (size_t)(bh / a.H) * s.b + (size_t)(bh % a.H) * s.h;
}

// word `i` (0..3) of a Philox block by selects (indexing a local array with a run-time index puts the
// array in scratch memory)
__device__ __forceinline__ uint32_t adell_word4(const uint4& r, unsigned i) {
  return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

// Keep decision of probability (bh, qrow, kcol): a pure function of the seed, so the two
// backward kernels regenerate the mask the forward applied.
__device__ __forceinline__ bool adell_att_keep(const AttArgs& a, int bh, int qrow, int kcol) {
  if (a.drop_p <= 0.f) return true;
  const uint64_t e = ((uint64_t)bh * a.T + qrow) * a.T + kcol;
  const uint4 r = adell_philox4((uint32_t)(e >> 2), (uint32_t)(e >> 34), a.rng_offset + g_adell_rng_step, 2u,
                                a.seed_lo, a.seed_hi);
  return (float)(adell_word4(r, (unsigned)(e & 3)) >> 8) * (1.0f / 16777216.0f) >= a.drop_p;
}

// The same decisions for the four consecutive keys kcol0 .. kcol0 + 3 of one query: element
// indices e0 .. e0 + 3 fall into at most two Philox blocks (one when e0 is a multiple of 4).
__device__ __forceinline__ void adell_att_keep4(const AttArgs& a, int bh, int qrow, int kcol0,
                                                bool keep[4]) {
  keep[0] = keep[1] = keep[2] = keep[3] = true;
  if (a.drop_p <= 0.f) return;
  const uint64_t e0 = ((uint64_t)bh * a.T + qrow) * a.T + kcol0;
  const uint64_t b0 = e0 >> 2, b1 = (e0 + 3) >> 2;
  const uint4 r0 = adell_philox4((uint32_t)b0, (uint32_t)(b0 >> 32), a.rng_offset + g_adell_rng_step, 2u, a.seed_lo,
                                 a.seed_hi);
  uint4 r1 = r0;
  if (b1 != b0)
    r1 = adell_philox4((uint32_t)b1, (uint32_t)(b1 >> 32), a.rng_offset + g_adell_rng_step, 2u, a.seed_lo, a.seed_hi);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t e = e0 + j;
    const bool first = (e >> 2) == b0;      // (component-wise: a select of the structs goes
    const unsigned i = (unsigned)(e & 3);   // through memory)
    const uint32_t w = i == 0 ? (first ? r0.x : r1.x) : (i == 1 ? (first ? r0.y : r1.y) :
                       (i == 2 ? (first ? r0.z : r1.z) : (first ? r0.w : r1.w)));
    keep[j] = (float)(w >> 8) * (1.0f / 16777216.0f) >= a.drop_p;
  }
}

__global__ __launch_bounds__(256) void adell_attention_fwd_kernel(AttArgs a) {
  extern __shared__ float sh[];
  const int AP = a.A + 1, DP = a.Dv;  // K rows padded against bank conflicts
  float* sK = sh;                       // [ATT_TK][AP]
  float* sV = sK + ATT_TK * AP;         // [ATT_TK][DP]
  float* sQ = sV + ATT_TK * DP;         // [ATT_ROWS][A]
  float* sP = sQ + ATT_ROWS * a.A;      // [4 waves][ATT_TK]
  const int bh = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ATT_ROWS;
  const float* qb = a.q + (size_t)bh * a.T * a.A;
  const float* kb = a.k + (size_t)bh * a.T * a.A;
  const float* vb = a.v + (size_t)bh * a.T * a.Dv;
  for (int i = threadIdx.x; i < ATT_ROWS * a.A; i += 256) {
    const int r = i / a.A, c = i - r * a.A;
    sQ[i] = (row0 + r < a.T) ? qb[(size_t)(row0 + r) * a.A + c] : 0.f;
  }
  constexpr int RPW = ATT_ROWS / 4;
  constexpr int MAXD = 4;  // output columns per lane: Dv <= 256
  float m[RPW], l[RPW], acc[RPW][MAXD];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) acc[r][d] = 0.f;
  }
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int k0 = 0; k0 < a.T; k0 += ATT_TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < ATT_TK * a.A; i += 256) {
      const int r = i / a.A, c = i - r * a.A;
      sK[r * AP + c] = (k0 + r < a.T) ? kb[(size_t)(k0 + r) * a.A + c] : 0.f;
    }
    for (int i = threadIdx.x; i < ATT_TK * a.Dv; i += 256) {
      const int r = i / a.Dv, c = i - r * a.Dv;
      sV[r * DP + c] = (k0 + r < a.T) ? vb[(size_t)(k0 + r) * a.Dv + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int qrow = row0 + wave * RPW + r;
      const float* qr = sQ + (wave * RPW + r) * a.A;
      float s = -INFINITY;
      if (k0 + lane < a.T && qrow < a.T) {
        float t = 0.f;
        for (int c = 0; c < a.A; ++c) t += qr[c] * sK[lane * AP + c];
        s = t * a.scale;
        if (biasb) s += biasb[(size_t)qrow * a.T + k0 + lane];
      }
      float mx = s;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float mnew = fmaxf(m[r], mx);
      const float p = (s == -INFINITY) ? 0.f : expf(s - mnew);
      const float corr = (m[r] == -INFINITY) ? 0.f : expf(m[r] - mnew);
      const float psum = adell_wave_sum(p);
      l[r] = l[r] * corr + psum;
      m[r] = mnew;
      // dropout acts on the normalised probabilities: l keeps the full sum, the PV product
      // sees the kept entries scaled by 1/(1-p)
      sP[wave * ATT_TK + lane] =
          (p != 0.f && adell_att_keep(a, bh, qrow, k0 + lane)) ? p * keep_scale : 0.f;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int d = 0; d < MAXD; ++d) {
        const int col = lane + 64 * d;
        if (col < a.Dv) {
          float t = 0.f;
          for (int j = 0; j < ATT_TK; ++j) t += sP[wave * ATT_TK + j] * sV[j * DP + col];
          acc[r][d] = acc[r][d] * corr + t;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int qrow = row0 + wave * RPW + r;
    if (qrow >= a.T) continue;
    const float inv = 1.0f / l[r];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      const int col = lane + 64 * d;
      if (col < a.Dv) a.out[((size_t)bh * a.T + qrow) * a.Dv + col] = acc[r][d] * inv;
    }
    if (lane == 0) a.lse_out[(size_t)bh * a.T + qrow] = m[r] + logf(l[r]);
  }
}

// dQ: one wave per query row (same structure as forward).
__global__ __launch_bounds__(256) void adell_attention_bwd_q_kernel(AttArgs a) {
  extern __shared__ float sh[];
  const int AP = a.A + 1, DP = a.Dv + 1;
  float* sK = sh;                        // [ATT_TK][AP]
  float* sV = sK + ATT_TK * AP;          // [ATT_TK][DP]
  float* sQ = sV + ATT_TK * DP;          // [ATT_ROWS][A]
  float* sdO = sQ + ATT_ROWS * a.A;      // [ATT_ROWS][Dv]
  float* sP = sdO + ATT_ROWS * a.Dv;     // [4][ATT_TK]  (holds dS)
  const int bh = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ATT_ROWS;
  const float* qb = a.q + (size_t)bh * a.T * a.A;
  const float* kb = a.k + (size_t)bh * a.T * a.A;
  const float* vb = a.v + (size_t)bh * a.T * a.Dv;
  const float* ob = a.o + (size_t)bh * a.T * a.Dv;
  const float* gb = a.dout + (size_t)bh * a.T * a.Dv;
  for (int i = threadIdx.x; i < ATT_ROWS * a.A; i += 256) {
    const int r = i / a.A, c = i - r * a.A;
    sQ[i] = (row0 + r < a.T) ? qb[(size_t)(row0 + r) * a.A + c] : 0.f;
  }
  for (int i = threadIdx.x; i < ATT_ROWS * a.Dv; i += 256) {
    const int r = i / a.Dv, c = i - r * a.Dv;
    sdO[i] = (row0 + r < a.T) ? gb[(size_t)(row0 + r) * a.Dv + c] : 0.f;
  }
  __syncthreads();
  constexpr int RPW = ATT_ROWS / 4;
  constexpr int MAXA = 4;  // A <= 256
  float Dr[RPW], lse[RPW], acc[RPW][MAXA];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int qrow = row0 + wave * RPW + r;
    float t = 0.f;
    if (qrow < a.T)
      for (int c = lane; c < a.Dv; c += 64)
        t += sdO[(wave * RPW + r) * a.Dv + c] * ob[(size_t)qrow * a.Dv + c];
    Dr[r] = adell_wave_sum(t);
    lse[r] = qrow < a.T ? a.lse[(size_t)bh * a.T + qrow] : 0.f;
#pragma unroll
    for (int d = 0; d < MAXA; ++d) acc[r][d] = 0.f;
  }
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int k0 = 0; k0 < a.T; k0 += ATT_TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < ATT_TK * a.A; i += 256) {
      const int r = i / a.A, c = i - r * a.A;
      sK[r * AP + c] = (k0 + r < a.T) ? kb[(size_t)(k0 + r) * a.A + c] : 0.f;
    }
    for (int i = threadIdx.x; i < ATT_TK * a.Dv; i += 256) {
      const int r = i / a.Dv, c = i - r * a.Dv;
      sV[r * DP + c] = (k0 + r < a.T) ? vb[(size_t)(k0 + r) * a.Dv + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int qrow = row0 + wave * RPW + r;
      const float* qr = sQ + (wave * RPW + r) * a.A;
      const float* gr = sdO + (wave * RPW + r) * a.Dv;
      float ds = 0.f;
      if (k0 + lane < a.T && qrow < a.T) {
        float t = 0.f;
        for (int c = 0; c < a.A; ++c) t += qr[c] * sK[lane * AP + c];
        float s = t * a.scale;
        if (biasb) s += biasb[(size_t)qrow * a.T + k0 + lane];
        const float p = expf(s - lse[r]);
        float dp = 0.f;
        for (int c = 0; c < a.Dv; ++c) dp += gr[c] * sV[lane * DP + c];
        dp = adell_att_keep(a, bh, qrow, k0 + lane) ? dp * keep_scale : 0.f;
        ds = p * (dp - Dr[r]);
      }
      sP[wave * ATT_TK + lane] = ds;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int d = 0; d < MAXA; ++d) {
        const int col = lane + 64 * d;
        if (col < a.A) {
          float t = 0.f;
          for (int j = 0; j < ATT_TK; ++j) t += sP[wave * ATT_TK + j] * sK[j * AP + col];
          acc[r][d] += t;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int qrow = row0 + wave * RPW + r;
    if (qrow >= a.T) continue;
#pragma unroll
    for (int d = 0; d < MAXA; ++d) {
      const int col = lane + 64 * d;
      if (col < a.A) a.dq[((size_t)bh * a.T + qrow) * a.A + col] = acc[r][d] * a.scale;
    }
  }
}

// dK, dV: one wave per key row; queries streamed through LDS in tiles of 64.
__global__ __launch_bounds__(256) void adell_attention_bwd_kv_kernel(AttArgs a) {
  extern __shared__ float sh[];
  const int AP = a.A + 1, DP = a.Dv + 1;
  float* sQ = sh;                        // [ATT_TK][AP]   query tile
  float* sdO = sQ + ATT_TK * AP;         // [ATT_TK][DP]
  float* sK = sdO + ATT_TK * DP;         // [ATT_ROWS][A]  this block's keys
  float* sV = sK + ATT_ROWS * a.A;       // [ATT_ROWS][Dv]
  float* sP = sV + ATT_ROWS * a.Dv;      // [4][2][ATT_TK] (p, dS)
  float* sL = sP + 4 * 2 * ATT_TK;       // [ATT_TK][2] lse, D of the query tile
  const int bh = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ATT_ROWS;
  const float* qb = a.q + (size_t)bh * a.T * a.A;
  const float* kb = a.k + (size_t)bh * a.T * a.A;
  const float* vb = a.v + (size_t)bh * a.T * a.Dv;
  const float* ob = a.o + (size_t)bh * a.T * a.Dv;
  const float* gb = a.dout + (size_t)bh * a.T * a.Dv;
  for (int i = threadIdx.x; i < ATT_ROWS * a.A; i += 256) {
    const int r = i / a.A, c = i - r * a.A;
    sK[i] = (row0 + r < a.T) ? kb[(size_t)(row0 + r) * a.A + c] : 0.f;
  }
  for (int i = threadIdx.x; i < ATT_ROWS * a.Dv; i += 256) {
    const int r = i / a.Dv, c = i - r * a.Dv;
    sV[i] = (row0 + r < a.T) ? vb[(size_t)(row0 + r) * a.Dv + c] : 0.f;
  }
  constexpr int RPW = ATT_ROWS / 4;
  constexpr int MAXC = 4;
  float dk[RPW][MAXC], dv[RPW][MAXC];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int d = 0; d < MAXC; ++d) dk[r][d] = dv[r][d] = 0.f;
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  for (int q0 = 0; q0 < a.T; q0 += ATT_TK) {
    __syncthreads();
    for (int i = threadIdx.x; i < ATT_TK * a.A; i += 256) {
      const int r = i / a.A, c = i - r * a.A;
      sQ[r * AP + c] = (q0 + r < a.T) ? qb[(size_t)(q0 + r) * a.A + c] : 0.f;
    }
    for (int i = threadIdx.x; i < ATT_TK * a.Dv; i += 256) {
      const int r = i / a.Dv, c = i - r * a.Dv;
      sdO[r * DP + c] = (q0 + r < a.T) ? gb[(size_t)(q0 + r) * a.Dv + c] : 0.f;
    }
    __syncthreads();
    // D_i = dO_i . O_i for the query tile: wave w handles rows w, w+4, ...
    for (int r = wave; r < ATT_TK; r += 4) {
      float t = 0.f;
      if (q0 + r < a.T)
        for (int c = lane; c < a.Dv; c += 64) t += sdO[r * DP + c] * ob[(size_t)(q0 + r) * a.Dv + c];
      t = adell_wave_sum(t);
      if (lane == 0) {
        sL[r * 2 + 0] = (q0 + r < a.T) ? a.lse[(size_t)bh * a.T + q0 + r] : 0.f;
        sL[r * 2 + 1] = t;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int krow = row0 + wave * RPW + r;
      const float* kr = sK + (wave * RPW + r) * a.A;
      const float* vr = sV + (wave * RPW + r) * a.Dv;
      float p = 0.f, ds = 0.f;
      if (q0 + lane < a.T && krow < a.T) {
        float t = 0.f;
        for (int c = 0; c < a.A; ++c) t += kr[c] * sQ[lane * AP + c];
        float s = t * a.scale;
        if (biasb) s += biasb[(size_t)(q0 + lane) * a.T + krow];
        p = expf(s - sL[lane * 2 + 0]);
        float dp = 0.f;
        for (int c = 0; c < a.Dv; ++c) dp += vr[c] * sdO[lane * DP + c];
        const float km = adell_att_keep(a, bh, q0 + lane, krow) ? keep_scale : 0.f;
        ds = p * (dp * km - sL[lane * 2 + 1]);
        p *= km;  // dV sees the dropped probabilities
      }
      sP[(wave * 2 + 0) * ATT_TK + lane] = p;
      sP[(wave * 2 + 1) * ATT_TK + lane] = ds;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int d = 0; d < MAXC; ++d) {
        const int col = lane + 64 * d;
        if (col < a.Dv) {
          float t = 0.f;
          for (int j = 0; j < ATT_TK; ++j) t += sP[(wave * 2 + 0) * ATT_TK + j] * sdO[j * DP + col];
          dv[r][d] += t;
        }
        if (col < a.A) {
          float t = 0.f;
          for (int j = 0; j < ATT_TK; ++j) t += sP[(wave * 2 + 1) * ATT_TK + j] * sQ[j * AP + col];
          dk[r][d] += t;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int krow = row0 + wave * RPW + r;
    if (krow >= a.T) continue;
#pragma unroll
    for (int d = 0; d < MAXC; ++d) {
      const int col = lane + 64 * d;
      if (col < a.Dv) a.dv[((size_t)bh * a.T + krow) * a.Dv + col] = dv[r][d];
      if (col < a.A) a.dk[((size_t)bh * a.T + krow) * a.A + col] = dk[r][d] * a.scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA attention (head dims that are multiples of 32, up to 128): QK^T and PV -- and in the
// backward dO V^T, dS K, dS^T Q, P^T dO -- on v_mfma_f32_32x32x2_f32, i.e. exact fp32 FMA chains
// (the reference op is F.scaled_dot_product_attention on fp32 tensors, linear_blocks.py:358-417).
// One wave per 32-row tile, one tile per block: 7 x BH blocks at the 216 tokens of config 3.
//
// The score tile is always computed with the SUMMATION index of the product that follows in its
// accumulator ROWS (registers): a 32x32 accumulator holds column (lane & 31) and rows
// am_row(r, lane >> 5) in its 16 registers, and the A operand of the 32x32x2 MFMA wants
// A[i = lane & 31][k = lane >> 5] -- so register r of the tile IS the A operand of k-step r of the
// next product (k = rows am_row(r, 0), am_row(r, 1)), with no lane movement and no LDS trip:
//   forward / dQ kernel:  S^T = K Q^T (rows = keys, column = query)  ->  O = P V,  dQ = dS K
//   dK/dV kernel:         S   = Q K^T (rows = queries, column = key) ->  dV = P^T dO, dK = dS^T Q
// Softmax statistics are per column (= per lane) in the first form: a max / sum over the 16
// registers plus one exchange between the two lane halves; the forward makes two passes over the
// keys (statistics, then normalised probabilities), so the output tile is never rescaled.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int am_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// rows [r0, r0 + NR) x C columns of a [T][C] matrix -> LDS rows of C + 1 floats, rows >= T zero,
// by the 256 threads of the block. All 16-byte loads are issued before the first LDS store, from
// clamped row indices (a select on the loaded value, not a branch around the load: hipcc would
// otherwise wait for every load in turn -- dependent L2 round trips).
template <int C>
__device__ __forceinline__ void am_stage32(float* dst, const float* src, long rs, int r0, int T, int tid) {
  constexpr int N4 = 32 * C / 4 / 256;   // float4 per thread for a 32-row tile
  float4 v[N4];
#pragma unroll
  for (int u = 0; u < N4; ++u) {
    const int i = tid + 256 * u, r = i / (C / 4), c4 = i - r * (C / 4);
    const int row = r0 + r < T ? r0 + r : T - 1;
    v[u] = *reinterpret_cast<const float4*>(src + (size_t)row * rs + 4 * c4);
  }
#pragma unroll
  for (int u = 0; u < N4; ++u) {
    const int i = tid + 256 * u, r = i / (C / 4), c4 = i - r * (C / 4);
    const bool ok = r0 + r < T;
    float* d = dst + r * (C + 1) + 4 * c4;
    d[0] = ok ? v[u].x : 0.f;
    d[1] = ok ? v[u].y : 0.f;
    d[2] = ok ? v[u].z : 0.f;
    d[3] = ok ? v[u].w : 0.f;
  }
}
// the whole [T][C] matrix, padded to a multiple of 32 rows (resident variants): the loads of
// four tiles are in flight together
template <int C>
__device__ __forceinline__ void am_stage_all(float* dst, const float* src, long rs, int T, int tid) {
  constexpr int N4 = 32 * C / 4 / 256, G = 4;
  const int tiles = (T + 31) / 32;
  for (int t0 = 0; t0 < tiles; t0 += G) {
    float4 v[G][N4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int u = 0; u < N4; ++u) {
        const int i = tid + 256 * u, r = i / (C / 4), c4 = i - r * (C / 4);
        int row = (t0 + g) * 32 + r;
        row = row < T ? row : T - 1;
        v[g][u] = *reinterpret_cast<const float4*>(src + (size_t)row * rs + 4 * c4);
      }
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (t0 + g < tiles) {
#pragma unroll
        for (int u = 0; u < N4; ++u) {
          const int i = tid + 256 * u, r = i / (C / 4), c4 = i - r * (C / 4);
          const bool ok = (t0 + g) * 32 + r < T;
          float* d = dst + (size_t)((t0 + g) * 32 + r) * (C + 1) + 4 * c4;
          d[0] = ok ? v[g][u].x : 0.f;
          d[1] = ok ? v[g][u].y : 0.f;
          d[2] = ok ? v[g][u].z : 0.f;
          d[3] = ok ? v[g][u].w : 0.f;
        }
      }
  }
}

// S^T tile: rows = keys k0 + am_row(r, h), column = this lane's query; -inf outside the sequence
template <int A>
__device__ __forceinline__ void am_scores_t(const AttArgs& a, const float* sK, const float* qreg,
                                            const float* biasb, int qrow, bool qok, int k0, int li,
                                            int h, float* val) {
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
  for (int ss = 0; ss < A / 2; ++ss)
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[li * (A + 1) + 2 * ss + h], qreg[ss], s, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = k0 + am_row(r, h);
    float v = -INFINITY;
    if (qok && key < a.T) {
      v = s[r] * a.scale;
      if (biasb) v += biasb[(size_t)qrow * a.T + key];
    }
    val[r] = v;
  }
}

// Block = 4 waves = 4 query tiles of 32 rows sharing the staged K / V. RES: every K / V row of the
// sequence is staged ONCE and stays in LDS (config 3: 216 tokens x (64 + 64) floats = 110 KB), no
// further global loads or barriers in the key loops; otherwise 32-key tiles are streamed.
template <int AT, int DT, bool RES>
__global__ __launch_bounds__(256) void adell_attn_mfma_fwd_kernel(AttArgs a) {
  constexpr int A = AT * 32, Dv = DT * 32;
  extern __shared__ float sh[];
  const int Tpad = (a.T + 31) & ~31;
  float* sK = sh;
  float* sV = sK + (size_t)(RES ? Tpad : 32) * (A + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, q0 = (blockIdx.x * 4 + wave) * 32, qrow = q0 + li;
  const bool qok = qrow < a.T, active = q0 < a.T;   // `active` is wave-uniform
  const float* qb = a.q + att_base(a, a.sq, bh);
  const float* kb = a.k + att_base(a, a.sk, bh);
  const float* vb = a.v + att_base(a, a.sv, bh);
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  float qreg[A / 2];
#pragma unroll
  for (int ss = 0; ss < A / 2; ++ss) qreg[ss] = qok ? qb[(size_t)qrow * a.sq.r + 2 * ss + h] : 0.f;
  if (RES) {
    am_stage_all<A>(sK, kb, a.sk.r, a.T, tid);
    am_stage_all<Dv>(sV, vb, a.sv.r, a.T, tid);
    __syncthreads();
  }
  // pass 1: log-sum-exp of this lane's query
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < a.T; k0 += 32) {
    if (!RES) {
      __syncthreads();
      am_stage32<A>(sK, kb, a.sk.r, k0, a.T, tid);
      __syncthreads();
    }
    if (!active) continue;
    float val[16];
    am_scores_t<A>(a, sK + (size_t)(RES ? k0 : 0) * (A + 1), qreg, biasb, qrow, qok, k0, li, h, val);
    float mt = val[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, val[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mnew = fmaxf(m, mt);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ps += (val[r] == -INFINITY) ? 0.f : expf(val[r] - mnew);
    ps += __shfl_xor(ps, 32, 64);
    l = l * ((m == -INFINITY) ? 0.f : expf(m - mnew)) + ps;
    m = mnew;
  }
  const float lse = m + logf(l);
  // pass 2: O = dropout(softmax) V with the normalised probabilities as the MFMA A operand
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  for (int k0 = 0; k0 < a.T; k0 += 32) {
    if (!RES) {
      __syncthreads();
      am_stage32<A>(sK, kb, a.sk.r, k0, a.T, tid);
      am_stage32<Dv>(sV, vb, a.sv.r, k0, a.T, tid);
      __syncthreads();
    }
    if (!active) continue;
    const float* sVt = sV + (size_t)(RES ? k0 : 0) * (Dv + 1);
    float val[16];
    am_scores_t<A>(a, sK + (size_t)(RES ? k0 : 0) * (A + 1), qreg, biasb, qrow, qok, k0, li, h, val);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      bool keep[4];   // registers 4 r4 .. 4 r4 + 3 are four consecutive keys
      adell_att_keep4(a, bh, qrow, k0 + am_row(4 * r4, h), keep);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * r4 + j;
        const float p = (val[r] == -INFINITY) ? 0.f : expf(val[r] - lse);
        val[r] = keep[j] ? p * keep_scale : 0.f;
      }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
            val[r], sVt[am_row(r, h) * (Dv + 1) + dt * 32 + li], o[dt], 0, 0, 0);
  }
  if (!active) return;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = q0 + am_row(r, h);
      if (row < a.T) a.out[att_base(a, a.so, bh) + (size_t)row * a.so.r + dt * 32 + li] = o[dt][r];
    }
  if (h == 0 && qok) a.lse_out[(size_t)bh * a.T + qrow] = lse;
}

// dQ = scale * dS K with dS^T tiles (rows = keys, column = this lane's query)
template <int AT, int DT, bool RES>
__global__ __launch_bounds__(256) void adell_attn_mfma_bwd_q_kernel(AttArgs a) {
  constexpr int A = AT * 32, Dv = DT * 32;
  extern __shared__ float sh[];
  const int Tpad = (a.T + 31) & ~31;
  float* sK = sh;
  float* sV = sK + (size_t)(RES ? Tpad : 32) * (A + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, q0 = (blockIdx.x * 4 + wave) * 32, qrow = q0 + li;
  const bool qok = qrow < a.T, active = q0 < a.T;
  const float* qb = a.q + att_base(a, a.sq, bh);
  const float* kb = a.k + att_base(a, a.sk, bh);
  const float* vb = a.v + att_base(a, a.sv, bh);
  const float* ob = a.o + att_base(a, a.so, bh);
  const float* gb = a.dout + att_base(a, a.sg, bh);
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  float qreg[A / 2], doreg[Dv / 2];
  float D = 0.f;
#pragma unroll
  for (int ss = 0; ss < A / 2; ++ss) qreg[ss] = qok ? qb[(size_t)qrow * a.sq.r + 2 * ss + h] : 0.f;
#pragma unroll
  for (int ss = 0; ss < Dv / 2; ++ss) {
    doreg[ss] = qok ? gb[(size_t)qrow * a.sg.r + 2 * ss + h] : 0.f;
    D += qok ? doreg[ss] * ob[(size_t)qrow * a.so.r + 2 * ss + h] : 0.f;
  }
  D += __shfl_xor(D, 32, 64);
  const float lse = qok ? a.lse[(size_t)bh * a.T + qrow] : 0.f;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  f32x16 dq[AT];
#pragma unroll
  for (int at = 0; at < AT; ++at)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[at][r] = 0.f;
  if (RES) {
    am_stage_all<A>(sK, kb, a.sk.r, a.T, tid);
    am_stage_all<Dv>(sV, vb, a.sv.r, a.T, tid);
    __syncthreads();
  }
  for (int k0 = 0; k0 < a.T; k0 += 32) {
    if (!RES) {
      __syncthreads();
      am_stage32<A>(sK, kb, a.sk.r, k0, a.T, tid);
      am_stage32<Dv>(sV, vb, a.sv.r, k0, a.T, tid);
      __syncthreads();
    }
    if (!active) continue;
    const float* sKt = sK + (size_t)(RES ? k0 : 0) * (A + 1);
    const float* sVt = sV + (size_t)(RES ? k0 : 0) * (Dv + 1);
    float val[16];
    am_scores_t<A>(a, sKt, qreg, biasb, qrow, qok, k0, li, h, val);
    f32x16 dp;   // dP^T = V dO^T: rows = keys, column = query
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
    for (int ss = 0; ss < Dv / 2; ++ss)
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sVt[li * (Dv + 1) + 2 * ss + h], doreg[ss], dp, 0, 0, 0);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      bool keep[4];
      adell_att_keep4(a, bh, qrow, k0 + am_row(4 * r4, h), keep);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * r4 + j;
        const float p = (val[r] == -INFINITY) ? 0.f : expf(val[r] - lse);
        const float g = keep[j] ? dp[r] * keep_scale : 0.f;
        val[r] = p * (g - D);   // dS
      }
    }
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dq[at] = __builtin_amdgcn_mfma_f32_32x32x2f32(
            val[r], sKt[am_row(r, h) * (A + 1) + at * 32 + li], dq[at], 0, 0, 0);
  }
  if (!active) return;
#pragma unroll
  for (int at = 0; at < AT; ++at)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = q0 + am_row(r, h);
      if (row < a.T) a.dq[att_base(a, a.sdq, bh) + (size_t)row * a.sdq.r + at * 32 + li] = dq[at][r] * a.scale;
    }
}

// dV = P_drop^T dO, dK = scale * dS^T Q with S tiles (rows = queries, column = this lane's key);
// block = 4 key tiles sharing the staged Q / dO (+ D = rowsum(dO * O) and lse per query)
template <int AT, int DT, bool RES>
__global__ __launch_bounds__(256) void adell_attn_mfma_bwd_kv_kernel(AttArgs a) {
  constexpr int A = AT * 32, Dv = DT * 32;
  extern __shared__ float sh[];
  const int Tpad = (a.T + 31) & ~31;
  const int rows = RES ? Tpad : 32;
  float* sQ = sh;
  float* sG = sQ + (size_t)rows * (A + 1);     // dO
  float* sD = sG + (size_t)rows * (Dv + 1);
  float* sL = sD + rows;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, k0 = (blockIdx.x * 4 + wave) * 32, krow = k0 + li;
  const bool kok = krow < a.T, active = k0 < a.T;
  const float* qb = a.q + att_base(a, a.sq, bh);
  const float* kb = a.k + att_base(a, a.sk, bh);
  const float* vb = a.v + att_base(a, a.sv, bh);
  const float* ob = a.o + att_base(a, a.so, bh);
  const float* gb = a.dout + att_base(a, a.sg, bh);
  const float* biasb = a.bias ? a.bias + (size_t)(bh % a.nbias) * a.T * a.T : nullptr;
  float kreg[A / 2], vreg[Dv / 2];
#pragma unroll
  for (int ss = 0; ss < A / 2; ++ss) kreg[ss] = kok ? kb[(size_t)krow * a.sk.r + 2 * ss + h] : 0.f;
#pragma unroll
  for (int ss = 0; ss < Dv / 2; ++ss) vreg[ss] = kok ? vb[(size_t)krow * a.sv.r + 2 * ss + h] : 0.f;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  f32x16 dv[DT], dk[AT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[t][r] = 0.f;
#pragma unroll
  for (int t = 0; t < AT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[t][r] = 0.f;
  // D[q] = sum_dv dO[q][dv] O[q][dv] and lse[q] of queries [r0, r0 + n): 8 threads per query
  auto stage_stats = [&](int r0, int n, float* dD, float* dL) {
    for (int qi = tid >> 3; qi < n; qi += 32) {
      const int qr = r0 + qi;
      float d = 0.f;
      if (qr < a.T)
        for (int c = (tid & 7) * 4; c < Dv; c += 32) {
          const float4 g4 = *reinterpret_cast<const float4*>(gb + (size_t)qr * a.sg.r + c);
          const float4 o4 = *reinterpret_cast<const float4*>(ob + (size_t)qr * a.so.r + c);
          d += g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
        }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if ((tid & 7) == 0) {
        dD[qi] = d;
        dL[qi] = qr < a.T ? a.lse[(size_t)bh * a.T + qr] : 0.f;
      }
    }
  };
  if (RES) {
    am_stage_all<A>(sQ, qb, a.sq.r, a.T, tid);
    am_stage_all<Dv>(sG, gb, a.sg.r, a.T, tid);
    stage_stats(0, Tpad, sD, sL);
    __syncthreads();
  }
  for (int q0 = 0; q0 < a.T; q0 += 32) {
    if (!RES) {
      __syncthreads();
      am_stage32<A>(sQ, qb, a.sq.r, q0, a.T, tid);
      am_stage32<Dv>(sG, gb, a.sg.r, q0, a.T, tid);
      stage_stats(q0, 32, sD, sL);
      __syncthreads();
    }
    if (!active) continue;
    const int base = RES ? q0 : 0;
    const float* sQt = sQ + (size_t)base * (A + 1);
    const float* sGt = sG + (size_t)base * (Dv + 1);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ss = 0; ss < A / 2; ++ss)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(sQt[li * (A + 1) + 2 * ss + h], kreg[ss], s, 0, 0, 0);
#pragma unroll
    for (int ss = 0; ss < Dv / 2; ++ss)
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sGt[li * (Dv + 1) + 2 * ss + h], vreg[ss], dp, 0, 0, 0);
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = am_row(r, h), qr = q0 + row;
      float p = 0.f;
      if (kok && qr < a.T) {
        float v = s[r] * a.scale;
        if (biasb) v += biasb[(size_t)qr * a.T + krow];
        p = expf(v - sL[base + row]);
      }
      const bool keep = adell_att_keep(a, bh, qr, krow);
      pd[r] = keep ? p * keep_scale : 0.f;
      ds[r] = p * ((keep ? dp[r] * keep_scale : 0.f) - sD[base + row]);
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
            pd[r], sGt[am_row(r, h) * (Dv + 1) + t * 32 + li], dv[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
            ds[r], sQt[am_row(r, h) * (A + 1) + t * 32 + li], dk[t], 0, 0, 0);
  }
  if (!active) return;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = k0 + am_row(r, h);
      if (row < a.T) a.dv[att_base(a, a.sdv, bh) + (size_t)row * a.sdv.r + t * 32 + li] = dv[t][r];
    }
#pragma unroll
  for (int t = 0; t < AT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = k0 + am_row(r, h);
      if (row < a.T) a.dk[att_base(a, a.sdk, bh) + (size_t)row * a.sdk.r + t * 32 + li] = dk[t][r] * a.scale;
    }
}

static bool adell_att_mfma_ok(int T, int A, int Dv) {
  return T >= 16 && (A == 32 || A == 64 || A == 128) && (Dv == 32 || Dv == 64 || Dv == 128);
}

// which: 0 forward, 1 backward dQ, 2 backward dK/dV
template <int AT, int DT, bool RES>
static int adell_att_mfma_launch3(int which, const AttArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  const void* fn = which == 0 ? reinterpret_cast<const void*>(adell_attn_mfma_fwd_kernel<AT, DT, RES>)
                   : which == 1 ? reinterpret_cast<const void*>(adell_attn_mfma_bwd_q_kernel<AT, DT, RES>)
                                : reinterpret_cast<const void*>(adell_attn_mfma_bwd_kv_kernel<AT, DT, RES>);
  if (lds > 48 * 1024)
    ADELL_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (which == 0)
    hipLaunchKernelGGL((adell_attn_mfma_fwd_kernel<AT, DT, RES>), grid, dim3(256), lds, st, a);
  else if (which == 1)
    hipLaunchKernelGGL((adell_attn_mfma_bwd_q_kernel<AT, DT, RES>), grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((adell_attn_mfma_bwd_kv_kernel<AT, DT, RES>), grid, dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
template <int AT, int DT>
static int adell_att_mfma_launch2(int which, const AttArgs& a, int BH, hipStream_t st) {
  const dim3 grid((unsigned)adell_cdiv(a.T, 128), (unsigned)BH);
  const size_t tpad = (size_t)((a.T + 31) & ~31);
  const size_t per_row = (size_t)(AT * 32 + 1) + (size_t)(DT * 32 + 1) + (which == 2 ? 2 : 0);
  const size_t res = sizeof(float) * tpad * per_row;
  if (res <= 150 * 1024)   // the whole sequence stays in LDS
    return adell_att_mfma_launch3<AT, DT, true>(which, a, grid, res, st);
  return adell_att_mfma_launch3<AT, DT, false>(which, a, grid, sizeof(float) * 32 * per_row, st);
}
template <int AT>
static int adell_att_mfma_launch1(int which, const AttArgs& a, int BH, hipStream_t st) {
  switch (a.Dv) {
    case 32: return adell_att_mfma_launch2<AT, 1>(which, a, BH, st);
    case 64: return adell_att_mfma_launch2<AT, 2>(which, a, BH, st);
    default: return adell_att_mfma_launch2<AT, 4>(which, a, BH, st);
  }
}
static int adell_att_mfma_launch(int which, const AttArgs& a, int BH, hipStream_t st) {
  switch (a.A) {
    case 32: return adell_att_mfma_launch1<1>(which, a, BH, st);
    case 64: return adell_att_mfma_launch1<2>(which, a, BH, st);
    default: return adell_att_mfma_launch1<4>(which, a, BH, st);
  }
}

static int adell_att_check(int BH, int T, int A, int Dv, int nbias, const float* bias) {
  ADELL_REQUIRE(BH > 0 && T > 0 && A > 0 && Dv > 0, "attention: bad dims");
  ADELL_REQUIRE(A <= 256 && Dv <= 256, "attention: head dims up to 256 supported");
  ADELL_REQUIRE(bias == nullptr || nbias > 0, "attention: bias needs nbias > 0");
  return ADELL_OK;
}

template <typename K>
static int adell_att_launch(K kern, const AttArgs& a, int BH, size_t lds, hipStream_t st) {
  ADELL_REQUIRE(lds <= 160 * 1024, "attention: head dims need %zu B of LDS (> 160 KiB)", lds);
  ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(kern, dim3(adell_cdiv(a.T, ATT_ROWS), BH), dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// contiguous [BH][T][A | Dv] operands as strides
static void adell_att_contiguous(AttArgs* a) {
  const AttStride sa = {(long)a->T * a->A, 0, a->A}, sd = {(long)a->T * a->Dv, 0, a->Dv};
  a->H = 1;
  a->sq = a->sk = a->sdq = a->sdk = sa;
  a->sv = a->so = a->sg = a->sdv = sd;
}

static int adell_att_stride(const char* what, const void* p, const long* s, int C, AttStride* out) {
  ADELL_REQUIRE(p && (((uintptr_t)p) & 15) == 0, "attention (strided): %s must be 16-byte aligned", what);
  ADELL_REQUIRE(s[0] >= 0 && s[1] >= 0 && s[2] >= C && s[0] % 4 == 0 && s[1] % 4 == 0 && s[2] % 4 == 0,
                "attention (strided): %s strides (%ld, %ld, %ld) must be multiples of 4 elements, "
                "rows at least %d apart", what, s[0], s[1], s[2], C);
  out->b = s[0]; out->h = s[1]; out->r = s[2];
  return ADELL_OK;
}

extern "C" int adell_attention_fwd(const float* q, const float* k, const float* v,
                                   const float* bias, int nbias, int BH, int T, int A, int Dv,
                                   float scale, float drop_p, unsigned long long seed,
                                   unsigned int rng_offset, float* out, float* lse,
                                   void* stream) {
  int rc = adell_att_check(BH, T, A, Dv, nbias, bias);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(q && k && v && out && lse, "attention_fwd: null pointer");
  AttArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.out = out; a.lse_out = lse;
  a.T = T; a.A = A; a.Dv = Dv; a.nbias = nbias > 0 ? nbias : 1; a.scale = scale;
  ADELL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention: bad dropout probability");
  a.drop_p = drop_p; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.rng_offset = rng_offset;
  if (adell_att_mfma_ok(T, A, Dv) && !g_adell_tune.attn_nomfma) {
    adell_att_contiguous(&a);
    return adell_att_mfma_launch(0, a, BH, (hipStream_t)stream);
  }
  const size_t lds = sizeof(float) * ((size_t)ATT_TK * (A + 1) + (size_t)ATT_TK * Dv +
                                      (size_t)ATT_ROWS * A + 4 * ATT_TK);
  return adell_att_launch(adell_attention_fwd_kernel, a, BH, lds, (hipStream_t)stream);
}

extern "C" int adell_attention_bwd(const float* q, const float* k, const float* v,
                                   const float* bias, int nbias, const float* out,
                                   const float* dout, const float* lse, int BH, int T, int A,
                                   int Dv, float scale, float drop_p, unsigned long long seed,
                                   unsigned int rng_offset, float* dq, float* dk, float* dv,
                                   void* stream) {
  int rc = adell_att_check(BH, T, A, Dv, nbias, bias);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv, "attention_bwd: null pointer");
  AttArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = out; a.dout = dout; a.lse = lse;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.T = T; a.A = A; a.Dv = Dv; a.nbias = nbias > 0 ? nbias : 1; a.scale = scale;
  ADELL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention: bad dropout probability");
  a.drop_p = drop_p; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.rng_offset = rng_offset;
  if (adell_att_mfma_ok(T, A, Dv) && !g_adell_tune.attn_nomfma) {
    adell_att_contiguous(&a);
    rc = adell_att_mfma_launch(1, a, BH, (hipStream_t)stream);
    if (rc != ADELL_OK) return rc;
    return adell_att_mfma_launch(2, a, BH, (hipStream_t)stream);
  }
  const size_t lds_q = sizeof(float) * ((size_t)ATT_TK * (A + 1) + (size_t)ATT_TK * (Dv + 1) +
                                        (size_t)ATT_ROWS * A + (size_t)ATT_ROWS * Dv + 4 * ATT_TK);
  rc = adell_att_launch(adell_attention_bwd_q_kernel, a, BH, lds_q, (hipStream_t)stream);
  if (rc != ADELL_OK) return rc;
  const size_t lds_kv = sizeof(float) * ((size_t)ATT_TK * (A + 1) + (size_t)ATT_TK * (Dv + 1) +
                                         (size_t)ATT_ROWS * A + (size_t)ATT_ROWS * Dv +
                                         8 * ATT_TK + 2 * ATT_TK);
  return adell_att_launch(adell_attention_bwd_kv_kernel, a, BH, lds_kv, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// The same attention with every operand addressed by (item, head, token row) element strides:
// sequence bh = b * H + h. Q / K / V can stay inside the packed [B][T][H][q | k | v] projection
// output, O / dO are read and written as [B][T][H * Dv] token rows, dQ / dK / dV land where the
// caller's next kernel wants them -- none of the permute / slice copies of
// linear_blocks.py:372-417 exist as launches. MFMA-shaped heads only (adell_attention_strided_ok).
// ---------------------------------------------------------------------------------------------
extern "C" int adell_attention_strided_ok(int T, int A, int Dv) {
  return adell_att_mfma_ok(T, A, Dv) && !g_adell_tune.attn_nomfma ? 1 : 0;
}

extern "C" int adell_attention_fwd_strided(const float* q, const float* k, const float* v,
                                           const float* bias, int nbias, int B, int H, int T,
                                           int A, int Dv, const long* strides, float scale,
                                           float drop_p, unsigned long long seed,
                                           unsigned int rng_offset, float* out, float* lse,
                                           void* stream) {
  ADELL_REQUIRE(B > 0 && H > 0 && strides, "attention_fwd_strided: bad dims");
  int rc = adell_att_check(B * H, T, A, Dv, nbias, bias);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(adell_attention_strided_ok(T, A, Dv),
                "attention_fwd_strided: head dims (%d, %d) at %d tokens have no MFMA instance", A, Dv, T);
  ADELL_REQUIRE(lse, "attention_fwd_strided: null pointer");
  AttArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.out = out; a.lse_out = lse;
  a.T = T; a.A = A; a.Dv = Dv; a.nbias = nbias > 0 ? nbias : 1; a.scale = scale; a.H = H;
  ADELL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention: bad dropout probability");
  a.drop_p = drop_p; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.rng_offset = rng_offset;
  if ((rc = adell_att_stride("q", q, strides, A, &a.sq)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("k", k, strides + 3, A, &a.sk)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("v", v, strides + 6, Dv, &a.sv)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("out", out, strides + 9, Dv, &a.so)) != ADELL_OK) return rc;
  return adell_att_mfma_launch(0, a, B * H, (hipStream_t)stream);
}

extern "C" int adell_attention_bwd_strided(const float* q, const float* k, const float* v,
                                           const float* bias, int nbias, const float* out,
                                           const float* dout, const float* lse, int B, int H,
                                           int T, int A, int Dv, const long* strides, float scale,
                                           float drop_p, unsigned long long seed,
                                           unsigned int rng_offset, float* dq, float* dk,
                                           float* dv, void* stream) {
  ADELL_REQUIRE(B > 0 && H > 0 && strides, "attention_bwd_strided: bad dims");
  int rc = adell_att_check(B * H, T, A, Dv, nbias, bias);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(adell_attention_strided_ok(T, A, Dv),
                "attention_bwd_strided: head dims (%d, %d) at %d tokens have no MFMA instance", A, Dv, T);
  ADELL_REQUIRE(lse, "attention_bwd_strided: null pointer");
  AttArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = out; a.dout = dout; a.lse = lse;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.T = T; a.A = A; a.Dv = Dv; a.nbias = nbias > 0 ? nbias : 1; a.scale = scale; a.H = H;
  ADELL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention: bad dropout probability");
  a.drop_p = drop_p; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.rng_offset = rng_offset;
  if ((rc = adell_att_stride("q", q, strides, A, &a.sq)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("k", k, strides + 3, A, &a.sk)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("v", v, strides + 6, Dv, &a.sv)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("out", out, strides + 9, Dv, &a.so)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("dout", dout, strides + 12, Dv, &a.sg)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("dq", dq, strides + 15, A, &a.sdq)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("dk", dk, strides + 18, A, &a.sdk)) != ADELL_OK) return rc;
  if ((rc = adell_att_stride("dv", dv, strides + 21, Dv, &a.sdv)) != ADELL_OK) return rc;
  rc = adell_att_mfma_launch(1, a, B * H, (hipStream_t)stream);
  if (rc != ADELL_OK) return rc;
  return adell_att_mfma_launch(2, a, B * H, (hipStream_t)stream);
}
