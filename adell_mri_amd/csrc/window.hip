// Kernels of the shifted-window (SWIN) token path (adell_mri/modules/layers/vit.py:33-45,
// 95-129, 1005-1256; linear_blocks.py:358-417 with window_size set):
//   * adell_gather_nd: every einops rearrange / torch.roll of that path as ONE gather --
//     window partition (+ cyclic shift), its inverse, space-to-depth (einops_rescale);
//   * adell_layernorm_rows_fwd/bwd: LayerNorm over short rows (C <= 512: the 2..32-channel
//     per-voxel norms, 4..32-wide per-head q/k norms, 128..2048-feature token norms of
//     SWIN), several rows per wave, strided input rows so the q / k slices of a QKV buffer
//     are normalised in place of a copy;
//   * adell_winattn_fwd/bwd: attention inside windows of T <= 64 tokens with the relative
//     position bias, the shifted-window mask and attention dropout -- one thread per
//     (window, head, query row); HBM/latency bound (QK^T is 2*T*A flops per row).
#include "common.h"

ADELL_RNG_STEP_DEFINE(window)

// ---------------------------------------------------------------------------
// out (contiguous over sizes[0..nd)) = in[offset(coords)]. Each out dim d feeds input axis
// axis[d] with multiplier mult[d]; input axis a has extent / stride / cyclic shift:
//   in_coord[a] = (sum_d coords[d] * mult[d] + shift[a]) mod extent[a].
// ---------------------------------------------------------------------------
#define ADELL_GATHER_MAX 12
struct GatherArgs {
  const float* in;
  float* out;
  int nd, na;
  int size[ADELL_GATHER_MAX];
  int axis[ADELL_GATHER_MAX];
  long mult[ADELL_GATHER_MAX];
  long extent[ADELL_GATHER_MAX];
  long stride[ADELL_GATHER_MAX];
  long shift[ADELL_GATHER_MAX];
  long total;   // in units of `vec` elements
  int vec;      // 1 or 4 (the innermost out dim is contiguous in the input)
};

__global__ __launch_bounds__(256) void adell_gather_nd_kernel(GatherArgs a) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < a.total; i += (long)gridDim.x * 256L) {
    long rem = i * a.vec;
    long coord[ADELL_GATHER_MAX];
#pragma unroll
    for (int q = 0; q < ADELL_GATHER_MAX; ++q) coord[q] = 0;
    for (int d = a.nd - 1; d >= 0; --d) {
      const long c = rem % a.size[d];
      rem /= a.size[d];
      const int ax = a.axis[d];
      const long v = c * a.mult[d];
#pragma unroll
      for (int q = 0; q < ADELL_GATHER_MAX; ++q)
        if (q == ax) coord[q] += v;
    }
    long off = 0;
#pragma unroll
    for (int q = 0; q < ADELL_GATHER_MAX; ++q)
      if (q < a.na) {
        long c = coord[q] + a.shift[q];
        if (a.shift[q] != 0) {
          c %= a.extent[q];
          if (c < 0) c += a.extent[q];
        }
        off += c * a.stride[q];
      }
    if (a.vec == 4)
      reinterpret_cast<f32x4*>(a.out)[i] = *reinterpret_cast<const f32x4*>(a.in + off);
    else
      a.out[i] = a.in[off];
  }
}

// The same gather in 32-bit arithmetic (every problem of the SWIN path: < 2^31 elements on either
// side). The generic kernel above spends ~60 instructions per 64-bit division and 12 selects per
// dimension on coordinates it rarely needs (0.66 TB/s on the 2-channel window partition of config 5):
//   * divisions by the (launch-constant) sizes are multiply-high + shift (host-made magic numbers);
//   * a dimension that feeds an UNSHIFTED input axis contributes coordinate * (mult * stride)
//     straight to the offset -- no per-axis coordinate; only the (at most four) cyclically shifted
//     axes keep one, wrapped by one conditional subtraction (0 <= shift < extent);
//   * adjacent unshifted dimensions whose offsets nest (lin[d] == size[d+1] * lin[d+1]) are merged on
//     the host, so 'z c' runs of a channels-last volume move as 16- or 8-byte pieces.
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
struct Gather32Args {
  const float* in;
  float* out;
  int nd, ns;
  unsigned size[ADELL_GATHER_MAX], magic[ADELL_GATHER_MAX], shr[ADELL_GATHER_MAX];
  unsigned lin[ADELL_GATHER_MAX];       // offset per coordinate (unshifted axes), else 0
  int sidx[ADELL_GATHER_MAX];           // shifted axis this dimension feeds, or -1
  unsigned smult[ADELL_GATHER_MAX];
  unsigned sext[4], sshift[4], sstride[4];
  unsigned total;                       // in units of `vec` elements
};

template <int VEC>
__global__ __launch_bounds__(256) void adell_gather_nd32_kernel(Gather32Args a) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < a.total; i += gridDim.x * 256u) {
    unsigned rem = i * VEC, off = 0, sc[4] = {0u, 0u, 0u, 0u};
    for (int d = a.nd - 1; d >= 0; --d) {
      const unsigned q = (__umulhi(rem, a.magic[d]) + rem) >> a.shr[d];
      const unsigned c = rem - q * a.size[d];
      rem = q;
      const int sx = a.sidx[d];                     // launch constants: scalar branches
      if (sx < 0) {
        off += c * a.lin[d];
      } else {
        const unsigned v = c * a.smult[d];
        sc[0] += sx == 0 ? v : 0u;
        sc[1] += sx == 1 ? v : 0u;
        sc[2] += sx == 2 ? v : 0u;
        sc[3] += sx == 3 ? v : 0u;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < a.ns) {
        unsigned c = sc[q] + a.sshift[q];
        c = c >= a.sext[q] ? c - a.sext[q] : c;
        off += c * a.sstride[q];
      }
    if constexpr (VEC == 4)
      reinterpret_cast<f32x4*>(a.out)[i] = *reinterpret_cast<const f32x4*>(a.in + off);
    else if constexpr (VEC == 2)
      reinterpret_cast<g_f32x2*>(a.out)[i] = *reinterpret_cast<const g_f32x2*>(a.in + off);
    else
      a.out[i] = a.in[off];
  }
}

// plans the 32-bit form; false when the problem does not fit it
static bool adell_gather32_plan(const float* in, float* out, int nd, const int* sizes, const int* axis,
                                const long* mult, int na, const long* extent, const long* stride,
                                const long* shift, Gather32Args* a, int* vec_out) {
  long total = 1, in_span = 1;
  for (int d = 0; d < nd; ++d) total *= sizes[d];
  for (int q = 0; q < na; ++q) in_span += (extent[q] - 1) * (stride[q] < 0 ? -stride[q] : stride[q]);
  if (total >= (1L << 31) || in_span >= (1L << 31)) return false;
  // shifted axes (shift normalised into [0, extent)); at most four; negative strides not taken
  int smap[ADELL_GATHER_MAX];
  a->ns = 0;
  for (int q = 0; q < na; ++q) {
    smap[q] = -1;
    if (stride[q] < 0) return false;
    long sh = shift[q] % extent[q];
    if (sh < 0) sh += extent[q];
    if (sh != 0) {
      if (a->ns == 4) return false;
      // the wrap is ONE conditional subtraction: the axis coordinate must stay below its extent
      long mx = 0;
      for (int d = 0; d < nd; ++d)
        if (axis[d] == q) { if (mult[d] < 0) return false; mx += (long)(sizes[d] - 1) * mult[d]; }
      if (mx >= extent[q]) return false;
      smap[q] = a->ns;
      a->sext[a->ns] = (unsigned)extent[q];
      a->sshift[a->ns] = (unsigned)sh;
      a->sstride[a->ns] = (unsigned)stride[q];
      ++a->ns;
    }
  }
  for (int q = a->ns; q < 4; ++q) a->sext[q] = 1u, a->sshift[q] = 0u, a->sstride[q] = 0u;
  // dimensions, innermost last; merge nested unshifted neighbours
  long sz[ADELL_GATHER_MAX], lin[ADELL_GATHER_MAX], sm[ADELL_GATHER_MAX];
  int sx[ADELL_GATHER_MAX], n = 0;
  for (int d = 0; d < nd; ++d) {
    const int ax = axis[d];
    const long l = smap[ax] < 0 ? mult[d] * stride[ax] : 0;
    if (smap[ax] < 0 && l < 0) return false;
    if (sizes[d] == 1) continue;
    if (n > 0 && smap[ax] < 0 && sx[n - 1] < 0 && lin[n - 1] == (long)sizes[d] * l) {
      sz[n - 1] *= sizes[d];
      lin[n - 1] = l;
      continue;
    }
    sz[n] = sizes[d]; lin[n] = l; sx[n] = smap[ax]; sm[n] = smap[ax] < 0 ? 0 : mult[d];
    ++n;
  }
  if (n == 0) { sz[0] = 1; lin[0] = 0; sx[0] = -1; sm[0] = 0; n = 1; }
  // vector width: the innermost dimension walks unit-stride, unshifted memory
  int vec = 1;
  if (sx[n - 1] < 0 && lin[n - 1] == 1) {
    for (int v = 4; v >= 2 && vec == 1; v >>= 1) {
      bool ok = sz[n - 1] % v == 0 && ((uintptr_t)in % (4 * v)) == 0 && ((uintptr_t)out % (4 * v)) == 0;
      for (int d = 0; d + 1 < n && ok; ++d)
        if (sx[d] < 0 && lin[d] % v != 0) ok = false;
      for (int q = 0; q < a->ns && ok; ++q)
        if (a->sstride[q] % v != 0) ok = false;
      if (ok) vec = v;
    }
  }
  a->in = in; a->out = out; a->nd = n;
  for (int d = 0; d < ADELL_GATHER_MAX; ++d) {
    const unsigned dv = d < n ? (unsigned)sz[d] : 1u;
    int l = 0;
    while ((1ul << l) < dv) ++l;
    a->size[d] = dv;
    a->magic[d] = (unsigned)(((1ul << 32) * ((1ul << l) - dv)) / dv + 1);
    a->shr[d] = (unsigned)l;
    a->lin[d] = d < n ? (unsigned)lin[d] : 0u;
    a->sidx[d] = d < n ? sx[d] : -1;
    a->smult[d] = d < n ? (unsigned)sm[d] : 0u;
  }
  a->total = (unsigned)(total / vec);
  *vec_out = vec;
  return true;
}

extern "C" int adell_gather_nd(const float* in, float* out, int nd, const int* sizes,
                               const int* axis, const long* mult, int na, const long* extent,
                               const long* stride, const long* shift, void* stream) {
  ADELL_REQUIRE(in && out && sizes && axis && mult && extent && stride && shift,
                "gather_nd: null pointer");
  ADELL_REQUIRE(nd >= 1 && nd <= ADELL_GATHER_MAX && na >= 1 && na <= ADELL_GATHER_MAX,
                "gather_nd: at most 12 dims / axes");
  GatherArgs a;
  a.in = in; a.out = out; a.nd = nd; a.na = na;
  long total = 1;
  for (int d = 0; d < ADELL_GATHER_MAX; ++d) {
    a.size[d] = d < nd ? sizes[d] : 1;
    a.axis[d] = d < nd ? axis[d] : 0;
    a.mult[d] = d < nd ? mult[d] : 0;
    a.extent[d] = d < na ? extent[d] : 1;
    a.stride[d] = d < na ? stride[d] : 0;
    a.shift[d] = d < na ? shift[d] : 0;
    if (d < nd) {
      ADELL_REQUIRE(sizes[d] > 0 && axis[d] >= 0 && axis[d] < na, "gather_nd: bad dim");
      total *= sizes[d];
    }
  }
  // the largest reachable coordinate of every axis must stay inside it (or wrap)
  for (int q = 0; q < na; ++q) {
    long mx = 0;
    for (int d = 0; d < nd; ++d)
      if (axis[d] == q) mx += (long)(sizes[d] - 1) * mult[d];
    ADELL_REQUIRE(extent[q] > 0 && mx < extent[q], "gather_nd: coordinates exceed the input axis");
  }
  // the 32-bit form (fast divisions, merged runs) whenever the problem fits it
  {
    Gather32Args g;
    int v = 1;
    if (adell_gather32_plan(in, out, nd, sizes, axis, mult, na, extent, stride, shift, &g, &v)) {
      unsigned blocks = (g.total + 255u) / 256u;
      if (blocks > 16384u) blocks = 16384u;
      if (blocks < 1u) blocks = 1u;
      hipStream_t st = (hipStream_t)stream;
      if (v == 4) hipLaunchKernelGGL(adell_gather_nd32_kernel<4>, dim3(blocks), dim3(256), 0, st, g);
      else if (v == 2) hipLaunchKernelGGL(adell_gather_nd32_kernel<2>, dim3(blocks), dim3(256), 0, st, g);
      else hipLaunchKernelGGL(adell_gather_nd32_kernel<1>, dim3(blocks), dim3(256), 0, st, g);
      ADELL_CHECK_HIP(hipGetLastError());
      return ADELL_OK;
    }
  }
  // 16-byte path: innermost out dim walks a unit-stride, unshifted axis of its own
  const int last = nd - 1, la = axis[last];
  bool vec = (sizes[last] % 4 == 0) && mult[last] == 1 && stride[la] == 1 && shift[la] == 0 &&
             ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  for (int d = 0; d < last && vec; ++d)
    if (axis[d] == la && mult[d] % 4 != 0) vec = false;
  for (int q = 0; q < na && vec; ++q)
    if (q != la && stride[q] % 4 != 0) vec = false;
  a.vec = vec ? 4 : 1;
  a.total = total / a.vec;
  long blocks = (a.total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_gather_nd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// LayerNorm over rows of C <= 512 values. A row is handled by LPR lanes (power of two,
// E = ceil(C / LPR) <= 8 values per lane), so a wave covers 64 / LPR rows at once. Row r
// of the input starts at (r / inner) * so + (r % inner) * si (contiguous rows: inner = 1,
// so = C); y / dy / mean / rstd are contiguous.
// ---------------------------------------------------------------------------
#define ADELL_LNR_MAXE 8
struct LnRowsArgs {
  const float* x;
  const float* dy;
  const float* gamma;
  const float* beta;
  const float* mean_in;
  const float* rstd_in;
  float* y;      // forward: y; backward: dx (strided with dso / dsi)
  float* mean;
  float* rstd;
  float* part;   // backward: [blocks][2][C] partial dgamma / dbeta (or null)
  long rows;
  int C, inner, lpr, rows_per_block;
  long so, si, dso, dsi;
  float eps;
};

__device__ __forceinline__ float adell_group_sum(float v, int lpr) {
  for (int m = lpr >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ __launch_bounds__(256) void adell_layernorm_rows_fwd_kernel(LnRowsArgs a) {
  const int lpr = a.lpr, gl = threadIdx.x % lpr, grp = threadIdx.x / lpr, ngrp = 256 / lpr;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.rows) r1 = a.rows;
  for (long rb = r0; rb < r1; rb += ngrp) {
    const long row = rb + grp;
    const bool live = row < r1;
    const float* xr = a.x + (live ? (row / a.inner) * a.so + (row % a.inner) * a.si : 0);
    float v[ADELL_LNR_MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
      const int c = gl + e * lpr;
      v[e] = (live && c < a.C) ? xr[c] : 0.f;
      s += v[e];
    }
    const float m = adell_group_sum(s, lpr) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
      const int c = gl + e * lpr;
      const float d = (c < a.C) ? v[e] - m : 0.f;
      q += d * d;
    }
    const float r = rsqrtf(adell_group_sum(q, lpr) / (float)a.C + a.eps);
    if (!live) continue;
    float* yr = a.y + row * a.C;
#pragma unroll
    for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
      const int c = gl + e * lpr;
      if (c < a.C) {
        float t = (v[e] - m) * r;
        if (a.gamma) t *= a.gamma[c];
        if (a.beta) t += a.beta[c];
        yr[c] = t;
      }
    }
    if (gl == 0) {
      a.mean[row] = m;
      a.rstd[row] = r;
    }
  }
}

__global__ __launch_bounds__(256) void adell_layernorm_rows_bwd_kernel(LnRowsArgs a) {
  extern __shared__ float sh[];  // [ngrp][2][C]
  const int lpr = a.lpr, gl = threadIdx.x % lpr, grp = threadIdx.x / lpr, ngrp = 256 / lpr;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.rows) r1 = a.rows;
  float dg[ADELL_LNR_MAXE], db[ADELL_LNR_MAXE];
#pragma unroll
  for (int e = 0; e < ADELL_LNR_MAXE; ++e) dg[e] = db[e] = 0.f;
  for (long rb = r0; rb < r1; rb += ngrp) {
    const long row = rb + grp;
    const bool live = row < r1;
    const float* xr = a.x + (live ? (row / a.inner) * a.so + (row % a.inner) * a.si : 0);
    const float* gr = a.dy + (live ? row * a.C : 0);
    const float m = live ? a.mean_in[row] : 0.f, r = live ? a.rstd_in[row] : 0.f;
    float xh[ADELL_LNR_MAXE], g[ADELL_LNR_MAXE];
    float sa = 0.f, sbb = 0.f;
#pragma unroll
    for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
      const int c = gl + e * lpr;
      const bool ok = live && c < a.C;
      xh[e] = ok ? (xr[c] - m) * r : 0.f;
      const float d = ok ? gr[c] : 0.f;
      g[e] = d * ((a.gamma && ok) ? a.gamma[c] : 1.f);
      sa += g[e];
      sbb += g[e] * xh[e];
      dg[e] += d * xh[e];
      db[e] += d;
    }
    sa = adell_group_sum(sa, lpr) / (float)a.C;
    sbb = adell_group_sum(sbb, lpr) / (float)a.C;
    if (!live) continue;
    float* dr = a.y + (row / a.inner) * a.dso + (row % a.inner) * a.dsi;
#pragma unroll
    for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
      const int c = gl + e * lpr;
      if (c < a.C) dr[c] = r * (g[e] - sa - xh[e] * sbb);
    }
  }
  if (!a.part) return;
#pragma unroll
  for (int e = 0; e < ADELL_LNR_MAXE; ++e) {
    const int c = gl + e * lpr;
    if (c < a.C) {
      sh[(grp * 2) * a.C + c] = dg[e];
      sh[(grp * 2 + 1) * a.C + c] = db[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * a.C; c += 256) {
    const int which = c / a.C, cc = c % a.C;
    float s = 0.f;
    for (int k = 0; k < ngrp; ++k) s += sh[(k * 2 + which) * a.C + cc];
    a.part[(size_t)blockIdx.x * 2 * a.C + c] = s;
  }
}

static int adell_lnr_plan(long rows, int C, int* lpr, int* rpb, int* blocks) {
  int l = 1;
  while (l < 64 && (C + l - 1) / l > ADELL_LNR_MAXE) l <<= 1;
  if ((C + l - 1) / l > ADELL_LNR_MAXE) return 0;
  const int ngrp = 256 / l;
  long nb = (rows + ngrp - 1) / ngrp;  // one pass of the block per block at most ...
  if (nb > 2048) nb = 2048;             // ... capped: blocks loop over their chunk
  long per = (rows + nb - 1) / nb;
  per = (per + ngrp - 1) / ngrp * ngrp;
  *lpr = l;
  *rpb = (int)per;
  *blocks = (int)((rows + per - 1) / per);
  return 1;
}

extern "C" long adell_layernorm_rows_bwd_workspace(long rows, int C) {
  int lpr, rpb, blocks;
  if (rows <= 0 || C <= 0 || !adell_lnr_plan(rows, C, &lpr, &rpb, &blocks)) return 0;
  return (long)blocks * 2 * C * (long)sizeof(float);
}

extern "C" int adell_layernorm_rows_fwd(const float* x, long rows, int C, int inner, long so,
                                        long si, const float* gamma, const float* beta, float eps,
                                        float* y, float* mean, float* rstd, void* stream) {
  ADELL_REQUIRE(x && y && mean && rstd, "layernorm_rows_fwd: null pointer");
  ADELL_REQUIRE(rows > 0 && C > 0 && inner > 0, "layernorm_rows_fwd: bad dims");
  LnRowsArgs a = {};
  int blocks;
  ADELL_REQUIRE(adell_lnr_plan(rows, C, &a.lpr, &a.rows_per_block, &blocks),
                "layernorm_rows: C must be <= 512");
  a.x = x; a.gamma = gamma; a.beta = beta; a.y = y; a.mean = mean; a.rstd = rstd;
  a.rows = rows; a.C = C; a.inner = inner; a.so = so; a.si = si; a.eps = eps;
  hipLaunchKernelGGL(adell_layernorm_rows_fwd_kernel, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// dgamma / dbeta = sum of the per-block partial rows [nb][2][C]: 16 columns x 64 row groups per
// block (the sums in double, groups added in order: one fixed tree). With 64 columns x 16 groups
// a C = 96 layer had THREE blocks walking 2 048 partial rows: 25 us per call, 17 calls per VICReg
// ConvNeXt step.
__global__ __launch_bounds__(1024) void adell_lnr_final_kernel(const float* __restrict__ part,
                                                               int nb, int C,
                                                               float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
  __shared__ double sh[64][16];
  const int cl = threadIdx.x & 15, vl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < 2 * C) {
    int b = vl;
    for (; b + 64 < nb; b += 128) {
      s0 += (double)part[(size_t)b * 2 * C + c];
      s1 += (double)part[(size_t)(b + 64) * 2 * C + c];
    }
    if (b < nb) s0 += (double)part[(size_t)b * 2 * C + c];
  }
  sh[vl][cl] = s0 + s1;
  __syncthreads();
  if (vl != 0 || c >= 2 * C) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 64; ++k) s += sh[k][cl];
  if (c < C) {
    if (dgamma) dgamma[c] = (float)s;
  } else if (dbeta) {
    dbeta[c - C] = (float)s;
  }
}

// dx is written at row offsets (r / inner) * dso + (r % inner) * dsi (contiguous: dso = C).
extern "C" int adell_layernorm_rows_bwd(const float* x, const float* dy, const float* gamma,
                                        const float* mean, const float* rstd, long rows, int C,
                                        int inner, long so, long si, float* dx, long dso, long dsi,
                                        float* dgamma, float* dbeta, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(x && dy && mean && rstd && dx, "layernorm_rows_bwd: null pointer");
  ADELL_REQUIRE(rows > 0 && C > 0 && inner > 0, "layernorm_rows_bwd: bad dims");
  LnRowsArgs a = {};
  int blocks;
  ADELL_REQUIRE(adell_lnr_plan(rows, C, &a.lpr, &a.rows_per_block, &blocks),
                "layernorm_rows: C must be <= 512");
  const bool want = dgamma || dbeta;
  if (want)
    ADELL_REQUIRE(workspace && (long)workspace_bytes >= adell_layernorm_rows_bwd_workspace(rows, C),
                  "layernorm_rows_bwd: workspace too small");
  a.x = x; a.dy = dy; a.gamma = gamma; a.mean_in = mean; a.rstd_in = rstd; a.y = dx;
  a.part = want ? (float*)workspace : nullptr;
  a.rows = rows; a.C = C; a.inner = inner; a.so = so; a.si = si; a.dso = dso; a.dsi = dsi;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = want ? (size_t)(256 / a.lpr) * 2 * C * sizeof(float) : 0;
  hipLaunchKernelGGL(adell_layernorm_rows_bwd_kernel, dim3(blocks), dim3(256), lds, st, a);
  if (want)
    hipLaunchKernelGGL(adell_lnr_final_kernel, dim3(adell_cdiv(2 * C, 16)), dim3(1024), 0, st,
                       (const float*)workspace, blocks, C, dgamma, dbeta);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Window attention. Tokens are rows t = w * T + i (window w, position i); Q, K, O are
// [tokens][H][A|Dv] contiguous, V is read (and dV written) in place inside a QKV buffer:
// element (t, h, d) at v[t * v_ts + h * v_hs + d].
//   S = scale * Q K^T + rel[h] + mask[w % n_mask];  P = softmax(S);  O = dropout(P) V
// One thread per (w, h, i). lse[(w*H + h)*T + i] = logsumexp(S_i) is kept for the backward.
// ---------------------------------------------------------------------------
struct WinAttnArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* rel;    // [H][T][T] or null
  const float* mask;   // [n_mask][T][T] or null
  const float* o;      // backward
  const float* dout;   // backward
  const float* lse_in; // backward
  float* out;          // forward: O
  float* lse;          // forward
  float* dq;
  float* dk;
  float* dv;           // strided like v
  float* ds;           // [W][H][T][T] or null (gradient of the additive bias)
  long W;
  int H, T, A, Dv, n_mask;
  long v_ts, v_hs;
  float scale, drop_p;
  uint32_t seed_lo, seed_hi, rng_offset;
};

__device__ __forceinline__ bool adell_wa_keep(const WinAttnArgs& a, long seq, int i, int j) {
  if (a.drop_p <= 0.f) return true;
  const unsigned long e = ((unsigned long)seq * a.T + i) * a.T + j;
  const uint4 r = adell_philox4((uint32_t)(e >> 2), (uint32_t)(e >> 34), a.rng_offset + g_adell_rng_step, 1u,
                                a.seed_lo, a.seed_hi);
  const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
  return (float)(rr[e & 3] >> 8) * (1.0f / 16777216.0f) >= a.drop_p;
}

template <int DMAX>
__device__ __forceinline__ float adell_wa_score(const WinAttnArgs& a, const float (&qi)[DMAX],
                                                const float* __restrict__ kj, int h, long w, int i,
                                                int j) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DMAX; ++d)
    if (d < a.A) s = fmaf(qi[d], kj[d], s);
  s *= a.scale;
  if (a.rel) s += a.rel[((size_t)h * a.T + i) * a.T + j];
  if (a.mask) s += a.mask[((size_t)(w % a.n_mask) * a.T + i) * a.T + j];
  return s;
}

template <int DMAX>
__global__ __launch_bounds__(256) void adell_winattn_fwd_kernel(WinAttnArgs a) {
  const long total = a.W * a.H * a.T;
  const long g = blockIdx.x * 256L + threadIdx.x;
  if (g >= total) return;
  const int i = (int)(g % a.T);
  const long seq = g / a.T;
  const int h = (int)(seq % a.H);
  const long w = seq / a.H;
  const long tok0 = w * a.T;
  float qi[DMAX];
  const float* qp = a.q + ((tok0 + i) * a.H + h) * a.A;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) qi[d] = d < a.A ? qp[d] : 0.f;
  float m = -INFINITY;
  for (int j = 0; j < a.T; ++j)
    m = fmaxf(m, adell_wa_score<DMAX>(a, qi, a.k + ((tok0 + j) * a.H + h) * a.A, h, w, i, j));
  float l = 0.f;
  for (int j = 0; j < a.T; ++j)
    l += __expf(adell_wa_score<DMAX>(a, qi, a.k + ((tok0 + j) * a.H + h) * a.A, h, w, i, j) - m);
  const float inv = 1.f / l, keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  float o[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) o[d] = 0.f;
  for (int j = 0; j < a.T; ++j) {
    float p = __expf(adell_wa_score<DMAX>(a, qi, a.k + ((tok0 + j) * a.H + h) * a.A, h, w, i, j) - m) *
              inv;
    p = adell_wa_keep(a, seq, i, j) ? p * keep_scale : 0.f;
    const float* vp = a.v + (tok0 + j) * a.v_ts + h * a.v_hs;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < a.Dv) o[d] = fmaf(p, vp[d], o[d]);
  }
  float* op = a.out + ((tok0 + i) * a.H + h) * a.Dv;
#pragma unroll
  for (int d = 0; d < DMAX; ++d)
    if (d < a.Dv) op[d] = o[d];
  a.lse[g] = m + __logf(l);
}

// thread (w, h, r): dQ of query r, then dK / dV of key r (scores are recomputed).
template <int DMAX>
__global__ __launch_bounds__(256) void adell_winattn_bwd_kernel(WinAttnArgs a) {
  const long total = a.W * a.H * a.T;
  const long g = blockIdx.x * 256L + threadIdx.x;
  if (g >= total) return;
  const int r = (int)(g % a.T);
  const long seq = g / a.T;
  const int h = (int)(seq % a.H);
  const long w = seq / a.H;
  const long tok0 = w * a.T;
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  float qr[DMAX], kr[DMAX], gr[DMAX], acc[DMAX];
  {
    const float* qp = a.q + ((tok0 + r) * a.H + h) * a.A;
    const float* kp = a.k + ((tok0 + r) * a.H + h) * a.A;
    const float* gp = a.dout + ((tok0 + r) * a.H + h) * a.Dv;
    const float* op = a.o + ((tok0 + r) * a.H + h) * a.Dv;
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      qr[d] = d < a.A ? qp[d] : 0.f;
      kr[d] = d < a.A ? kp[d] : 0.f;
      gr[d] = d < a.Dv ? gp[d] : 0.f;
      if (d < a.Dv) D = fmaf(gr[d], op[d], D);
      acc[d] = 0.f;
    }
    const float lse = a.lse_in[g];
    for (int j = 0; j < a.T; ++j) {
      const float* kj = a.k + ((tok0 + j) * a.H + h) * a.A;
      const float p = __expf(adell_wa_score<DMAX>(a, qr, kj, h, w, r, j) - lse);
      const float* vp = a.v + (tok0 + j) * a.v_ts + h * a.v_hs;
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < a.Dv) dp = fmaf(gr[d], vp[d], dp);
      dp = adell_wa_keep(a, seq, r, j) ? dp * keep_scale : 0.f;
      const float dsv = p * (dp - D);
      if (a.ds) a.ds[(seq * a.T + r) * a.T + j] = dsv;
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < a.A) acc[d] = fmaf(dsv * a.scale, kj[d], acc[d]);
    }
    float* dqp = a.dq + ((tok0 + r) * a.H + h) * a.A;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < a.A) dqp[d] = acc[d];
  }
  float dkr[DMAX], dvr[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) dkr[d] = dvr[d] = 0.f;
  const float* vr = a.v + (tok0 + r) * a.v_ts + h * a.v_hs;
  for (int i = 0; i < a.T; ++i) {
    const float* qp = a.q + ((tok0 + i) * a.H + h) * a.A;
    const float* gp = a.dout + ((tok0 + i) * a.H + h) * a.Dv;
    const float* op = a.o + ((tok0 + i) * a.H + h) * a.Dv;
    float qi[DMAX];
    float D = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      qi[d] = d < a.A ? qp[d] : 0.f;
      if (d < a.Dv) {
        D = fmaf(gp[d], op[d], D);
        dp = fmaf(gp[d], vr[d], dp);
      }
    }
    const float p = __expf(adell_wa_score<DMAX>(a, qi, kr, h, w, i, r) - a.lse_in[seq * a.T + i]);
    const bool keep = adell_wa_keep(a, seq, i, r);
    const float pt = keep ? p * keep_scale : 0.f;
    const float dsv = p * ((keep ? dp * keep_scale : 0.f) - D);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      if (d < a.A) dkr[d] = fmaf(dsv * a.scale, qi[d], dkr[d]);
      if (d < a.Dv) dvr[d] = fmaf(pt, gp[d], dvr[d]);
    }
  }
  float* dkp = a.dk + ((tok0 + r) * a.H + h) * a.A;
  float* dvp = a.dv + (tok0 + r) * a.v_ts + h * a.v_hs;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    if (d < a.A) dkp[d] = dkr[d];
    if (d < a.Dv) dvp[d] = dvr[d];
  }
}

static int adell_wa_check(const WinAttnArgs& a) {
  ADELL_REQUIRE(a.W > 0 && a.H > 0 && a.T > 0 && a.T <= 64, "winattn: need 1 <= T <= 64");
  ADELL_REQUIRE(a.A > 0 && a.A <= 32 && a.Dv > 0 && a.Dv <= 32, "winattn: head dims must be <= 32");
  ADELL_REQUIRE(!a.mask || a.n_mask > 0, "winattn: mask needs n_mask > 0");
  ADELL_REQUIRE(a.drop_p >= 0.f && a.drop_p < 1.f, "winattn: bad dropout probability");
  return ADELL_OK;
}

#define ADELL_WA_LAUNCH(KERN, a, st)                                                         \
  do {                                                                                       \
    const long total_ = (a).W * (a).H * (a).T;                                               \
    const unsigned blocks_ = (unsigned)((total_ + 255) / 256);                               \
    const int dm_ = (a).A > (a).Dv ? (a).A : (a).Dv;                                         \
    if (dm_ <= 4) hipLaunchKernelGGL(KERN<4>, dim3(blocks_), dim3(256), 0, st, a);           \
    else if (dm_ <= 8) hipLaunchKernelGGL(KERN<8>, dim3(blocks_), dim3(256), 0, st, a);      \
    else if (dm_ <= 16) hipLaunchKernelGGL(KERN<16>, dim3(blocks_), dim3(256), 0, st, a);    \
    else hipLaunchKernelGGL(KERN<32>, dim3(blocks_), dim3(256), 0, st, a);                   \
  } while (0)

extern "C" int adell_winattn_fwd(const float* q, const float* k, const float* v, long v_ts,
                                 long v_hs, const float* rel, const float* mask, int n_mask,
                                 long W, int H, int T, int A, int Dv, float scale, float drop_p,
                                 unsigned long seed, unsigned rng_offset, float* out, float* lse,
                                 void* stream) {
  ADELL_REQUIRE(q && k && v && out && lse, "winattn_fwd: null pointer");
  WinAttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.rel = rel; a.mask = mask; a.out = out; a.lse = lse;
  a.W = W; a.H = H; a.T = T; a.A = A; a.Dv = Dv; a.n_mask = n_mask;
  a.v_ts = v_ts; a.v_hs = v_hs; a.scale = scale; a.drop_p = drop_p;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.rng_offset = rng_offset;
  int rc = adell_wa_check(a);
  if (rc != ADELL_OK) return rc;
  ADELL_WA_LAUNCH(adell_winattn_fwd_kernel, a, (hipStream_t)stream);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ds (optional, [W][H][T][T]): gradient of the additive bias of every window / head.
extern "C" int adell_winattn_bwd(const float* q, const float* k, const float* v, long v_ts,
                                 long v_hs, const float* rel, const float* mask, int n_mask,
                                 const float* o, const float* dout, const float* lse, long W,
                                 int H, int T, int A, int Dv, float scale, float drop_p,
                                 unsigned long seed, unsigned rng_offset, float* dq, float* dk,
                                 float* dv, float* ds, void* stream) {
  ADELL_REQUIRE(q && k && v && o && dout && lse && dq && dk && dv, "winattn_bwd: null pointer");
  WinAttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.rel = rel; a.mask = mask; a.o = o; a.dout = dout; a.lse_in = lse;
  a.dq = dq; a.dk = dk; a.dv = dv; a.ds = ds;
  a.W = W; a.H = H; a.T = T; a.A = A; a.Dv = Dv; a.n_mask = n_mask;
  a.v_ts = v_ts; a.v_hs = v_hs; a.scale = scale; a.drop_p = drop_p;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.rng_offset = rng_offset;
  int rc = adell_wa_check(a);
  if (rc != ADELL_OK) return rc;
  ADELL_WA_LAUNCH(adell_winattn_bwd_kernel, a, (hipStream_t)stream);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
