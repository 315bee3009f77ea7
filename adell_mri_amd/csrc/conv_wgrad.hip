// Backward-weight of the 3D convolution family on the fp32 MFMA.
//
//   dW[tap][ci][co] = sum_{n,v} X[n, S*v + tap - P][ci] * dY[n, v][co]
//
// GEMM view: M = ci (A rows), N = co (B cols), K = output voxels. Both operands
// are staged in their natural NDHWC layout (rows = voxels, channels inner), so
// the 32 lanes of an MFMA row/column group read 32 consecutive channels of one
// voxel (conflict-free ds_read_b32).
//
// Work split: blockIdx.x = region (an interleaved subset of the spatial bricks,
// split-K), blockIdx.y = (ci tile, co tile), blockIdx.z = tap group. A wave
// owns up to MAXJ "jobs" = (tap, 32x32 channel sub-tile) accumulators. Every
// region writes its partial dW slab to the workspace; a second kernel reduces
// the slabs in fixed order (deterministic) straight into torch's canonical
// [Cout][Cin][kD][kH][kW] layout.
#include "common.h"

struct WgradArgs {
  const float* x0;
  const float* x1;
  const float* dy;
  float* ws;          // [R][ntap][Cin][Cout]
  float* wsdb;        // [R][Cout] column sums of dY (bias gradient) or null
  int N, D, H, W;     // X dims
  int C0, C1, Cin, Cout;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int Do, Ho, Wo;     // dY dims
  int lTX, lTY, lTZ;
  int ntx, nty, ntz;  // bricks per dim (per batch item)
  int HX, HY, HZ;     // halo brick of X for one tap group
  int TCI, TCO;       // channel tile (32 or 64)
  int nci, nco;       // channel tiles
  int KDg;            // kz planes per tap group (1 or KD)
  int R;              // regions
  int vecx, vecy;
};

template <int MAXJ>
__global__ __launch_bounds__(256, 2) void adell_conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ float smem[];
  const int TV = 1 << (a.lTX + a.lTY + a.lTZ);
  const int HV = a.HX * a.HY * a.HZ;
  float* sX = smem;                  // [HV][TCI]
  float* sY = smem + HV * a.TCI;     // [TV][TCO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int region = blockIdx.x;
  const int cit = blockIdx.y % a.nci, cot = blockIdx.y / a.nci;
  const int ci0 = cit * a.TCI, co0 = cot * a.TCO;
  const int grp = blockIdx.z;
  const int kz0 = grp * a.KDg;
  const int tapsg = a.KDg * a.KH * a.KW;
  const int sci = a.TCI >> 5;
  const int nsub = sci * (a.TCO >> 5);

  // A wave owns ONE 32x32 channel sub-tile and every (4/nsub)-th tap of the
  // group: the dY operand is read once per k-step and shared by all its taps.
  const int sub = wave % nsub;
  const int cis = sub % sci, cos = sub / sci;
  const int tstride = 4 / nsub, tfirst = wave / nsub;
  int aoffj[MAXJ];
  bool jok[MAXJ];
#pragma unroll
  for (int q = 0; q < MAXJ; ++q) {
    int tl = tfirst + tstride * q;
    jok[q] = tl < tapsg;
    if (!jok[q]) tl = 0;  // duplicate work into a discarded accumulator
    const int kx = tl % a.KW, ky = (tl / a.KW) % a.KH, kz = tl / (a.KW * a.KH);
    aoffj[q] = ((kz * a.HY + ky) * a.HX + kx) * a.TCI + cis * 32 + lh * a.SW * a.TCI + li;
  }
  const int boff = cos * 32 + lh * a.TCO + li;

  f32x16 acc[MAXJ];
#pragma unroll
  for (int q = 0; q < MAXJ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int TX = 1 << a.lTX, TY = 1 << a.lTY;
  const long tiles_per_item = (long)a.ntx * a.nty * a.ntz;
  const long ntiles = tiles_per_item * a.N;
  const int HXY = a.HX * a.HY;
  const int c4x = a.TCI >> 2, c4y = a.TCO >> 2;
  const int ksteps = TV >> 1;
  const int lhx = a.lTX - 1;  // log2 of k-steps per brick row
  // bias gradient: thread t always stages channel quad t % c4y, so it can keep a
  // private running sum of everything it stages (only the ci-tile-0 / group-0 blocks)
  const bool do_db = a.wsdb != nullptr && cit == 0 && grp == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  for (long tile = region; tile < ntiles; tile += a.R) {
    long t = tile;
    const int tx = (int)(t % a.ntx); t /= a.ntx;
    const int ty = (int)(t % a.nty); t /= a.nty;
    const int tz = (int)(t % a.ntz);
    const int nb = (int)(t / a.ntz);
    const int ox0 = tx << a.lTX, oy0 = ty << a.lTY, oz0 = tz << a.lTZ;
    const int lx0 = ox0 * a.SW - a.PW, ly0 = oy0 * a.SH - a.PH,
              lz0 = oz0 * a.SD - a.PD + kz0;
    __syncthreads();
    // ---- stage X halo brick [HV][TCI] -------------------------------------
    for (int it = tid; it < HV * c4x; it += 256) {
      const int c4 = it % c4x, hv = it / c4x;
      const int hz = hv / HXY;
      const int rem = hv - hz * HXY;
      const int hy = rem / a.HX, hx = rem - hy * a.HX;
      const int rx = lx0 + hx, ry = ly0 + hy, rz = lz0 + hz;
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rx >= 0 && ry >= 0 && rz >= 0 && rx < a.W && ry < a.H && rz < a.D) {
        const size_t gv = ((size_t)(nb * a.D + rz) * a.H + ry) * a.W + rx;
        const int c = ci0 + 4 * c4;
        if (a.vecx) {
          if (c < a.C0)
            f = *reinterpret_cast<const float4*>(a.x0 + gv * a.C0 + c);
          else if (c < a.Cin)
            f = *reinterpret_cast<const float4*>(a.x1 + gv * a.C1 + (c - a.C0));
        } else {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cc = c + j;
            v[j] = cc < a.C0 ? a.x0[gv * a.C0 + cc]
                             : (cc < a.Cin ? a.x1[gv * a.C1 + (cc - a.C0)] : 0.f);
          }
          f = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      *reinterpret_cast<float4*>(&sX[hv * a.TCI + 4 * c4]) = f;
    }
    // ---- stage dY brick [TV][TCO] -----------------------------------------
    for (int it = tid; it < TV * c4y; it += 256) {
      const int c4 = it % c4y, v = it / c4y;
      const int x = ox0 + (v & (TX - 1));
      const int y = oy0 + ((v >> a.lTX) & (TY - 1));
      const int z = oz0 + (v >> (a.lTX + a.lTY));
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x < a.Wo && y < a.Ho && z < a.Do) {
        const size_t gv = ((size_t)(nb * a.Do + z) * a.Ho + y) * a.Wo + x;
        const int c = co0 + 4 * c4;
        const float* p = a.dy + gv * a.Cout + c;
        if (a.vecy) {
          if (c < a.Cout) f = *reinterpret_cast<const float4*>(p);
        } else {
          if (c + 0 < a.Cout) f.x = p[0];
          if (c + 1 < a.Cout) f.y = p[1];
          if (c + 2 < a.Cout) f.z = p[2];
          if (c + 3 < a.Cout) f.w = p[3];
        }
      }
      *reinterpret_cast<float4*>(&sY[v * a.TCO + 4 * c4]) = f;
      dbacc.x += f.x; dbacc.y += f.y; dbacc.z += f.z; dbacc.w += f.w;
    }
    __syncthreads();
    // ---- K loop over the brick's voxels, two per MFMA, register double-buffered
    auto a_row = [&](int ks) {
      const int xp = (ks & ((1 << lhx) - 1)) << 1;
      const int r = ks >> lhx;
      const int y = r & (TY - 1), z = r >> a.lTY;
      return (((z * a.SD) * a.HY + y * a.SH) * a.HX + xp * a.SW) * a.TCI;
    };
    float av0[MAXJ], av1[MAXJ], bv0, bv1;
    {
      const int ab = a_row(0);
#pragma unroll
      for (int q = 0; q < MAXJ; ++q) av0[q] = sX[ab + aoffj[q]];
      bv0 = sY[boff];
    }
    for (int ks = 0; ks < ksteps; ks += 2) {
      {
        const int ab = a_row(ks + 1);
#pragma unroll
        for (int q = 0; q < MAXJ; ++q) av1[q] = sX[ab + aoffj[q]];
        bv1 = sY[(ks + 1) * 2 * a.TCO + boff];
      }
#pragma unroll
      for (int q = 0; q < MAXJ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], bv0, acc[q], 0, 0, 0);
      if (ks + 2 < ksteps) {
        const int ab = a_row(ks + 2);
#pragma unroll
        for (int q = 0; q < MAXJ; ++q) av0[q] = sX[ab + aoffj[q]];
        bv0 = sY[(ks + 2) * 2 * a.TCO + boff];
      }
#pragma unroll
      for (int q = 0; q < MAXJ; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], bv1, acc[q], 0, 0, 0);
    }
  }

  if (do_db) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < c4y) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = tid; k < 256; k += c4y) {
        const float4 u = red[k];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const int c = co0 + 4 * tid;
      float* o = a.wsdb + (size_t)region * a.Cout + c;
      if (c + 0 < a.Cout) o[0] = t.x;
      if (c + 1 < a.Cout) o[1] = t.y;
      if (c + 2 < a.Cout) o[2] = t.z;
      if (c + 3 < a.Cout) o[3] = t.w;
    }
  }
  // ---- write this region's partial slab ------------------------------------
  const int ntap = a.KD * a.KH * a.KW;
  const int co = co0 + cos * 32 + li;
#pragma unroll
  for (int q = 0; q < MAXJ; ++q) {
    const int tl = tfirst + tstride * q;
    const int tap = kz0 * a.KH * a.KW + tl;
    const int cib = ci0 + cis * 32 + 4 * lh;
    float* base = a.ws + (((size_t)region * ntap + tap) * a.Cin + cib) * a.Cout + co;
    if (jok[q] && co < a.Cout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        if (cib + row < a.Cin) base[(size_t)row * a.Cout] = acc[q][r];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// out[(co*Cin + ci)*ntap + tap] = sum_r ws[r][tap][ci][co]   (fixed order).
// db[co] = sum_r wsdb[r][co] by one 1024-thread block: 64 channels x 16 slab lanes with 4
// independent partial sums each (a serial chain over R = 512 slabs costs > 100 us), combined in
// fixed order through LDS.
__device__ __forceinline__ void adell_db_fold(const float* __restrict__ wsdb,
                                              float* __restrict__ db, int R, int Cout) {
  __shared__ float shdb[16][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int c0 = 0; c0 < Cout; c0 += 64) {
    const int co = c0 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (co < Cout) {
      int r = q;
      for (; r + 48 < R; r += 64) {
        s0 += wsdb[(size_t)r * Cout + co];
        s1 += wsdb[(size_t)(r + 16) * Cout + co];
        s2 += wsdb[(size_t)(r + 32) * Cout + co];
        s3 += wsdb[(size_t)(r + 48) * Cout + co];
      }
      for (; r < R; r += 16) s0 += wsdb[(size_t)r * Cout + co];
    }
    __syncthreads();
    shdb[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && co < Cout) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += shdb[k][e];
      db[co] = s;
    }
  }
}

// 1024 threads = 64 consecutive elements x 16 interleaved slab lanes (4 independent partial
// sums each, so 64 slab rows are in flight per element); the partial sums are combined in a
// fixed order through LDS.
__global__ __launch_bounds__(1024) void adell_wgrad_reduce_kernel(
    const float* __restrict__ ws, float* __restrict__ out, int R, int ntap, int Cin,
    int Cout, const float* __restrict__ wsdb, float* __restrict__ db) {
  __shared__ float sh[16][64];
  const long total = (long)ntap * Cin * Cout;
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long i = base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
      int r = q;
      for (; r + 48 < R; r += 64) {
        s0 += ws[(size_t)r * total + i];
        s1 += ws[(size_t)(r + 16) * total + i];
        s2 += ws[(size_t)(r + 32) * total + i];
        s3 += ws[(size_t)(r + 48) * total + i];
      }
      for (; r < R; r += 16) s0 += ws[(size_t)r * total + i];
    }
    sh[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < total) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += sh[k][e];
      const int co = (int)(i % Cout);
      const long rest = i / Cout;
      const int ci = (int)(rest % Cin);
      const int tap = (int)(rest / Cin);
      out[((size_t)co * Cin + ci) * ntap + tap] = s;
    }
    __syncthreads();
  }
  if (db != nullptr && blockIdx.x == 0) adell_db_fold(wsdb, db, R, Cout);
}

// The same fold for Cout % 4 == 0 (every f16x3 / MFMA tile): 1024 threads = 64 float4 lanes
// (256 consecutive elements, 1 KiB of every slab) x 16 slab lanes, 8 independent 16-byte loads
// in flight per thread -- a streaming read of the slabs instead of 4-byte strided gathers
// (R = 512 slabs of 27 x 32 x 32: 124 us -> the HBM time of 56 MB).
__global__ __launch_bounds__(1024) void adell_wgrad_reduce4_kernel(
    const float* __restrict__ ws, float* __restrict__ out, int R, int ntap, int Cin,
    int Cout, const float* __restrict__ wsdb, float* __restrict__ db) {
  __shared__ float4 sh[16][64];
  const long total = (long)ntap * Cin * Cout;
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + e) * 4;
  float4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total) {
    int r = q;
    for (; r + 7 * 16 < R; r += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)(r + 16 * u) * total + i);
        acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
      }
    }
    for (; r < R; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)r * total + i);
      acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
  }
  float4 t;
  t.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
  t.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
  t.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
  t.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
  sh[q][e] = t;
  __syncthreads();
  if (q == 0 && i < total) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 v = sh[k][e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int co = (int)(i % Cout);
    const long rest = i / Cout;
    const int ci = (int)(rest % Cin);
    const int tap = (int)(rest / Cin);
    float* o = out + ((size_t)co * Cin + ci) * ntap + tap;
    const size_t cs = (size_t)Cin * ntap;
    o[0] = s.x; o[cs] = s.y; o[2 * cs] = s.z; o[3 * cs] = s.w;
  }
  if (db != nullptr && blockIdx.x == 0) adell_db_fold(wsdb, db, R, Cout);
}

// shared with conv_wgrad_f16.hip
extern "C" int adell_wgrad_reduce_launch(const float* ws, float* out, int R, int ntap, int Cin,
                                         int Cout, const float* wsdb, float* db, void* stream) {
  const long total = (long)ntap * Cin * Cout;
  if (Cout % 4 == 0 && (((uintptr_t)ws) & 15) == 0) {
    hipLaunchKernelGGL(adell_wgrad_reduce4_kernel, dim3((unsigned)((total / 4 + 63) / 64)),
                       dim3(1024), 0, (hipStream_t)stream, ws, out, R, ntap, Cin, Cout, wsdb, db);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  int blocks = (int)((total + 63) / 64);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adell_wgrad_reduce_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream,
                     ws, out, R, ntap, Cin, Cout, wsdb, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

struct WgradPlan {
  int lTX, lTY, lTZ, HX, HY, HZ, TCI, TCO, nci, nco, KDg, ngrp, R, maxj;
  size_t lds;
  long ntiles;
};

static int adell_wgrad_plan(int N, int Cin, int Cout, int KD, int KH, int KW, int SD,
                            int SH, int SW, int Do, int Ho, int Wo, WgradPlan* p) {
  p->TCI = Cin > 32 ? 64 : 32;
  p->TCO = Cout > 32 ? 64 : 32;
  p->nci = adell_cdiv(Cin, p->TCI);
  p->nco = adell_cdiv(Cout, p->TCO);
  const int nsub = (p->TCI / 32) * (p->TCO / 32);
  p->KDg = (nsub >= 2 && KD > 1) ? 1 : KD;
  p->ngrp = KD / p->KDg;
  const int J = p->KDg * KH * KW * nsub;
  p->maxj = adell_cdiv(J, 4);
  if (p->maxj > 9) {
    adell_set_error("wgrad: %d jobs per wave unsupported", p->maxj);
    return ADELL_E_UNSUPPORTED;
  }
  // brick: start from 8x8x2 and shrink until the LDS budget (<= 80 KiB, two
  // blocks per CU) is met; the brick keeps >= 4 voxels (two k-steps per loop trip).
  int l[3] = {3, 3, 1};
  const int cap[3] = {adell_ilog2(Wo) < 1 ? 1 : adell_ilog2(Wo), adell_ilog2(Ho),
                      adell_ilog2(Do)};
  for (int d = 0; d < 3; ++d)
    if (l[d] > cap[d]) l[d] = cap[d];
  while (l[0] + l[1] + l[2] < 2) ++l[0];
  for (;;) {
    const int TX = 1 << l[0], TY = 1 << l[1], TZ = 1 << l[2];
    p->HX = (TX - 1) * SW + KW;
    p->HY = (TY - 1) * SH + KH;
    p->HZ = (TZ - 1) * SD + p->KDg;
    p->lds = ((size_t)p->HX * p->HY * p->HZ * p->TCI + (size_t)TX * TY * TZ * p->TCO) *
             sizeof(float);
    if (p->lds <= 80 * 1024) break;
    // shrink the largest dim (prefer z, then y, then x)
    int d = 2;
    if (l[1] > l[d]) d = 1;
    if (l[0] > l[d] && l[0] > 1) d = 0;
    if (l[d] == 0 || (d == 0 && l[0] == 1) || l[0] + l[1] + l[2] <= 2) {
      if (p->lds <= 160 * 1024) break;
      adell_set_error("wgrad: cannot fit LDS");
      return ADELL_E_UNSUPPORTED;
    }
    --l[d];
  }
  p->lTX = l[0]; p->lTY = l[1]; p->lTZ = l[2];
  p->ntiles = (long)N * adell_cdiv(Wo, 1 << l[0]) * adell_cdiv(Ho, 1 << l[1]) *
              adell_cdiv(Do, 1 << l[2]);
  // One resident wave of blocks: 2 blocks per CU by registers (<= 256 per lane at
  // MAXJ <= 9), fewer if LDS does not allow it; R regions fill exactly that.
  const long chan_blocks = (long)p->nci * p->nco * p->ngrp;
  int per_cu = (int)((160 * 1024) / p->lds);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  long R = (256L * per_cu) / chan_blocks;
  if (R > p->ntiles) R = p->ntiles;
  if (R < 1) R = 1;
  p->R = (int)R;
  return ADELL_OK;
}

static size_t adell_wgrad_ws_bytes(const WgradPlan& p, int ntap, int Cin, int Cout) {
  return ((size_t)p.R * ntap * Cin * Cout + (size_t)p.R * Cout) * sizeof(float);
}

template <int MAXJ>
static int adell_launch_wgrad(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  auto kern = adell_conv_wgrad_kernel<MAXJ>;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// Core: X is the strided/haloed operand (its channels index dW's "Cin" axis), dY
// the dense one. out is [CoutY][CinX][ntap].
static int adell_wgrad_core(int N, int D, int H, int W, int C0, int C1, const float* x0,
                            const float* x1, int Cout, int Do, int Ho, int Wo,
                            const float* dy, int KD, int KH, int KW, int SD, int SH,
                            int SW, int PD, int PH, int PW, float* out, float* db, void* ws,
                            size_t ws_bytes, hipStream_t st) {
  const int Cin = C0 + C1;
  WgradPlan p;
  int rc = adell_wgrad_plan(N, Cin, Cout, KD, KH, KW, SD, SH, SW, Do, Ho, Wo, &p);
  if (rc != ADELL_OK) return rc;
  const int ntap = KD * KH * KW;
  const size_t need = adell_wgrad_ws_bytes(p, ntap, Cin, Cout);
  ADELL_REQUIRE(ws != nullptr && ws_bytes >= need, "wgrad: workspace too small (%zu < %zu)",
                ws_bytes, need);
  WgradArgs a = {};
  a.x0 = x0; a.x1 = x1; a.dy = dy; a.ws = (float*)ws;
  a.wsdb = db ? (float*)ws + (size_t)p.R * ntap * Cin * Cout : nullptr;
  a.N = N; a.D = D; a.H = H; a.W = W;
  a.C0 = C0; a.C1 = C1; a.Cin = Cin; a.Cout = Cout;
  a.KD = KD; a.KH = KH; a.KW = KW; a.SD = SD; a.SH = SH; a.SW = SW;
  a.PD = PD; a.PH = PH; a.PW = PW;
  a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.lTX = p.lTX; a.lTY = p.lTY; a.lTZ = p.lTZ;
  a.ntx = adell_cdiv(Wo, 1 << p.lTX);
  a.nty = adell_cdiv(Ho, 1 << p.lTY);
  a.ntz = adell_cdiv(Do, 1 << p.lTZ);
  a.HX = p.HX; a.HY = p.HY; a.HZ = p.HZ;
  a.TCI = p.TCI; a.TCO = p.TCO; a.nci = p.nci; a.nco = p.nco;
  a.KDg = p.KDg; a.R = p.R;
  a.vecx = (C0 % 4 == 0) && (C1 % 4 == 0) && (((uintptr_t)x0 & 15) == 0) &&
           (((uintptr_t)x1 & 15) == 0);
  a.vecy = (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
  dim3 grid((unsigned)p.R, (unsigned)(p.nci * p.nco), (unsigned)p.ngrp);
  if (p.maxj <= 2)
    rc = adell_launch_wgrad<2>(a, grid, p.lds, st);
  else if (p.maxj <= 5)
    rc = adell_launch_wgrad<5>(a, grid, p.lds, st);
  else if (p.maxj <= 7)
    rc = adell_launch_wgrad<7>(a, grid, p.lds, st);
  else
    rc = adell_launch_wgrad<9>(a, grid, p.lds, st);
  if (rc != ADELL_OK) return rc;
  return adell_wgrad_reduce_launch((const float*)ws, out, p.R, ntap, Cin, Cout,
                                   (const float*)a.wsdb, db, st);
}

// small-channel paths (conv_small.hip)
extern "C" long adell_wgrad_small_workspace(const adell_conv3d_desc* d);
extern "C" int adell_wgrad_small(const adell_conv3d_desc* d, const float* x0, const float* x1,
                                 const float* dy, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream);

extern "C" long adell_conv3d_bwd_weight_workspace(const adell_conv3d_desc* d) {
  if (!d) return ADELL_E_BADARG;
  if (adell_wgrad_small_workspace(d) > 0) return adell_wgrad_small_workspace(d);
  WgradPlan p;
  const int Cin = d->C0 + d->C1;
  if (adell_wgrad_plan(d->N, Cin, d->Cout, d->KD, d->KH, d->KW, d->SD, d->SH, d->SW, d->Do,
                       d->Ho, d->Wo, &p) != ADELL_OK)
    return ADELL_E_UNSUPPORTED;
  return (long)adell_wgrad_ws_bytes(p, d->KD * d->KH * d->KW, Cin, d->Cout);
}

extern "C" int adell_conv3d_bwd_weight(const adell_conv3d_desc* d, const float* x0,
                                       const float* x1, const float* dy, float* dw,
                                       float* db, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && x0 && dy && dw, "conv_bwd_weight: null pointer");
  ADELL_REQUIRE(d->C1 == 0 || x1, "conv_bwd_weight: C1 > 0 needs x1");
  if (adell_wgrad_small_workspace(d) > 0)
    return adell_wgrad_small(d, x0, x1, dy, dw, db, workspace, workspace_bytes, stream);
  return adell_wgrad_core(d->N, d->D, d->H, d->W, d->C0, d->C1, x0, x1, d->Cout, d->Do,
                          d->Ho, d->Wo, dy, d->KD, d->KH, d->KW, d->SD, d->SH, d->SW, d->PD,
                          d->PH, d->PW, dw, db, workspace, workspace_bytes, (hipStream_t)stream);
}

// dW[ci][co][tap] = sum_v x[v][ci] * dy[2v+tap][co]: the same GEMM with the roles
// swapped (dy is the strided operand, x the dense one).
extern "C" long adell_convtranspose3d_bwd_weight_workspace(int N, int D, int H, int W, int Cin,
                                                           int Cout, int FD, int FH, int FW) {
  WgradPlan p;
  if (adell_wgrad_plan(N, Cout, Cin, FD, FH, FW, FD, FH, FW, D, H, W, &p) != ADELL_OK)
    return ADELL_E_UNSUPPORTED;
  return (long)adell_wgrad_ws_bytes(p, FD * FH * FW, Cout, Cin);
}

extern "C" long adell_convtranspose3d_k2s2_bwd_weight_workspace(int N, int D, int H, int W,
                                                                int Cin, int Cout) {
  return adell_convtranspose3d_bwd_weight_workspace(N, D, H, W, Cin, Cout, 2, 2, 2);
}

extern "C" int adell_convtranspose3d_bwd_weight(int N, int D, int H, int W, int Cin, int Cout,
                                                int FD, int FH, int FW, const float* x,
                                                const float* dy, float* dw, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(x && dy && dw, "convT_bwd_weight: null pointer");
  ADELL_REQUIRE((FD == 1 || FD == 2) && (FH == 1 || FH == 2) && (FW == 1 || FW == 2),
                "convT_bwd_weight: kernel=stride must be 1 or 2 per dim");
  return adell_wgrad_core(N, FD * D, FH * H, FW * W, Cout, 0, dy, nullptr, Cin, D, H, W, x, FD,
                          FH, FW, FD, FH, FW, 0, 0, 0, dw, nullptr, workspace, workspace_bytes,
                          (hipStream_t)stream);
}

extern "C" int adell_convtranspose3d_k2s2_bwd_weight(int N, int D, int H, int W, int Cin,
                                                     int Cout, const float* x,
                                                     const float* dy, float* dw,
                                                     void* workspace, size_t workspace_bytes,
                                                     void* stream) {
  return adell_convtranspose3d_bwd_weight(N, D, H, W, Cin, Cout, 2, 2, 2, x, dy, dw, workspace,
                                          workspace_bytes, stream);
}

// ---------------------------------------------------------------------------
// Bias gradient: column sums of dy [rows][C], two deterministic phases.
// ---------------------------------------------------------------------------
// grid (row chunks, groups of 64 columns); block = 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void adell_colsum_partial_kernel(
    const float* __restrict__ dy, long rows, int C, int chunk, float* __restrict__ part) {
  __shared__ float sh[4][64];
  const long r0 = (long)blockIdx.x * chunk;
  long r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    long r = r0 + vl;
    for (; r + 4 < r1; r += 8) {
      s0 += dy[r * C + c];
      s1 += dy[(r + 4) * C + c];
    }
    if (r < r1) s0 += dy[r * C + c];
  }
  sh[vl][cl] = s0 + s1;
  __syncthreads();
  if (vl == 0 && c < C)
    part[(size_t)blockIdx.x * C + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// block = 64 columns x 16 lanes, each lane a fixed share of the partials (fp64, fixed order)
__global__ __launch_bounds__(1024) void adell_colsum_final_kernel(
    const float* __restrict__ part, int nb, int C, float* __restrict__ out) {
  __shared__ double sh[16][64];
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < C) {
    int b = vl;
    for (; b + 7 * 16 < nb; b += 8 * 16) {             // eight loads in flight, same order of additions
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nb; b += 16) s += (double)part[(size_t)b * C + c];
  }
  sh[vl][cl] = s;
  __syncthreads();
  if (vl != 0 || c >= C) return;
  s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[k][cl];
  out[c] = (float)s;
}

// Narrow tensors (C a power of two <= 64: the 2 .. 32-channel per-voxel layers of SWIN / U-Net heads):
// the kernel above keeps C of its 64 column lanes busy (C = 2: 8 threads of 256, 63 us for a 67 MB
// tensor). Here dy is a flat stream of float4: the grid stride is a multiple of C elements, so a
// thread's four lanes stay on the same four columns; block fold = a fixed tree over the threads that
// share columns. part[block][C] as above.
__global__ __launch_bounds__(256) void adell_colsum_flat_kernel(const f32x4* __restrict__ x4, long n4,
                                                                int C, float* __restrict__ part) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x;
  const long stride = (long)gridDim.x * 256L;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  long i = blockIdx.x * 256L + tid;
  for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-byte loads in flight
    const f32x4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < n4; i += stride) a0 += x4[i];
  sh[tid] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  const int q = C >= 4 ? C >> 2 : 1;                      // threads per period of the column pattern
  for (int s = 128; s >= q; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  if (tid < C) {
    float v;
    if (C >= 4) v = sh[tid >> 2][tid & 3];
    else if (C == 2) v = sh[0][tid] + sh[0][tid + 2];
    else v = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    part[(size_t)blockIdx.x * C + tid] = v;
  }
}

static bool adell_colsum_flat_ok(const float* dy, long rows, int C) {
  return C >= 1 && C <= 64 && (C & (C - 1)) == 0 && ((rows * C) & 3) == 0 && rows * (long)C >= 4096 &&
         (((uintptr_t)dy) & 15) == 0;
}
static int adell_colsum_flat_blocks(long rows, int C) {
  long nb = (rows * C / 4 + 1023) / 1024;                // >= 4 float4 per thread
  if (nb > 1024) nb = 1024;
  return (int)(nb < 1 ? 1 : nb);
}

// rows per block: about 2048 blocks in total, at least 16 rows each
static int adell_bias_grad_chunk(long rows, int C) {
  const long colgroups = adell_cdiv(C, 64);
  long nchunks = 2048 / colgroups;
  if (nchunks < 1) nchunks = 1;
  long chunk = (rows + nchunks - 1) / nchunks;
  if (chunk < 16) chunk = 16;
  chunk = (chunk + 3) / 4 * 4;
  if (chunk > 65536) chunk = 65536;
  return (int)chunk;
}

extern "C" long adell_bias_grad_workspace(long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  const int chunk = adell_bias_grad_chunk(rows, C);
  long nb = (rows + chunk - 1) / chunk;
  if (C <= 64 && (C & (C - 1)) == 0 && adell_colsum_flat_blocks(rows, C) > nb)
    nb = adell_colsum_flat_blocks(rows, C);              // either kernel may run (alignment decides)
  return nb * C * (long)sizeof(float);
}

extern "C" int adell_bias_grad(const float* dy, long rows, int C, float* db, void* workspace,
                               size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(dy && db && workspace, "bias_grad: null pointer");
  ADELL_REQUIRE(rows > 0 && C > 0, "bias_grad: bad dims");
  ADELL_REQUIRE((long)workspace_bytes >= adell_bias_grad_workspace(rows, C),
                "bias_grad: workspace too small");
  const int chunk = adell_bias_grad_chunk(rows, C);
  int nb = (int)((rows + chunk - 1) / chunk);
  if (adell_colsum_flat_ok(dy, rows, C)) {
    nb = adell_colsum_flat_blocks(rows, C);
    hipLaunchKernelGGL(adell_colsum_flat_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(dy), rows * C / 4, C, (float*)workspace);
  } else {
    hipLaunchKernelGGL(adell_colsum_partial_kernel, dim3(nb, adell_cdiv(C, 64)), dim3(256), 0,
                       (hipStream_t)stream, dy, rows, C, chunk, (float*)workspace);
  }
  hipLaunchKernelGGL(adell_colsum_final_kernel, dim3(adell_cdiv(C, 64)), dim3(1024), 0,
                     (hipStream_t)stream, (const float*)workspace, nb, C, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
