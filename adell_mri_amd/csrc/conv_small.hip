// Convolutions whose channel counts are too small for an MFMA tile (the 2-channel input
// layers and the 1-channel logits head of the U-Nets: unet.py:245-273, 712-731):
//
//  * adell_wgrad_small: dW / db of a k in {1, 3} stride-1 conv with Cin <= 4 (any Cout). An
//    MFMA tile would pad Cin to 16 per tap; here the 27*Cin products per (voxel, co) run on
//    the vector ALU out of an LDS tile -- thread = (co, kz, ky) with Cin x K accumulators
//    over every tile its block visits, deterministic split partials + fixed-order fold.
//  * adell_conv1_small_fwd / _bwd_data / _bwd_weight: 1x1x1 conv with Cout <= 4 (the logits
//    head): one pass over the activation, HBM-bound.
//
// Called from the conv entry points of conv3d.hip / conv_wgrad*.hip when the shapes qualify.
#include "common.h"

// ---------------------------------------------------------------------------
// small-Cin weight gradient
// ---------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WgSmallArgs {
  const float* x0;
  const float* x1;
  const float* dy;
  float* part;   // [splits][coPad][4*K3 + 1]
  int N, D, H, W, C0, C1, Cout, Do, Ho, Wo, P;
  int tilesX, tilesY, tilesZ, coBlocks;
  long items;
  int itemsPerSplit;
};

template <int K>
struct WgSmallCfg {
  static constexpr int WT = 16;                 // outputs per x segment
  static constexpr int HX = WT + K - 1;         // input voxels per row
  static constexpr int HZ = 4 + K - 1, HY = 4 + K - 1;
  static constexpr int K3 = K * K * K;
  static constexpr int XT_FLOATS = HZ * HY * HX * 4;      // [row][x][4 ci]
  static constexpr int DY_FLOATS = 16 * WT * 16;          // [row][x][16 co]
  static constexpr int THREADS = ((16 * K * K + 63) / 64) * 64;
};

template <int K, int CIN>   // CIN: channel slots that carry data (2 or 4)
__global__ __launch_bounds__(((16 * K * K + 63) / 64) * 64) void adell_wgrad_small_kernel(
    WgSmallArgs a) {
  using Cf = WgSmallCfg<K>;
  constexpr int NT = Cf::THREADS;
  __shared__ __attribute__((aligned(16))) float xt[Cf::XT_FLOATS];
  __shared__ __attribute__((aligned(16))) float dyt[Cf::DY_FLOATS];
  const int tid = threadIdx.x;
  const int cb = blockIdx.x % a.coBlocks, split = blockIdx.x / a.coBlocks;
  const int co0 = cb * 16;
  const int co = tid & 15, kk = tid >> 4;
  const bool active = kk < K * K;
  const int kz = kk / K, ky = kk % K;
  const int Cin = a.C0 + a.C1;
  f32x2 acc2[CIN / 2][K];                       // (ci, ci + 1) pairs
#pragma unroll
  for (int c = 0; c < CIN / 2; ++c)
#pragma unroll
    for (int q = 0; q < K; ++q) acc2[c][q] = f32x2{0.f, 0.f};
  float sb = 0.f;
  const long first = (long)split * a.itemsPerSplit;
  long last = first + a.itemsPerSplit;
  if (last > a.items) last = a.items;
  for (long item = first; item < last; ++item) {
    long t = item;
    const int tx = (int)(t % a.tilesX); t /= a.tilesX;
    const int ty = (int)(t % a.tilesY); t /= a.tilesY;
    const int tz = (int)(t % a.tilesZ);
    const int n = (int)(t / a.tilesZ);
    const int z0 = tz * 4, y0 = ty * 4, x0o = tx * Cf::WT;
    __syncthreads();
    // input halo tile, 4 channel slots per voxel (zero beyond Cin and beyond the volume)
    for (int i = tid; i < Cf::HZ * Cf::HY * Cf::HX; i += NT) {
      const int j = i % Cf::HX, r = i / Cf::HX;
      const int z = z0 - a.P + r / Cf::HY, y = y0 - a.P + r % Cf::HY, x = x0o - a.P + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (z >= 0 && z < a.D && y >= 0 && y < a.H && x >= 0 && x < a.W) {
        const size_t vox = (((size_t)n * a.D + z) * a.H + y) * a.W + x;
        float c4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < Cin; ++c)
          c4[c] = c < a.C0 ? a.x0[vox * a.C0 + c] : a.x1[vox * a.C1 + (c - a.C0)];
        v = f32x4{c4[0], c4[1], c4[2], c4[3]};
      }
      *reinterpret_cast<f32x4*>(xt + (size_t)i * 4) = v;
    }
    // dy tile: 16 output rows x WT voxels x 16 output channels
    for (int i = tid; i < 16 * Cf::WT * 4; i += NT) {
      const int q = i & 3, j = (i >> 2) % Cf::WT, r = (i >> 2) / Cf::WT;
      const int z = z0 + (r >> 2), y = y0 + (r & 3), x = x0o + j;
      const int c = co0 + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (z < a.Do && y < a.Ho && x < a.Wo && c < a.Cout) {
        const float* p = a.dy + ((((size_t)n * a.Do + z) * a.Ho + y) * a.Wo + x) * a.Cout + c;
        const int nv = a.Cout - c;
        if (nv >= 4 && (a.Cout & 3) == 0) {
          v = *reinterpret_cast<const f32x4*>(p);
        } else {
          if (nv > 0) v.x = p[0];
          if (nv > 1) v.y = p[1];
          if (nv > 2) v.z = p[2];
          if (nv > 3) v.w = p[3];
        }
      }
      *reinterpret_cast<f32x4*>(dyt + (r * Cf::WT + j) * 16 + q * 4) = v;
    }
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int row = 0; row < 16; ++row) {
        const int rz = row >> 2, ry = row & 3;
        float g[Cf::WT];
        const float* gr = dyt + row * Cf::WT * 16 + co;
#pragma unroll
        for (int j = 0; j < Cf::WT; ++j) g[j] = gr[j * 16];
        const f32x4* xr =
            reinterpret_cast<const f32x4*>(xt) + ((rz + kz) * Cf::HY + ry + ky) * Cf::HX;
#pragma unroll
        for (int jj = 0; jj < Cf::HX; ++jj) {
          const f32x4 xv = xr[jj];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const int j = jj - kx;
            if (j >= 0 && j < Cf::WT) {
              // channel pairs as packed fp32 FMAs (v_pk_fma_f32: two lanes' worth per issue;
              // the kernel is bound by vector-ALU issue)
              const f32x2 gg = {g[j], g[j]};
              acc2[0][kx] = __builtin_elementwise_fma(gg, f32x2{xv.x, xv.y}, acc2[0][kx]);
              if (CIN > 2) acc2[1][kx] = __builtin_elementwise_fma(gg, f32x2{xv.z, xv.w}, acc2[1][kx]);
            }
          }
        }
        if (kk == 0) {
#pragma unroll
          for (int j = 0; j < Cf::WT; ++j) sb += g[j];
        }
      }
    }
  }
  if (active) {
    float* dst = a.part + ((size_t)split * a.coBlocks * 16 + co0 + co) * (4 * Cf::K3 + 1);
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
        dst[c * Cf::K3 + (kz * K + ky) * K + kx] = (c & 1) ? acc2[c / 2][kx].y : acc2[c / 2][kx].x;
    if (kk == 0) dst[4 * Cf::K3] = sb;
  }
}

// dw[co][ci][tap] = sum over splits (fixed order); the last column of a partial row is db.
// block = 64 outputs x 16 split lanes.
__global__ __launch_bounds__(1024) void adell_wgrad_small_reduce_kernel(
    const float* __restrict__ part, int splits, int coPad, int Cout, int Cin, int K3,
    float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double sh[16][64];
  const int rowlen = 4 * K3 + 1;
  const long total = (long)Cout * (Cin * K3 + 1);
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + cl;
  int co = 0, e = 0;
  double s = 0.0;
  if (i < total) {
    co = (int)(i / (Cin * K3 + 1));
    e = (int)(i % (Cin * K3 + 1));
    const int col = e < Cin * K3 ? e : 4 * K3;   // (ci, tap) slots are laid out ci-major
    int sp = vl;
    for (; sp + 7 * 16 < splits; sp += 8 * 16) {       // eight loads in flight, same order of additions
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(sp + 16 * u) * coPad + co) * rowlen + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; sp < splits; sp += 16) s += (double)part[((size_t)sp * coPad + co) * rowlen + col];
  }
  sh[vl][cl] = s;
  __syncthreads();
  if (vl != 0 || i >= total) return;
  s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[k][cl];
  if (e < Cin * K3) dw[(size_t)co * Cin * K3 + e] = (float)s;
  else if (db) db[co] = (float)s;
}

// Cin <= 2 and Cout <= 2, k = 3 (Conv3d(2 -> 2) of the input block): the tile kernel above keeps
// 2 of its 16 output-channel lanes busy. Here a thread owns whole voxels and all 27 * CIN * COUT
// products in registers; one shuffle + LDS fold per block at the end. Partial rows have the
// layout the reduce kernel above reads (one split per block, 16 channel rows per split).
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void adell_wgrad_tiny_kernel(WgSmallArgs a, int nbricks, int ntx,
                                                               int nty, int ntz) {
  // A block walks 8 x 8 x 4 output bricks (thread = voxel); the brick's 10 x 10 x 6 input halo goes
  // through LDS, so a tap is one ds_read at a compile-time offset from the thread's base (the
  // first version addressed every tap in global memory: 27 bounds checks and 64-bit addresses and
  // three 64-bit divisions per voxel, 0.19 ms for 70 MB at 2 x 128^3).
  __shared__ float sx[6 * 10 * 10 * CIN];
  __shared__ float sred[4][27 * CIN * COUT + COUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[27][CIN][COUT];
  float sb[COUT];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int o = 0; o < COUT; ++o) acc[t][c][o] = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) sb[o] = 0.f;
  const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
  const int base = ((lz * 10 + ly) * 10 + lx) * CIN;
  // registers of the NEXT brick: three halo voxels per thread (600 over 256 threads) and the
  // thread's own dY voxel, fetched while the products of the current brick run
  float hx_[3][CIN], gn[COUT];
  auto fetch = [&](int b) {
    int t = b;
    const int tx = t % ntx; t /= ntx;
    const int ty = t % nty; t /= nty;
    const int tz = t % ntz;
    const int n = t / ntz;
    const int ox0 = tx * 8, oy0 = ty * 8, oz0 = tz * 4;
    const float* xb = a.x0 + (size_t)n * a.D * a.H * a.W * CIN;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + 256 * u;
      const int hz = i / 100, r = i - hz * 100, hy = r / 10, hx = r - hy * 10;
      const int iz = oz0 - a.P + hz, iy = oy0 - a.P + hy, ix = ox0 - a.P + hx;
      const bool ok = (i < 600) & (iz >= 0) & (iz < a.D) & (iy >= 0) & (iy < a.H) & (ix >= 0) &
                      (ix < a.W);
      const float* p = xb + (ok ? ((size_t)(iz * a.H + iy) * a.W + ix) * CIN : 0);
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float v = p[c];
        hx_[u][c] = ok ? v : 0.f;
      }
    }
    const int x = ox0 + lx, y = oy0 + ly, z = oz0 + lz;
    const bool valid = (x < a.Wo) & (y < a.Ho) & (z < a.Do);
    const float* gp = a.dy + (valid ? ((((size_t)n * a.Do + z) * a.Ho + y) * a.Wo + x) * COUT : 0);
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float v = gp[o];
      gn[o] = valid ? v : 0.f;
    }
  };
  int b = blockIdx.x;
  if (b < nbricks) fetch(b);
  for (; b < nbricks; b += gridDim.x) {
    __syncthreads();   // the previous brick's taps are read
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + 256 * u;
      if (i < 600) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) sx[i * CIN + c] = hx_[u][c];
      }
    }
    float g[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      g[o] = gn[o];
      sb[o] += g[o];
    }
    __syncthreads();
    if (b + (int)gridDim.x < nbricks) fetch(b + gridDim.x);
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            const float xv = sx[base + ((kz * 10 + ky) * 10 + kx) * CIN + c];
#pragma unroll
            for (int o = 0; o < COUT; ++o)
              acc[(kz * 3 + ky) * 3 + kx][c][o] = fmaf(xv, g[o], acc[(kz * 3 + ky) * 3 + kx][c][o]);
          }
  }
  // wave fold (fixed butterfly order), then the four waves through LDS
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        float s = acc[t][c][o];
        for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
        if (lane == 0) sred[wave][(t * CIN + c) * COUT + o] = s;
      }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    float s = sb[o];
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);
    if (lane == 0) sred[wave][27 * CIN * COUT + o] = s;
  }
  __syncthreads();
  constexpr int NV = 27 * CIN * COUT + COUT;
  if (tid < NV) {
    const float s = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
    int co, col;
    if (tid < 27 * CIN * COUT) {
      const int o = tid % COUT, c = (tid / COUT) % CIN, t = tid / (COUT * CIN);
      co = o;
      col = c * 27 + t;
    } else {
      co = tid - 27 * CIN * COUT;
      col = 4 * 27;
    }
    a.part[((size_t)blockIdx.x * 16 + co) * (4 * 27 + 1) + col] = s;
  }
}

// The same weight gradient for 2 input channels and rows of a multiple of four voxels (the 2 -> 2
// conv at 128^3): a thread owns FOUR consecutive voxels of a row and reads its taps straight from
// global memory as 16-byte pieces (the six voxels x - 1 .. x + 4 of a (kz, ky) row: 4 loads, 2 bounds
// checks, shared by 24 products each) -- no LDS halo, no barrier per brick, grid-stride over the
// volume with the next group's dY in flight. Partial rows as adell_wgrad_tiny_kernel writes them.
template <int COUT>
__global__ __launch_bounds__(256) void adell_wgrad_cin2_rows_kernel(WgSmallArgs a, long groups) {
  __shared__ float sred[4][27 * 2 * COUT + COUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[27][2][COUT];
  float sb[COUT];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int o = 0; o < COUT; ++o) acc[t][c][o] = 0.f;
#pragma unroll
  for (int o = 0; o < COUT; ++o) sb[o] = 0.f;
  const int W = a.W, H = a.H, D = a.D;
  const long vox = (long)D * H * W, gpi = vox >> 2;            // groups of four voxels per item
  for (long gi = blockIdx.x * 256L + tid; gi < groups; gi += gridDim.x * 256L) {
    const int n = (int)(gi / gpi);
    const long v0 = (gi - (long)n * gpi) << 2;
    const int x0 = (int)(v0 % W), y = (int)((v0 / W) % H), z = (int)(v0 / ((long)W * H));
    const float* xb = a.x0 + (size_t)n * vox * 2;
    const float* gp = a.dy + ((size_t)n * vox + v0) * COUT;
    float g[4][COUT];
    if (COUT == 2) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
      g[0][0] = g0.x; g[0][COUT - 1] = g0.y; g[1][0] = g0.z; g[1][COUT - 1] = g0.w;
      g[2][0] = g1.x; g[2][COUT - 1] = g1.y; g[3][0] = g1.z; g[3][COUT - 1] = g1.w;
    } else {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
      g[0][0] = g0.x; g[1][0] = g0.y; g[2][0] = g0.z; g[3][0] = g0.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int o = 0; o < COUT; ++o) sb[o] += g[q][o];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const int iz = z - 1 + kz;
      if (iz < 0 || iz >= D) continue;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = y - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        const float* row = xb + ((size_t)(iz * H + iy) * W) * 2;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(row + (size_t)x0 * 2);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(row + (size_t)(x0 + 2) * 2);
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (x0 > 0) lo = *reinterpret_cast<const f32x4*>(row + (size_t)(x0 - 2) * 2);
        if (x0 + 4 < W) hi = *reinterpret_cast<const f32x4*>(row + (size_t)(x0 + 4) * 2);
        const float in[6][2] = {{lo.z, lo.w}, {m0.x, m0.y}, {m0.z, m0.w}, {m1.x, m1.y}, {m1.z, m1.w},
                                {hi.x, hi.y}};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int o = 0; o < COUT; ++o)
                acc[(kz * 3 + ky) * 3 + kx][c][o] = fmaf(in[q + kx][c], g[q][o], acc[(kz * 3 + ky) * 3 + kx][c][o]);
      }
    }
  }
  // wave fold (fixed butterfly order), then the four waves through LDS
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        float v = acc[t][c][o];
        for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh, 64);
        if (lane == 0) sred[wave][(t * 2 + c) * COUT + o] = v;
      }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    float v = sb[o];
    for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh, 64);
    if (lane == 0) sred[wave][27 * 2 * COUT + o] = v;
  }
  __syncthreads();
  constexpr int NV = 27 * 2 * COUT + COUT;
  if (tid < NV) {
    const float v = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
    int co, col;
    if (tid < 27 * 2 * COUT) {
      const int o = tid % COUT, c = (tid / COUT) % 2, t = tid / (COUT * 2);
      co = o;
      col = c * 27 + t;
    } else {
      co = tid - 27 * 2 * COUT;
      col = 4 * 27;
    }
    a.part[((size_t)blockIdx.x * 16 + co) * (4 * 27 + 1) + col] = v;
  }
}

template <int CIN, int COUT>
static void adell_wgrad_tiny_launch(const WgSmallArgs& a, int blocks, int nbricks, int ntx, int nty,
                                    int ntz, hipStream_t st) {
  hipLaunchKernelGGL((adell_wgrad_tiny_kernel<CIN, COUT>), dim3((unsigned)blocks), dim3(256), 0, st,
                     a, nbricks, ntx, nty, ntz);
}

static bool adell_wgrad_tiny_ok(const adell_conv3d_desc* d) {
  return d->C1 == 0 && d->C0 <= 2 && d->Cout <= 2 && d->KD == 3 && d->KH == 3 && d->KW == 3;
}

static bool adell_wgrad_small_ok(const adell_conv3d_desc* d) {
  const int Cin = d->C0 + d->C1;
  return Cin <= 4 && d->KD == d->KH && d->KH == d->KW &&
         (d->KD == 1 || d->KD == 3 || d->KD == 5 || d->KD == 7) &&
         d->SD == 1 && d->SH == 1 && d->SW == 1 && d->PD == d->PH && d->PH == d->PW;
}

static void adell_wgrad_small_plan(const adell_conv3d_desc* d, WgSmallArgs* a, int* splits) {
  a->N = d->N; a->D = d->D; a->H = d->H; a->W = d->W; a->C0 = d->C0; a->C1 = d->C1;
  a->Cout = d->Cout; a->Do = d->Do; a->Ho = d->Ho; a->Wo = d->Wo; a->P = d->PD;
  a->tilesX = adell_cdiv(d->Wo, 16);
  a->tilesY = adell_cdiv(d->Ho, 4);
  a->tilesZ = adell_cdiv(d->Do, 4);
  a->coBlocks = adell_cdiv(d->Cout, 16);
  a->items = (long)d->N * a->tilesZ * a->tilesY * a->tilesX;
  long s = adell_cdiv(1536, a->coBlocks);
  if (s > a->items) s = a->items;
  if (s < 1) s = 1;
  a->itemsPerSplit = (int)((a->items + s - 1) / s);
  *splits = (int)((a->items + a->itemsPerSplit - 1) / a->itemsPerSplit);
}

// bytes of workspace, or 0 when the shape does not take this path
extern "C" long adell_wgrad_small_workspace(const adell_conv3d_desc* d) {
  if (!d || !adell_wgrad_small_ok(d)) return 0;
  WgSmallArgs a;
  int splits;
  adell_wgrad_small_plan(d, &a, &splits);
  const int K3 = d->KD * d->KH * d->KW;
  return (long)sizeof(float) * splits * a.coBlocks * 16 * (4 * K3 + 1);
}

extern "C" int adell_wgrad_small(const adell_conv3d_desc* d, const float* x0, const float* x1,
                                 const float* dy, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && x0 && dy && dw && workspace, "wgrad_small: null pointer");
  ADELL_REQUIRE(adell_wgrad_small_ok(d), "wgrad_small: shape not supported");
  ADELL_REQUIRE((long)workspace_bytes >= adell_wgrad_small_workspace(d),
                "wgrad_small: workspace too small");
  WgSmallArgs a;
  int splits;
  adell_wgrad_small_plan(d, &a, &splits);
  a.x0 = x0; a.x1 = x1; a.dy = dy; a.part = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (adell_wgrad_tiny_ok(d)) {
    const int ntx = adell_cdiv(d->Wo, 8), nty = adell_cdiv(d->Ho, 8), ntz = adell_cdiv(d->Do, 4);
    const long bricks = (long)d->N * ntx * nty * ntz;
    ADELL_REQUIRE(bricks < 0x7fffffffL, "wgrad_small: too many bricks");
    // three resident blocks per CU (159 registers at 2 x 2 channels): one round of blocks
    int nb = splits < 768 ? splits : 768;            // the workspace holds `splits` partial rows
    if (nb > bricks) nb = (int)bricks;
    // rows of a multiple of four voxels, padding 1, 16-byte aligned tensors: four voxels per thread
    if (d->C0 == 2 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->W % 4 == 0 && d->Do == d->D &&
        d->Ho == d->H && d->Wo == d->W && ((((uintptr_t)x0) | ((uintptr_t)dy)) & 15) == 0) {
      const long groups = (long)d->N * d->D * d->H * d->W / 4;
      long gb = (groups + 255) / 256;
      if (gb < nb) nb = (int)gb;
      if (d->Cout == 2)
        hipLaunchKernelGGL(adell_wgrad_cin2_rows_kernel<2>, dim3((unsigned)nb), dim3(256), 0, st, a, groups);
      else
        hipLaunchKernelGGL(adell_wgrad_cin2_rows_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, a, groups);
    } else
    if (d->C0 == 2 && d->Cout == 2) adell_wgrad_tiny_launch<2, 2>(a, nb, (int)bricks, ntx, nty, ntz, st);
    else if (d->C0 == 2) adell_wgrad_tiny_launch<2, 1>(a, nb, (int)bricks, ntx, nty, ntz, st);
    else if (d->Cout == 2) adell_wgrad_tiny_launch<1, 2>(a, nb, (int)bricks, ntx, nty, ntz, st);
    else adell_wgrad_tiny_launch<1, 1>(a, nb, (int)bricks, ntx, nty, ntz, st);
    const int Cin = d->C0;
    const long blocks = ((long)d->Cout * (Cin * 27 + 1) + 63) / 64;
    hipLaunchKernelGGL(adell_wgrad_small_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st,
                       (const float*)workspace, nb, 16, d->Cout, Cin, 27, dw, db);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  const unsigned grid = (unsigned)(splits * a.coBlocks);
  const bool two = d->C0 + d->C1 <= 2;
  // k = 5 / 7: the stems of the ResNet backbones (res_net.py:60-130; 7^3 x 2 channels at 128^3)
  if (d->KD == 7 && two)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<7, 2>), dim3(grid), dim3(WgSmallCfg<7>::THREADS),
                       0, st, a);
  else if (d->KD == 7)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<7, 4>), dim3(grid), dim3(WgSmallCfg<7>::THREADS),
                       0, st, a);
  else if (d->KD == 5 && two)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<5, 2>), dim3(grid), dim3(WgSmallCfg<5>::THREADS),
                       0, st, a);
  else if (d->KD == 5)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<5, 4>), dim3(grid), dim3(WgSmallCfg<5>::THREADS),
                       0, st, a);
  else if (d->KD == 3 && two)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<3, 2>), dim3(grid), dim3(WgSmallCfg<3>::THREADS),
                       0, st, a);
  else if (d->KD == 3)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<3, 4>), dim3(grid), dim3(WgSmallCfg<3>::THREADS),
                       0, st, a);
  else if (two)
    hipLaunchKernelGGL((adell_wgrad_small_kernel<1, 2>), dim3(grid), dim3(WgSmallCfg<1>::THREADS),
                       0, st, a);
  else
    hipLaunchKernelGGL((adell_wgrad_small_kernel<1, 4>), dim3(grid), dim3(WgSmallCfg<1>::THREADS),
                       0, st, a);
  const int K3 = d->KD * d->KH * d->KW, Cin = d->C0 + d->C1;
  const long blocks = ((long)d->Cout * (Cin * K3 + 1) + 63) / 64;
  hipLaunchKernelGGL(adell_wgrad_small_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st,
                     (const float*)workspace, splits, a.coBlocks * 16, d->Cout, Cin, K3, dw, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// 1x1x1 convolution with Cout <= 4 (logits head): y[v][o] = sum_c x[v][c] w[o][c] + b[o]
// ---------------------------------------------------------------------------
#define ADELL_C1_MAXO 4
struct Conv1Args {
  const float* x0;
  const float* x1;
  const float* w;     // [Cout][Cin]
  const float* bias;
  const float* dy;
  float* y;
  float* dx0;
  float* dx1;
  float* part;        // bwd_weight partials [blocks][Cout][Cin + 1]
  long V;
  int C0, C1, Cout;
};

// LPR lanes per voxel, each holding CE = ceil(Cin / LPR) <= 8 channels (like layernorm_rows)
__global__ __launch_bounds__(256) void adell_conv1_small_fwd_kernel(Conv1Args a, int lpr) {
  const int Cin = a.C0 + a.C1;
  const int gl = threadIdx.x % lpr, grp = threadIdx.x / lpr, ngrp = 256 / lpr;
  const bool vec8 = a.C1 == 0 && (a.C0 & 7) == 0 && ((reinterpret_cast<uintptr_t>(a.x0) & 15) == 0);
  float wr[ADELL_C1_MAXO][8];
#pragma unroll
  for (int o = 0; o < ADELL_C1_MAXO; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = gl * 8 + e;   // a lane owns 8 consecutive channels
      wr[o][e] = (o < a.Cout && c < Cin) ? a.w[o * Cin + c] : 0.f;
    }
  for (long vb = (long)blockIdx.x * ngrp; vb < a.V; vb += (long)gridDim.x * ngrp) {
    const long v = vb + grp;
    const bool live = v < a.V;
    float s[ADELL_C1_MAXO] = {0.f, 0.f, 0.f, 0.f};
    float xv[8];
    if (vec8) {   // one source, whole 8-channel groups: two 16-byte loads per lane
      float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
      if (live && gl * 8 < Cin) {
        const float4* p = reinterpret_cast<const float4*>(a.x0 + v * a.C0 + gl * 8);
        f0 = p[0];
        f1 = p[1];
      }
      xv[0] = f0.x; xv[1] = f0.y; xv[2] = f0.z; xv[3] = f0.w;
      xv[4] = f1.x; xv[5] = f1.y; xv[6] = f1.z; xv[7] = f1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = gl * 8 + e;
        xv[e] = 0.f;
        if (live && c < Cin) xv[e] = c < a.C0 ? a.x0[v * a.C0 + c] : a.x1[v * a.C1 + (c - a.C0)];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int o = 0; o < ADELL_C1_MAXO; ++o) s[o] = fmaf(xv[e], wr[o][e], s[o]);
#pragma unroll
    for (int o = 0; o < ADELL_C1_MAXO; ++o)
      for (int m = lpr >> 1; m > 0; m >>= 1) s[o] += __shfl_xor(s[o], m, 64);
    if (live && gl == 0)
      for (int o = 0; o < a.Cout; ++o) a.y[v * a.Cout + o] = s[o] + (a.bias ? a.bias[o] : 0.f);
  }
}

// dx[v][c] = sum_o dy[v][o] w[o][c]
__global__ __launch_bounds__(256) void adell_conv1_small_bwd_data_kernel(Conv1Args a) {
  const int Cin = a.C0 + a.C1;
  if (a.C1 == 0 && (a.C0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.dx0) & 15) == 0)) {
    // one destination, whole channel quads: 16-byte stores
    const int cq = Cin >> 2;
    const long quads = a.V * cq;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < quads; i += (long)gridDim.x * 256L) {
      const long v = i / cq;
      const int c = (int)(i - v * cq) * 4;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o < a.Cout; ++o) {
        const float g = a.dy[v * a.Cout + o];
        const float4 wv = *reinterpret_cast<const float4*>(a.w + o * Cin + c);
        s.x = fmaf(g, wv.x, s.x); s.y = fmaf(g, wv.y, s.y);
        s.z = fmaf(g, wv.z, s.z); s.w = fmaf(g, wv.w, s.w);
      }
      *reinterpret_cast<float4*>(a.dx0 + v * a.C0 + c) = s;
    }
    return;
  }
  const long total = a.V * Cin;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long v = i / Cin;
    const int c = (int)(i - v * Cin);
    float s = 0.f;
    for (int o = 0; o < a.Cout; ++o) s = fmaf(a.dy[v * a.Cout + o], a.w[o * Cin + c], s);
    if (c < a.C0) a.dx0[v * a.C0 + c] = s;
    else a.dx1[v * a.C1 + (c - a.C0)] = s;
  }
}

// partial dW[o][c] = sum_v dy[v][o] x[v][c], db[o] = sum_v dy[v][o] over the block's voxels.
// block = CL channel lanes (power of two >= min(Cin + 1, 64)) x 256 / CL voxel lanes, the
// voxel loop unrolled by 4 with independent loads (channel groups of CL loop).
__global__ __launch_bounds__(256) void adell_conv1_small_wgrad_kernel(Conv1Args a, int chunk,
                                                                      int CL) {
  __shared__ float sh[256];
  const int Cin = a.C0 + a.C1;
  const int VL = 256 / CL;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL;
  const long v0 = (long)blockIdx.x * chunk;
  long v1 = v0 + chunk;
  if (v1 > a.V) v1 = a.V;
  float* prow = a.part + (size_t)blockIdx.x * a.Cout * (Cin + 1);
  for (int cb = 0; cb < Cin + 1; cb += CL) {
    const int c = cb + cl;   // c == Cin: the bias column (x := 1)
    float s[ADELL_C1_MAXO] = {0.f, 0.f, 0.f, 0.f};
    if (c <= Cin) {
      long v = v0 + vl;
      for (; v + 3L * VL < v1; v += 4L * VL) {
        float xv[4], g[4][ADELL_C1_MAXO];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long vv = v + (long)u * VL;
          xv[u] = 1.f;
          if (c < Cin) xv[u] = c < a.C0 ? a.x0[vv * a.C0 + c] : a.x1[vv * a.C1 + (c - a.C0)];
#pragma unroll
          for (int o = 0; o < ADELL_C1_MAXO; ++o) g[u][o] = o < a.Cout ? a.dy[vv * a.Cout + o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int o = 0; o < ADELL_C1_MAXO; ++o) s[o] = fmaf(g[u][o], xv[u], s[o]);
      }
      for (; v < v1; v += VL) {
        float xv = 1.f;
        if (c < Cin) xv = c < a.C0 ? a.x0[v * a.C0 + c] : a.x1[v * a.C1 + (c - a.C0)];
#pragma unroll
        for (int o = 0; o < ADELL_C1_MAXO; ++o)
          if (o < a.Cout) s[o] = fmaf(a.dy[v * a.Cout + o], xv, s[o]);
      }
    }
    for (int o = 0; o < a.Cout; ++o) {
      sh[threadIdx.x] = s[o];
      __syncthreads();
      if (vl == 0 && c <= Cin) {
        float t = 0.f;
        for (int k = 0; k < VL; ++k) t += sh[k * CL + cl];
        prow[o * (Cin + 1) + c] = t;
      }
      __syncthreads();
    }
  }
}

// The same partials for one source with whole channel quads: thread = (channel quad, voxel lane),
// 16-byte loads of x, the quad after the last one carries the bias column (x := 1).
__global__ __launch_bounds__(256) void adell_conv1_small_wgrad4_kernel(Conv1Args a, int chunk,
                                                                       int QL) {
  __shared__ float4 sh[256];
  const int Cin = a.C0, cq = Cin >> 2;
  const int VL = 256 / QL;
  const int ql = threadIdx.x % QL, vl = threadIdx.x / QL;
  const long v0 = (long)blockIdx.x * chunk;
  long v1 = v0 + chunk;
  if (v1 > a.V) v1 = a.V;
  float* prow = a.part + (size_t)blockIdx.x * a.Cout * (Cin + 1);
  float4 s[ADELL_C1_MAXO];
#pragma unroll
  for (int o = 0; o < ADELL_C1_MAXO; ++o) s[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ql <= cq) {
    long v = v0 + vl;
    for (; v + 3L * VL < v1; v += 4L * VL) {
      float4 xv[4];
      float g[4][ADELL_C1_MAXO];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long vv = v + (long)u * VL;
        xv[u] = ql < cq ? *reinterpret_cast<const float4*>(a.x0 + vv * Cin + 4 * ql)
                        : make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < ADELL_C1_MAXO; ++o) g[u][o] = o < a.Cout ? a.dy[vv * a.Cout + o] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int o = 0; o < ADELL_C1_MAXO; ++o) {
          s[o].x = fmaf(g[u][o], xv[u].x, s[o].x);
          s[o].y = fmaf(g[u][o], xv[u].y, s[o].y);
          s[o].z = fmaf(g[u][o], xv[u].z, s[o].z);
          s[o].w = fmaf(g[u][o], xv[u].w, s[o].w);
        }
    }
    for (; v < v1; v += VL) {
      const float4 xv = ql < cq ? *reinterpret_cast<const float4*>(a.x0 + v * Cin + 4 * ql)
                                : make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int o = 0; o < ADELL_C1_MAXO; ++o)
        if (o < a.Cout) {
          const float g = a.dy[v * a.Cout + o];
          s[o].x = fmaf(g, xv.x, s[o].x);
          s[o].y = fmaf(g, xv.y, s[o].y);
          s[o].z = fmaf(g, xv.z, s[o].z);
          s[o].w = fmaf(g, xv.w, s[o].w);
        }
    }
  }
  for (int o = 0; o < a.Cout; ++o) {
    sh[threadIdx.x] = s[o];
    __syncthreads();
    if (vl == 0 && ql <= cq) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < VL; ++k) {
        const float4 u = sh[k * QL + ql];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float* o4 = prow + o * (Cin + 1) + 4 * ql;
      o4[0] = t.x;                       // ql == cq: the bias column (only x is meaningful)
      if (ql < cq) { o4[1] = t.y; o4[2] = t.z; o4[3] = t.w; }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void adell_conv1_small_wgrad_fold_kernel(
    const float* __restrict__ part, int nb, int Cout, int Cin, float* __restrict__ dw,
    float* __restrict__ db) {
  __shared__ double sh[16][64];
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const int n = Cout * (Cin + 1);
  const int e = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (e < n) {
    // (eight independent loads in flight, added in the same order: 128 dependent round trips made
    // this fold of 33 values 43 us long)
    int b = vl;
    for (; b + 7 * 16 < nb; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nb; b += 16) s += (double)part[(size_t)b * n + e];
  }
  sh[vl][cl] = s;
  __syncthreads();
  if (vl != 0 || e >= n) return;
  s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[k][cl];
  const int o = e / (Cin + 1), c = e % (Cin + 1);
  if (c < Cin) dw[o * Cin + c] = (float)s;
  else if (db) db[o] = (float)s;
}

static bool adell_conv1_small_ok(const adell_conv3d_desc* d) {
  return d->KD == 1 && d->KH == 1 && d->KW == 1 && d->SD == 1 && d->SH == 1 && d->SW == 1 &&
         d->PD == 0 && d->PH == 0 && d->PW == 0 && d->Cout <= ADELL_C1_MAXO &&
         d->C0 + d->C1 <= 512;
}

static int adell_conv1_chunk(long V) {
  long nb = 2048;
  long chunk = (V + nb - 1) / nb;
  if (chunk < 64) chunk = 64;
  return (int)((chunk + 3) / 4 * 4);
}

extern "C" int adell_conv1_small_applicable(const adell_conv3d_desc* d) {
  return d && adell_conv1_small_ok(d) ? 1 : 0;
}

extern "C" long adell_conv1_small_wgrad_workspace(const adell_conv3d_desc* d) {
  if (!d || !adell_conv1_small_ok(d)) return 0;
  const long V = (long)d->N * d->D * d->H * d->W;
  const int chunk = adell_conv1_chunk(V);
  return (long)sizeof(float) * ((V + chunk - 1) / chunk) * d->Cout * (d->C0 + d->C1 + 1);
}

extern "C" int adell_conv1_small_fwd(const adell_conv3d_desc* d, const float* x0, const float* x1,
                                     const float* w, const float* bias, float* y, void* stream) {
  ADELL_REQUIRE(d && x0 && w && y && adell_conv1_small_ok(d), "conv1_small_fwd: bad arguments");
  Conv1Args a = {};
  a.x0 = x0; a.x1 = x1; a.w = w; a.bias = bias; a.y = y;
  a.V = (long)d->N * d->D * d->H * d->W; a.C0 = d->C0; a.C1 = d->C1; a.Cout = d->Cout;
  const int Cin = d->C0 + d->C1;
  int lpr = 1;
  while (lpr < 64 && (Cin + lpr - 1) / lpr > 8) lpr <<= 1;
  const int ngrp = 256 / lpr;
  long blocks = (a.V + ngrp - 1) / ngrp;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adell_conv1_small_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a, lpr);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv1_small_bwd_data(const adell_conv3d_desc* d, const float* dy,
                                          const float* w, float* dx0, float* dx1, void* stream) {
  ADELL_REQUIRE(d && dy && w && dx0 && adell_conv1_small_ok(d), "conv1_small_bwd_data: bad arguments");
  Conv1Args a = {};
  a.dy = dy; a.w = w; a.dx0 = dx0; a.dx1 = dx1;
  a.V = (long)d->N * d->D * d->H * d->W; a.C0 = d->C0; a.C1 = d->C1; a.Cout = d->Cout;
  long blocks = (a.V * (d->C0 + d->C1) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(adell_conv1_small_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv1_small_bwd_weight(const adell_conv3d_desc* d, const float* x0,
                                            const float* x1, const float* dy, float* dw, float* db,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && x0 && dy && dw && workspace && adell_conv1_small_ok(d),
                "conv1_small_bwd_weight: bad arguments");
  ADELL_REQUIRE((long)workspace_bytes >= adell_conv1_small_wgrad_workspace(d),
                "conv1_small_bwd_weight: workspace too small");
  Conv1Args a = {};
  a.x0 = x0; a.x1 = x1; a.dy = dy; a.part = (float*)workspace;
  a.V = (long)d->N * d->D * d->H * d->W; a.C0 = d->C0; a.C1 = d->C1; a.Cout = d->Cout;
  const int chunk = adell_conv1_chunk(a.V);
  const int nb = (int)((a.V + chunk - 1) / chunk);
  hipStream_t st = (hipStream_t)stream;
  if (d->C1 == 0 && d->C0 % 4 == 0 && d->C0 / 4 + 1 <= 64 && (((uintptr_t)x0) & 15) == 0) {
    int QL = 2;
    while (QL < d->C0 / 4 + 1) QL <<= 1;
    hipLaunchKernelGGL(adell_conv1_small_wgrad4_kernel, dim3(nb), dim3(256), 0, st, a, chunk, QL);
  } else {
    int CL = 4;
    while (CL < 64 && CL < d->C0 + d->C1 + 1) CL <<= 1;
    hipLaunchKernelGGL(adell_conv1_small_wgrad_kernel, dim3(nb), dim3(256), 0, st, a, chunk, CL);
  }
  const int n = d->Cout * (d->C0 + d->C1 + 1);
  hipLaunchKernelGGL(adell_conv1_small_wgrad_fold_kernel, dim3(adell_cdiv(n, 64)), dim3(1024), 0, st,
                     (const float*)workspace, nb, d->Cout, d->C0 + d->C1, dw, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// small-Cin forward and backward-data (the 2-channel input block of the U-Nets,
// unet.py:260-273: Conv3d(2 -> 2), Conv3d(2 -> 32)): exact fp32 on the vector ALU.
// An MFMA tile would pad Cin to a 16-channel chunk and Cout to 32 columns (2 -> 2 ran at 1.3
// TFLOP/s, i.e. 0.34 ms for a 34 MB pass).
//   forward      : thread = one output voxel, its KD*9*Cin inputs in registers, all output
//                  channels of a 32-wide tile accumulated from LDS-broadcast weights; bias and
//                  the per-channel (sum, sum of squares) partials in the epilogue.
//   backward-data: thread = one input voxel, dX[ci] += dY[v + P - tap][co] * w[co][ci][tap],
//                  rows of dY by 16-byte loads, weights [tap][co][ci] broadcast from LDS.
// k = 3 in H and W, 1 or 3 in D (the 2-D U-Net is the D = 1 case), stride 1, one source.
// ---------------------------------------------------------------------------
struct CinSmallArgs {
  const float* x;      // fwd: input [N][D][H][W][Cin];  dgrad: dY [N][Do][Ho][Wo][Cout]
  const float* w;      // canonical [Cout][Cin][KD][3][3]
  const float* bias;
  float* y;            // fwd: [N][Do][Ho][Wo][Cout];     dgrad: dX [N][D][H][W][Cin]
  float* part;         // fwd: [N][tiles][Cout][2] or null
  int N, D, H, W, Cout, Do, Ho, Wo, PD, PH, PW;
  int tiles;           // blocks per batch item
};

// Work mapping of both kernels: a lane is (voxel, quad of 4 output channels); the LPV = 2^k
// lanes of a voxel sit next to each other, so a wave touches 64 / LPV consecutive voxels and
// every global access is a run of whole NDHWC rows (thread-per-voxel mappings measured 2-3x
// slower than the MFMA path they replace: 64 partial lines per access).
template <int CIN, int KD>
__global__ __launch_bounds__(256) void adell_cin_small_fwd_kernel(CinSmallArgs a, int lpv) {
  constexpr int NTAP = KD * 9, COT = 64;
  __shared__ __attribute__((aligned(16))) float sw[NTAP * CIN * COT];   // [tap][ci][co]
  __shared__ float sred[4][COT][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = blockIdx.z, co0 = blockIdx.y * COT;
  const int nco = (a.Cout - co0) < COT ? (a.Cout - co0) : COT;
  // only the 4 * lpv columns the lanes read are staged (Cout = 2: 216 values, not 3456)
  const int cw = 4 * lpv, lcw = __ffs(cw) - 1;
  for (int i = tid; i < NTAP * CIN * cw; i += 256) {
    const int co = i & (cw - 1), r = i >> lcw;           // r = tap * CIN + ci
    const int ci = r % CIN, tap = r / CIN;
    sw[r * COT + co] = co < nco ? a.w[((size_t)(co0 + co) * CIN + ci) * NTAP + tap] : 0.f;
  }
  const int vpb = 256 / lpv;                       // voxels per block
  const int quad = tid & (lpv - 1);
  const long vox = (long)a.Do * a.Ho * a.Wo;
  const long v = (long)blockIdx.x * vpb + tid / lpv;
  const bool vok = v < vox;
  const int ox = (int)(v % a.Wo), oy = (int)((v / a.Wo) % a.Ho), oz = (int)(v / ((long)a.Wo * a.Ho));
  float in[NTAP][CIN];
  const float* xb = a.x + (size_t)nb * a.D * a.H * a.W * CIN;
#pragma unroll
  for (int kz = 0; kz < KD; ++kz)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int t = (kz * 3 + ky) * 3 + kx;
        const int iz = oz - a.PD + kz, iy = oy - a.PH + ky, ix = ox - a.PW + kx;
        const bool ok = vok & (iz >= 0) & (iz < a.D) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
        const float* p = xb + ((size_t)(iz * a.H + iy) * a.W + ix) * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) in[t][c] = ok ? p[c] : 0.f;
      }
  __syncthreads();
  // output channel pairs as packed fp32 FMAs; a layer of <= 2 channels skips the second pair
  f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
  if (nco > 2) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float4 wv = *reinterpret_cast<const float4*>(sw + (t * CIN + c) * COT + 4 * quad);
        const f32x2 iv = {in[t][c], in[t][c]};
        acc01 = __builtin_elementwise_fma(iv, f32x2{wv.x, wv.y}, acc01);
        acc23 = __builtin_elementwise_fma(iv, f32x2{wv.z, wv.w}, acc23);
      }
  } else {
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const f32x2 wv = *reinterpret_cast<const f32x2*>(sw + (t * CIN + c) * COT + 4 * quad);
        acc01 = __builtin_elementwise_fma(f32x2{in[t][c], in[t][c]}, wv, acc01);
      }
  }
  const int c = 4 * quad;                              // first channel of this lane inside the tile
  float val[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = vok && c + j < nco;
    val[j] = ok ? val[j] + (a.bias ? a.bias[co0 + c + j] : 0.f) : 0.f;
  }
  if (vok && c < nco) {
    float* yp = a.y + ((size_t)nb * vox + v) * a.Cout + co0 + c;
    if (c + 3 < nco && (a.Cout & 3) == 0) {
      *reinterpret_cast<float4*>(yp) = make_float4(val[0], val[1], val[2], val[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < nco) yp[j] = val[j];
    }
  }
  if (a.part) {
    float s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] = val[j];
      s2[j] = val[j] * val[j];
      for (int o = 32; o >= lpv; o >>= 1) {   // over the voxels of the wave (same quad)
        s1[j] += __shfl_xor(s1[j], o, 64);
        s2[j] += __shfl_xor(s2[j], o, 64);
      }
    }
    if (lane < lpv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sred[wave][c + j][0] = s1[j];
        sred[wave][c + j][1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < nco) {
      float* p = a.part + (((size_t)nb * a.tiles + blockIdx.x) * a.Cout + co0 + tid) * 2;
      p[0] = (sred[0][tid][0] + sred[1][tid][0]) + (sred[2][tid][0] + sred[3][tid][0]);
      p[1] = (sred[0][tid][1] + sred[1][tid][1]) + (sred[2][tid][1] + sred[3][tid][1]);
    }
  }
}

// Two input channels, at most two output channels, 3x3x3, rows of a multiple of four voxels (the
// Conv3d(2 -> 2) of the U-Net input block, unet.py:260-273: 67 MB at 2 x 128^3). The kernel above
// gives it one voxel per thread: 27 bounds-checked 8-byte loads per voxel and a block that stages the
// weights, folds its statistics and writes a partial row for 256 voxels -- 131 us, 0.06 of HBM time.
// Here a thread owns FOUR consecutive voxels of a row: per (kz, ky) it loads the six voxels x - 1 ..
// x + 4 as 16-byte pieces (36 loads per four voxels instead of 108), keeps them in registers across
// the three kx taps, reads the 108 weights as LDS broadcasts and stores its 4 x Cout outputs as
// 16-byte pieces; a block covers 1024 voxels (= one partial row of the statistics).
template <int COUT>
__global__ __launch_bounds__(256) void adell_cin2_rows_fwd_kernel(CinSmallArgs a) {
  __shared__ __attribute__((aligned(16))) float sw[27 * 2 * 2];    // [tap][ci][co], co padded to 2
  __shared__ float sred[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = blockIdx.z;
  if (tid < 108) {
    const int co = tid & 1, ci = (tid >> 1) & 1, tap = tid >> 2;
    sw[tid] = co < COUT ? a.w[((size_t)co * 2 + ci) * 27 + tap] : 0.f;
  }
  __syncthreads();
  const long vox = (long)a.Do * a.Ho * a.Wo;            // == D H W (stride 1, "same" padding)
  const long v0 = ((long)blockIdx.x * 256 + tid) * 4;
  const bool vok = v0 < vox;
  const int W = a.W, H = a.H, D = a.D;
  const int x0 = (int)(v0 % W), y = (int)((v0 / W) % H), z = (int)(v0 / ((long)W * H));
  const float* xb = a.x + (size_t)nb * vox * 2;
  float acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = 0.f;
  if (vok) {
    // all 36 loads of the nine (kz, ky) rows first, branch-free (rows / end pieces outside the
    // volume read a clamped address and are zeroed by a select): loads behind a branch per row
    // waited for each other -- nine memory round trips per group of four voxels
    f32x4 pc[9][4];
    float rowok[9];
    const float lok = x0 > 0 ? 1.f : 0.f, hok = x0 + 4 < W ? 1.f : 0.f;
    const int xl = x0 > 0 ? x0 - 2 : x0, xh = x0 + 4 < W ? x0 + 4 : x0;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iz = z - 1 + kz, iy = y - 1 + ky;
        const bool ok = (iz >= 0) & (iz < D) & (iy >= 0) & (iy < H);
        const float* row = xb + ((size_t)((ok ? iz : z) * H + (ok ? iy : y)) * W) * 2;
        rowok[kz * 3 + ky] = ok ? 1.f : 0.f;
        pc[kz * 3 + ky][0] = *reinterpret_cast<const f32x4*>(row + (size_t)xl * 2);
        pc[kz * 3 + ky][1] = *reinterpret_cast<const f32x4*>(row + (size_t)x0 * 2);
        pc[kz * 3 + ky][2] = *reinterpret_cast<const f32x4*>(row + (size_t)(x0 + 2) * 2);
        pc[kz * 3 + ky][3] = *reinterpret_cast<const f32x4*>(row + (size_t)xh * 2);
      }
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float rk = rowok[kz * 3 + ky];
        const f32x4 lo = pc[kz * 3 + ky][0] * (rk * lok), m0 = pc[kz * 3 + ky][1] * rk,
                    m1 = pc[kz * 3 + ky][2] * rk, hi = pc[kz * 3 + ky][3] * (rk * hok);
        float in[8][2];
        in[1][0] = lo.z; in[1][1] = lo.w;
        in[2][0] = m0.x; in[2][1] = m0.y; in[3][0] = m0.z; in[3][1] = m0.w;
        in[4][0] = m1.x; in[4][1] = m1.y; in[5][0] = m1.z; in[5][1] = m1.w;
        in[6][0] = hi.x; in[6][1] = hi.y;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + ((kz * 3 + ky) * 3 + kx) * 4);  // [ci][co]
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float i0 = in[1 + q + kx][0], i1 = in[1 + q + kx][1];
            acc[q][0] = fmaf(i0, wv.x, fmaf(i1, wv.z, acc[q][0]));
            if (COUT > 1) acc[q][1] = fmaf(i0, wv.y, fmaf(i1, wv.w, acc[q][1]));
          }
        }
      }
    }
  }
  const float b0 = a.bias ? a.bias[0] : 0.f, b1 = (a.bias && COUT > 1) ? a.bias[1] : 0.f;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  if (vok) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q][0] += b0;
      acc[q][1] += b1;
      s1[0] += acc[q][0]; s2[0] += acc[q][0] * acc[q][0];
      s1[1] += acc[q][1]; s2[1] += acc[q][1] * acc[q][1];
    }
    float* yp = a.y + ((size_t)nb * vox + v0) * COUT;
    if (COUT == 2) {
      *reinterpret_cast<f32x4*>(yp) = f32x4{acc[0][0], acc[0][1], acc[1][0], acc[1][1]};
      *reinterpret_cast<f32x4*>(yp + 4) = f32x4{acc[2][0], acc[2][1], acc[3][0], acc[3][1]};
    } else {
      *reinterpret_cast<f32x4*>(yp) = f32x4{acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
    }
  }
  if (a.part) {
    float r[4] = {s1[0], s2[0], s1[1], s2[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int o = 32; o > 0; o >>= 1) r[k] += __shfl_xor(r[k], o, 64);
    if (lane == 0) {
      sred[wave][0] = r[0]; sred[wave][1] = r[1]; sred[wave][2] = r[2]; sred[wave][3] = r[3];
    }
    __syncthreads();
    if (tid < COUT) {
      float* p = a.part + (((size_t)nb * a.tiles + blockIdx.x) * COUT + tid) * 2;
      p[0] = (sred[0][2 * tid] + sred[1][2 * tid]) + (sred[2][2 * tid] + sred[3][2 * tid]);
      p[1] = (sred[0][2 * tid + 1] + sred[1][2 * tid + 1]) + (sred[2][2 * tid + 1] + sred[3][2 * tid + 1]);
    }
  }
}

// the rows form takes: 2 input channels, <= 2 output channels, 3x3x3 taps with padding 1 (output =
// input extents), rows of a multiple of 4 voxels, 16-byte aligned tensors (checked at the call)
static bool adell_cin2_rows_ok(const adell_conv3d_desc* d) {
  return d->C0 == 2 && d->C1 == 0 && d->Cout >= 1 && d->Cout <= 2 && d->KD == 3 && d->KH == 3 &&
         d->KW == 3 && d->PD == 1 && d->PH == 1 && d->PW == 1 && d->SD == 1 && d->SH == 1 &&
         d->SW == 1 && d->W % 4 == 0 && d->Do == d->D && d->Ho == d->H && d->Wo == d->W;
}

template <int CIN, int KD>
__global__ __launch_bounds__(256) void adell_cin_small_bwd_data_kernel(CinSmallArgs a, int lpv) {
  constexpr int NTAP = KD * 9;
  extern __shared__ float swd[];   // [tap][co][CIN]
  const int tid = threadIdx.x;
  const int nb = blockIdx.z;
  for (int i = tid; i < NTAP * a.Cout * CIN; i += 256) {
    const int ci = i % CIN, co = (i / CIN) % a.Cout, tap = i / (CIN * a.Cout);
    swd[i] = a.w[((size_t)co * CIN + ci) * NTAP + tap];
  }
  __syncthreads();
  const int vpb = 256 / lpv;
  const int quad = tid & (lpv - 1);
  const long vox = (long)a.D * a.H * a.W;
  const long v = (long)blockIdx.x * vpb + tid / lpv;
  const bool vok = v < vox;
  const int ix = (int)(v % a.W), iy = (int)((v / a.W) % a.H), iz = (int)(v / ((long)a.W * a.H));
  const float* yb = a.x + (size_t)nb * a.Do * a.Ho * a.Wo * a.Cout;
  const int nq = a.Cout >> 2;      // quads of the dY row; lane `quad` takes quad, quad + lpv, ...
  float acc[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
#pragma unroll
  for (int kz = 0; kz < KD; ++kz)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int t = (kz * 3 + ky) * 3 + kx;
        const int oz = iz + a.PD - kz, oy = iy + a.PH - ky, ox = ix + a.PW - kx;
        if (vok & (oz >= 0) & (oz < a.Do) & (oy >= 0) & (oy < a.Ho) & (ox >= 0) & (ox < a.Wo)) {
          const float4* dyr = reinterpret_cast<const float4*>(
              yb + ((size_t)(oz * a.Ho + oy) * a.Wo + ox) * a.Cout);
          const float* wt = swd + t * a.Cout * CIN;
          for (int q = quad; q < nq; q += lpv) {
            const float4 g = dyr[q];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int c = 0; c < CIN; ++c) acc[c] += gv[j] * wt[(4 * q + j) * CIN + c];
          }
        }
      }
#pragma unroll
  for (int c = 0; c < CIN; ++c)
    for (int o = 1; o < lpv; o <<= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  if (vok && quad == 0) {
    float* o = a.y + ((size_t)nb * vox + v) * CIN;
#pragma unroll
    for (int c = 0; c < CIN; ++c) o[c] = acc[c];
  }
}

static bool adell_cin_small_ok(const adell_conv3d_desc* d) {
  const int Cin = d->C0 + d->C1;
  return d->C1 == 0 && Cin >= 1 && Cin <= 4 && (d->KD == 1 || d->KD == 3) && d->KH == 3 &&
         d->KW == 3 && d->SD == 1 && d->SH == 1 && d->SW == 1;
}

extern "C" int adell_conv_cin_small_applicable(const adell_conv3d_desc* d) {
  return d && adell_cin_small_ok(d) ? 1 : 0;
}

/* rows per batch item of the statistics-partials buffer the forward writes */
// lanes per voxel: one per quad of output channels of a 64-channel tile, a power of two
static int adell_cin_small_lpv(int Cout) {
  const int q = ((Cout < 64 ? Cout : 64) + 3) / 4;
  int l = 1;
  while (l < q) l <<= 1;
  return l;
}

extern "C" int adell_conv_cin_small_ntiles(const adell_conv3d_desc* d) {
  if (!d || !adell_cin_small_ok(d)) return ADELL_E_BADARG;
  if (adell_cin2_rows_ok(d)) return (int)(((long)d->Do * d->Ho * d->Wo + 1023) / 1024);
  const int vpb = 256 / adell_cin_small_lpv(d->Cout);
  return (int)(((long)d->Do * d->Ho * d->Wo + vpb - 1) / vpb);
}

template <int KD>
static int adell_cin_small_fwd_launch(const CinSmallArgs& a, int Cin, dim3 grid, int lpv,
                                      hipStream_t st) {
  switch (Cin) {
    case 1: hipLaunchKernelGGL((adell_cin_small_fwd_kernel<1, KD>), grid, dim3(256), 0, st, a, lpv); break;
    case 2: hipLaunchKernelGGL((adell_cin_small_fwd_kernel<2, KD>), grid, dim3(256), 0, st, a, lpv); break;
    case 3: hipLaunchKernelGGL((adell_cin_small_fwd_kernel<3, KD>), grid, dim3(256), 0, st, a, lpv); break;
    default: hipLaunchKernelGGL((adell_cin_small_fwd_kernel<4, KD>), grid, dim3(256), 0, st, a, lpv); break;
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv_cin_small_fwd(const adell_conv3d_desc* d, const float* x, const float* w,
                                        const float* bias, float* y, float* stat_partials,
                                        int partial_rows, void* stream) {
  ADELL_REQUIRE(d && x && w && y && adell_cin_small_ok(d), "conv_cin_small_fwd: bad arguments");
  ADELL_REQUIRE_ROWS(stat_partials, partial_rows, adell_conv_cin_small_ntiles(d), "conv_cin_small_fwd");
  ADELL_REQUIRE(d->N <= 65535, "conv_cin_small_fwd: batch too large");
  CinSmallArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.part = stat_partials;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo; a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  a.tiles = adell_conv_cin_small_ntiles(d);
  if (adell_cin2_rows_ok(d)) {
    // (the plan -- and with it the statistics rows -- depends on the shape only; the operands of
    // this path are whole NDHWC tensors from the allocator: 16-byte aligned or refused)
    ADELL_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0,
                  "conv_cin_small_fwd: x and y must be 16-byte aligned");
    dim3 g((unsigned)a.tiles, 1, (unsigned)d->N);
    if (d->Cout == 2)
      hipLaunchKernelGGL(adell_cin2_rows_fwd_kernel<2>, g, dim3(256), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(adell_cin2_rows_fwd_kernel<1>, g, dim3(256), 0, (hipStream_t)stream, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  const int lpv = adell_cin_small_lpv(d->Cout);
  dim3 grid((unsigned)a.tiles, (unsigned)adell_cdiv(d->Cout, 64), (unsigned)d->N);
  return d->KD == 1 ? adell_cin_small_fwd_launch<1>(a, d->C0, grid, lpv, (hipStream_t)stream)
                    : adell_cin_small_fwd_launch<3>(a, d->C0, grid, lpv, (hipStream_t)stream);
}

template <int KD>
static int adell_cin_small_bwd_launch(const CinSmallArgs& a, int Cin, dim3 grid, size_t lds,
                                      int lpv, hipStream_t st) {
  switch (Cin) {
    case 1: hipLaunchKernelGGL((adell_cin_small_bwd_data_kernel<1, KD>), grid, dim3(256), lds, st, a, lpv); break;
    case 2: hipLaunchKernelGGL((adell_cin_small_bwd_data_kernel<2, KD>), grid, dim3(256), lds, st, a, lpv); break;
    case 3: hipLaunchKernelGGL((adell_cin_small_bwd_data_kernel<3, KD>), grid, dim3(256), lds, st, a, lpv); break;
    default: hipLaunchKernelGGL((adell_cin_small_bwd_data_kernel<4, KD>), grid, dim3(256), lds, st, a, lpv); break;
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_conv_cin_small_bwd_data(const adell_conv3d_desc* d, const float* dy,
                                             const float* w, float* dx, void* stream) {
  ADELL_REQUIRE(d && dy && w && dx && adell_cin_small_ok(d), "conv_cin_small_bwd_data: bad arguments");
  ADELL_REQUIRE(d->Cout % 4 == 0 && (((uintptr_t)dy) & 15) == 0,
                "conv_cin_small_bwd_data: Cout must be a multiple of 4");
  const size_t lds = (size_t)d->KD * 9 * d->Cout * d->C0 * sizeof(float);
  ADELL_REQUIRE(lds <= 64 * 1024 && d->N <= 65535, "conv_cin_small_bwd_data: Cout too large");
  CinSmallArgs a = {};
  a.x = dy; a.w = w; a.y = dx;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo; a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  int lpv = 1;                       // lanes per voxel: one per quad of dY, at most 16
  while (lpv < d->Cout / 4 && lpv < 16) lpv <<= 1;
  const int vpb = 256 / lpv;
  dim3 grid((unsigned)(((long)d->D * d->H * d->W + vpb - 1) / vpb), 1, (unsigned)d->N);
  return d->KD == 1 ? adell_cin_small_bwd_launch<1>(a, d->C0, grid, lds, lpv, (hipStream_t)stream)
                    : adell_cin_small_bwd_launch<3>(a, d->C0, grid, lds, lpv, (hipStream_t)stream);
}
