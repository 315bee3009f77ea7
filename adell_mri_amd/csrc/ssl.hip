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

// ---------------------------------------------------------------------------
// LDS-tiled depthwise kernels for cubic K in {3, 5, 7} (the ConvNeXt shapes).
//
// A block owns 16 channels x a 4 x 4 (z, y) tile of output rows x one x-segment of WT
// voxels. The input halo tile ((4+K-1)^2 rows x WT voxels x 16 channels) is staged in LDS
// in the memory order of the activation ([row][x][channel], 64-byte channel segments
// copied with 16-byte loads, all of a thread's loads in flight together); a thread =
// (channel, output row) keeps its WT outputs in registers and, per (kz, ky), reads one
// input row and K weights from LDS and issues the fully unrolled K x WT FMA stencil along
// x. The row stride is 16 words past a multiple of 64, so the 4 rows x 16 channels of a
// wave hit 64 distinct banks. VALU-bound: 2*K^3 flops per output at the fp32 vector rate.
// When W > WT the x axis is cut into segments of WT - 2P outputs (the row carries its own
// halo); rows, columns and channels beyond the volume are zero in LDS and never stored.
// ---------------------------------------------------------------------------
template <int K, int WT>
struct DwCfg {
  static constexpr int P = K / 2;
  static constexpr int TD = 4, TH = 4;
  static constexpr int HZ = TD + K - 1, HY = TH + K - 1;
  static constexpr int RS = WT * 16 + 16;   // row stride (words)
  static constexpr int K3 = K * K * K;
  static constexpr int XT_FLOATS = HZ * HY * RS;
  static constexpr int WT_FLOATS = 16 * K3;
  static constexpr int DY_FLOATS = TD * TH * RS;
  static constexpr int WG_THREADS = ((16 * K * K + 63) / 64) * 64;
};

struct DwTile {
  int N, C, D, H, W;
  int tilesX, tilesY, tilesZ, chanBlocks;
  int seg;      // outputs per x segment (W when one segment covers the row)
  int single;   // 1: one segment, LDS row starts at x = 0
  int vec;      // 16-byte channel loads are legal (C % 4 == 0, aligned base)
};

__device__ __forceinline__ void adell_dw_decode(const DwTile& t, long item, int& n, int& z0,
                                                int& y0, int& tx) {
  tx = (int)(item % t.tilesX); item /= t.tilesX;
  const int ty = (int)(item % t.tilesY); item /= t.tilesY;
  const int tz = (int)(item % t.tilesZ);
  n = (int)(item / t.tilesZ);
  z0 = tz * 4;
  y0 = ty * 4;
}

__device__ __forceinline__ f32x4 adell_dw_load_quad(const float* __restrict__ p, int nvalid,
                                                    int vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

// Stage ROWS_Z x ROWS_Y rows (from (zs, ys)) x WT voxels x 16 channels of `src` into LDS.
// jlo/jhi: only x slots in [jlo, jhi) are taken from memory (the rest are zero).
template <int ROWS_Z, int ROWS_Y, int WT, int NT, int MAXB>
__device__ __forceinline__ void adell_dw_stage(const float* __restrict__ src, const DwTile& t,
                                               int n, int zs, int ys, int xin0, int jlo, int jhi,
                                               int c0, float* lds, int tid) {
  constexpr int RS = WT * 16 + 16;
  constexpr int TOTAL = ROWS_Z * ROWS_Y * WT * 4;
  constexpr int PER = (TOTAL + NT - 1) / NT;
  constexpr int BATCH = PER < MAXB ? PER : MAXB;
  for (int base = 0; base < PER; base += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = tid + (base + u) * NT;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (base + u < PER && idx < TOTAL) {
        const int q = idx & 3, j = (idx >> 2) % WT, r = (idx >> 2) / WT;
        const int z = zs + r / ROWS_Y, y = ys + r % ROWS_Y, x = xin0 + j;
        const int c = c0 + q * 4;
        if (z >= 0 && z < t.D && y >= 0 && y < t.H && x >= 0 && x < t.W && j >= jlo && j < jhi &&
            c < t.C)
          v[u] = adell_dw_load_quad(
              src + ((((size_t)n * t.D + z) * t.H + y) * t.W + x) * t.C + c, t.C - c, t.vec);
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int idx = tid + (base + u) * NT;
      if (base + u < PER && idx < TOTAL) {
        const int q = idx & 3, j = (idx >> 2) % WT, r = (idx >> 2) / WT;
        *reinterpret_cast<f32x4*>(lds + r * RS + j * 16 + q * 4) = v[u];
      }
    }
  }
}

struct DwTileArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  DwTile t;
  int flip;
};

template <int K, int WT>
__global__ __launch_bounds__(256) void adell_dw_tile_kernel(DwTileArgs a) {
  using Cf = DwCfg<K, WT>;
  extern __shared__ float smem[];
  float* xt = smem;
  float* wt = smem + Cf::XT_FLOATS;
  const int tid = threadIdx.x;
  const DwTile& t = a.t;
  const int cb = blockIdx.x % t.chanBlocks;     // channel blocks of one tile run together:
  const long item = blockIdx.x / t.chanBlocks;  // they share the tile's cache lines
  int n, z0, y0, tx;
  adell_dw_decode(t, item, n, z0, y0, tx);
  const int c0 = cb * 16;
  const int xin0 = t.single ? 0 : tx * t.seg - Cf::P;
  for (int i = tid; i < 16 * Cf::K3; i += 256) {
    const int c = i / Cf::K3, tap = i % Cf::K3;
    const int src = a.flip ? Cf::K3 - 1 - tap : tap;
    wt[tap * 16 + c] = (c0 + c < t.C) ? a.w[(size_t)(c0 + c) * Cf::K3 + src] : 0.f;
  }
  adell_dw_stage<Cf::HZ, Cf::HY, WT, 256, 16>(a.x, t, n, z0 - Cf::P, y0 - Cf::P, xin0, 0, WT, c0, xt,
                                          tid);
  __syncthreads();
  const int c = tid & 15, row = tid >> 4, rz = row >> 2, ry = row & 3;
  // Packed fp32 FMAs (v_pk_fma_f32, two outputs per issue: the kernel is bound by vector-ALU issue,
  // 2 K^3 flops per output). Output pair m = (2m, 2m + 1) and tap shift s = kx - P read the input
  // pair starting at 2m + s: an even-aligned pair E for even s, an odd-aligned pair O for odd s --
  // both sets are assembled once per (kz, ky) and shared by the K taps along x.
  typedef float dw_f2 __attribute__((ext_vector_type(2)));
  static_assert(WT % 2 == 0, "x segment in output pairs");
  dw_f2 acc2[WT / 2];
  const float bias = (a.b && c0 + c < t.C) ? a.b[c0 + c] : 0.f;
#pragma unroll
  for (int m = 0; m < WT / 2; ++m) acc2[m] = dw_f2{bias, bias};
#pragma unroll 1
  for (int kz = 0; kz < K; ++kz) {
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      const float* xr = xt + ((rz + kz) * Cf::HY + ry + ky) * Cf::RS + c;
      const float* wr = wt + (kz * K + ky) * K * 16 + c;
      // both pair sets straight from LDS (two dwords per ds_read2_b32): assembling the odd pairs
      // from the even ones cost six v_mov per packed FMA
      dw_f2 E[WT / 2], O[WT / 2 + 1];      // E[i] = (in[2i], in[2i+1]); O[i] = (in[2i-1], in[2i])
      float wk[K];
#pragma unroll
      for (int i = 0; i < WT / 2; ++i) E[i] = dw_f2{xr[(2 * i) * 16], xr[(2 * i + 1) * 16]};
#pragma unroll
      for (int i = 0; i <= WT / 2; ++i)
        O[i] = dw_f2{i > 0 ? xr[(2 * i - 1) * 16] : 0.f, i < WT / 2 ? xr[(2 * i) * 16] : 0.f};
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wk[kx] = wr[kx * 16];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const dw_f2 w2 = dw_f2{wk[kx], wk[kx]};
        const int sft = kx - Cf::P;
#pragma unroll
        for (int m = 0; m < WT / 2; ++m) {
          const int start = 2 * m + sft;           // first input of the pair (compile-time)
          if (start < -1 || start > WT - 1) continue;   // both inputs outside the row
          if ((start & 1) == 0)
            acc2[m] = __builtin_elementwise_fma(w2, E[start / 2], acc2[m]);
          else
            acc2[m] = __builtin_elementwise_fma(w2, O[(start + 1) / 2], acc2[m]);
        }
      }
    }
  }
  const int z = z0 + rz, y = y0 + ry;
  if (z < t.D && y < t.H && c0 + c < t.C) {
    const int jlo = t.single ? 0 : Cf::P;
    float* out = a.y + ((((size_t)n * t.D + z) * t.H + y) * t.W) * t.C + c0 + c;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int x = xin0 + j;
      if (j >= jlo && j < jlo + t.seg && x < t.W)
        out[(size_t)x * t.C] = (j & 1) ? acc2[j / 2].y : acc2[j / 2].x;
    }
  }
}

struct DwWgradArgs {
  const float* x;
  const float* dy;
  float* part;   // [splits][chanBlocks*16][K3 + 1]
  DwTile t;
  long items;
  int itemsPerSplit;
};

// thread = (channel, (kz, ky)): K accumulators (kx) in registers over every tile the block
// visits; per output row it reads the dy row and the matching shifted input row from LDS.
template <int K, int WT>
__global__ __launch_bounds__(((16 * K * K + 63) / 64) * 64) void adell_dw_wgrad_tile_kernel(
    DwWgradArgs a) {
  using Cf = DwCfg<K, WT>;
  constexpr int NT = Cf::WG_THREADS;
  extern __shared__ float smem[];
  float* xt = smem;
  float* dyt = smem + Cf::XT_FLOATS;
  const int tid = threadIdx.x;
  const DwTile& t = a.t;
  const int cb = blockIdx.x % t.chanBlocks, split = blockIdx.x / t.chanBlocks;
  const int c0 = cb * 16;
  const int c = tid & 15, kk = tid >> 4;
  const bool active = kk < K * K;
  const int kz = kk / K, ky = kk % K;
  float acc[K];
#pragma unroll
  for (int q = 0; q < K; ++q) acc[q] = 0.f;
  float sb = 0.f;
  const long first = (long)split * a.itemsPerSplit;
  long last = first + a.itemsPerSplit;
  if (last > a.items) last = a.items;
  for (long item = first; item < last; ++item) {
    int n, z0, y0, tx;
    adell_dw_decode(t, item, n, z0, y0, tx);
    const int xin0 = t.single ? 0 : tx * t.seg - Cf::P;
    const int jlo = t.single ? 0 : Cf::P;
    __syncthreads();
    adell_dw_stage<Cf::HZ, Cf::HY, WT, NT, 4>(a.x, t, n, z0 - Cf::P, y0 - Cf::P, xin0, 0, WT, c0, xt,
                                           tid);
    adell_dw_stage<Cf::TD, Cf::TH, WT, NT, 4>(a.dy, t, n, z0, y0, xin0, jlo, jlo + t.seg, c0, dyt,
                                           tid);
    __syncthreads();
    if (active) {
#pragma unroll 1
      for (int row = 0; row < 16; ++row) {
        const int rz = row >> 2, ry = row & 3;
        float g[WT], in[WT];
        const float* gr = dyt + row * Cf::RS + c;
        const float* xr = xt + ((rz + kz) * Cf::HY + ry + ky) * Cf::RS + c;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          g[j] = gr[j * 16];
          in[j] = xr[j * 16];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int j = 0; j < WT; ++j) {
            const int jj = j + kx - Cf::P;
            if (jj >= 0 && jj < WT) acc[kx] = fmaf(g[j], in[jj], acc[kx]);
          }
        if (kk == 0) {
#pragma unroll
          for (int j = 0; j < WT; ++j) sb += g[j];
        }
      }
    }
  }
  if (active) {
    float* dst = a.part + ((size_t)split * t.chanBlocks * 16 + c0 + c) * (Cf::K3 + 1);
#pragma unroll
    for (int kx = 0; kx < K; ++kx) dst[(kz * K + ky) * K + kx] = acc[kx];
    if (kk == 0) dst[Cf::K3] = sb;
  }
}

// dw[c][tap] = sum over splits (fixed order); column K3 of the partials is db
__global__ __launch_bounds__(256) void adell_dw_wgrad_reduce_kernel(
    const float* __restrict__ part, int splits, int cpad, int C, int K3, float* __restrict__ dw,
    float* __restrict__ db) {
  const long total = (long)C * (K3 + 1);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int c = (int)(i / (K3 + 1)), tap = (int)(i % (K3 + 1));
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += part[((size_t)sp * cpad + c) * (K3 + 1) + tap];
    if (tap < K3) dw[(size_t)c * K3 + tap] = s;
    else if (db) db[c] = s;
  }
}

static int adell_dw_check(int N, int C, int D, int H, int W, int KD, int KH, int KW) {
  ADELL_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, "dwconv: bad dims");
  ADELL_REQUIRE(KD >= 1 && KH >= 1 && KW >= 1 && (KD & 1) && (KH & 1) && (KW & 1),
                "dwconv: odd kernel sizes ('same' padding) only");
  return ADELL_OK;
}

// ---------------------------------------------------------------------------------------------
// Depthwise 7^3 marching along z (rows of 9 ... 16 voxels: ConvNeXt's first stage, 64 crops x 16^3 x
// 96 channels). The tile kernel above needs a 10 x 10 halo of rows for its 4 x 4 (z, y) outputs --
// 6.25x the input bytes, 109 KB of LDS, ONE block per CU whose staging (~12 us of L2 reads) and
// stencil (~9.5 us) run one after the other: 546 us per launch where the tensors move in 50.
// Here a block owns 16 channels x 8 output rows (y) x the whole x row and walks z: a ring of 8
// planes of 14 rows in LDS (7 live + the one being written: one barrier per step), one new plane
// staged per output plane -- halo only in y (1.75x) -- and its loads issued two steps ahead, under
// the stencil. Thread = (channel, row, x half): 8 outputs as 4 packed pairs, 28 v_pk_fma_f32 per
// (kz, ky).
constexpr int DZ_K = 7, DZ_P = 3, DZ_WT = 16, DZ_TY = 8, DZ_HY = DZ_TY + DZ_K - 1;
constexpr int DZ_RS = DZ_WT * 16 + 16, DZ_PLANE = DZ_HY * DZ_RS, DZ_SLOTS = 8, DZ_K3 = 343;
constexpr int DZ_QUADS = DZ_HY * DZ_WT * 4, DZ_PER = (DZ_QUADS + 255) / 256;

struct DwZrArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int N, C, D, H, W, flip;
  int tilesY, chanBlocks, nseg, seglen, vec;
};

template <int XH>
__device__ __forceinline__ void adell_dz_stencil(const float* __restrict__ ring, const float* __restrict__ wt,
                                                 int first_slot, int row, int c,
                                                 float (&acc)[8]) {
  typedef float dz_f2 __attribute__((ext_vector_type(2)));
  dz_f2 acc2[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc2[m] = dz_f2{acc[2 * m], acc[2 * m + 1]};
#pragma unroll 1
  for (int kz = 0; kz < DZ_K; ++kz) {
    const float* pl = ring + ((first_slot + kz) & (DZ_SLOTS - 1)) * DZ_PLANE;
    // (ky NOT unrolled: measured 457 -> 759 us per launch with the seven rows unrolled)
#pragma unroll 1
    for (int ky = 0; ky < DZ_K; ++ky) {
      // inputs t = 0 .. 13 are x = 8 XH - 3 + t; outside [0, 16) they are zero (compile time)
      const float* xr = pl + (row + ky) * DZ_RS + (8 * XH - DZ_P) * 16 + c;
      const float* wr = wt + (kz * DZ_K + ky) * DZ_K * 16 + c;
      float in[14];
#pragma unroll
      for (int t = 0; t < 14; ++t) {
        const int x = 8 * XH - DZ_P + t;
        in[t] = (x >= 0 && x < DZ_WT) ? xr[t * 16] : 0.f;
      }
      dz_f2 E[7], O[6];
#pragma unroll
      for (int i = 0; i < 7; ++i) E[i] = dz_f2{in[2 * i], in[2 * i + 1]};
#pragma unroll
      for (int i = 0; i < 6; ++i) O[i] = dz_f2{in[2 * i + 1], in[2 * i + 2]};
#pragma unroll
      for (int kx = 0; kx < DZ_K; ++kx) {
        const float wv = wr[kx * 16];
        const dz_f2 w2 = dz_f2{wv, wv};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int start = 2 * m + kx;
          if ((start & 1) == 0)
            acc2[m] = __builtin_elementwise_fma(w2, E[start / 2], acc2[m]);
          else
            acc2[m] = __builtin_elementwise_fma(w2, O[(start - 1) / 2], acc2[m]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc[2 * m] = acc2[m].x;
    acc[2 * m + 1] = acc2[m].y;
  }
}

__global__ __launch_bounds__(256) void adell_dw_zring_kernel(DwZrArgs a) {
  extern __shared__ float smem[];
  float* ring = smem;                            // [8 slots][14 rows][16 x][16 ch] (+16 pad per row)
  float* wt = smem + DZ_SLOTS * DZ_PLANE;        // [343 taps][16 ch]
  const int tid = threadIdx.x;
  const int c = tid & 15, q = tid >> 4, row = q & 7, xh = q >> 3;
  int item = blockIdx.x;
  const int cb = item % a.chanBlocks; item /= a.chanBlocks;   // channel blocks of a tile run together
  const int seg = item % a.nseg; item /= a.nseg;
  const int ty = item % a.tilesY;
  const int n = item / a.tilesY;
  const int c0 = cb * 16, y0 = ty * DZ_TY;
  const int z0 = seg * a.seglen;
  const int z1 = (z0 + a.seglen < a.D) ? z0 + a.seglen : a.D;
  for (int i = tid; i < 16 * DZ_K3; i += 256) {
    const int cc = i / DZ_K3, tap = i - cc * DZ_K3;
    const int src = a.flip ? DZ_K3 - 1 - tap : tap;
    wt[tap * 16 + cc] = (c0 + cc < a.C) ? a.w[(size_t)(c0 + cc) * DZ_K3 + src] : 0.f;
  }
  // staging roles: quads idx = tid + 256 u of the plane's 14 x 16 x 4
  unsigned goff[DZ_PER], loff[DZ_PER];
  bool gok[DZ_PER];
#pragma unroll
  for (int u = 0; u < DZ_PER; ++u) {
    const int idx = tid + 256 * u;
    const int qd = idx & 3, j = (idx >> 2) % DZ_WT, r = (idx >> 2) / DZ_WT;
    const int yy = y0 - DZ_P + r, cc = c0 + 4 * qd;
    gok[u] = idx < DZ_QUADS && yy >= 0 && yy < a.H && j < a.W && cc < a.C;
    goff[u] = gok[u] ? (unsigned)((yy * a.W + j) * a.C + cc) : 0u;
    loff[u] = (unsigned)(r * DZ_RS + j * 16 + 4 * qd);
  }
  const float* xn = a.x + (size_t)n * a.D * a.H * a.W * a.C;
  f32x4 pre[2][DZ_PER];
  auto fetch = [&](f32x4* v, int p) {
    const bool pok = p >= 0 && p < a.D;
    const float* base = xn + (size_t)(pok ? p : 0) * a.H * a.W * a.C;
#pragma unroll
    for (int u = 0; u < DZ_PER; ++u) {
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      if (pok && gok[u]) f = adell_dw_load_quad(base + goff[u], a.C - (c0 + 4 * ((tid + 256 * u) & 3)), a.vec);
      v[u] = f;
    }
  };
  auto put = [&](int slot, const f32x4* v) {
    float* dst = ring + slot * DZ_PLANE;
#pragma unroll
    for (int u = 0; u < DZ_PER; ++u)
      if (tid + 256 * u < DZ_QUADS) *reinterpret_cast<f32x4*>(dst + loff[u]) = v[u];
  };
  const float bias = (a.b && c0 + c < a.C) ? a.b[c0 + c] : 0.f;
  const int p0 = z0 - DZ_P;
  const int nsteps = (z1 - z0) + DZ_K - 1;
  fetch(pre[0], p0);
  fetch(pre[1], p0 + 1);
  for (int ib = 0; ib < nsteps; ib += 2) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const int i = ib + jb;
      if (i >= nsteps) break;
      put(i & (DZ_SLOTS - 1), pre[jb]);
      fetch(pre[jb], p0 + i + 2);        // under the stencils of this step and the next
      __syncthreads();
      if (i < DZ_K - 1) continue;
      const int z = z0 + i - (DZ_K - 1);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bias;
      if (xh == 0)
        adell_dz_stencil<0>(ring, wt, i - (DZ_K - 1), row, c, acc);
      else
        adell_dz_stencil<1>(ring, wt, i - (DZ_K - 1), row, c, acc);
      const int y = y0 + row;
      if (y < a.H && c0 + c < a.C) {
        float* out = a.y + ((((size_t)n * a.D + z) * a.H + y) * a.W + 8 * xh) * a.C + c0 + c;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (8 * xh + j < a.W) out[(size_t)j * a.C] = acc[j];
      }
    }
  }
}

// 1 when the z-marching kernel takes the problem; fills the launch geometry
static int adell_dw_zring_plan(int N, int C, int D, int H, int W, int KD, int KH, int KW, DwZrArgs* z) {
  if (KD != 7 || KH != 7 || KW != 7 || W > DZ_WT || W <= 8 || D < 4) return 0;
  z->N = N; z->C = C; z->D = D; z->H = H; z->W = W;
  z->tilesY = adell_cdiv(H, DZ_TY);
  z->chanBlocks = adell_cdiv(C, 16);
  const long base = (long)N * z->tilesY * z->chanBlocks;
  // segments: fill ~2 x the CUs when the batch alone does not, at least 8 planes each (6 priming planes)
  long nseg = 1;
  while (base * nseg < 512 && D / (nseg + 1) >= 8) ++nseg;
  z->seglen = adell_cdiv(D, (int)nseg);
  z->nseg = adell_cdiv(D, z->seglen);
  if (base * z->nseg > 0x7fffffffL) return 0;
  return 1;
}

static int adell_dw_zring_launch(DwZrArgs z, hipStream_t st) {
  static bool attr_done = false;
  const size_t lds = (size_t)(DZ_SLOTS * DZ_PLANE + 16 * DZ_K3) * sizeof(float);
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dw_zring_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const long blocks = (long)z.N * z.tilesY * z.chanBlocks * z.nseg;
  hipLaunchKernelGGL(adell_dw_zring_kernel, dim3((unsigned)blocks), dim3(256), lds, st, z);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// tiled path: cubic K in {3,5,7}. Returns the x-row width WT (4 / 8 / 16) or 0.
static int adell_dw_plan(int N, int C, int D, int H, int W, int KD, int KH, int KW, DwTile* t) {
  if (KD != KH || KH != KW || (KD != 3 && KD != 5 && KD != 7)) return 0;
  const int K = KD, P = K / 2;
  int WT = W <= 4 ? 4 : (W <= 8 ? 8 : 16);
  t->N = N; t->C = C; t->D = D; t->H = H; t->W = W;
  t->vec = (C % 4) == 0;
  t->single = W <= WT;
  if (!t->single && WT - 2 * P < 4) return 0;
  t->seg = t->single ? W : WT - 2 * P;
  t->tilesX = t->single ? 1 : adell_cdiv(W, t->seg);
  t->tilesY = adell_cdiv(H, 4);
  t->tilesZ = adell_cdiv(D, 4);
  t->chanBlocks = adell_cdiv(C, 16);
  const long blocks = (long)N * t->tilesZ * t->tilesY * t->tilesX * t->chanBlocks;
  if (blocks > 0x7fffffffL) return 0;
  return WT;
}

template <int K, int WT>
static int adell_dw_tile_launch(const DwTileArgs& a, hipStream_t st) {
  using Cf = DwCfg<K, WT>;
  static bool attr_done = false;
  auto kern = adell_dw_tile_kernel<K, WT>;
  const size_t lds = (size_t)(Cf::XT_FLOATS + Cf::WT_FLOATS) * sizeof(float);
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const DwTile& t = a.t;
  const long blocks = (long)t.N * t.tilesZ * t.tilesY * t.tilesX * t.chanBlocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

#define ADELL_DW_DISPATCH(FN, K, WT, ...)                                  \
  do {                                                                     \
    if (K == 3) {                                                          \
      if (WT == 4) return FN<3, 4>(__VA_ARGS__);                           \
      if (WT == 8) return FN<3, 8>(__VA_ARGS__);                           \
      return FN<3, 16>(__VA_ARGS__);                                       \
    }                                                                      \
    if (K == 5) {                                                          \
      if (WT == 4) return FN<5, 4>(__VA_ARGS__);                           \
      if (WT == 8) return FN<5, 8>(__VA_ARGS__);                           \
      return FN<5, 16>(__VA_ARGS__);                                       \
    }                                                                      \
    if (WT == 4) return FN<7, 4>(__VA_ARGS__);                             \
    if (WT == 8) return FN<7, 8>(__VA_ARGS__);                             \
    return FN<7, 16>(__VA_ARGS__);                                         \
  } while (0)

extern "C" int adell_dw_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                const float* x, const float* y);
extern "C" int adell_dw_mfma_launch(const float* x, const float* w, const float* b, float* y, int N,
                                    int C, int D, int H, int W, int flip, void* stream);

extern "C" int adell_dw_dense_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                 const float* x, const float* y);
extern "C" int adell_dw_dense_launch(const float* x, const float* w, const float* b, float* y, int N,
                                     int C, int D, int H, int W, int flip, void* stream);

static int adell_dw_launch(DwArgs a, hipStream_t st) {
  // 7^3 taps on volumes of at most 4^3 voxels: a dense per-channel matrix (csrc/dw_dense.hip)
  if (adell_dw_dense_ok(a.N, a.C, a.D, a.H, a.W, a.KD, a.KH, a.KW, a.x, a.y))
    return adell_dw_dense_launch(a.x, a.w, a.b, a.y, a.N, a.C, a.D, a.H, a.W, a.flip, st);
  // 7^3 taps on rows of 9 .. 16 voxels: the Toeplitz form on the f16x3 MFMA (csrc/dw_mfma.hip)
  if (adell_dw_mfma_ok(a.N, a.C, a.D, a.H, a.W, a.KD, a.KH, a.KW, a.x, a.y))
    return adell_dw_mfma_launch(a.x, a.w, a.b, a.y, a.N, a.C, a.D, a.H, a.W, a.flip, st);
  DwZrArgs zr = {};
  if (adell_dw_zring_plan(a.N, a.C, a.D, a.H, a.W, a.KD, a.KH, a.KW, &zr)) {
    zr.x = a.x; zr.w = a.w; zr.b = a.b; zr.y = a.y; zr.flip = a.flip;
    zr.vec = (a.C % 4 == 0) && ((uintptr_t)a.x % 16 == 0);
    return adell_dw_zring_launch(zr, st);
  }
  DwTileArgs ta;
  const int WT = adell_dw_plan(a.N, a.C, a.D, a.H, a.W, a.KD, a.KH, a.KW, &ta.t);
  if (WT) {
    ta.x = a.x; ta.w = a.w; ta.b = a.b; ta.y = a.y; ta.flip = a.flip;
    ta.t.vec = ta.t.vec && ((uintptr_t)a.x % 16 == 0);
    ADELL_DW_DISPATCH(adell_dw_tile_launch, a.KD, WT, ta, st);
  }
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
// Generic fallback (non-cubic kernels).
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

static int adell_dw_wgrad_splits(const DwTile& t, long* items_out, int* ips_out) {
  const long items = (long)t.N * t.tilesZ * t.tilesY * t.tilesX;
  long splits = adell_cdiv(512, t.chanBlocks);
  if (splits > items) splits = items;
  if (splits < 1) splits = 1;
  const int ips = (int)((items + splits - 1) / splits);
  splits = (items + ips - 1) / ips;
  *items_out = items;
  *ips_out = ips;
  return (int)splits;
}

template <int K, int WT>
static int adell_dw_wgrad_tile_launch(DwWgradArgs a, int splits, hipStream_t st) {
  using Cf = DwCfg<K, WT>;
  static bool attr_done = false;
  auto kern = adell_dw_wgrad_tile_kernel<K, WT>;
  const size_t lds = (size_t)(Cf::XT_FLOATS + Cf::DY_FLOATS) * sizeof(float);
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(splits * a.t.chanBlocks)), dim3(Cf::WG_THREADS), lds, st,
                     a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_dw_wgrad_dispatch(int K, int WT, const DwWgradArgs& a, int splits,
                                   hipStream_t st) {
  ADELL_DW_DISPATCH(adell_dw_wgrad_tile_launch, K, WT, a, splits, st);
}

extern "C" int adell_dw_wgrad_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                      const float* x, const float* dy);
extern "C" long adell_dw_wgrad_mfma_workspace_floats(int N, int C);
extern "C" int adell_dw_wgrad_mfma_launch(const float* x, const float* dy, float* workspace, int N,
                                          int C, int D, int H, int W, int* chunks_out, void* stream);

// floats of workspace adell_dwconv3d_bwd_weight needs (0: none)
extern "C" long adell_dwconv3d_bwd_weight_workspace_floats(int N, int C, int D, int H, int W,
                                                           int KD, int KH, int KW) {
  DwTile t;
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  long need = 0;
  // (the MFMA form is chosen per call from the operands' alignment: size for either)
  if (KD == 7 && KH == 7 && KW == 7 && C % 4 == 0) need = adell_dw_wgrad_mfma_workspace_floats(N, C);
  if (!adell_dw_plan(N, C, D, H, W, KD, KH, KW, &t)) return need;
  long items;
  int ips;
  const int splits = adell_dw_wgrad_splits(t, &items, &ips);
  const long tile = (long)splits * t.chanBlocks * 16 * ((long)KD * KH * KW + 1);
  return tile > need ? tile : need;
}

extern "C" int adell_dwconv3d_bwd_weight(int N, int C, int D, int H, int W, int KD, int KH,
                                         int KW, const float* x, const float* dy, float* dw,
                                         float* db, float* workspace, void* stream) {
  int rc = adell_dw_check(N, C, D, H, W, KD, KH, KW);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(x && dy && dw, "dwconv_bwd_weight: null pointer");
  hipStream_t st = (hipStream_t)stream;
  // 7^3 taps on rows of 9 .. 16 voxels: rows as the reduction dimension of f16x3 MFMA products
  // (csrc/dw_wgrad_mfma.hip), per-chunk partial sums folded by the reduce kernel below
  if (adell_dw_wgrad_mfma_ok(N, C, D, H, W, KD, KH, KW, x, dy)) {
    ADELL_REQUIRE(workspace, "dwconv_bwd_weight: workspace of "
                             "adell_dwconv3d_bwd_weight_workspace_floats() floats required");
    int chunks = 0;
    rc = adell_dw_wgrad_mfma_launch(x, dy, workspace, N, C, D, H, W, &chunks, stream);
    if (rc != ADELL_OK) return rc;
    long blocks = ((long)C * 344 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adell_dw_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       workspace, chunks, C, C, 343, dw, db);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  DwWgradArgs a;
  const int WT = adell_dw_plan(N, C, D, H, W, KD, KH, KW, &a.t);
  if (WT) {
    ADELL_REQUIRE(workspace, "dwconv_bwd_weight: workspace of "
                             "adell_dwconv3d_bwd_weight_workspace_floats() floats required");
    a.x = x; a.dy = dy; a.part = workspace;
    a.t.vec = a.t.vec && (((uintptr_t)x | (uintptr_t)dy) % 16 == 0);
    const int splits = adell_dw_wgrad_splits(a.t, &a.items, &a.itemsPerSplit);
    rc = adell_dw_wgrad_dispatch(KD, WT, a, splits, st);
    if (rc != ADELL_OK) return rc;
    const int K3 = KD * KH * KW;
    long blocks = ((long)C * (K3 + 1) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adell_dw_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       workspace, splits, a.t.chanBlocks * 16, C, K3, dw, db);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  hipLaunchKernelGGL(adell_dwconv3d_wgrad_kernel, dim3(KD * KH * KW, adell_cdiv(C, 64)), dim3(256),
                     0, st, x, dy, dw, db, N, C, D, H, W, KD, KH, KW);
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

// ---- the same forward on many blocks (B <= 64): a block owns a chunk of 128 feature columns ------
// (the one-block kernel above took 1.0 ms of the VICReg ConvNeXt step at B = 32: B^2 D products on
// one CU). Pass 1: per chunk and view the column means / variances, the centred chunk in LDS and its
// B x B partial Gram matrix; pass 2 (one block) folds the partials in block order (deterministic).
// scratch (after the 4 D + 2 B^2 floats the backward reads): [2 views][nblk][B^2] partial Gram
// matrices, then [nblk][4] partial (hinge, var^2, inv, -).
constexpr int VIC_CH = 128, VIC_MAXB = 64;
__global__ __launch_bounds__(256) void adell_vicreg_fwd_chunk_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, int B, int D, float min_var,
    float eps, float* __restrict__ scratch) {
  extern __shared__ float vsm[];            // [B][VIC_CH + 1] centred values of the chunk
  __shared__ float sh[16];
  const int tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x;
  const int j0 = blk * VIC_CH, j = j0 + tid;
  const bool col = tid < VIC_CH && j < D;
  const float* xs[2] = {x1, x2};
  float* gpart = scratch + 4L * D + 2L * B * B;
  float* part = gpart + 2L * nblk * B * B + (long)blk * 4;
  float hinge = 0.f, var2 = 0.f, inv = 0.f;
  for (int view = 0; view < 2; ++view) {
    const float* x = xs[view];
    float* mean = scratch + view * 2 * D;
    float* var = mean + D;
    if (tid < VIC_CH) {
      float m = 0.f, v = 0.f;
      if (col) {
        for (int b = 0; b < B; ++b) m += x[(size_t)b * D + j];
        m /= (float)B;
        for (int b = 0; b < B; ++b) {
          const float d = x[(size_t)b * D + j] - m;
          v += d * d;
          vsm[b * (VIC_CH + 1) + tid] = d;
        }
        v /= (float)(B - 1);
        mean[j] = m;
        var[j] = v;
        hinge += fmaxf(min_var - sqrtf(v + eps), 0.f);
        var2 += v * v;
      } else {
        for (int b = 0; b < B; ++b) vsm[b * (VIC_CH + 1) + tid] = 0.f;
      }
    }
    __syncthreads();
    float* G = gpart + ((long)view * nblk + blk) * B * B;
    for (int p = tid; p < B * B; p += 256) {
      const int a = p / B, b = p - a * B;
      const float* ra = vsm + a * (VIC_CH + 1);
      const float* rb = vsm + b * (VIC_CH + 1);
      float g = 0.f;
#pragma unroll 8
      for (int q = 0; q < VIC_CH; ++q) g += ra[q] * rb[q];
      G[p] = g;
    }
    __syncthreads();
  }
  if (col)
    for (int b = 0; b < B; ++b) {
      const float d = x1[(size_t)b * D + j] - x2[(size_t)b * D + j];
      inv += d * d;
    }
  hinge = adell_block_sum(hinge, sh);
  var2 = adell_block_sum(var2, sh);
  inv = adell_block_sum(inv, sh);
  if (tid == 0) {
    part[0] = hinge;
    part[1] = var2;
    part[2] = inv;
    part[3] = 0.f;
  }
}

__global__ __launch_bounds__(1024) void adell_vicreg_fwd_fold_kernel(int B, int D, int nblk,
                                                                    float* __restrict__ scratch,
                                                                    float* __restrict__ out) {
  __shared__ float sh[16];
  const float* gpart = scratch + 4L * D + 2L * B * B;
  const float* part = gpart + 2L * nblk * B * B;
  float g2 = 0.f;
  for (int view = 0; view < 2; ++view) {
    float* G = scratch + 4 * D + (size_t)view * B * B;
    for (int p = threadIdx.x; p < B * B; p += blockDim.x) {
      float g = 0.f;
      for (int k = 0; k < nblk; ++k) g += gpart[((long)view * nblk + k) * B * B + p];
      G[p] = g;
      g2 += g * g;
    }
  }
  g2 = adell_block_sum(g2, sh);
  if (threadIdx.x == 0) {
    float hinge = 0.f, var2 = 0.f, inv = 0.f;
    for (int k = 0; k < nblk; ++k) {
      hinge += part[4 * k];
      var2 += part[4 * k + 1];
      inv += part[4 * k + 2];
    }
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

static int adell_vicreg_chunks(int B, int D) {
  return (B <= VIC_MAXB && D >= 2 * VIC_CH) ? (D + VIC_CH - 1) / VIC_CH : 0;   // 0: the one-block kernel
}

extern "C" long adell_vicreg_scratch_floats(int B, int D) {
  const long nblk = adell_vicreg_chunks(B, D);
  return 4L * D + 2L * B * B + nblk * (2L * B * B + 4);
}

extern "C" int adell_vicreg_fwd(const float* x1, const float* x2, int B, int D, float min_var,
                                float eps, float* scratch, float* out3, void* stream) {
  ADELL_REQUIRE(x1 && x2 && scratch && out3, "vicreg_fwd: null pointer");
  ADELL_REQUIRE(B > 1 && D > 0, "vicreg_fwd: need B > 1, D > 0");
  const int nblk = adell_vicreg_chunks(B, D);
  if (nblk > 0) {
    const size_t lds = (size_t)B * (VIC_CH + 1) * sizeof(float);
    hipLaunchKernelGGL(adell_vicreg_fwd_chunk_kernel, dim3((unsigned)nblk), dim3(256), lds,
                       (hipStream_t)stream, x1, x2, B, D, min_var, eps, scratch);
    hipLaunchKernelGGL(adell_vicreg_fwd_fold_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, D,
                       nblk, scratch, out3);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
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

// ---------------------------------------------------------------------------
// Local contrastive loss of the semi-supervised U-Net (adell_mri/modules/
// semi_supervised_segmentation/losses.py:480-526, used by UNetContrastiveSemiSL.step_semi_sl_loco,
// semi_supervised_segmentation/pl.py:244-281): f1, f2 are the decoder features of two views,
// NDHWC [B][S][C]. Per voxel s the B x B matrix cos(f2[i,s,:], f1[j,s,:]) / T is soft-maxed over
// j and loss_i = mean_s -log(max(softmax_i[i], eps)). HBM-bound: a group of G = min(64, pow2 <=
// C/4) lanes owns one voxel, each lane a channel quad (coalesced float4 loads), the B^2 + 2B
// partial dot products are folded across the group with xor shuffles, every lane then holds the
// full matrix. The backward recomputes the matrix and writes its own quads of df1 (and df2).
// torch's cosine_similarity clamps each norm at 1e-8 separately (checked against torch 2.10).
// ---------------------------------------------------------------------------
#define ADELL_LOCO_MAXB 8
#define ADELL_LOCO_COS_EPS 1e-8f

template <int B>
struct LocoMat {
  float dot[B][B];  // dot[i][j] = f2[i] . f1[j]
  float n1[B], n2[B];
};

template <int B>
__device__ __forceinline__ void adell_loco_dots(LocoMat<B>& m, const float4* __restrict__ f1,
                                                const float4* __restrict__ f2, long S, long s,
                                                int C4, int G, int gl, bool ok) {
#pragma unroll
  for (int i = 0; i < B; ++i) {
    m.n1[i] = 0.f;
    m.n2[i] = 0.f;
#pragma unroll
    for (int j = 0; j < B; ++j) m.dot[i][j] = 0.f;
  }
  if (ok) {
    for (int q = gl; q < C4; q += G) {
      float4 a[B], b[B];
#pragma unroll
      for (int i = 0; i < B; ++i) {
        a[i] = f2[((size_t)i * S + s) * C4 + q];
        b[i] = f1[((size_t)i * S + s) * C4 + q];
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        m.n2[i] += a[i].x * a[i].x + a[i].y * a[i].y + a[i].z * a[i].z + a[i].w * a[i].w;
        m.n1[i] += b[i].x * b[i].x + b[i].y * b[i].y + b[i].z * b[i].z + b[i].w * b[i].w;
#pragma unroll
        for (int j = 0; j < B; ++j)
          m.dot[i][j] += a[i].x * b[j].x + a[i].y * b[j].y + a[i].z * b[j].z + a[i].w * b[j].w;
      }
    }
  }
  for (int o = G >> 1; o > 0; o >>= 1) {
#pragma unroll
    for (int i = 0; i < B; ++i) {
      m.n1[i] += __shfl_xor(m.n1[i], o, 64);
      m.n2[i] += __shfl_xor(m.n2[i], o, 64);
#pragma unroll
      for (int j = 0; j < B; ++j) m.dot[i][j] += __shfl_xor(m.dot[i][j], o, 64);
    }
  }
}

// grid-strided over voxels; part: [gridDim.x][B] block sums of the per-voxel losses
template <int B>
__global__ __launch_bounds__(256) void adell_loco_fwd_kernel(const float* __restrict__ f1,
                                                             const float* __restrict__ f2, long S,
                                                             int C, int G, float invT, float eps,
                                                             float* __restrict__ part) {
  __shared__ float sh[256];
  const int C4 = C >> 2, gl = threadIdx.x & (G - 1), vpb = 256 / G;
  const long stride = (long)gridDim.x * vpb;
  const long rounds = (S + stride - 1) / stride;  // every thread runs every round (shuffles)
  float lsum[B];
#pragma unroll
  for (int i = 0; i < B; ++i) lsum[i] = 0.f;
  for (long r = 0; r < rounds; ++r) {
    const long s = r * stride + (long)blockIdx.x * vpb + threadIdx.x / G;
    const bool ok = s < S;
    LocoMat<B> m;
    adell_loco_dots<B>(m, reinterpret_cast<const float4*>(f1), reinterpret_cast<const float4*>(f2),
                       S, s, C4, G, gl, ok);
    if (!ok || gl != 0) continue;
    float r1[B], r2[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      r1[i] = 1.f / fmaxf(sqrtf(m.n1[i]), ADELL_LOCO_COS_EPS);
      r2[i] = 1.f / fmaxf(sqrtf(m.n2[i]), ADELL_LOCO_COS_EPS);
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      float z[B], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < B; ++j) {
        z[j] = m.dot[i][j] * r2[i] * r1[j] * invT;
        mx = fmaxf(mx, z[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < B; ++j) den += expf(z[j] - mx);
      const float p = expf(z[i] - mx) / den;
      lsum[i] += -logf(fmaxf(p, eps));
    }
  }
#pragma unroll
  for (int i = 0; i < B; ++i) {
    const float t = adell_block_sum(lsum[i], sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * B + i] = t;
  }
}

// loss[i] = (sum over blocks, fixed order) / S
__global__ void adell_loco_finalize_kernel(const float* __restrict__ part, int nblocks, int B,
                                           double invS, float* __restrict__ loss) {
  const int i = threadIdx.x;
  if (i >= B) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += (double)part[(size_t)b * B + i];
  loss[i] = (float)(t * invS);
}

// df1[j] = sum_i c_ij d cos_ij / d f1[j], df2[i] = sum_j c_ij d cos_ij / d f2[i] with
// c_ij = gloss[i] / S * (softmax_ij - delta_ij) / T where softmax_ii > eps (else the max() of
// the reference passes no gradient), d cos / d b = ra rb (a - [|b| > 1e-8] (a.b) rb^2 b).
template <int B>
__global__ __launch_bounds__(256) void adell_loco_bwd_kernel(const float* __restrict__ f1,
                                                             const float* __restrict__ f2,
                                                             const float* __restrict__ gloss,
                                                             long S, int C, int G, float invT,
                                                             float eps, float invS,
                                                             float* __restrict__ df1,
                                                             float* __restrict__ df2) {
  const int C4 = C >> 2, gl = threadIdx.x & (G - 1), vpb = 256 / G;
  const long stride = (long)gridDim.x * vpb;
  const long rounds = (S + stride - 1) / stride;
  float g[B];
#pragma unroll
  for (int i = 0; i < B; ++i) g[i] = gloss[i] * invS * invT;
  const float4* F1 = reinterpret_cast<const float4*>(f1);
  const float4* F2 = reinterpret_cast<const float4*>(f2);
  for (long r = 0; r < rounds; ++r) {
    const long s = r * stride + (long)blockIdx.x * vpb + threadIdx.x / G;
    const bool ok = s < S;
    LocoMat<B> m;
    adell_loco_dots<B>(m, F1, F2, S, s, C4, G, gl, ok);
    if (!ok) continue;
    float r1[B], r2[B], k1[B], k2[B];  // k: 1/|.|^2 when the norm is above the clamp, else 0
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const float a1 = sqrtf(m.n1[i]), a2 = sqrtf(m.n2[i]);
      r1[i] = 1.f / fmaxf(a1, ADELL_LOCO_COS_EPS);
      r2[i] = 1.f / fmaxf(a2, ADELL_LOCO_COS_EPS);
      k1[i] = a1 > ADELL_LOCO_COS_EPS ? r1[i] * r1[i] : 0.f;
      k2[i] = a2 > ADELL_LOCO_COS_EPS ? r2[i] * r2[i] : 0.f;
    }
    float c[B][B];  // c_ij * ra_i * rb_j
#pragma unroll
    for (int i = 0; i < B; ++i) {
      float z[B], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < B; ++j) {
        z[j] = m.dot[i][j] * r2[i] * r1[j] * invT;
        mx = fmaxf(mx, z[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < B; ++j) {
        z[j] = expf(z[j] - mx);
        den += z[j];
      }
      const float live = (z[i] / den > eps) ? g[i] : 0.f;
#pragma unroll
      for (int j = 0; j < B; ++j)
        c[i][j] = live * (z[j] / den - (i == j ? 1.f : 0.f)) * r2[i] * r1[j];
    }
    for (int q = gl; q < C4; q += G) {
      float4 a[B], b[B];
#pragma unroll
      for (int i = 0; i < B; ++i) {
        a[i] = F2[((size_t)i * S + s) * C4 + q];
        b[i] = F1[((size_t)i * S + s) * C4 + q];
      }
      if (df1) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          float self = 0.f;
#pragma unroll
          for (int i = 0; i < B; ++i) {
            o.x += c[i][j] * a[i].x; o.y += c[i][j] * a[i].y;
            o.z += c[i][j] * a[i].z; o.w += c[i][j] * a[i].w;
            self += c[i][j] * m.dot[i][j];
          }
          self *= k1[j];
          o.x -= self * b[j].x; o.y -= self * b[j].y; o.z -= self * b[j].z; o.w -= self * b[j].w;
          reinterpret_cast<float4*>(df1)[((size_t)j * S + s) * C4 + q] = o;
        }
      }
      if (df2) {
#pragma unroll
        for (int i = 0; i < B; ++i) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          float self = 0.f;
#pragma unroll
          for (int j = 0; j < B; ++j) {
            o.x += c[i][j] * b[j].x; o.y += c[i][j] * b[j].y;
            o.z += c[i][j] * b[j].z; o.w += c[i][j] * b[j].w;
            self += c[i][j] * m.dot[i][j];
          }
          self *= k2[i];
          o.x -= self * a[i].x; o.y -= self * a[i].y; o.z -= self * a[i].z; o.w -= self * a[i].w;
          reinterpret_cast<float4*>(df2)[((size_t)i * S + s) * C4 + q] = o;
        }
      }
    }
  }
}

static int adell_loco_group(int C) {
  int G = 1;
  while (2 * G <= C / 4 && 2 * G <= 64) G *= 2;
  return G;
}
static int adell_loco_blocks(long S, int G) {
  const long vpb = 256 / G;
  long b = (S + vpb - 1) / vpb;
  if (b > 4096) b = 4096;
  return (int)(b < 1 ? 1 : b);
}

extern "C" long adell_loco_loss_workspace(int B, long S, int C) {
  ADELL_REQUIRE(B >= 1 && B <= ADELL_LOCO_MAXB, "loco_loss: batch must be 1..%d (got %d)",
                ADELL_LOCO_MAXB, B);
  ADELL_REQUIRE(S > 0 && C >= 4 && (C & 3) == 0, "loco_loss: need S > 0 and C %% 4 == 0 (C = %d)", C);
  return (long)sizeof(float) * adell_loco_blocks(S, adell_loco_group(C)) * B;
}

#define ADELL_LOCO_DISPATCH(KERN, B, ...)                                                      \
  switch (B) {                                                                                 \
    case 1: hipLaunchKernelGGL(KERN<1>, __VA_ARGS__); break;                                   \
    case 2: hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); break;                                   \
    case 3: hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); break;                                   \
    case 4: hipLaunchKernelGGL(KERN<4>, __VA_ARGS__); break;                                   \
    case 5: hipLaunchKernelGGL(KERN<5>, __VA_ARGS__); break;                                   \
    case 6: hipLaunchKernelGGL(KERN<6>, __VA_ARGS__); break;                                   \
    case 7: hipLaunchKernelGGL(KERN<7>, __VA_ARGS__); break;                                   \
    default: hipLaunchKernelGGL(KERN<8>, __VA_ARGS__); break;                                  \
  }

extern "C" int adell_loco_loss_fwd(const float* f1, const float* f2, int B, long S, int C,
                                   float temperature, float eps, float* loss, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(f1 && f2 && loss && workspace, "loco_loss_fwd: null pointer");
  ADELL_REQUIRE(B >= 1 && B <= ADELL_LOCO_MAXB, "loco_loss_fwd: batch must be 1..%d (got %d)",
                ADELL_LOCO_MAXB, B);
  ADELL_REQUIRE(S > 0 && C >= 4 && (C & 3) == 0, "loco_loss_fwd: need S > 0 and C %% 4 == 0");
  ADELL_REQUIRE(temperature > 0.f, "loco_loss_fwd: temperature must be positive");
  ADELL_REQUIRE((((uintptr_t)f1 | (uintptr_t)f2) & 15) == 0, "loco_loss_fwd: unaligned features");
  ADELL_REQUIRE((long)workspace_bytes >= adell_loco_loss_workspace(B, S, C),
                "loco_loss_fwd: workspace too small");
  const int G = adell_loco_group(C), blocks = adell_loco_blocks(S, G);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  ADELL_LOCO_DISPATCH(adell_loco_fwd_kernel, B, dim3(blocks), dim3(256), 0, st, f1, f2, S, C, G,
                      1.f / temperature, eps, part);
  hipLaunchKernelGGL(adell_loco_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)part,
                     blocks, B, 1.0 / (double)S, loss);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_loco_loss_bwd(const float* f1, const float* f2, const float* gloss, int B,
                                   long S, int C, float temperature, float eps, float* df1,
                                   float* df2, void* stream) {
  ADELL_REQUIRE(f1 && f2 && gloss && (df1 || df2), "loco_loss_bwd: null pointer");
  ADELL_REQUIRE(B >= 1 && B <= ADELL_LOCO_MAXB, "loco_loss_bwd: batch must be 1..%d (got %d)",
                ADELL_LOCO_MAXB, B);
  ADELL_REQUIRE(S > 0 && C >= 4 && (C & 3) == 0, "loco_loss_bwd: need S > 0 and C %% 4 == 0");
  ADELL_REQUIRE(temperature > 0.f, "loco_loss_bwd: temperature must be positive");
  ADELL_REQUIRE((((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)df1 | (uintptr_t)df2) & 15) == 0,
                "loco_loss_bwd: unaligned pointer");
  const int G = adell_loco_group(C), blocks = adell_loco_blocks(S, G);
  ADELL_LOCO_DISPATCH(adell_loco_bwd_kernel, B, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                      f1, f2, gloss, S, C, G, 1.f / temperature, eps, (float)(1.0 / (double)S),
                      df1, df2);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Cosine-similarity losses between two [B][D] embedding batches (the non-VICReg methods of
// SelfSLBasePL.init_loss, adell_mri/modules/self_supervised/pl.py:202-212):
//   kind 0  simsiam_loss  = -mean_i cos(x1_i, x2_i)              (losses/functional.py:138-150)
//   kind 1  byol_loss     = 2 * simsiam_loss + 2                 (losses/functional.py:153-164)
//   kind 2  NTXentLoss    (SimCLR, losses/ntxent.py:11-46): Z = [relu(x1); relu(x2)] (2B rows),
//           S = cos(Z_i, Z_j) / T, loss = mean_i ( -S[i][(i + B) % 2B] + logsumexp_{j != i} S[i][j] )
// cosine = dot / (max(|a|, 1e-8) max(|b|, 1e-8)) as torch.nn.functional.cosine_similarity.
// scratch (floats): G [R][R] raw dot products (R = 2B), W [R][R] = dL/dC. Small tensors: one block
// per row for the Gram matrix and for the gradient, one block for the loss itself.
// ---------------------------------------------------------------------------
#define ADELL_PAIR_MAXR 256
#define ADELL_PAIR_EPS 1e-8f

struct PairArgs {
  const float* x1;
  const float* x2;
  float* G;
  float* W;
  float* loss;
  const float* g;
  float* dx1;
  float* dx2;
  int B, D, kind, relu;
  float invT;
};

__device__ __forceinline__ float adell_pair_elem(const PairArgs& a, int row, int d) {
  const float v = row < a.B ? a.x1[(size_t)row * a.D + d] : a.x2[(size_t)(row - a.B) * a.D + d];
  return a.relu ? fmaxf(v, 0.f) : v;
}

// grid R, block 256: G[i][j] for the j this kind needs (kinds 0 / 1: itself and its partner)
__global__ __launch_bounds__(256) void adell_pair_gram_kernel(PairArgs a) {
  const int R = 2 * a.B, i = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int partner = (i + a.B) % R;
  for (int j = wave; j < R; j += 4) {
    if (a.kind != 2 && j != i && j != partner) continue;   // wave-uniform
    float s = 0.f;
    for (int d = lane; d < a.D; d += 64) s += adell_pair_elem(a, i, d) * adell_pair_elem(a, j, d);
    s = adell_wave_sum(s);
    if (lane == 0) a.G[(size_t)i * R + j] = s;
  }
}

// one block: loss and W = dL/dC (C = cosine matrix)
__global__ __launch_bounds__(256) void adell_pair_loss_kernel(PairArgs a) {
  __shared__ float sh[4];
  __shared__ float rn[ADELL_PAIR_MAXR];
  const int R = 2 * a.B, tid = threadIdx.x;
  for (int i = tid; i < R; i += 256) rn[i] = 1.f / fmaxf(sqrtf(a.G[(size_t)i * R + i]), ADELL_PAIR_EPS);
  for (int i = tid; i < R * R; i += 256) a.W[i] = 0.f;
  __syncthreads();
  float part = 0.f;
  if (a.kind != 2) {
    const float coef = a.kind == 0 ? -1.f / a.B : -2.f / a.B;
    for (int i = tid; i < a.B; i += 256) {
      const float c = a.G[(size_t)i * R + i + a.B] * rn[i] * rn[i + a.B];
      part += coef * c;
      a.W[(size_t)i * R + i + a.B] = coef;
    }
  } else {
    for (int i = tid; i < R; i += 256) {
      const int p = (i + a.B) % R;
      float mx = -INFINITY;
      for (int j = 0; j < R; ++j)
        if (j != i) mx = fmaxf(mx, a.G[(size_t)i * R + j] * rn[i] * rn[j] * a.invT);
      float den = 0.f;
      for (int j = 0; j < R; ++j)
        if (j != i) den += expf(a.G[(size_t)i * R + j] * rn[i] * rn[j] * a.invT - mx);
      const float sp = a.G[(size_t)i * R + p] * rn[i] * rn[p] * a.invT;
      part += (-sp + mx + logf(den)) / R;
      for (int j = 0; j < R; ++j)
        if (j != i) {
          const float sm = expf(a.G[(size_t)i * R + j] * rn[i] * rn[j] * a.invT - mx) / den;
          a.W[(size_t)i * R + j] = (sm - (j == p ? 1.f : 0.f)) * a.invT / R;
        }
    }
  }
  const float total = adell_block_sum(part, sh);
  if (tid == 0) a.loss[0] = total + (a.kind == 1 ? 2.f : 0.f);
}

// grid R, block 256: dz_i = g * (v - [|z_i| > eps] (v . zh_i) zh_i) / max(|z_i|, eps) with
// v = sum_j (W[i][j] + W[j][i]) zh_j, zh = z / max(|z|, eps); ReLU mask on top for kind 2
__global__ __launch_bounds__(256) void adell_pair_bwd_kernel(PairArgs a) {
  extern __shared__ float v[];   // [D]
  __shared__ float sh[4];
  __shared__ float wt[ADELL_PAIR_MAXR];
  const int R = 2 * a.B, i = blockIdx.x, tid = threadIdx.x;
  float* out = i < a.B ? (a.dx1 ? a.dx1 + (size_t)i * a.D : nullptr)
                       : (a.dx2 ? a.dx2 + (size_t)(i - a.B) * a.D : nullptr);
  if (!out) return;   // whole block
  for (int j = tid; j < R; j += 256) {
    const float w = a.W[(size_t)i * R + j] + a.W[(size_t)j * R + i];
    wt[j] = w == 0.f ? 0.f : w / fmaxf(sqrtf(a.G[(size_t)j * R + j]), ADELL_PAIR_EPS);
  }
  __syncthreads();
  const float ni = sqrtf(a.G[(size_t)i * R + i]);
  const float ri = 1.f / fmaxf(ni, ADELL_PAIR_EPS);
  float s = 0.f;
  for (int d = tid; d < a.D; d += 256) {
    float t = 0.f;
    for (int j = 0; j < R; ++j)
      if (wt[j] != 0.f) t += wt[j] * adell_pair_elem(a, j, d);
    v[d] = t;
    s += t * adell_pair_elem(a, i, d) * ri;
  }
  s = adell_block_sum(s, sh);
  if (!(ni > ADELL_PAIR_EPS)) s = 0.f;
  const float g = a.g[0];
  const float* raw = i < a.B ? a.x1 + (size_t)i * a.D : a.x2 + (size_t)(i - a.B) * a.D;
  for (int d = tid; d < a.D; d += 256) {
    float r = g * (v[d] - s * adell_pair_elem(a, i, d) * ri) * ri;
    if (a.relu && !(raw[d] > 0.f)) r = 0.f;
    out[d] = r;
  }
}

static int adell_pair_check(const float* x1, const float* x2, int B, int D, int kind) {
  ADELL_REQUIRE(x1 && x2, "pair_loss: null pointer");
  ADELL_REQUIRE(kind >= 0 && kind <= 2, "pair_loss: kind must be 0 (simsiam), 1 (byol), 2 (nt-xent)");
  ADELL_REQUIRE(B >= 1 && 2 * B <= ADELL_PAIR_MAXR, "pair_loss: batch must be 1..%d (got %d)",
                ADELL_PAIR_MAXR / 2, B);
  ADELL_REQUIRE(D >= 1 && D <= 36000, "pair_loss: embedding size must be 1..36000 (got %d)", D);
  return ADELL_OK;
}

extern "C" long adell_pair_loss_scratch_floats(int B, int D) { return 8L * B * B; }

extern "C" int adell_pair_loss_fwd(const float* x1, const float* x2, int B, int D, int kind,
                                   float temperature, int apply_relu, float* scratch,
                                   float* loss, void* stream) {
  int rc = adell_pair_check(x1, x2, B, D, kind);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(scratch && loss, "pair_loss_fwd: null pointer");
  ADELL_REQUIRE(kind != 2 || temperature > 0.f, "pair_loss_fwd: temperature must be positive");
  PairArgs a = {};
  const int R = 2 * B;
  a.x1 = x1; a.x2 = x2; a.G = scratch; a.W = scratch + (size_t)R * R; a.loss = loss;
  a.B = B; a.D = D; a.kind = kind; a.relu = (kind == 2 && apply_relu) ? 1 : 0;
  a.invT = kind == 2 ? 1.f / temperature : 1.f;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adell_pair_gram_kernel, dim3(R), dim3(256), 0, st, a);
  hipLaunchKernelGGL(adell_pair_loss_kernel, dim3(1), dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_pair_loss_bwd(const float* x1, const float* x2, int B, int D, int kind,
                                   float temperature, int apply_relu, const float* scratch,
                                   const float* g, float* dx1, float* dx2, void* stream) {
  int rc = adell_pair_check(x1, x2, B, D, kind);
  if (rc != ADELL_OK) return rc;
  ADELL_REQUIRE(scratch && g && (dx1 || dx2), "pair_loss_bwd: null pointer");
  PairArgs a = {};
  const int R = 2 * B;
  a.x1 = x1; a.x2 = x2; a.G = const_cast<float*>(scratch);
  a.W = const_cast<float*>(scratch) + (size_t)R * R; a.g = g; a.dx1 = dx1; a.dx2 = dx2;
  a.B = B; a.D = D; a.kind = kind; a.relu = (kind == 2 && apply_relu) ? 1 : 0;
  a.invT = kind == 2 ? 1.f / temperature : 1.f;
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_pair_bwd_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(adell_pair_bwd_kernel, dim3(R), dim3(256), (size_t)D * sizeof(float),
                     (hipStream_t)stream, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
