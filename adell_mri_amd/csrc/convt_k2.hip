// ConvTranspose3d with kernel = stride = 2 on every axis, 32 / 64 input and 16 / 32 / 64 output channels (the
// decoder upscaling of the high-resolution U-Net levels, unet.py:445-458) as streaming GEMMs on the
// fp32 MFMA (exact fp32 products). Every input voxel v feeds exactly the 8 output voxels 2 v + f:
//
//   forward   y[2v + f][co]  = b[co] + sum_ci x[v][ci] w[ci][co][f]      M = v, K = Cin,   N = (f, co)
//   dX        dx[v][ci]      = sum_{f, co} dy[2v + f][co] w[ci][co][f]   M = v, K = (f, co), N = ci
//   dW        dw[ci][co][f]  = sum_v x[v][ci] dy[2v + f][co]             M = ci, K = v,     N = (f, co)
//
// The implicit-GEMM kernels treat these as convolutions with a one-voxel halo and a pixel-shuffle
// store, 4 blocks per CU of staging-bound work (0.28 / 0.39 / 0.34 ms at 2 x 64^3 -> 128^3, 32
// channels); nothing here needs a halo, so a wave takes 32 consecutive input voxels, the (fz, fy)
// pair of its wave index (both fx: 2 Cout contiguous floats per voxel on the fine grid) and streams:
// operands go through a wave-private LDS tile (coalesced float4 loads, conflict-free column reads),
// weights live in registers, and the only traffic is the one pass over the fine-grid tensor.
#include "common.h"

struct ConvTK2Args {
  const float* x;     // [N][D][H][W][Cin]
  const float* w;     // canonical ConvTranspose weight [Cin][Cout][2][2][2]
  const float* bias;  // [Cout] or null
  const float* dy;    // [N][2D][2H][2W][Cout]
  float* y;           // [N][2D][2H][2W][Cout]
  float* dx;          // [N][D][H][W][Cin]
  float* ws;          // dW partials [blocks][pairs][8 f][32 ci][32 co]
  float* wsdb;        // db partials [blocks][4 waves][Cout] or null
  int N, D, H, W, Cin, Cout;
  int FX;             // factor along x: 2, or 1 (factors (2, 2, 1): the fine grid keeps W)
  long V;             // N D H W
  int ntiles;         // ceil(V / 32)
};

// fine-grid row (in units of voxels) of coarse voxel v at offset (fz, fy, fx = 0); v < 2^31
// (checked on the host): 32-bit divisions
__device__ __forceinline__ size_t adell_ctk2_fine(const ConvTK2Args& a, long v64, int fz, int fy) {
  const unsigned v = (unsigned)v64, W = (unsigned)a.W, H = (unsigned)a.H, D = (unsigned)a.D;
  const unsigned t1 = v / W, x = v - t1 * W;
  const unsigned t2 = t1 / H, y = t1 - t2 * H;
  const unsigned n = t2 / D, z = t2 - n * D;
  return (((size_t)n * 2 * D + 2 * z + fz) * 2 * H + 2 * y + fy) * (size_t)(a.FX * W) + (size_t)a.FX * x;
}

__device__ __forceinline__ size_t adell_ctk2_shfl(size_t v, int src) {
  const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
  return ((size_t)hi << 32) | lo;
}

// wave-private copy of 32 rows x WIDTH floats into an LDS tile with row stride `ld`, in two halves
// so that the loads of the NEXT tile are in flight while the MFMAs of the current one run: lane l
// holds the element offset of row l & 31 in `myrow` (the address arithmetic of a row is done once,
// by one lane); rows >= valid are zero-filled.
template <int WIDTH>
struct CtK2Regs {
  static constexpr int W4 = WIDTH / 4, PER = 32 * W4 / 64;
  float4 f[PER];
};
template <int WIDTH>
__device__ __forceinline__ void adell_ctk2_fetch(CtK2Regs<WIDTH>& g, int valid, int lane,
                                                 const float* src, size_t myrow, int col0 = 0) {
  constexpr int W4 = WIDTH / 4, PER = 32 * W4 / 64;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = lane + 64 * u, r = i / W4, c4 = i - r * W4;
    const size_t off = adell_ctk2_shfl(myrow, r);
    g.f[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < valid) g.f[u] = *reinterpret_cast<const float4*>(src + off + col0 + 4 * c4);
  }
}
template <int WIDTH>
__device__ __forceinline__ void adell_ctk2_put(float* tile, int ld, int lane,
                                               const CtK2Regs<WIDTH>& g) {
  constexpr int W4 = WIDTH / 4, PER = 32 * W4 / 64;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = lane + 64 * u, r = i / W4, c4 = i - r * W4;
    float* q = tile + r * ld + 4 * c4;
    q[0] = g.f[u].x; q[1] = g.f[u].y; q[2] = g.f[u].z; q[3] = g.f[u].w;
  }
}

// ---- forward, 16 output channels (UNETR's full-resolution upscaling): the 32 MFMA columns are
// (fx, co) -- the two fine-grid voxels of a coarse voxel are 32 contiguous floats -- one tile -------
template <int CIN>
__global__ __launch_bounds__(256) void adell_convt_k2_fwd16_kernel(ConvTK2Args a) {
  constexpr int LD = CIN + 1, KS = CIN / 2;
  __shared__ float sx[4][32 * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int fz = wave >> 1, fy = wave & 1, fx = li >> 4, co = li & 15;
  float bw[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    bw[s] = a.w[((size_t)(2 * s + lh) * 16 + co) * 8 + (fz * 2 + fy) * 2 + fx];
  const float bcol = a.bias ? a.bias[co] : 0.f;
  float* tile = sx[wave];
  CtK2Regs<CIN> regs;
  auto fetch = [&](int t) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_fetch<CIN>(regs, valid, lane, a.x + (size_t)v0 * CIN, (size_t)li * CIN);
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_put<CIN>(tile, LD, lane, regs);
    const size_t yrow = li < valid ? adell_ctk2_fine(a, v0 + li, fz, fy) : 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done
    if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[li * LD + 2 * s + lh], bw[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < valid) a.y[adell_ctk2_shfl(yrow, row) * 16 + li] = acc[r] + bcol;
    }
  }
}

// ---- forward: grid (blocks, Cout / 32); wave w = (fz, fy), both fx (FX = 1: the one) -----------
template <int CIN, int FX = 2>
__global__ __launch_bounds__(256) void adell_convt_k2_fwd_kernel(ConvTK2Args a) {
  constexpr int LD = CIN + 1, KS = CIN / 2, F = 4 * FX;
  __shared__ float sx[4][32 * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int fz = wave >> 1, fy = wave & 1, n0 = blockIdx.y * 32, col = n0 + li;
  float bw[FX][KS];
#pragma unroll
  for (int fx = 0; fx < FX; ++fx)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      bw[fx][s] = a.w[((size_t)(2 * s + lh) * a.Cout + col) * F + (fz * 2 + fy) * FX + fx];
  const float bcol = a.bias ? a.bias[col] : 0.f;
  float* tile = sx[wave];
  CtK2Regs<CIN> regs;
  auto fetch = [&](int t) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_fetch<CIN>(regs, valid, lane, a.x + (size_t)v0 * CIN, (size_t)li * CIN);
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_put<CIN>(tile, LD, lane, regs);
    const size_t yrow = li < valid ? adell_ctk2_fine(a, v0 + li, fz, fy) : 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done
    if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);   // in flight under the MFMAs / stores
    f32x16 acc[FX];
#pragma unroll
    for (int fx = 0; fx < FX; ++fx)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fx][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float av = tile[li * LD + 2 * s + lh];
#pragma unroll
      for (int fx = 0; fx < FX; ++fx)
        acc[fx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[fx][s], acc[fx], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < valid) {
        float* o = a.y + adell_ctk2_shfl(yrow, row) * a.Cout + col;
        o[0] = acc[0][r] + bcol;
        if constexpr (FX == 2) o[a.Cout] = acc[1][r] + bcol;
      }
    }
  }
}

// ---- backward-data: grid (blocks, Cin / 32); wave w = (fz, fy) owns a K slice of 2 Cout ---------
template <int COUT, int FX = 2>
__global__ __launch_bounds__(256) void adell_convt_k2_dx_kernel(ConvTK2Args a) {
  constexpr int KW = FX * COUT, LD = KW + 1, KS = KW / 2, F = 4 * FX;
  extern __shared__ float smem[];
  float* sred = smem + 4 * 32 * LD;   // [3][32][33] partial tiles of waves 1..3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int fz = wave >> 1, fy = wave & 1, n0 = blockIdx.y * 32, ci = n0 + li;
  float bw[KS];   // B[k = (fx, co)][j = ci]
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + lh, fx = k / COUT, co = k - fx * COUT;
    bw[s] = a.w[((size_t)ci * COUT + co) * F + (fz * 2 + fy) * FX + fx];
  }
  float* tile = smem + wave * 32 * LD;
  CtK2Regs<KW> regs;
  auto fetch = [&](int t) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_fetch<KW>(regs, valid, lane, a.dy,
                         li < valid ? adell_ctk2_fine(a, v0 + li, fz, fy) * COUT : 0);
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    adell_ctk2_put<KW>(tile, LD, lane, regs);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[li * LD + 2 * s + lh], bw[s], acc, 0, 0, 0);
    // fold the four K slices in wave order: waves 1..3 park theirs, wave 0 adds and stores
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sred[((wave - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < valid)
          a.dx[(size_t)(v0 + row) * a.Cin + ci] =
              ((acc[r] + sred[row * 33 + li]) + sred[(32 + row) * 33 + li]) +
              sred[(64 + row) * 33 + li];
      }
    }
    __syncthreads();   // sred is free again
  }
}

// ---- weight gradient: grid (blocks, (Cin / 32) (Cout / 32)); wave w = (fz, fy), both fx --------
template <int FX = 2>
__global__ __launch_bounds__(256) void adell_convt_k2_dw_kernel(ConvTK2Args a) {
  constexpr int LDX = 33, LDY = 65;
  __shared__ float sx[4][32 * LDX];
  __shared__ float sy[4][32 * LDY];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int fz = wave >> 1, fy = wave & 1;
  const int nct = a.Cout / 32, ci0 = (blockIdx.y / nct) * 32, co0 = (blockIdx.y % nct) * 32;
  f32x16 acc[FX];
#pragma unroll
  for (int fx = 0; fx < FX; ++fx)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[fx][r] = 0.f;
  float* tx = sx[wave];
  float* tyl = sy[wave];
  CtK2Regs<32> rx, ry0, ry1;
  // bias gradient as a by-product (blocks of the first ci tile): this lane's pieces of dY are always
  // the channels co0 + 4 (lane & 7) .. + 3, rows past the tensor are zero
  const bool do_db = a.wsdb != nullptr && ci0 == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int t) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    const size_t yrow = li < valid ? adell_ctk2_fine(a, v0 + li, fz, fy) * a.Cout : 0;
    adell_ctk2_fetch<32>(rx, valid, lane, a.x + (size_t)v0 * a.Cin + ci0, (size_t)li * a.Cin);
    // the two fine-grid voxels (fx = 0, 1) of a coarse voxel are adjacent rows of Cout floats
    adell_ctk2_fetch<32>(ry0, valid, lane, a.dy, yrow, co0);
    if constexpr (FX == 2) adell_ctk2_fetch<32>(ry1, valid, lane, a.dy, yrow, a.Cout + co0);
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    adell_ctk2_put<32>(tx, LDX, lane, rx);
    adell_ctk2_put<32>(tyl, LDY, lane, ry0);
    if constexpr (FX == 2) adell_ctk2_put<32>(tyl + 32, LDY, lane, ry1);
    if (do_db) {
#pragma unroll
      for (int u = 0; u < CtK2Regs<32>::PER; ++u) {
        dbacc.x += ry0.f[u].x; dbacc.y += ry0.f[u].y; dbacc.z += ry0.f[u].z; dbacc.w += ry0.f[u].w;
        if constexpr (FX == 2) {
          dbacc.x += ry1.f[u].x; dbacc.y += ry1.f[u].y; dbacc.z += ry1.f[u].z; dbacc.w += ry1.f[u].w;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int v = 2 * s + lh;
      const float av = tx[v * LDX + li];                 // A[i = ci][k = v]
#pragma unroll
      for (int fx = 0; fx < FX; ++fx)
        acc[fx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, tyl[v * LDY + fx * 32 + li], acc[fx],
                                                        0, 0, 0);
    }
  }
  // every wave owns its own FX sub-positions: no fold inside the block
  float* out = a.ws + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (4 * FX)) * 1024;
#pragma unroll
  for (int fx = 0; fx < FX; ++fx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // ci
      out[(size_t)((fz * 2 + fy) * FX + fx) * 1024 + row * 32 + li] = acc[fx][r];
    }
  if (do_db) {
    // lanes with the same lane & 7 hold the same four channels: fold over lane bits 3, 4, 5
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      dbacc.x += __shfl_xor(dbacc.x, o, 64);
      dbacc.y += __shfl_xor(dbacc.y, o, 64);
      dbacc.z += __shfl_xor(dbacc.z, o, 64);
      dbacc.w += __shfl_xor(dbacc.w, o, 64);
    }
    if (lane < 8) {
      float* o = a.wsdb + ((size_t)blockIdx.x * 4 + wave) * a.Cout + co0 + 4 * lane;
      o[0] = dbacc.x; o[1] = dbacc.y; o[2] = dbacc.z; o[3] = dbacc.w;
    }
  }
}

// ---- weight gradient, 16 output channels: N = (fx, co), one tile per (fz, fy) wave ---------------
// partials [blocks][Cin / 32][4 (fz, fy)][32 ci][32 (fx, co)]
__global__ __launch_bounds__(256) void adell_convt_k2_dw16_kernel(ConvTK2Args a) {
  constexpr int LDX = 33, LDY = 33;
  __shared__ float sx[4][32 * LDX];
  __shared__ float sy[4][32 * LDY];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const int fz = wave >> 1, fy = wave & 1;
  const int ci0 = blockIdx.y * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float* tx = sx[wave];
  float* tyl = sy[wave];
  CtK2Regs<32> rx, ry;
  const bool do_db = a.wsdb != nullptr && ci0 == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int t) {
    const long v0 = (long)t * 32;
    const int valid = (a.V - v0) < 32 ? (int)(a.V - v0) : 32;
    const size_t yrow = li < valid ? adell_ctk2_fine(a, v0 + li, fz, fy) * 16 : 0;
    adell_ctk2_fetch<32>(rx, valid, lane, a.x + (size_t)v0 * a.Cin + ci0, (size_t)li * a.Cin);
    adell_ctk2_fetch<32>(ry, valid, lane, a.dy, yrow);   // both fx: 32 contiguous floats
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    adell_ctk2_put<32>(tx, LDX, lane, rx);
    adell_ctk2_put<32>(tyl, LDY, lane, ry);
    if (do_db) {
#pragma unroll
      for (int u = 0; u < CtK2Regs<32>::PER; ++u) {
        dbacc.x += ry.f[u].x; dbacc.y += ry.f[u].y; dbacc.z += ry.f[u].z; dbacc.w += ry.f[u].w;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (t + (int)gridDim.x < a.ntiles) fetch(t + gridDim.x);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int v = 2 * s + lh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tx[v * LDX + li], tyl[v * LDY + li], acc, 0, 0, 0);
    }
  }
  float* out = a.ws + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + (fz * 2 + fy)) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // ci
    out[row * 32 + li] = acc[r];
  }
  if (do_db) {
    // this lane's pieces are always floats 4 (lane & 7) .. + 3 of the (fx, co) row: channels
    // 4 (lane & 3) .. + 3 -- fold over lane bits 2 (fx), 3, 4, 5
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      dbacc.x += __shfl_xor(dbacc.x, o, 64);
      dbacc.y += __shfl_xor(dbacc.y, o, 64);
      dbacc.z += __shfl_xor(dbacc.z, o, 64);
      dbacc.w += __shfl_xor(dbacc.w, o, 64);
    }
    if (lane < 4) {
      float* o = a.wsdb + ((size_t)blockIdx.x * 4 + wave) * 16 + 4 * lane;
      o[0] = dbacc.x; o[1] = dbacc.y; o[2] = dbacc.z; o[3] = dbacc.w;
    }
  }
}

__global__ __launch_bounds__(256) void adell_convt_k2_dw16_reduce_kernel(
    const float* __restrict__ ws, int blocks, int Cin, float* __restrict__ dw,
    const float* __restrict__ wsdb, float* __restrict__ db) {
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long ndw = (long)Cin * 16 * 8;
  if (i >= ndw) {
    const long co = i - ndw;
    if (db == nullptr || co >= 16) return;
    float s = 0.f;
    for (int r = lane; r < blocks * 4; r += 64) s += wsdb[(size_t)r * 16 + co];
    s = adell_wave_sum(s);
    if (lane == 0) db[co] = s;
    return;
  }
  const int f = (int)(i & 7);
  const long cc = i >> 3;
  const int co = (int)(cc & 15), ci = (int)(cc >> 4);
  const int pairs = Cin / 32;
  const float* p = ws + ((size_t)(ci >> 5) * 4 + (f >> 1)) * 1024 + (ci & 31) * 32 + (f & 1) * 16 + co;
  const size_t stride = (size_t)pairs * 4 * 1024;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += p[(size_t)b * stride];
  s = adell_wave_sum(s);
  if (lane == 0) dw[i] = s;
}

// dw[ci][co][f] = sum over blocks: one wave per value (lane l adds blocks l, l + 64, ...)
__global__ __launch_bounds__(256) void adell_convt_k2_dw_reduce_kernel(
    const float* __restrict__ ws, int blocks, int Cin, int Cout, float* __restrict__ dw,
    const float* __restrict__ wsdb, float* __restrict__ db, int F) {
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long ndw = (long)Cin * Cout * F;
  if (i >= ndw) {   // whole wave: the waves past dW fold the bias-gradient rows, one channel each
    const long co = i - ndw;
    if (db == nullptr || co >= Cout) return;
    float s = 0.f;
    for (int r = lane; r < blocks * 4; r += 64) s += wsdb[(size_t)r * Cout + co];
    s = adell_wave_sum(s);
    if (lane == 0) db[co] = s;
    return;
  }
  const int f = (int)(i % F);
  const long cc = i / F;
  const int co = (int)(cc % Cout), ci = (int)(cc / Cout);
  const int nct = Cout / 32, pairs = (Cin / 32) * nct;
  const int pair = (ci >> 5) * nct + (co >> 5);
  const float* p = ws + ((size_t)pair * F + f) * 1024 + (ci & 31) * 32 + (co & 31);
  const size_t stride = (size_t)pairs * F * 1024;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += p[(size_t)b * stride];
  s = adell_wave_sum(s);
  if (lane == 0) dw[i] = s;
}

static bool adell_convt_k2_ok(int N, int D, int H, int W, int Cin, int Cout, int FX = 2) {
  // (Cout == 16: the (fx, co) column forms above, factors 2 x 2 x 2 only)
  return N >= 1 && D >= 1 && H >= 1 && W >= 1 && (FX == 1 || FX == 2) && (Cin == 32 || Cin == 64) &&
         ((Cout == 16 && FX == 2) || Cout == 32 || Cout == 64) && (long)N * D * H * W >= 32768 &&
         (long)N * D * H * W < 0x7fffffe0L;
}

// 1 when the streaming kernels take this problem (factors 2x2x2, 32 / 64 channels, >= 32 K voxels)
extern "C" int adell_convt_k2_applicable(int N, int D, int H, int W, int Cin, int Cout) {
  return adell_convt_k2_ok(N, D, H, W, Cin, Cout) ? 1 : 0;
}
// ... factors (2, 2, 1): depth and height doubled, width kept (SWIN-UNet's anisotropic upscaling)
extern "C" int adell_convt_k221_applicable(int N, int D, int H, int W, int Cin, int Cout) {
  return adell_convt_k2_ok(N, D, H, W, Cin, Cout, 1) ? 1 : 0;
}

static void adell_ctk2_fill(ConvTK2Args* a, int N, int D, int H, int W, int Cin, int Cout, int FX = 2) {
  a->N = N; a->D = D; a->H = H; a->W = W; a->Cin = Cin; a->Cout = Cout;
  a->FX = FX;
  a->V = (long)N * D * H * W;
  a->ntiles = (int)((a->V + 31) / 32);
}
static int adell_ctk2_blocks(const ConvTK2Args& a, int per_cu) {
  const int want = 256 * per_cu;
  return a.ntiles < want ? a.ntiles : want;
}

static int adell_convt_k2_fwd_impl(int N, int D, int H, int W, int Cin, int Cout, int FX, const float* x,
                                   const float* w, const float* bias, float* y, void* stream) {
  ADELL_REQUIRE(x && w && y && adell_convt_k2_ok(N, D, H, W, Cin, Cout, FX),
                "convt_k2_fwd: factor-2 transposed conv with 32 / 64 -> 16 / 32 / 64 channels expected");
  ADELL_REQUIRE(((uintptr_t)x & 15) == 0, "convt_k2_fwd: x must be 16-byte aligned");
  ConvTK2Args a = {};
  adell_ctk2_fill(&a, N, D, H, W, Cin, Cout, FX);
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  dim3 grid((unsigned)adell_ctk2_blocks(a, 2), (unsigned)(Cout == 16 ? 1 : Cout / 32));
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 16 && Cin == 32)
    hipLaunchKernelGGL(adell_convt_k2_fwd16_kernel<32>, grid, dim3(256), 0, st, a);
  else if (Cout == 16)
    hipLaunchKernelGGL(adell_convt_k2_fwd16_kernel<64>, grid, dim3(256), 0, st, a);
  else if (Cin == 32 && FX == 2)
    hipLaunchKernelGGL((adell_convt_k2_fwd_kernel<32, 2>), grid, dim3(256), 0, st, a);
  else if (FX == 2)
    hipLaunchKernelGGL((adell_convt_k2_fwd_kernel<64, 2>), grid, dim3(256), 0, st, a);
  else if (Cin == 32)
    hipLaunchKernelGGL((adell_convt_k2_fwd_kernel<32, 1>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((adell_convt_k2_fwd_kernel<64, 1>), grid, dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_convt_k2_fwd(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                                  const float* w, const float* bias, float* y, void* stream) {
  return adell_convt_k2_fwd_impl(N, D, H, W, Cin, Cout, 2, x, w, bias, y, stream);
}
extern "C" int adell_convt_k221_fwd(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                                    const float* w, const float* bias, float* y, void* stream) {
  return adell_convt_k2_fwd_impl(N, D, H, W, Cin, Cout, 1, x, w, bias, y, stream);
}

template <int COUT, int FX>
static int adell_ctk2_launch_dx(const ConvTK2Args& a, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  auto kern = adell_convt_k2_dx_kernel<COUT, FX>;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_convt_k2_bwd_data_impl(int N, int D, int H, int W, int Cin, int Cout, int FX,
                                        const float* dy, const float* w, float* dx, void* stream) {
  ADELL_REQUIRE(dy && w && dx && adell_convt_k2_ok(N, D, H, W, Cin, Cout, FX),
                "convt_k2_bwd_data: factor-2 transposed conv with 32 / 64 channels expected");
  ADELL_REQUIRE(((uintptr_t)dy & 15) == 0, "convt_k2_bwd_data: dy must be 16-byte aligned");
  ConvTK2Args a = {};
  adell_ctk2_fill(&a, N, D, H, W, Cin, Cout, FX);
  a.dy = dy; a.w = w; a.dx = dx;
  const size_t lds = (size_t)(4 * 32 * (FX * Cout + 1) + 3 * 32 * 33) * sizeof(float);
  dim3 grid((unsigned)adell_ctk2_blocks(a, 2), (unsigned)(Cin / 32));
  hipStream_t st = (hipStream_t)stream;
  if (FX == 2) {
    if (Cout == 16) return adell_ctk2_launch_dx<16, 2>(a, grid, lds, st);
    if (Cout == 32) return adell_ctk2_launch_dx<32, 2>(a, grid, lds, st);
    return adell_ctk2_launch_dx<64, 2>(a, grid, lds, st);
  }
  if (Cout == 32) return adell_ctk2_launch_dx<32, 1>(a, grid, lds, st);
  return adell_ctk2_launch_dx<64, 1>(a, grid, lds, st);
}

extern "C" int adell_convt_k2_bwd_data(int N, int D, int H, int W, int Cin, int Cout,
                                       const float* dy, const float* w, float* dx, void* stream) {
  return adell_convt_k2_bwd_data_impl(N, D, H, W, Cin, Cout, 2, dy, w, dx, stream);
}
extern "C" int adell_convt_k221_bwd_data(int N, int D, int H, int W, int Cin, int Cout,
                                         const float* dy, const float* w, float* dx, void* stream) {
  return adell_convt_k2_bwd_data_impl(N, D, H, W, Cin, Cout, 1, dy, w, dx, stream);
}

static long adell_convt_k2_wgrad_ws_impl(int N, int D, int H, int W, int Cin, int Cout, int FX) {
  if (!adell_convt_k2_ok(N, D, H, W, Cin, Cout, FX)) return ADELL_E_BADARG;
  ConvTK2Args a = {};
  adell_ctk2_fill(&a, N, D, H, W, Cin, Cout, FX);
  const long blocks = adell_ctk2_blocks(a, 2);
  if (Cout == 16) return (long)sizeof(float) * (blocks * (Cin / 32) * 4 * 1024 + blocks * 4 * 16);
  return (long)sizeof(float) *
         (blocks * (Cin / 32) * (Cout / 32) * (4 * FX) * 1024 + blocks * 4 * Cout);
}
extern "C" long adell_convt_k2_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  return adell_convt_k2_wgrad_ws_impl(N, D, H, W, Cin, Cout, 2);
}
extern "C" long adell_convt_k221_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout) {
  return adell_convt_k2_wgrad_ws_impl(N, D, H, W, Cin, Cout, 1);
}

// db (optional): the bias gradient sum_v dy[v][co], a by-product of the same pass over dy.
static int adell_convt_k2_bwd_weight_impl(int N, int D, int H, int W, int Cin, int Cout, int FX,
                                          const float* x, const float* dy, float* dw, float* db,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(x && dy && dw && workspace && adell_convt_k2_ok(N, D, H, W, Cin, Cout, FX),
                "convt_k2_bwd_weight: factor-2 transposed conv with 32 / 64 channels expected");
  ADELL_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 15) == 0,
                "convt_k2_bwd_weight: x and dy must be 16-byte aligned");
  ADELL_REQUIRE((long)workspace_bytes >= adell_convt_k2_wgrad_ws_impl(N, D, H, W, Cin, Cout, FX),
                "convt_k2_bwd_weight: workspace too small");
  ConvTK2Args a = {};
  adell_ctk2_fill(&a, N, D, H, W, Cin, Cout, FX);
  a.x = x; a.dy = dy; a.ws = (float*)workspace;
  const int blocks = adell_ctk2_blocks(a, 2);
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 16) {
    a.wsdb = db ? a.ws + (size_t)blocks * (Cin / 32) * 4 * 1024 : nullptr;
    hipLaunchKernelGGL(adell_convt_k2_dw16_kernel, dim3((unsigned)blocks, (unsigned)(Cin / 32)), dim3(256),
                       0, st, a);
    const long outs16 = (long)Cin * 16 * 8 + (db ? 16 : 0);
    hipLaunchKernelGGL(adell_convt_k2_dw16_reduce_kernel, dim3((unsigned)((outs16 + 3) / 4)), dim3(256),
                       0, st, (const float*)workspace, blocks, Cin, dw, (const float*)a.wsdb, db);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  const int F = 4 * FX;
  a.wsdb = db ? a.ws + (size_t)blocks * (Cin / 32) * (Cout / 32) * F * 1024 : nullptr;
  dim3 grid((unsigned)blocks, (unsigned)((Cin / 32) * (Cout / 32)));
  if (FX == 2)
    hipLaunchKernelGGL(adell_convt_k2_dw_kernel<2>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(adell_convt_k2_dw_kernel<1>, grid, dim3(256), 0, st, a);
  const long outs = (long)Cin * Cout * F + (db ? Cout : 0);
  hipLaunchKernelGGL(adell_convt_k2_dw_reduce_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0,
                     st, (const float*)workspace, blocks, Cin, Cout, dw, (const float*)a.wsdb, db, F);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_convt_k2_bwd_weight(int N, int D, int H, int W, int Cin, int Cout,
                                         const float* x, const float* dy, float* dw, float* db,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return adell_convt_k2_bwd_weight_impl(N, D, H, W, Cin, Cout, 2, x, dy, dw, db, workspace,
                                        workspace_bytes, stream);
}
extern "C" int adell_convt_k221_bwd_weight(int N, int D, int H, int W, int Cin, int Cout,
                                           const float* x, const float* dy, float* dw, float* db,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  return adell_convt_k2_bwd_weight_impl(N, D, H, W, Cin, Cout, 1, x, dy, dw, db, workspace,
                                        workspace_bytes, stream);
}
