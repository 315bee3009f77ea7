// 3x3x3 stride-1 convolutions whose input has 1..4 channels and whose output is wide (the
// 2 -> 32 conv of the U-Net input block, unet.py:260-273; the 1 -> 16 conv of UNETR's first
// encoder, unetr.py:225-237), as ONE small GEMM per brick on the fp32 MFMA:
//
//   forward          y[v][co]      = sum_k  A[v][k] W[k][co],   k = (tap, ci), K = 27 Cin <= 108
//   weight gradient  dW[co][k]     = sum_v  dy[v][co] A[v][k]
//
// where A is the im2col row of voxel v. An MFMA tile of the general kernels pads Cin to a
// 16-channel chunk PER TAP (8x the MFMA work and staging for Cin = 2; the x taps folded into one
// chunk still leave it 2.7x), and the vector-ALU kernels issue 27 Cin Cout FMAs per voxel. Here
// K = 27 Cin is the whole contraction: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation) takes two k per step, the A values are gathered from a 10x10x6 input halo in LDS
// (4.8 KB for Cin = 2) with one ds_read_b32 per MFMA, and the other operand lives in registers
// (forward: the 27 Cin / 2 weight values of this lane's output channel; weight gradient: dY rows
// read straight from global memory, 128 contiguous bytes per half-wave). Both directions are then
// bound by the one pass over the wide tensor (y or dY).
#include <type_traits>
#include "common.h"

struct CinFoldArgs {
  const float* x;      // [N][D][H][W][Cin]
  const float* w;      // canonical [Cout][Cin][27]
  const float* bias;   // [Cout] or null
  const float* dy;     // [N][Do][Ho][Wo][Cout] (weight gradient)
  float* y;            // [N][Do][Ho][Wo][Cout]
  float* part;         // forward: [N][ntiles][Cout][2] statistics partials or null
  float* ws;           // weight gradient: [blocks][CoutPad][KP + 1] partial sums
  int N, D, H, W, Cout, Do, Ho, Wo, PD, PH, PW;
  int ntx, nty, ntz;
};

// The input halo of an 8x8x4 output brick, [6][10][10][CIN] floats, goes through registers:
// `fetch` issues this thread's global loads (clamped addresses, zeros outside the volume), `put`
// stores them to an LDS image later -- so the loads of the NEXT brick are in flight while the
// MFMAs of the current one run.
template <int CIN>
struct CinFoldHalo {
  static constexpr int PER = (600 * CIN + 255) / 256;
  float v[PER];
  __device__ __forceinline__ void fetch(const CinFoldArgs& a, int nb, int ox0, int oy0, int oz0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = threadIdx.x + 256 * u;
      const int ci = i % CIN, hv = i / CIN;
      const int hx = hv % 10, hy = (hv / 10) % 10, hz = hv / 100;
      const int x = ox0 - a.PW + hx, y = oy0 - a.PH + hy, z = oz0 - a.PD + hz;
      const bool ok = i < 600 * CIN && x >= 0 && x < a.W && y >= 0 && y < a.H && z >= 0 && z < a.D;
      const size_t off = ok ? ((((size_t)nb * a.D + z) * a.H + y) * a.W + x) * CIN + ci : 0;
      const float t = a.x[off];
      v[u] = ok ? t : 0.f;
    }
  }
  __device__ __forceinline__ void put(float* xh) const {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < 600 * CIN) xh[i] = v[u];
    }
  }
};

// LDS offset (floats) of im2col column k = tap * CIN + ci relative to the voxel's halo origin
template <int CIN>
__device__ __forceinline__ int adell_cinfold_koff(int k) {
  const int tap = k / CIN, ci = k - tap * CIN;
  const int kz = tap / 9, ky = (tap - 9 * kz) / 3, kx = tap - 9 * kz - 3 * ky;
  return ((kz * 10 + ky) * 10 + kx) * CIN + ci;
}

struct CinFoldBrick {
  int nb, tile, ox0, oy0, oz0;
};
__device__ __forceinline__ CinFoldBrick adell_cinfold_brick(const CinFoldArgs& a, int b, int nsp) {
  CinFoldBrick k;
  int t = b % nsp;
  k.nb = b / nsp;
  k.tile = t;
  const int tx = t % a.ntx;
  t /= a.ntx;
  k.ox0 = tx * 8;
  k.oy0 = (t % a.nty) * 8;
  k.oz0 = (t / a.nty) * 4;
  return k;
}

// grid (blocks, ceil(Cout / 32)), 256 threads. A block walks `per` consecutive bricks (over all
// batch items) with its 27 Cin / 2 weight values per lane loaded once; wave w owns the z = w slice
// of a brick (two 32-voxel M tiles), all waves the same 32 output channels. The halo image is
// double buffered.
template <int CIN>
__global__ __launch_bounds__(256) void adell_cinfold_fwd_kernel(CinFoldArgs a, int total_bricks,
                                                                int per) {
  constexpr int KT = 27 * CIN, KS = (KT + 1) / 2;
  __shared__ float xh[2][600 * CIN];
  __shared__ float red[2][4][32][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32, col = n0 + li;
  const bool colok = col < a.Cout;
  const int nsp = a.ntx * a.nty * a.ntz;
  const int b0 = blockIdx.x * per;
  const int b1 = (b0 + per) < total_bricks ? (b0 + per) : total_bricks;
  if (b0 >= b1) return;
  float bf[KS];
  int aoff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + lh;
    const bool ok = k < KT;
    aoff[s] = ok ? adell_cinfold_koff<CIN>(k) : 0;
    const int tap = ok ? k / CIN : 0, ci = ok ? k - tap * CIN : 0;
    const float t = a.w[((size_t)(colok ? col : 0) * CIN + ci) * 27 + tap];
    bf[s] = (ok && colok) ? t : 0.f;
  }
  int abase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) abase[j] = ((wave * 10 + 4 * j + (li >> 3)) * 10 + (li & 7)) * CIN;
  const float bcol = (a.bias && colok) ? a.bias[col] : 0.f;
  CinFoldHalo<CIN> halo;
  CinFoldBrick cur = adell_cinfold_brick(a, b0, nsp);
  halo.fetch(a, cur.nb, cur.ox0, cur.oy0, cur.oz0);
  halo.put(xh[0]);
  __syncthreads();
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    CinFoldBrick nxt = cur;
    if (b + 1 < b1) {
      nxt = adell_cinfold_brick(a, b + 1, nsp);
      halo.fetch(a, nxt.nb, nxt.ox0, nxt.oy0, nxt.oz0);   // in flight during the MFMAs below
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* xb = xh[buf];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[abase[j] + aoff[s]], bf[s], acc[j], 0, 0, 0);
    }
    float s1 = 0.f, s2 = 0.f;
    const int oz = cur.oz0 + wave;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // voxel of the M tile
        const int ox = cur.ox0 + (row & 7), oy = cur.oy0 + 4 * j + (row >> 3);
        if (colok && ox < a.Wo && oy < a.Ho && oz < a.Do) {
          const float v = acc[j][r] + bcol;
          a.y[((((size_t)cur.nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.Cout + col] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (a.part) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        red[buf][wave][li][0] = s1;
        red[buf][wave][li][1] = s2;
      }
    }
    if (b + 1 < b1) halo.put(xh[buf ^ 1]);
    __syncthreads();   // next halo image complete, this brick's statistics visible
    if (a.part && tid < 32 && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += red[buf][w][tid][0];
        t2 += red[buf][w][tid][1];
      }
      float* p = a.part + (((size_t)cur.nb * nsp + cur.tile) * a.Cout + n0 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// The forward for TWO input channels on the f16 MFMA (f16x3 splits, conv_igemm_f16.h): in the
// [z][y][x][2 ch] halo image the six im2col values of one (kz, ky) pair -- (kx, ci) = (0,0) (0,1)
// (1,0) (1,1) (2,0) (2,1) -- are three consecutive half2 words, so with the contraction ordered
// (kz, ky) major an 8-half MFMA fragment is ONE pair (+ two zero-weight slots): K = 9 pairs x 8 =
// 72, padded to 80 = five v_mfma_f32_32x32x16_f16 steps x 3 split products = 15 MFMAs of 32 cycles
// per 32 voxels where the fp32 MFMA form above issues 27 of 64. The image holds split halves
// (hi | lo planes of half2 words) scaled by a power of two from the brick's own absmax (one more
// barrier per brick); the weights sit in registers as split fragments with a per-column scale.
// ---------------------------------------------------------------------------
typedef _Float16 cf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 cf_half2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int cf_scale_exp(float mx) {
  const int ebits = (__float_as_int(mx) >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);   // max lands in [2^13, 2^14)
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

__global__ __launch_bounds__(256) void adell_cinfold2_fwd_f16_kernel(CinFoldArgs a, int total_bricks,
                                                                     int per) {
  constexpr int HV = 600, HP = 608;                 // halo voxels, padded plane (reads run 1 word past a row)
  __shared__ uint32_t xhi[2][HP], xlo[2][HP];       // half2 (ch 0, ch 1) per voxel
  __shared__ float red[2][4][32][2];
  __shared__ float smax[2][4];
  // per-wave transpose tile of the epilogue: the MFMA result holds ONE column per lane (a dword per
  // lane and row: 256 B per store instruction, in two pieces); through this tile a lane stores four
  // consecutive columns of a voxel, a wave 8 voxels x 128 B = 1 KB per instruction
  __shared__ __attribute__((aligned(16))) float tile[4][32][36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32, col = n0 + li;
  const bool colok = col < a.Cout;
  const int nsp = a.ntx * a.nty * a.ntz;
  const int b0 = blockIdx.x * per;
  const int b1 = (b0 + per) < total_bricks ? (b0 + per) : total_bricks;
  if (b0 >= b1) return;
  // ---- this lane's weights: column `col`, pairs p = 2 s + lh, split with the column's scale -----
  const float* wc = a.w + (size_t)(colok ? col : 0) * 54;      // [ci][27]
  // (this lane's 30 values, all loads in flight at once: a rolled loop of dependent round trips
  // took ~25 us per block; the two lane halves hold the even / odd pairs, i.e. all 54 between them)
  float wv[5][6];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int p = 2 * s + lh;                         // (kz, ky) pair; p = 9 is padding
#pragma unroll
    for (int j = 0; j < 6; ++j)                       // (kx = j >> 1, ci = j & 1)
      wv[s][j] = (p < 9 && colok) ? wc[(j & 1) * 27 + (p < 9 ? p : 0) * 3 + (j >> 1)] : 0.f;
  }
  float wmax = 0.f;
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int j = 0; j < 6; ++j) wmax = fmaxf(wmax, fabsf(wv[s][j]));
  wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
  const int kw = cf_scale_exp(wmax);
  const float wsc = __int_as_float((kw + 127) << 23);
  cf_half8 bh[5], bl[5];
  int aoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int p = 2 * s + lh;
    aoff[s] = p < 9 ? ((p / 3) * 10 + (p % 3)) * 10 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = j < 6 ? wv[s][j] * wsc : 0.f;
      const _Float16 h = (_Float16)v;
      bh[s][j] = h;
      bl[s][j] = (_Float16)(v - (float)h);
    }
  }
  int abase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) abase[j] = (wave * 10 + 4 * j + (li >> 3)) * 10 + (li & 7);
  const float bcol = (a.bias && colok) ? a.bias[col] : 0.f;
  const float wun = __int_as_float((127 - kw) << 23);

  float2 hv[3];                                       // this thread's halo voxels tid, tid + 256, tid + 512
  auto fetch = [&](const CinFoldBrick& k) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + 256 * u;
      const int hx = i % 10, hy = (i / 10) % 10, hz = i / 100;
      const int x = k.ox0 - a.PW + hx, y = k.oy0 - a.PH + hy, z = k.oz0 - a.PD + hz;
      const bool ok = i < HV && x >= 0 && x < a.W && y >= 0 && y < a.H && z >= 0 && z < a.D;
      const size_t off = ok ? ((((size_t)k.nb * a.D + z) * a.H + y) * a.W + x) * 2 : 0;
      const float2 t = *reinterpret_cast<const float2*>(a.x + off);
      hv[u] = ok ? t : make_float2(0.f, 0.f);
    }
  };
  // block-wide absmax of the fetched halo -> its power-of-two scale (all threads return the same)
  auto halo_scale = [&](int buf) -> int {
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) mx = fmaxf(mx, fmaxf(fabsf(hv[u].x), fabsf(hv[u].y)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) smax[buf][wave] = mx;
    __syncthreads();
    return cf_scale_exp(fmaxf(fmaxf(smax[buf][0], smax[buf][1]), fmaxf(smax[buf][2], smax[buf][3])));
  };
  auto put = [&](int buf, int kx) {
    const float sc = __int_as_float((kx + 127) << 23);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + 256 * u;
      if (i < HV) {
        const float t0 = hv[u].x * sc, t1 = hv[u].y * sc;
        cf_half2 h, l;
        h[0] = (_Float16)t0; h[1] = (_Float16)t1;
        l[0] = (_Float16)(t0 - (float)h[0]); l[1] = (_Float16)(t1 - (float)h[1]);
        xhi[buf][i] = *reinterpret_cast<uint32_t*>(&h);
        xlo[buf][i] = *reinterpret_cast<uint32_t*>(&l);
      }
    }
  };
  if (tid < HP - HV) {   // the padding words are read (against zero weights): keep them finite
    xhi[0][HV + tid] = xhi[1][HV + tid] = xlo[0][HV + tid] = xlo[1][HV + tid] = 0u;
  }
  CinFoldBrick cur = adell_cinfold_brick(a, b0, nsp);
  fetch(cur);
  int kcur = halo_scale(0);
  put(0, kcur);
  __syncthreads();
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    CinFoldBrick nxt = cur;
    const bool more = b + 1 < b1;
    if (more) {
      nxt = adell_cinfold_brick(a, b + 1, nsp);
      fetch(nxt);                                       // in flight during the MFMAs below
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const uint32_t* ph = xhi[buf];
    const uint32_t* pl = xlo[buf];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = abase[j] + aoff[s];
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wh[q] = ph[o + q];
          wl[q] = pl[o + q];
        }
        const cf_half8 ah = *reinterpret_cast<const cf_half8*>(wh);
        const cf_half8 al = *reinterpret_cast<const cf_half8*>(wl);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc[j], 0, 0, 0);
      }
    }
    const float oscale = __int_as_float((127 - kcur) << 23) * wun;
    float s1 = 0.f, s2 = 0.f;
    const int oz = cur.oz0 + wave;
    // whole 8 x 8 slice inside the volume and a whole, 16-byte aligned column tile: wide stores
    const bool wide = cur.ox0 + 8 <= a.Wo && cur.oy0 + 8 <= a.Ho && oz < a.Do &&
                      n0 + 32 <= a.Cout && (a.Cout & 3) == 0;
    if (wide) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float v = acc[j][r] * oscale + bcol;
          tile[wave][row][li] = v;
          s1 += v;
          s2 += v * v;
        }
        // rows 8 i .. 8 i + 7 are the eight x-neighbours of brick row y = 4 j + i
        float* ybase = a.y + ((((size_t)cur.nb * a.Do + oz) * a.Ho + cur.oy0 + 4 * j) * a.Wo + cur.ox0) *
                                 a.Cout + n0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 q = *reinterpret_cast<const float4*>(&tile[wave][8 * i + (lane >> 3)][4 * (lane & 7)]);
          *reinterpret_cast<float4*>(ybase + ((size_t)i * a.Wo + (lane >> 3)) * a.Cout + 4 * (lane & 7)) = q;
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // voxel of the M tile
        const int ox = cur.ox0 + (row & 7), oy = cur.oy0 + 4 * j + (row >> 3);
        if (colok && ox < a.Wo && oy < a.Ho && oz < a.Do) {
          const float v = acc[j][r] * oscale + bcol;
          a.y[((((size_t)cur.nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.Cout + col] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    }
    if (a.part) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        red[buf][wave][li][0] = s1;
        red[buf][wave][li][1] = s2;
      }
    }
    int knext = kcur;
    if (more) {           // (block-uniform)
      knext = halo_scale(buf ^ 1);
      put(buf ^ 1, knext);
    }
    __syncthreads();   // next halo image complete, this brick's statistics visible
    if (a.part && tid < 32 && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += red[buf][w][tid][0];
        t2 += red[buf][w][tid][1];
      }
      float* p = a.part + (((size_t)cur.nb * nsp + cur.tile) * a.Cout + n0 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
    cur = nxt;
    kcur = knext;
  }
}

// Weight gradient. grid (blocks, ceil(Cout / 32)); a block walks bricks b = blockIdx.x,
// blockIdx.x + gridDim.x, ... over all batch items and keeps dW[32 co][KP] in accumulators;
// wave w takes the z = w slice of each brick (32 K steps of two voxels). The input halo image is
// double buffered and the dY values of the next brick are fetched before the MFMAs of the
// current one. Partial sums (and the bias gradient in column KP) go to ws[block][co][KP + 1];
// adell_cinfold_wgrad_reduce folds them in block order.
template <int CIN>
__global__ __launch_bounds__(256, 2) void adell_cinfold_wgrad_kernel(CinFoldArgs a, int total_bricks) {
  constexpr int KT = 27 * CIN, NTL = (KT + 31) / 32, KP = NTL * 32;
  __shared__ float xh[2][600 * CIN];
  __shared__ float red[32][KP + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32, col = n0 + li;
  const bool colok = col < a.Cout;
  const int nsp = a.ntx * a.nty * a.ntz;
  int boff[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int k = nt * 32 + li;
    boff[nt] = k < KT ? adell_cinfold_koff<CIN>(k) : -1;
  }
  f32x16 acc[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  float sdy = 0.f;
  // this lane's dY values of a brick: voxel 2 s + lh of the wave's 8x8 slice, output channel col
  auto fetch_dy = [&](const CinFoldBrick& k, float* av) {
    const int oz = k.oz0 + wave;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int v = 2 * s + lh, ox = k.ox0 + (v & 7), oy = k.oy0 + (v >> 3);
      const bool ok = colok && ox < a.Wo && oy < a.Ho && oz < a.Do;
      const size_t off =
          ok ? ((((size_t)k.nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.Cout + col : 0;
      const float t = a.dy[off];
      av[s] = ok ? t : 0.f;
    }
  };
  CinFoldHalo<CIN> halo;
  float av[2][32];
  if (blockIdx.x < total_bricks) {
    const CinFoldBrick k0 = adell_cinfold_brick(a, blockIdx.x, nsp);
    halo.fetch(a, k0.nb, k0.ox0, k0.oy0, k0.oz0);
    fetch_dy(k0, av[0]);
    halo.put(xh[0]);
  }
  __syncthreads();
  int it = 0;
  for (int b = blockIdx.x; b < total_bricks; b += gridDim.x, ++it) {
    const int buf = it & 1;
    const bool more = b + (int)gridDim.x < total_bricks;
    if (more) {
      const CinFoldBrick kn = adell_cinfold_brick(a, b + gridDim.x, nsp);
      halo.fetch(a, kn.nb, kn.ox0, kn.oy0, kn.oz0);
    }
    // (two static copies of the loop body so that av[buf] stays in registers)
    auto body = [&](const float* cur, float* nxt) {
      if (more) fetch_dy(adell_cinfold_brick(a, b + gridDim.x, nsp), nxt);
      const float* xb = xh[buf];
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const int v = 2 * s + lh;
        const int vbase = ((wave * 10 + (v >> 3)) * 10 + (v & 7)) * CIN;
        sdy += cur[s];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          const float bv = boff[nt] >= 0 ? xb[vbase + boff[nt]] : 0.f;
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], bv, acc[nt], 0, 0, 0);
        }
      }
    };
    if (buf == 0)
      body(av[0], av[1]);
    else
      body(av[1], av[0]);
    if (more) halo.put(xh[buf ^ 1]);
    __syncthreads();
  }
  // fold the four waves in wave order (fixed order), then one partial row set per block
  sdy += __shfl_xor(sdy, 32, 64);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;      // output channel of the tile
          float* q = &red[row][nt * 32 + li];
          *q = w == 0 ? acc[nt][r] : *q + acc[nt][r];
        }
      if (lh == 0) red[li][KP] = w == 0 ? sdy : red[li][KP] + sdy;
    }
    __syncthreads();
  }
  float* out = a.ws + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 32 * (KP + 1);
  for (int i = tid; i < 32 * (KP + 1); i += 256) out[i] = (&red[0][0])[i];
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient on the f16 MFMA with the error-compensated split (f16x3): the fp32-MFMA
// kernel above is bound by its 2 x 32 MFMAs of 64 cycles per 64 voxels (0.36 ms at 2 x 128^3, 2 -> 32,
// where dY moves in 0.13 ms). GEMM view: M = co (32), N = im2col columns k = (tap, ci), K = voxels.
//   A (dY): the wave's 64-voxel z slice, split to fp16 hi / lo planes [voxel][32 co] in wave-private
//           LDS, read transposed (ds_read_b64_tr_b16) like the dY operand of conv_wgrad_zring.hip;
//           one power-of-two scale per slice (its absmax), the accumulators rescaled by the exact
//           ratio when it changes.
//   B (x):  the brick's 6 x 10 x 10 halo split once (scale from the tensor's absmax, a pre-pass over
//           the narrow input) into THREE x-shifted copies [kx][ci][z][y][8 x] so that the eight
//           consecutive voxels of a k-block are one aligned 16-byte read for every tap.
// 12 / 24 MFMAs of 32 cycles per wave and brick instead of 64 of 64: HBM-bound.
typedef _Float16 cf_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 cf_half4 __attribute__((ext_vector_type(4)));
typedef __fp16 cf_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ cf_half8 adell_cf_trfrag(const char* p) {
  typedef __attribute__((address_space(3))) cf_fp16x4* lds_p;
  const cf_fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p));
  const cf_fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p + 4 * 64));
  cf_half8 r;
  r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
  r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
  return r;
}

__device__ __forceinline__ int adell_cf_exp(float mx) {
  const int ebits = (__float_as_int(mx) >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 8 * ((13 - (ebits - 127)) >> 3);
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

template <int CIN>
__global__ __launch_bounds__(256, 2) void adell_cinfold_wgrad_f16_kernel(CinFoldArgs a, int total_bricks,
                                                                         const unsigned* xmax) {
  constexpr int KT = 27 * CIN, NTL = (KT + 31) / 32, KP = NTL * 32;
  constexpr int XCOPY = CIN * 6 * 10 * 8;                 // halfs of one shifted copy
  constexpr int XIMG = 3 * XCOPY * 2 + 16;                // bytes-in-halfs of (hi | lo) + a zero row
  __shared__ __attribute__((aligned(16))) _Float16 xs[2][XIMG];   // [buf][hi 3 copies | lo 3 copies | 8 zeros]
  __shared__ __attribute__((aligned(16))) char dys[4][2][64 * 64];  // [wave][hi | lo][64 voxels][32 co]
  __shared__ float red[32][KP + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const int nsp = a.ntx * a.nty * a.ntz;
  const int kX = adell_cf_exp(__uint_as_float(xmax[0]));
  const float sX = __int_as_float((kX + 127) << 23);
  // B fragment addresses (halfs) of this lane's im2col column in every N tile: brick row 0 of slice z
  int boff[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int k = nt * 32 + li;
    if (k < KT) {
      const int tap = k / CIN, ci = k - tap * CIN;
      const int kz = tap / 9, ky = (tap - 9 * kz) / 3, kx = tap - 9 * kz - 3 * ky;
      boff[nt] = (((kx * CIN + ci) * 6 + (wave + kz)) * 10 + ky) * 8;
    } else {
      boff[nt] = -1;
    }
  }
  // A fragment (transposed read of the wave's dY planes): lane roles of conv_wgrad_zring.hip
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const int abase = (lh * 8 + tq) * 64 + (16 * cg + 4 * tp) * 2;
  f32x16 acc[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  int kprev = 0;
  bool first = true;
  // dY slice of a brick: lane = (row lane >> 3 + 8 u, channel quad lane & 7). TWO bricks of loads in
  // flight (register sets 0 / 1): with one, an iteration lasted a memory round trip under load
  // (~10 us per brick and block: 1.7 TB/s, the same as the fp32-MFMA kernel it replaces).
  float4 yr[2][8];
  // (voxel v = (lane >> 3) + 8 u of the 8 x 8 slice: x = lane >> 3, y = u -- one 64-bit base per
  // brick and a uniform row stride instead of eight per-lane offsets)
  const int vx = lane >> 3, cq = n0 + 4 * (lane & 7);
  const bool cok = cq < a.Cout;
  const bool yvec = (a.Cout & 3) == 0;
  auto fetch_dy = [&](const CinFoldBrick& k, float4* y) {
    const int oz = k.oz0 + wave;
    const bool ok0 = cok && (k.ox0 + vx < a.Wo) && oz < a.Do;
    const float* base = a.dy + ((((size_t)k.nb * a.Do + (oz < a.Do ? oz : 0)) * a.Ho + k.oy0) * a.Wo +
                                (k.ox0 + vx < a.Wo ? k.ox0 + vx : 0)) * a.Cout + (cok ? cq : 0);
    const size_t rstride = (size_t)a.Wo * a.Cout;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = ok0 && (k.oy0 + u < a.Ho);
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const float* p = base + u * rstride;
        if (yvec) {
          f = *reinterpret_cast<const float4*>(p);
        } else {
          f.x = p[0];
          if (cq + 1 < a.Cout) f.y = p[1];
          if (cq + 2 < a.Cout) f.z = p[2];
          if (cq + 3 < a.Cout) f.w = p[3];
        }
      }
      y[u] = f;
    }
  };
  // registers -> the wave's hi / lo planes; returns the slice's exponent
  auto put_dy = [&](const float4* y) -> int {
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(y[u].x), fabsf(y[u].y)), fmaxf(fabsf(y[u].z), fabsf(y[u].w))));
      dbacc.x += y[u].x; dbacc.y += y[u].y; dbacc.z += y[u].z; dbacc.w += y[u].w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const int kY = adell_cf_exp(mx);
    const float sY = __int_as_float((kY + 127) << 23);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = (lane >> 3) + 8 * u;
      const float t[4] = {y[u].x * sY, y[u].y * sY, y[u].z * sY, y[u].w * sY};
      cf_half4 h, l;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        h[jj] = (_Float16)t[jj];
        l[jj] = (_Float16)(t[jj] - (float)h[jj]);
      }
      *reinterpret_cast<cf_half4*>(dys[wave][0] + v * 64 + (lane & 7) * 8) = h;
      *reinterpret_cast<cf_half4*>(dys[wave][1] + v * 64 + (lane & 7) * 8) = l;
    }
    return kY;
  };
  // halo registers -> the three shifted hi / lo copies of buffer `buf`
  CinFoldHalo<CIN> halo[2];
  auto put_x = [&](int buf, const CinFoldHalo<CIN>& hreg) {
    _Float16* hi = xs[buf];
    _Float16* lo = xs[buf] + 3 * XCOPY;
#pragma unroll
    for (int u = 0; u < CinFoldHalo<CIN>::PER; ++u) {
      const int i2 = tid + 256 * u;
      if (i2 < 600 * CIN) {
        const int ci = i2 % CIN, hv = i2 / CIN;
        const int hx = hv % 10, hy = (hv / 10) % 10, hz = hv / 100;
        const float t = hreg.v[u] * sX;
        const _Float16 h = (_Float16)t, l = (_Float16)(t - (float)h);
#pragma unroll
        for (int sh = 0; sh < 3; ++sh) {
          const int xq = hx - sh;
          if (xq >= 0 && xq < 8) {
            const int o = (((sh * CIN + ci) * 6 + hz) * 10 + hy) * 8 + xq;
            hi[o] = h;
            lo[o] = l;
          }
        }
      }
    }
    if (tid < 8) xs[buf][6 * XCOPY + tid] = (_Float16)0.f;     // the zero row of columns >= KT
  };
  const int stride = (int)gridDim.x;
  if (blockIdx.x < total_bricks) {
    const CinFoldBrick k0 = adell_cinfold_brick(a, blockIdx.x, nsp);
    halo[0].fetch(a, k0.nb, k0.ox0, k0.oy0, k0.oz0);
    fetch_dy(k0, yr[0]);
    if ((int)blockIdx.x + stride < total_bricks) {
      const CinFoldBrick k1 = adell_cinfold_brick(a, blockIdx.x + stride, nsp);
      halo[1].fetch(a, k1.nb, k1.ox0, k1.oy0, k1.oz0);
      fetch_dy(k1, yr[1]);
    }
    put_x(0, halo[0]);
  }
  __syncthreads();
  // iteration `it` (register set S = it & 1): brick it out of LDS, brick it + 2 into set S, the
  // halo of brick it + 1 (set S ^ 1) into the other image
  auto iteration = [&](int b, int it, auto S) {
    constexpr int s0 = decltype(S)::value, s1 = s0 ^ 1;
    const int buf = it & 1;
    const int kY = put_dy(yr[s0]);
    const int ksum = kX + kY;
    if (!first && ksum != kprev) {
      const float f = __int_as_float((ksum - kprev + 127) << 23);
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] *= f;
    }
    kprev = ksum;
    first = false;
    if (b + 2 * stride < total_bricks) {
      const CinFoldBrick kn = adell_cinfold_brick(a, b + 2 * stride, nsp);
      halo[s0].fetch(a, kn.nb, kn.ox0, kn.oy0, kn.oz0);
      fetch_dy(kn, yr[s0]);
    }
    const _Float16* xh = xs[buf];
    const _Float16* xl = xs[buf] + 3 * XCOPY;
    const _Float16* zero = xs[buf] + 6 * XCOPY;
    // 4 k-steps of 16 voxels (two brick rows): lane half lh takes row 2 s + lh
    // (two at a time: all four in flight spilled next to the two prefetched bricks)
#pragma unroll 2
    for (int s4 = 0; s4 < 4; ++s4) {
      const cf_half8 ah = adell_cf_trfrag(dys[wave][0] + abase + s4 * 16 * 64);
      const cf_half8 al = adell_cf_trfrag(dys[wave][1] + abase + s4 * 16 * 64);
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        const int o = boff[nt] + (2 * s4 + lh) * 8;
        const cf_half8 bh = *reinterpret_cast<const cf_half8*>(boff[nt] >= 0 ? xh + o : zero);
        const cf_half8 bl = *reinterpret_cast<const cf_half8*>(boff[nt] >= 0 ? xl + o : zero);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
    if (b + stride < total_bricks) put_x(buf ^ 1, halo[s1]);
    __syncthreads();
  };
  {
    int it = 0;
    for (int b = blockIdx.x; b < total_bricks; b += 2 * stride, it += 2) {
      iteration(b, it, std::integral_constant<int, 0>{});
      if (b + stride < total_bricks) iteration(b + stride, it + 1, std::integral_constant<int, 1>{});
    }
  }
  // undo the scales, fold the four waves in wave order, one partial row set per block
  const float unscale = __int_as_float((127 - kprev) << 23);
  // db: lanes with the same lane & 7 hold the same channel quad
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    dbacc.x += __shfl_xor(dbacc.x, o, 64);
    dbacc.y += __shfl_xor(dbacc.y, o, 64);
    dbacc.z += __shfl_xor(dbacc.z, o, 64);
    dbacc.w += __shfl_xor(dbacc.w, o, 64);
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;      // output channel of the tile
          float* q = &red[row][nt * 32 + li];
          const float v = first ? 0.f : acc[nt][r] * unscale;
          *q = w == 0 ? v : *q + v;
        }
      if (lane < 8) {
        float* q = &red[4 * lane][KP];
        const float d4[4] = {dbacc.x, dbacc.y, dbacc.z, dbacc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j * (KP + 1)] = w == 0 ? d4[j] : q[j * (KP + 1)] + d4[j];
      }
    }
    __syncthreads();
  }
  float* out = a.ws + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 32 * (KP + 1);
  for (int i = tid; i < 32 * (KP + 1); i += 256) out[i] = (&red[0][0])[i];
}

__global__ __launch_bounds__(256) void adell_cinfold_absmax_kernel(const float* __restrict__ x, long n,
                                                                   unsigned* __restrict__ out) {
  // 16-byte loads, four in flight per thread (the scalar grid-stride form took 100 us for 33 MB)
  float mx = 0.f;
  const long n4 = ((((uintptr_t)x) & 15) == 0) ? (n >> 2) : 0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const long stride = (long)gridDim.x * 256L;
  long i = blockIdx.x * 256L + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                         fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                         fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
  }
  for (; i < n4; i += stride) {
    const f32x4 a = x4[i];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
  }
  for (long j = 4 * n4 + blockIdx.x * 256L + threadIdx.x; j < n; j += stride) mx = fmaxf(mx, fabsf(x[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  // ONE atomic per block: with one per wave, 8192 waves queued on the same word and the pass over a
  // 33 MB tensor took 98 us (an atomic to one address retires every ~12 ns)
  __shared__ float smx[4];
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]))));
}

// dw[co][ci][tap] (and db[co]) = sum over the blocks' partials: one wave per output value, lane l
// adds blocks l, l + 64, ... and the wave folds its 64 sums with the fixed xor-shuffle tree (a
// thread per value walking 1024 strided partials in turn took longer than the MFMA kernel).
__global__ __launch_bounds__(256) void adell_cinfold_wgrad_reduce_kernel(
    const float* __restrict__ ws, int blocks, int ntile, int KP, int Cin, int Cout,
    float* __restrict__ dw, float* __restrict__ db) {
  const int KT = 27 * Cin;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= Cout * (KT + 1)) return;   // whole wave
  const int co = i / (KT + 1), k = i - co * (KT + 1);
  const bool is_db = k == KT;
  if (is_db && !db) return;
  const int kk = is_db ? KP : k;
  const float* p = ws + ((size_t)(co >> 5) * 32 + (co & 31)) * (KP + 1) + kk;
  const size_t stride = (size_t)ntile * 32 * (KP + 1);
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += p[(size_t)b * stride];
  s = adell_wave_sum(s);
  if (lane == 0) {
    if (is_db) {
      db[co] = s;
    } else {
      const int tap = k / Cin, ci = k - tap * Cin;
      dw[((size_t)co * Cin + ci) * 27 + tap] = s;
    }
  }
}

static bool adell_cinfold_ok(const adell_conv3d_desc* d) {
  return d && d->C1 == 0 && d->C0 >= 1 && d->C0 <= 4 && d->KD == 3 && d->KH == 3 && d->KW == 3 &&
         d->SD == 1 && d->SH == 1 && d->SW == 1 && d->PD <= 1 && d->PH <= 1 && d->PW <= 1 &&
         d->PD >= 0 && d->PH >= 0 && d->PW >= 0 && d->Cout >= 1 && d->N >= 1 && d->N <= 65535 &&
         d->Do == d->D + 2 * d->PD - 2 && d->Ho == d->H + 2 * d->PH - 2 &&
         d->Wo == d->W + 2 * d->PW - 2 && d->Do > 0 && d->Ho > 0 && d->Wo > 0;
}

extern "C" int adell_conv_cinfold_applicable(const adell_conv3d_desc* d) {
  return adell_cinfold_ok(d) ? 1 : 0;
}

static void adell_cinfold_fill(CinFoldArgs* a, const adell_conv3d_desc* d) {
  a->N = d->N; a->D = d->D; a->H = d->H; a->W = d->W; a->Cout = d->Cout;
  a->Do = d->Do; a->Ho = d->Ho; a->Wo = d->Wo; a->PD = d->PD; a->PH = d->PH; a->PW = d->PW;
  a->ntx = adell_cdiv(d->Wo, 8);
  a->nty = adell_cdiv(d->Ho, 8);
  a->ntz = adell_cdiv(d->Do, 4);
}

extern "C" int adell_conv_cinfold_ntiles(const adell_conv3d_desc* d) {
  if (!adell_cinfold_ok(d)) return ADELL_E_BADARG;
  return adell_cdiv(d->Wo, 8) * adell_cdiv(d->Ho, 8) * adell_cdiv(d->Do, 4);
}

static int adell_cinfold_fwd_impl(const adell_conv3d_desc* d, const float* x, const float* w,
                                  const float* bias, float* y, float* stat_partials,
                                  int partial_rows, int f16x3, void* stream);

extern "C" int adell_conv_cinfold_fwd(const adell_conv3d_desc* d, const float* x, const float* w,
                                      const float* bias, float* y, float* stat_partials,
                                      int partial_rows, void* stream) {
  return adell_cinfold_fwd_impl(d, x, w, bias, y, stat_partials, partial_rows, 0, stream);
}

// The same on the f16 MFMA with error-compensated splits (~2^-22 per product, the precision of
// the other f16x3 conv entry points) for two input channels; other channel counts run the exact
// fp32-MFMA kernel.
extern "C" int adell_conv_cinfold_fwd_f16x3(const adell_conv3d_desc* d, const float* x,
                                            const float* w, const float* bias, float* y,
                                            float* stat_partials, int partial_rows,
                                            void* stream) {
  return adell_cinfold_fwd_impl(d, x, w, bias, y, stat_partials, partial_rows, 1, stream);
}

static int adell_cinfold_fwd_impl(const adell_conv3d_desc* d, const float* x, const float* w,
                                  const float* bias, float* y, float* stat_partials,
                                  int partial_rows, int f16x3, void* stream) {
  ADELL_REQUIRE(x && w && y && adell_cinfold_ok(d),
                "conv_cinfold_fwd: 3x3x3 stride-1 conv with 1..4 input channels expected");
  CinFoldArgs a = {};
  adell_cinfold_fill(&a, d);
  ADELL_REQUIRE_ROWS(stat_partials, partial_rows, (long)a.ntx * a.nty * a.ntz, "conv_cinfold_fwd");
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.part = stat_partials;
  const long total = (long)d->N * a.ntx * a.nty * a.ntz;
  ADELL_REQUIRE(total < 0x7fffffffL, "conv_cinfold_fwd: too many bricks");
  // consecutive bricks per block: enough blocks to fill the chip several times over, few enough
  // that the per-block weight loads and the first (exposed) halo fetch are amortised
  int per = (int)(total / 4096);
  if (per < 1) per = 1;
  if (per > 16) per = 16;
  dim3 grid((unsigned)((total + per - 1) / per), (unsigned)adell_cdiv(d->Cout, 32));
  hipStream_t st = (hipStream_t)stream;
  if (f16x3 && d->C0 == 2 && (((uintptr_t)x & 7) == 0)) {
    hipLaunchKernelGGL(adell_cinfold2_fwd_f16_kernel, grid, dim3(256), 0, st, a, (int)total, per);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  switch (d->C0) {
    case 1: hipLaunchKernelGGL(adell_cinfold_fwd_kernel<1>, grid, dim3(256), 0, st, a, (int)total, per); break;
    case 2: hipLaunchKernelGGL(adell_cinfold_fwd_kernel<2>, grid, dim3(256), 0, st, a, (int)total, per); break;
    case 3: hipLaunchKernelGGL(adell_cinfold_fwd_kernel<3>, grid, dim3(256), 0, st, a, (int)total, per); break;
    default: hipLaunchKernelGGL(adell_cinfold_fwd_kernel<4>, grid, dim3(256), 0, st, a, (int)total, per); break;
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_cinfold_wgrad_blocks(const adell_conv3d_desc* d) {
  const long total = (long)d->N * adell_cdiv(d->Wo, 8) * adell_cdiv(d->Ho, 8) * adell_cdiv(d->Do, 4);
  return (int)(total < 512 ? total : 512);   // two resident blocks per CU: one round
}

extern "C" long adell_conv_cinfold_wgrad_workspace(const adell_conv3d_desc* d) {
  if (!adell_cinfold_ok(d)) return ADELL_E_BADARG;
  const int KP = adell_cdiv(27 * d->C0, 32) * 32;
  return (long)sizeof(float) * ((long)adell_cinfold_wgrad_blocks(d) * adell_cdiv(d->Cout, 32) * 32 * (KP + 1) + 4);
}

extern "C" int adell_conv_cinfold_bwd_weight(const adell_conv3d_desc* d, const float* x,
                                             const float* dy, float* dw, float* db,
                                             void* workspace, size_t workspace_bytes,
                                             void* stream) {
  ADELL_REQUIRE(x && dy && dw && workspace && adell_cinfold_ok(d),
                "conv_cinfold_bwd_weight: 3x3x3 stride-1 conv with 1..4 input channels expected");
  ADELL_REQUIRE((long)workspace_bytes >= adell_conv_cinfold_wgrad_workspace(d),
                "conv_cinfold_bwd_weight: workspace too small");
  CinFoldArgs a = {};
  adell_cinfold_fill(&a, d);
  a.x = x; a.dy = dy; a.ws = (float*)workspace;
  const long total = (long)d->N * a.ntx * a.nty * a.ntz;
  ADELL_REQUIRE(total < 0x7fffffffL, "conv_cinfold_bwd_weight: too many bricks");
  const int blocks = adell_cinfold_wgrad_blocks(d), ntile = adell_cdiv(d->Cout, 32);
  dim3 grid((unsigned)blocks, (unsigned)ntile);
  hipStream_t st = (hipStream_t)stream;
  switch (d->C0) {
    case 1: hipLaunchKernelGGL(adell_cinfold_wgrad_kernel<1>, grid, dim3(256), 0, st, a, (int)total); break;
    case 2: hipLaunchKernelGGL(adell_cinfold_wgrad_kernel<2>, grid, dim3(256), 0, st, a, (int)total); break;
    case 3: hipLaunchKernelGGL(adell_cinfold_wgrad_kernel<3>, grid, dim3(256), 0, st, a, (int)total); break;
    default: hipLaunchKernelGGL(adell_cinfold_wgrad_kernel<4>, grid, dim3(256), 0, st, a, (int)total); break;
  }
  const int KP = adell_cdiv(27 * d->C0, 32) * 32;
  const int outs = d->Cout * (27 * d->C0 + 1);
  hipLaunchKernelGGL(adell_cinfold_wgrad_reduce_kernel, dim3(adell_cdiv(outs, 4)), dim3(256), 0, st,
                     (const float*)workspace, blocks, ntile, KP, d->C0, d->Cout, dw, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// The same on the f16 MFMA with the error-compensated split (adell_cinfold_wgrad_f16_kernel): same
// arguments, same workspace.
extern "C" int adell_conv_cinfold_bwd_weight_f16x3(const adell_conv3d_desc* d, const float* x,
                                                   const float* dy, float* dw, float* db,
                                                   void* workspace, size_t workspace_bytes,
                                                   void* stream) {
  ADELL_REQUIRE(x && dy && dw && workspace && adell_cinfold_ok(d),
                "conv_cinfold_bwd_weight_f16x3: 3x3x3 stride-1 conv with 1..4 input channels expected");
  ADELL_REQUIRE((long)workspace_bytes >= adell_conv_cinfold_wgrad_workspace(d),
                "conv_cinfold_bwd_weight_f16x3: workspace too small");
  CinFoldArgs a = {};
  adell_cinfold_fill(&a, d);
  a.x = x; a.dy = dy; a.ws = (float*)workspace;
  const long total = (long)d->N * a.ntx * a.nty * a.ntz;
  ADELL_REQUIRE(total < 0x7fffffffL, "conv_cinfold_bwd_weight_f16x3: too many bricks");
  const int blocks = adell_cinfold_wgrad_blocks(d), ntile = adell_cdiv(d->Cout, 32);
  const int KP = adell_cdiv(27 * d->C0, 32) * 32;
  unsigned* xmax = reinterpret_cast<unsigned*>(a.ws + (size_t)blocks * ntile * 32 * (KP + 1));
  hipStream_t st = (hipStream_t)stream;
  ADELL_CHECK_HIP(hipMemsetAsync(xmax, 0, 4 * sizeof(unsigned), st));
  const long nx = (long)d->N * d->D * d->H * d->W * d->C0;
  long ab = (nx / 4 + 1023) / 1024;
  if (ab > 512) ab = 512;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(adell_cinfold_absmax_kernel, dim3((unsigned)ab), dim3(256), 0, st, x, nx, xmax);
  dim3 grid((unsigned)blocks, (unsigned)ntile);
  switch (d->C0) {
    case 1: hipLaunchKernelGGL(adell_cinfold_wgrad_f16_kernel<1>, grid, dim3(256), 0, st, a, (int)total, xmax); break;
    case 2: hipLaunchKernelGGL(adell_cinfold_wgrad_f16_kernel<2>, grid, dim3(256), 0, st, a, (int)total, xmax); break;
    case 3: hipLaunchKernelGGL(adell_cinfold_wgrad_f16_kernel<3>, grid, dim3(256), 0, st, a, (int)total, xmax); break;
    default: hipLaunchKernelGGL(adell_cinfold_wgrad_f16_kernel<4>, grid, dim3(256), 0, st, a, (int)total, xmax); break;
  }
  const int outs = d->Cout * (27 * d->C0 + 1);
  hipLaunchKernelGGL(adell_cinfold_wgrad_reduce_kernel, dim3(adell_cdiv(outs, 4)), dim3(256), 0, st,
                     (const float*)workspace, blocks, ntile, KP, d->C0, d->Cout, dw, db);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// ---------------------------------------------------------------------------
// Backward-data of the same convs: dx has 1..4 channels, so an implicit-GEMM tile would carry
// 1..4 useful columns out of 32. Instead T[v][k] = sum_co dy[v][co] w[co][k], k = (tap, ci), is
// ONE GEMM per dY voxel (K = Cout, N = 27 Cin) and dx[u][ci] = sum_tap T[u + pad - tap][(tap, ci)]
// is a 27-term gather ("col2im"). A block owns an 8x8 column of dx and marches along z: per step
// it stages the 10x10 dY halo of ONE new plane (coalesced float4 loads issued one step ahead),
// runs the GEMM for its 100 voxels on the fp32 MFMA (weights in registers), parks T in LDS and adds
// the nine (ky, kx) terms of each kz to the three dx planes that plane feeds -- the running sums of
// those planes live in the registers of the 64 Cin gather threads; the plane whose last dY plane
// this was is stored. Cout <= 64.
// ---------------------------------------------------------------------------
struct CinFoldDxArgs {
  const float* dy;   // [N][Do][Ho][Wo][Cout]
  const float* w;    // canonical [Cout][Cin][27]
  float* dx;         // [N][D][H][W][Cin]
  int N, D, H, W, Cout, Do, Ho, Wo, PD, PH, PW;
  int ntx, nty, nseg, seglen;
};

template <int CIN, int CO>   // CO: Cout padded to 32 or 64
__global__ __launch_bounds__(256, 2) void adell_cinfold_dx_kernel(CinFoldDxArgs a) {
  constexpr int KT = 27 * CIN, NTL = (KT + 31) / 32, KP = NTL * 32, KS = CO / 2;
  constexpr int DYS = CO + 1;        // row stride of the dY plane image (conflict-free columns)
  constexpr int TS = KP + 1;         // row stride of a T plane
  extern __shared__ float smem[];
  float* sdy = smem;                 // [128][DYS]
  float* sT = sdy + 128 * DYS;       // [128][TS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty, seg = t / a.nty;
  const int nb = blockIdx.y;
  const int x0 = tx * 8, y0 = ty * 8;
  const int z_beg = seg * a.seglen;
  const int z_end = (z_beg + a.seglen) < a.D ? (z_beg + a.seglen) : a.D;
  // weights of this lane: B[k = co (2 s + lh)][j = column nt * 32 + li]
  float bw[NTL][KS];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int k = nt * 32 + li;
    const int tap = k < KT ? k / CIN : 0, ci = k < KT ? k - tap * CIN : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int co = 2 * s + lh;
      const bool ok = k < KT && co < a.Cout;
      const float v = a.w[((size_t)(ok ? co : 0) * CIN + ci) * 27 + tap];
      bw[nt][s] = ok ? v : 0.f;
    }
  }
  // dY halo plane p (dY z index): voxel (hy, hx) <-> dY (y0 - 1 + PH ... ) see below. Row r of the
  // image = hy * 10 + hx, r < 100; the float4 pieces of a plane are spread over the threads.
  constexpr int C4 = CO / 4, PIECES = (100 * C4 + 255) / 256;
  float4 pre[PIECES];
  auto fetch_plane = [&](int p) {
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
      const int i = tid + 256 * u;
      const int c4 = i % C4, r = i / C4;
      const int hy = r / 10, hx = r - hy * 10;
      const int yy = y0 + a.PH - 2 + hy, xx = x0 + a.PW - 2 + hx;   // dY coordinates
      const bool ok = r < 100 && p >= 0 && p < a.Do && yy >= 0 && yy < a.Ho && xx >= 0 &&
                      xx < a.Wo && 4 * c4 < a.Cout;
      const size_t off =
          ok ? ((((size_t)nb * a.Do + p) * a.Ho + yy) * a.Wo + xx) * a.Cout + 4 * c4 : 0;
      const float4 f = *reinterpret_cast<const float4*>(a.dy + off);
      pre[u] = ok ? f : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put_plane = [&]() {
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
      const int i = tid + 256 * u;
      const int c4 = i % C4, r = i / C4;
      if (r < 100) {
        float* q = sdy + r * DYS + 4 * c4;
        q[0] = pre[u].x; q[1] = pre[u].y; q[2] = pre[u].z; q[3] = pre[u].w;
      }
    }
  };
  // rows 100..127 of the image are never written: zero them once (their T rows are never read)
  for (int i = tid; i < 28 * DYS; i += 256) sdy[100 * DYS + i] = 0.f;
  // dx plane z gathers dY planes z + PD - kz, kz = 0..2: the first plane needed is z_beg + PD - 2
  const int p_first = z_beg + a.PD - 2, p_last = z_end - 1 + a.PD;
  float run[3] = {0.f, 0.f, 0.f};
  fetch_plane(p_first);
  for (int p = p_first; p <= p_last; ++p) {
    __syncthreads();            // previous step's readers of sdy and of T are done
    put_plane();
    __syncthreads();
    if (p + 1 <= p_last) fetch_plane(p + 1);   // in flight during the MFMAs
    // T rows of this wave: M tile = wave (rows wave * 32 + li)
    f32x16 acc[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const float* arow = sdy + (wave * 32 + li) * DYS + lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float av = arow[2 * s];
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[nt][s], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sT[row * TS + nt * 32 + li] = acc[nt][r];
      }
    __syncthreads();
    // dY plane p feeds dx planes z = p - PD + kz: run[kz] is the running sum of that plane
    if (tid < 64 * CIN) {
      const int ci = tid % CIN, v = tid / CIN, vy = v >> 3, vx = v & 7;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        float t9 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            // dY (yy + PH - ky, xx + PW - kx) = halo (vy + 2 - ky, vx + 2 - kx)
            const int r = (vy + 2 - ky) * 10 + (vx + 2 - kx);
            t9 += sT[r * TS + ((kz * 3 + ky) * 3 + kx) * CIN + ci];
          }
        run[kz] += t9;
      }
      // kz = 0 closes plane z = p - PD; the others move one slot down
      const int z = p - a.PD, yy = y0 + vy, xx = x0 + vx;
      if (z >= z_beg && z < z_end && yy < a.H && xx < a.W)
        a.dx[((((size_t)nb * a.D + z) * a.H + yy) * a.W + xx) * CIN + ci] = run[0];
      run[0] = run[1];
      run[1] = run[2];
      run[2] = 0.f;
    }
  }
}

// The same kernel with the per-voxel GEMM T = dY W on the f16 MFMA (f16x3 splits): the dY plane
// image holds the 64-byte rows of conv_igemm_f16.h ([voxel][hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] per
// 16-channel chunk, slots XOR-swizzled by (row >> 2) & 3), scaled by a power of two from the
// PLANE's absmax (the reduction rides the barrier that already separates the steps); the weights
// sit in registers as split fragments with one power-of-two scale per column (tap, ci).
// K = Cout = 32: 2 chunks x 3 products x NTL tiles = 12 MFMAs of 32 cycles per 32 voxels where the
// fp32 form issues 32 of 64 (the profile had this kernel at 63 % matrix-pipe busy).
template <int CIN, int CO>
__global__ __launch_bounds__(256, 2) void adell_cinfold_dx_f16_kernel(CinFoldDxArgs a) {
  constexpr int KT = 27 * CIN, NTL = (KT + 31) / 32, KP = NTL * 32, NCH = CO / 16;
  constexpr int TS = KP + 1;         // row stride of a T plane
  extern __shared__ float smem[];
  char* sdy = reinterpret_cast<char*>(smem);                       // [NCH][128 rows][64 B]
  float* sT = smem + NCH * 128 * 16;                               // [128][TS]
  float* smax = sT + 128 * TS;                                     // [2][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty, seg = t / a.nty;
  const int nb = blockIdx.y;
  const int x0 = tx * 8, y0 = ty * 8;
  const int z_beg = seg * a.seglen;
  const int z_end = (z_beg + a.seglen) < a.D ? (z_beg + a.seglen) : a.D;
  // weights of this lane: column j = nt * 32 + li = (tap, ci); fragment (chunk c): co 16 c + 8 lh + q
  cf_half8 bh[NTL][NCH], bl[NTL][NCH];
  float wun[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int k = nt * 32 + li;
    const int tap = k < KT ? k / CIN : 0, ci = k < KT ? k - tap * CIN : 0;
    float wv[NCH][8];
    float mx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int co = 16 * c + 8 * lh + q;
        const bool ok = k < KT && co < a.Cout;
        const float v = a.w[((size_t)(ok ? co : 0) * CIN + ci) * 27 + tap];
        wv[c][q] = ok ? v : 0.f;
        mx = fmaxf(mx, fabsf(wv[c][q]));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));       // the column's other eight-channel halves
    const int kw = cf_scale_exp(mx);
    const float wsc = __int_as_float((kw + 127) << 23);
    wun[nt] = __int_as_float((127 - kw) << 23);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v = wv[c][q] * wsc;
        const _Float16 h = (_Float16)v;
        bh[nt][c][q] = h;
        bl[nt][c][q] = (_Float16)(v - (float)h);
      }
  }
  constexpr int C4 = CO / 4, PIECES = (100 * C4 + 255) / 256;
  float4 pre[PIECES];
  auto fetch_plane = [&](int p) {
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
      const int i = tid + 256 * u;
      const int c4 = i % C4, r = i / C4;
      const int hy = r / 10, hx = r - hy * 10;
      const int yy = y0 + a.PH - 2 + hy, xx = x0 + a.PW - 2 + hx;   // dY coordinates
      const bool ok = r < 100 && p >= 0 && p < a.Do && yy >= 0 && yy < a.Ho && xx >= 0 &&
                      xx < a.Wo && 4 * c4 < a.Cout;
      const size_t off =
          ok ? ((((size_t)nb * a.Do + p) * a.Ho + yy) * a.Wo + xx) * a.Cout + 4 * c4 : 0;
      const float4 f = *reinterpret_cast<const float4*>(a.dy + off);
      pre[u] = ok ? f : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto plane_max = [&](int slot) {
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < PIECES; ++u)
      mx = fmaxf(fmaxf(fmaxf(mx, fabsf(pre[u].x)), fmaxf(fabsf(pre[u].y), fabsf(pre[u].z))),
                 fabsf(pre[u].w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) smax[slot * 4 + wave] = mx;
  };
  auto put_plane = [&](float sc) {
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
      const int i = tid + 256 * u;
      const int c4 = i % C4, r = i / C4;
      if (r < 100) {
        // channels 4 c4 .. + 3 of row r: chunk c4 >> 2, halfs (c4 & 3) * 4 inside the chunk
        const float v[4] = {pre[u].x * sc, pre[u].y * sc, pre[u].z * sc, pre[u].w * sc};
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = (_Float16)v[q];
          l[q] = (_Float16)(v[q] - (float)h[q]);
        }
        const int sw = (r >> 2) & 3, slot = (c4 & 3) >> 1;
        char* row = sdy + ((size_t)(c4 >> 2) * 128 + r) * 64 + (c4 & 1) * 8;
        *reinterpret_cast<h4*>(row + ((slot ^ sw) << 4)) = h;
        *reinterpret_cast<h4*>(row + (((2 + slot) ^ sw) << 4)) = l;
      }
    }
  };
  // rows 100..127 of the image are never written: zero them once (their T rows are never read)
  for (int i = tid; i < NCH * 28 * 16; i += 256) {
    const int c = i / (28 * 16), w = i - c * 28 * 16;
    reinterpret_cast<float*>(sdy)[(c * 128 + 100) * 16 + w] = 0.f;
  }
  const int p_first = z_beg + a.PD - 2, p_last = z_end - 1 + a.PD;
  float run[3] = {0.f, 0.f, 0.f};
  fetch_plane(p_first);
  int it = 0;
  for (int p = p_first; p <= p_last; ++p, ++it) {
    plane_max(it & 1);
    __syncthreads();            // previous step's readers of sdy and of T are done; maxima visible
    const float* sm = smax + (it & 1) * 4;
    const int kd = cf_scale_exp(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])));
    put_plane(__int_as_float((kd + 127) << 23));
    __syncthreads();
    if (p + 1 <= p_last) fetch_plane(p + 1);   // in flight during the MFMAs
    f32x16 acc[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int rowi = wave * 32 + li;
    const int sw = (rowi >> 2) & 3;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const char* row = sdy + ((size_t)c * 128 + rowi) * 64;
      const cf_half8 ah = *reinterpret_cast<const cf_half8*>(row + ((lh ^ sw) << 4));
      const cf_half8 al = *reinterpret_cast<const cf_half8*>(row + (((2 + lh) ^ sw) << 4));
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nt][c], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nt][c], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt][c], acc[nt], 0, 0, 0);
      }
    }
    const float dun = __int_as_float((127 - kd) << 23);
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sT[row * TS + nt * 32 + li] = acc[nt][r] * (dun * wun[nt]);
      }
    __syncthreads();
    if (tid < 64 * CIN) {
      const int ci = tid % CIN, v = tid / CIN, vy = v >> 3, vx = v & 7;
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        float t9 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int r = (vy + 2 - ky) * 10 + (vx + 2 - kx);
            t9 += sT[r * TS + ((kz * 3 + ky) * 3 + kx) * CIN + ci];
          }
        run[kz] += t9;
      }
      const int z = p - a.PD, yy = y0 + vy, xx = x0 + vx;
      if (z >= z_beg && z < z_end && yy < a.H && xx < a.W)
        a.dx[((((size_t)nb * a.D + z) * a.H + yy) * a.W + xx) * CIN + ci] = run[0];
      run[0] = run[1];
      run[1] = run[2];
      run[2] = 0.f;
    }
  }
}

extern "C" int adell_conv_cinfold_dx_applicable(const adell_conv3d_desc* d) {
  return (adell_cinfold_ok(d) && d->Cout <= 64 && d->Cout % 4 == 0) ? 1 : 0;
}

template <int CIN, int CO>
static void adell_cinfold_dx_f16_launch(const CinFoldDxArgs& a, dim3 grid, hipStream_t st) {
  constexpr int KP = ((27 * CIN + 31) / 32) * 32;
  const size_t lds = (size_t)((CO / 16) * 128 * 16 + 128 * (KP + 1) + 8) * sizeof(float);
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adell_cinfold_dx_f16_kernel<CIN, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    done = true;
  }
  hipLaunchKernelGGL((adell_cinfold_dx_f16_kernel<CIN, CO>), grid, dim3(256), lds, st, a);
}

template <int CIN>
static void adell_cinfold_dx_launch(const CinFoldDxArgs& a, dim3 grid, hipStream_t st, int f16x3) {
  constexpr int KP = ((27 * CIN + 31) / 32) * 32;
  if (f16x3 && a.Cout <= 32) {
    adell_cinfold_dx_f16_launch<CIN, 32>(a, grid, st);
    return;
  }
  if constexpr (CIN < 4) {   // (4 channels x 64 columns: the split fragments do not fit 256 registers)
    if (f16x3) {
      adell_cinfold_dx_f16_launch<CIN, 64>(a, grid, st);
      return;
    }
  }
  if (a.Cout <= 32) {
    const size_t lds = (size_t)(128 * 33 + 128 * (KP + 1)) * sizeof(float);
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adell_cinfold_dx_kernel<CIN, 32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      done = true;
    }
    hipLaunchKernelGGL((adell_cinfold_dx_kernel<CIN, 32>), grid, dim3(256), lds, st, a);
  } else {
    const size_t lds = (size_t)(128 * 65 + 128 * (KP + 1)) * sizeof(float);
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adell_cinfold_dx_kernel<CIN, 64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      done = true;
    }
    hipLaunchKernelGGL((adell_cinfold_dx_kernel<CIN, 64>), grid, dim3(256), lds, st, a);
  }
}

static int adell_cinfold_bwd_data_impl(const adell_conv3d_desc* d, const float* dy, const float* w,
                                       float* dx, int f16x3, void* stream);

extern "C" int adell_conv_cinfold_bwd_data(const adell_conv3d_desc* d, const float* dy,
                                           const float* w, float* dx, void* stream) {
  return adell_cinfold_bwd_data_impl(d, dy, w, dx, 0, stream);
}

// the same with the per-voxel GEMM on the f16 MFMA (error-compensated splits, ~2^-22 per product)
extern "C" int adell_conv_cinfold_bwd_data_f16x3(const adell_conv3d_desc* d, const float* dy,
                                                 const float* w, float* dx, void* stream) {
  return adell_cinfold_bwd_data_impl(d, dy, w, dx, 1, stream);
}

static int adell_cinfold_bwd_data_impl(const adell_conv3d_desc* d, const float* dy, const float* w,
                                       float* dx, int f16x3, void* stream) {
  ADELL_REQUIRE(dy && w && dx && adell_conv_cinfold_dx_applicable(d),
                "conv_cinfold_bwd_data: 3x3x3 stride-1 conv, 1..4 input channels, Cout <= 64 "
                "(multiple of 4) expected");
  ADELL_REQUIRE((((uintptr_t)dy) & 15) == 0, "conv_cinfold_bwd_data: dy must be 16-byte aligned");
  CinFoldDxArgs a = {};
  a.dy = dy; a.w = w; a.dx = dx;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo; a.PD = d->PD; a.PH = d->PH; a.PW = d->PW;
  a.ntx = adell_cdiv(d->W, 8);
  a.nty = adell_cdiv(d->H, 8);
  // z segments: enough blocks for two per CU (each segment re-computes two T planes; four per CU
  // measured the same 0.39 ms at 2 x 128^3, 32 channels)
  const long cols = (long)a.ntx * a.nty * d->N;
  int nseg = (int)((512 + cols - 1) / cols);
  if (nseg < 1) nseg = 1;
  while (nseg > 1 && adell_cdiv(d->D, nseg) < 8) --nseg;
  a.nseg = nseg;
  a.seglen = adell_cdiv(d->D, nseg);
  a.nseg = adell_cdiv(d->D, a.seglen);
  dim3 grid((unsigned)(a.ntx * a.nty * a.nseg), (unsigned)d->N);
  hipStream_t st = (hipStream_t)stream;
  switch (d->C0) {
    case 1: adell_cinfold_dx_launch<1>(a, grid, st, f16x3); break;
    case 2: adell_cinfold_dx_launch<2>(a, grid, st, f16x3); break;
    case 3: adell_cinfold_dx_launch<3>(a, grid, st, f16x3); break;
    default: adell_cinfold_dx_launch<4>(a, grid, st, f16x3); break;
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
