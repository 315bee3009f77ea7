// Implicit-GEMM 3D convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One kernel serves: forward conv (k in 1..3 per dim, stride 1..2, any pad,
// virtual channel-concat of two sources), backward-data (same kernel on dY with
// tap-flipped / transposed weights, zero-inserted input when stride>1), the
// k=s=2 transposed conv (as a 1x1x1 conv to 8*Cs columns with a pixel-shuffle
// store) and its backward-data (k=2,s=2 conv).
//
// GEMM view:  M = output voxels (NDHWC rows), N = output channels,
//             K = taps x input channels.
// Block tile: BM = WM*MT*32 voxels (a TX x TY x TZ brick) x BN = WN*NT*32 chans.
// Per 8-channel chunk the block stages into LDS
//   sA[8][VP]          the input halo brick, channel-major (transposed from
//                      NDHWC) so that the 32 lanes of an MFMA row group read 32
//                      neighbouring voxels of one channel (ds_read_b32)
//   sB[ntap][8][BN]    the weight slice for every tap of this chunk
// then runs ntap x 4 k-steps of MT x NT MFMAs per wave.
#pragma once
#include "common.h"

struct ConvArgs {
  const float* x0;
  const float* x1;
  const float* w;     // packed [ntap][Cin][Cout]
  const float* bias;  // [Cout] (or [Cs] when shuffle) or null
  const float* res;   // residual added in the epilogue, output-shaped, or null
  float* y0;
  float* y1;
  float* part;        // per-block per-channel (sum, sumsq) partials or null
  int D, H, W;        // real input dims
  int C0, C1, Cin, Cout;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int UPS;            // zero-insertion factor applied to the input along x (1 = none)
  int UPSY, UPSZ;     // ... along y and z
  int Do, Ho, Wo;     // GEMM-M spatial dims
  int lTX, lTY, lTZ;  // log2 of the output brick dims
  int ntx, nty, ntz;
  int HX, HY, HZ, VP; // halo brick dims and LDS channel pitch (floats)
  int ysplit;         // columns [0,ysplit) -> y0, [ysplit,Cout) -> y1
  int shuffle;        // != 0: transposed-conv scatter store; bits 0/1/2 = factor 2 along x/y/z
  int GKH;            // f16 kernel: ky rows per staged weight group (KH, or fewer for big kernels)
  int Cs;             // channels of the shuffled destination
  int vecx, vecw;     // 16-byte global loads legal for input / weights
  int ksplit;         // f16 kernel: > 1 = blockIdx.z also indexes a share of the channel chunks
  long slab;          // ... whose partial outputs go to y0 + share * slab (plain [N][vox][Cout])
  int part_rows;      // host side: rows per batch item the caller sized `part` for (checked against
                      // the launch plan before anything is launched)
};

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(256) void adell_conv_igemm_kernel(ConvArgs a) {
  constexpr int BN = WN * NT * 32, CC = 8;
  extern __shared__ float smem[];
  float* sA = smem;
  float* sB = smem + CC * a.VP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  int t = blockIdx.x;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty;
  const int tz = t / a.nty;
  const int n0 = blockIdx.y * BN;
  const int nb = blockIdx.z;
  const int ox0 = tx << a.lTX, oy0 = ty << a.lTY, oz0 = tz << a.lTZ;
  const int ntap = a.KD * a.KH * a.KW;
  const int HV = a.HX * a.HY * a.HZ;
  const int HXY = a.HX * a.HY;
  const int lx0 = ox0 * a.SW - a.PW, ly0 = oy0 * a.SH - a.PH,
            lz0 = oz0 * a.SD - a.PD;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + li;
    const int x = m & ((1 << a.lTX) - 1);
    const int y = (m >> a.lTX) & ((1 << a.lTY) - 1);
    const int z = m >> (a.lTX + a.lTY);
    abase[mt] = ((z * a.SD) * a.HY + y * a.SH) * a.HX + x * a.SW + lh * a.VP;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = lh * BN + (wn * NT + nt) * 32 + li;

  const int nchunk = (a.Cin + CC - 1) / CC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int c0 = ch * CC;
    __syncthreads();
    // ---- stage the input halo brick (transpose to channel-major) ----------
    for (int hv = tid; hv < HV; hv += 256) {
      const int hz = hv / HXY;
      const int rem = hv - hz * HXY;
      const int hy = rem / a.HX;
      const int hx = rem - hy * a.HX;
      int rx = lx0 + hx, ry = ly0 + hy, rz = lz0 + hz;
      bool ok = (rx >= 0) & (ry >= 0) & (rz >= 0);
      if ((a.UPS | a.UPSY | a.UPSZ) > 1) {
        ok = ok & (rx % a.UPS == 0) & (ry % a.UPSY == 0) & (rz % a.UPSZ == 0);
        rx /= a.UPS;
        ry /= a.UPSY;
        rz /= a.UPSZ;
      }
      ok = ok & (rx < a.W) & (ry < a.H) & (rz < a.D);
      const size_t gv = ((size_t)(nb * a.D + rz) * a.H + ry) * a.W + rx;
      float v[CC];
#pragma unroll
      for (int j = 0; j < CC; ++j) v[j] = 0.f;
      if (ok) {
        if (a.vecx) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c = c0 + 4 * q;
            const float* p = nullptr;
            if (c < a.C0)
              p = a.x0 + gv * a.C0 + c;
            else if (c < a.Cin)
              p = a.x1 + gv * a.C1 + (c - a.C0);
            if (p) {
              const float4 f = *reinterpret_cast<const float4*>(p);
              v[4 * q + 0] = f.x;
              v[4 * q + 1] = f.y;
              v[4 * q + 2] = f.z;
              v[4 * q + 3] = f.w;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < CC; ++j) {
            const int c = c0 + j;
            if (c < a.C0)
              v[j] = a.x0[gv * a.C0 + c];
            else if (c < a.Cin)
              v[j] = a.x1[gv * a.C1 + (c - a.C0)];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < CC; ++j) sA[j * a.VP + hv] = v[j];
    }
    // ---- stage the weight slice [ntap][8][BN] -----------------------------
    {
      const int items = ntap * CC * (BN / 4);
      for (int it = tid; it < items; it += 256) {
        const int n4 = it % (BN / 4);
        const int rc = it / (BN / 4);
        const int c = rc % CC;
        const int tap = rc / CC;
        const int cg = c0 + c, ng = n0 + 4 * n4;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cg < a.Cin) {
          const float* p = a.w + ((size_t)tap * a.Cin + cg) * a.Cout + ng;
          if (a.vecw) {
            if (ng < a.Cout) f = *reinterpret_cast<const float4*>(p);
          } else {
            if (ng + 0 < a.Cout) f.x = p[0];
            if (ng + 1 < a.Cout) f.y = p[1];
            if (ng + 2 < a.Cout) f.z = p[2];
            if (ng + 3 < a.Cout) f.w = p[3];
          }
        }
        *reinterpret_cast<float4*>(&sB[(tap * CC + c) * BN + 4 * n4]) = f;
      }
    }
    __syncthreads();
    // ---- MFMA over taps x 4 k-steps ---------------------------------------
    int kx = 0, ky = 0, kz = 0;
    for (int tap = 0; tap < ntap; ++tap) {
      const int aoff = (kz * a.HY + ky) * a.HX + kx;
      const int boff = tap * CC * BN;
#pragma unroll
      for (int s = 0; s < CC / 2; ++s) {
        float av[MT], bv[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[mt] = sA[abase[mt] + aoff + 2 * s * a.VP];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = sB[boff + bbase[nt] + 2 * s * BN];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
      }
      if (++kx == a.KW) {
        kx = 0;
        if (++ky == a.KH) {
          ky = 0;
          ++kz;
        }
      }
    }
  }

  // ---- epilogue: bias, residual, store, per-channel partial statistics ----
  // dst(row, col) = colptr[col] + rowoff(row) * rowmul[col]; this one form
  // covers the plain store, the split store (two concat sources in
  // backward-data) and the 2x2x2 pixel-shuffle scatter of the transposed conv.
  float s1[NT], s2[NT], bcol[NT];
  float* colptr[NT];
  int rowmul[NT];
  bool nok[NT];
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = s2[nt] = 0.f;
    const int n = n0 + (wn * NT + nt) * 32 + li;
    ncol[nt] = n;
    nok[nt] = n < a.Cout;
    bcol[nt] = 0.f;
    colptr[nt] = a.y0;
    rowmul[nt] = 0;
    if (nok[nt]) {
      if (a.shuffle) {
        const int fx = (a.shuffle & 1) + 1, fy = ((a.shuffle >> 1) & 1) + 1;
        const int sub = n / a.Cs, co = n - sub * a.Cs;
        const int sx = sub % fx, sy = (sub / fx) % fy, sz = sub / (fx * fy);
        colptr[nt] = a.y0 + ((size_t)(sz * fy * a.Ho + sy) * (fx * a.Wo) + sx) * a.Cs + co;
        rowmul[nt] = a.Cs;
        if (a.bias) bcol[nt] = a.bias[co];
      } else {
        if (n < a.ysplit) {
          colptr[nt] = a.y0 + n;
          rowmul[nt] = a.ysplit;
        } else {
          colptr[nt] = a.y1 + (n - a.ysplit);
          rowmul[nt] = a.Cout - a.ysplit;
        }
        if (a.bias) bcol[nt] = a.bias[n];
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = (wm * MT + mt) * 32 + row;
      const int x = ox0 + (m & ((1 << a.lTX) - 1));
      const int y = oy0 + ((m >> a.lTX) & ((1 << a.lTY) - 1));
      const int z = oz0 + (m >> (a.lTX + a.lTY));
      const bool rok = (x < a.Wo) & (y < a.Ho) & (z < a.Do);
      const int ov = ((nb * a.Do + z) * a.Ho + y) * a.Wo + x;
      const int fx = (a.shuffle & 1) + 1, fy = ((a.shuffle >> 1) & 1) + 1,
                fz = ((a.shuffle >> 2) & 1) + 1;
      const int ovs = ((nb * fz * a.Do + fz * z) * (fy * a.Ho) + fy * y) * (fx * a.Wo) + fx * x;
      const int rowoff = a.shuffle ? ovs : ov;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (rok && nok[nt]) {
          float v = acc[mt][nt][r] + bcol[nt];
          if (a.res) v += a.res[(size_t)ov * a.Cout + ncol[nt]];
          colptr[nt][(size_t)rowoff * rowmul[nt]] = v;
          s1[nt] += v;
          s2[nt] += v * v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (a.part) {
    __syncthreads();
    float* red = smem;  // [WM][BN][2]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float t1 = s1[nt] + __shfl_xor(s1[nt], 32, 64);
      const float t2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
      if (lh == 0) {
        const int col = (wn * NT + nt) * 32 + li;
        red[(wm * BN + col) * 2 + 0] = t1;
        red[(wm * BN + col) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t1 += red[(w * BN + tid) * 2 + 0];
          t2 += red[(w * BN + tid) * 2 + 1];
        }
        const size_t ntiles = (size_t)a.ntx * a.nty * a.ntz;
        float* p = a.part + ((nb * ntiles + blockIdx.x) * a.Cout + n) * 2;
        p[0] = t1;
        p[1] = t2;
      }
    }
  }
}
