// Implicit-GEMM 3D convolution with fp32 operands computed on the f16 MFMA by
// error-compensated splitting ("f16x3"):
//
//     a = a_hi + a_lo,  b = b_hi + b_lo     (a_hi = fp16(a), a_lo = fp16(a - a_hi))
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (drops a_lo*b_lo ~ 2^-22 |ab|)
//
// Every product of two fp16 values is exact in the MFMA's fp32 accumulator, so the
// result carries ~22 bits per product -- fp32-class accuracy at 3 f16 MFMAs per
// K-block, i.e. 16/3 = 5.3x the rate of v_mfma_f32_32x32x2_f32.
//
// Range: fp16 has a 5-bit exponent, so each staged 16-channel chunk of the input
// brick is scaled by a power of two chosen from the block-wide absmax of that
// chunk (max lands in [2^13, 2^14)); when the scale changes between chunks the
// fp32 accumulators are rescaled by the exact power-of-two ratio. Weights carry
// one power-of-two scale per layer (computed by the pack kernel). Both scales
// are undone in the epilogue. No tensor-wide statistics are needed, so the
// kernel serves activations (forward) and gradients (backward-data) alike.
//
// Tiling is the fp32 kernel's (conv_igemm.h): M = TX x TY x TZ voxel brick, N = 32
// or 64 channels, 4 waves. LDS (all fp16, 64-byte rows, 16-byte slots XOR-swizzled
// by (row >> 2) & 3 so that ds_read_b128 fragments are conflict-free):
//   sA[halo voxel][hi c0..7 | hi c8..15 | lo c0..7 | lo c8..15]
//   sB[tap of one kz plane][n][same four slots]
#pragma once
#include <type_traits>
#include "conv_igemm.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// A pointer the caller knows to be wave-uniform, pinned into SGPRs so that `p + (unsigned)offset`
// becomes one scalar-base + 32-bit-VGPR-offset access instead of a 64-bit per-lane pointer (which
// the compiler would hoist out of the loops, one register pair per access, and spill).
// (ADELL_GLOBAL, common.h: global_load, not flat_load -- round 2 shipped the generic-pointer form,
// whose flat loads made every LDS fragment wait of the MFMA loop a wait for the weight prefetch)
__device__ __forceinline__ const ADELL_GLOBAL char* adell_uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const ADELL_GLOBAL char*>(((uint64_t)hi << 32) | lo);
}

// 16 bytes from global memory
__device__ __forceinline__ float4 adell_gload4(const ADELL_GLOBAL char* p) {
  const f32x4 v = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

struct ConvF16Extra {
  const _Float16* wh;    // packed split weights [tap][Cout][nchunk][32 halfs]
  const float* wscale;   // [Cout] 2^-kw[n]: undoes the per-output-channel weight scale
  unsigned* amax_out;    // optional: receives the absmax (float bits) of the input tensor(s)
  // Split-row sources (optional, per source; SPEC instances): the tensor holds, per voxel and
  // 16-channel chunk, the 64-byte row [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] of the LDS image, scaled
  // by 2^xk[item * (C / 16) + chunk] (same bytes per element as fp32; written by the producer:
  // adell_norm_act_fwd_split). Staging is then a copy: no block-wide absmax, no conversion VALU
  // work. xs0 / xs1 null: that source is fp32 (a.x0 / a.x1).
  const char* xs0;
  const char* xs1;
  const int* xk0;
  const int* xk1;
  // Backward of a norm -> dropout -> activation site fused into this (backward-data) launch
  // (EPI = 1 instances): destination d (0: columns [0, ysplit), 1: the rest) is the gradient with
  // respect to the OUTPUT of such a site whose pre-norm input is adn[d].y (same shape as the
  // destination). The epilogue then stores dt = dout * act'(u) * keep / (1 - p) instead of dout and
  // emits the per-brick partial sums (sum dt, sum dt * xhat) through ConvArgs::part: the first of the
  // two elementwise backward passes of the site (norm_act.hip) is gone. adn[d].y == null: plain store.
  struct Adn {
    const float* y;
    const float* mean;       // [N][C of the destination]
    const float* rstd;
    const unsigned* mask;    // keep bits written by the site's forward (norm_act.hip), or null
    float keep_scale;        // 1 / (1 - p)
    float act_p;
    int act;                 // ADELL_ACT_*
    int groups;              // 256-element groups per batch item in `mask`
  } adn[2];
  int dbg;               // timing experiments (-DADELL_DEBUG builds only: ADELL_IGEMM_DBG,
                         // tools/igemm_dbg.py): results are wrong when nonzero. 1: halo staged for chunk 0 only; 2: weights staged
                         // for the first tap group only; 8: no MFMAs; 16: no output stores; 32: no weight-group barriers
};

__device__ __forceinline__ void adell_split8(const float* v, float scale, half8* hi, half8* lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = v[j] * scale;
    const _Float16 h = (_Float16)t;
    (*hi)[j] = h;
    (*lo)[j] = (_Float16)(t - (float)h);
  }
}

// SPEC = 1 is the instance for the configuration that carries the FLOPs of a U-Net: 3x3x3 taps,
// stride 1, no zero insertion, 8x8x4 output brick (10x10x6 halo), 16-byte-aligned channel
// counts, plain store. Every index of the halo, tap and epilogue arithmetic is then a
// compile-time constant (no integer divisions in the loops, tap loop fully unrolled).
// SPEC = 0 takes all of them from ConvArgs.
// EPI = 1 (SPEC instances, launches whose bricks are all whole): the fused ADN backward of
// ConvF16Extra::adn in the epilogue.
// In-kernel stamps (debug build only, tools/igemm_stamps.py): thread 0 of every block writes
// s_memtime at the phase boundaries of its life into adell_g_stamps[block][24] when that pointer
// is set (adell_debug_set_stamps); no stamp executes in the product build.
#ifdef ADELL_DEBUG
extern __device__ unsigned long long* adell_g_stamps;
#define ADELL_STAMP(i)                                                                          \
  do {                                                                                          \
    if (adell_g_stamps != nullptr && threadIdx.x == 0)                                          \
      adell_g_stamps[((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 24 + (i)] = \
          __builtin_amdgcn_s_memtime();                                                         \
  } while (0)
#else
#define ADELL_STAMP(i) do { } while (0)
#endif

// ROWS (SPEC instances): 0 = fp32 sources; 2 = every source holds split rows (ConvF16Extra::xs0 /
// xs1: staging is a copy, the fp32 staging path is not compiled in); 1 = decided per source at run
// time (the two halves of a virtual concat in different formats: both paths compiled in).
template <int MT, int NT, int WM, int WN, int SPEC, int EPI = 0, int ROWS = 0>
__global__ __launch_bounds__(WM * WN * 64, SPEC == 2 ? 3 : (WM * WN >= 8 ? 2 : WM * WN / 2))
void adell_conv_igemm_f16_kernel(ConvArgs a, ConvF16Extra e) {
  constexpr int BN = WN * NT * 32, CC = 16;
  // SPEC = 2: the same with the 27 taps staged in 4 linear groups of <= 7 (not per kz plane):
  // 52.8 KB of LDS and <= 168 registers, i.e. three blocks per CU for the 32-channel tile
  // (measured +5 % over SPEC 1; superseded by SPEC 3 and not instantiated).
  constexpr int GT = SPEC >= 2 ? 7 : 9;               // taps per weight group (SPEC)
  constexpr int NGRP = (27 + GT - 1) / GT;
  constexpr int NW = WM * WN, NTHR = NW * 64;  // 4 waves (2 blocks per CU) or 8 (4 waves per SIMD)
  // SPEC = 3: 8x8x8 brick (10x10x10 halo), 4 m-tiles per wave, 7-tap weight groups: 78 KB of
  // LDS, two blocks per CU -- less halo per output, fewer LDS reads and barriers per MFMA.
  const int lTX = SPEC ? 3 : a.lTX, lTY = SPEC ? 3 : a.lTY, lTZ = SPEC == 3 ? 3 : (SPEC ? 2 : a.lTZ);
  const int HX = SPEC ? 10 : a.HX, HY = SPEC ? 10 : a.HY, HZ = SPEC == 3 ? 10 : (SPEC ? 6 : a.HZ);
  const int KD = SPEC ? 3 : a.KD, KH = SPEC ? 3 : a.KH, KW = SPEC ? 3 : a.KW;
  const int SD = SPEC ? 1 : a.SD, SH = SPEC ? 1 : a.SH, SW = SPEC ? 1 : a.SW;
  const int GKH = SPEC ? 3 : a.GKH;
  const int shuffle = SPEC ? 0 : a.shuffle;
  extern __shared__ float smem[];
  char* sA = reinterpret_cast<char*>(smem);
  const int HV = HX * HY * HZ;
  char* sB = sA + (size_t)HV * 64;
  float* sMax = reinterpret_cast<float*>(sB + (size_t)(SPEC ? GT : GKH * KW) * BN * 64);  // [NW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // consecutive blocks go round-robin to the 8 XCDs (one L2 each): give every XCD one
  // contiguous range of bricks so that neighbouring halos meet in the same L2
  const int nsp = a.ntx * a.nty * a.ntz;
  int t = (blockIdx.x & 7) * ((nsp + 7) >> 3) + (blockIdx.x >> 3);
  if (t >= nsp) return;
  const int tile_id = t;
  const int tx = t % a.ntx;
  t /= a.ntx;
  const int ty = t % a.nty;
  const int tz = t / a.nty;
  const int n0 = blockIdx.y * BN;
  // split-K (low-resolution layers: too few bricks to fill the chip): blockIdx.z = item * ksplit
  // + share; a share accumulates its channel chunks and stores raw partial outputs to its slab
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  const int nb = blockIdx.z / ksplit, kshare = blockIdx.z - nb * ksplit;
  const int ox0 = tx << lTX, oy0 = ty << lTY, oz0 = tz << lTZ;
  const int HXY = HX * HY;
  const int lx0 = ox0 * SW - a.PW, ly0 = oy0 * SH - a.PH, lz0 = oz0 * SD - a.PD;
  const int ngy = (KH + GKH - 1) / GKH;  // weight groups per kz plane

  const int nchunk = (a.Cin + CC - 1) / CC;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int arow[MT];  // halo voxel index of this lane's A row at tap (0,0,0)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + li;
    const int x = m & ((1 << lTX) - 1);
    const int y = (m >> lTX) & ((1 << lTY) - 1);
    const int z = m >> (lTX + lTY);
    arow[mt] = ((z * SD) * HY + y * SH) * HX + x * SW;
  }
  int boffh[NT], boffl[NT];  // byte offsets of this lane's B fragments inside one tap
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (wn * NT + nt) * 32 + li;
    const int sw = (n >> 2) & 3;
    boffh[nt] = n * 64 + ((lh ^ sw) << 4);
    boffl[nt] = n * 64 + (((2 + lh) ^ sw) << 4);
  }

  // input voxel behind halo voxel hv: its DHW index inside batch item nb, or -1 outside the
  // tensor / on an inserted zero. Independent of the channel chunk.
  auto halo_voxel = [&](int hv) -> int {
    const int hz = hv / HXY;
    const int rem = hv - hz * HXY;
    const int hy = rem / HX;
    const int hx = rem - hy * HX;
    int rx = lx0 + hx, ry = ly0 + hy, rz = lz0 + hz;
    bool ok = (rx >= 0) & (ry >= 0) & (rz >= 0);
    if (!SPEC && (a.UPS | a.UPSY | a.UPSZ) > 1) {
      ok = ok & (rx % a.UPS == 0) & (ry % a.UPSY == 0) & (rz % a.UPSZ == 0);
      rx /= a.UPS;
      ry /= a.UPSY;
      rz /= a.UPSZ;
    }
    ok = ok & (rx < a.W) & (ry < a.H) & (rz < a.D);
    return ok ? (rz * a.H + ry) * a.W + rx : -1;
  };
  const size_t vox0 = (size_t)nb * a.D * a.H * a.W;
  const float* x0n = a.x0 + vox0 * a.C0;                           // this batch item
  const float* x1n = a.x1 ? a.x1 + vox0 * a.C1 : nullptr;
  // the 16 channels [c0, c0+16) of input voxel gv (zeros for gv < 0)
  auto load16 = [&](int gvi, int c0, float* v) {
#pragma unroll
    for (int j = 0; j < CC; ++j) v[j] = 0.f;
    if (gvi < 0) return;
    if constexpr (SPEC) {
      // both sources hold whole chunks (C0, C1 multiples of 16): one uniform base per chunk
      const bool first = c0 < a.C0;
      const ADELL_GLOBAL char* src = adell_uniform_ptr(first ? x0n + c0 : x1n + (c0 - a.C0));
      const unsigned cs = first ? a.C0 : a.C1;
      // byte offset inside the batch item: < 2^32 (checked on the host)
      const ADELL_GLOBAL char* p = src + (unsigned)gvi * cs * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 f = adell_gload4(p + 16 * q);
        v[4 * q + 0] = f.x;
        v[4 * q + 1] = f.y;
        v[4 * q + 2] = f.z;
        v[4 * q + 3] = f.w;
      }
      return;
    }
    const size_t gv = (size_t)gvi;
    if (a.vecx) {
      // (vecx also says that a batch item spans < 2^32 bytes: uniform base + 32-bit byte offset per
      // load instead of a 64-bit pointer per load, which the compiler hoists and spills)
      const ADELL_GLOBAL char* b0 = adell_uniform_ptr(x0n);
      const ADELL_GLOBAL char* b1 = adell_uniform_ptr(x1n);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + 4 * q;
        if (c < a.Cin) {
          const bool first = c < a.C0;
          const unsigned off = first ? ((unsigned)gvi * (unsigned)a.C0 + (unsigned)c) * 4u
                                     : ((unsigned)gvi * (unsigned)a.C1 + (unsigned)(c - a.C0)) * 4u;
          const float4 f = adell_gload4((first ? b0 : b1) + off);
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
          v[j] = x0n[gv * a.C0 + c];
        else if (c < a.Cin)
          v[j] = x1n[gv * a.C1 + (c - a.C0)];
      }
    }
  };

  int kA_prev = 0;
  // The halo brick of a chunk goes through registers in one piece when it has at most KEEP
  // voxels per thread ("resident"): one load, absmax, split, store.
  // (generic single-tile instances have registers to spare: 6 voxels per thread keep the
  // 17 x 9 x 9 halo of a stride-2 3^3 layer out of the two-pass loop below, whose loads are
  // issued one dependent round at a time -- 2 x 64^3 x 32 -> 32 stride 2: 0.76 ms before)
  // (six voxels only on the two-wave instance, which runs those layers: on the four-wave single-tile
  // instances the 96 registers spilled)
  constexpr int KEEP = SPEC ? ((SPEC == 3 ? 1000 : 600) + NTHR - 1) / NTHR
                            : (MT * NT == 1 && NW == 2 ? 6 : 3);
  const bool resident = SPEC || HV <= KEEP * NTHR;
  float keep[KEEP][CC];
  int gvk[KEEP];
#pragma unroll
  for (int u = 0; u < KEEP; ++u) {
    const int hv = tid + NTHR * u;
    gvk[u] = (resident && hv < HV) ? halo_voxel(hv) : -1;
  }
  // Weight slices go through a register prefetch: the slice of the next tap group is fetched
  // while the MFMAs of the current one run (a kz plane of a 3^3 kernel: 9 taps x BN ch x 4
  // slots = 4.5 (BN 32) / 9 (BN 64) x 256 slots of 16 bytes).
  // (the 256-column transposed-conv instance, WN = 4, has one tap per group)
  constexpr int WPF = ((WN == 4 ? 1 : GT) * BN * 4 + NTHR - 1) / NTHR;  // a kz plane of a 3^3 kernel
  // (not on the generic 2x2-tile instance: with 64 accumulators and one fragment set the 36 prefetch
  // registers spilled; it stages its weight slices straight from global memory)
  const bool wpipe = SPEC || ((MT * NT < 4 || WN == 4) && GKH * KW * BN * 4 <= WPF * NTHR);
  float4 wreg[WPF];
  const int ngroups = SPEC ? NGRP : KD * ngy;
  // SPEC: slot index it = tid + u * NTHR means column (tid >> 2) % BN, tap u * TPU + (tid >> 2) / BN
  // of the plane: a per-thread 32-bit offset plus a uniform stride per u, on both sides
  constexpr int TPU = NTHR / 4 / BN;
  const int wcol = (tid >> 2) % BN, wtap = (tid >> 2) / BN;
  const unsigned wgoff = ((unsigned)(wtap * a.Cout + n0 + wcol) * nchunk) * 64 + (tid & 3) * 16;
  const unsigned wloff = ((wtap * BN + wcol) * 4 + ((tid & 3) ^ ((wcol >> 2) & 3))) * 16;
  const bool wcolok = n0 + wcol < a.Cout;
  auto wfetch = [&](int ch_, int grp_) {
    if constexpr (SPEC) {
      const char* base = reinterpret_cast<const char*>(e.wh) +
                         ((size_t)(grp_ * GT) * a.Cout * nchunk + ch_) * 64;
      const size_t ustride = (size_t)TPU * a.Cout * nchunk * 64;
      const int tpg_ = (27 - grp_ * GT) < GT ? (27 - grp_ * GT) : GT;
#pragma unroll
      for (int u = 0; u < WPF; ++u) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wcolok && u * TPU + wtap < tpg_)
          f = adell_gload4(adell_uniform_ptr(base + u * ustride) + wgoff);
        wreg[u] = f;
      }
      return;
    }
    const int kz = grp_ / ngy, ky0 = (grp_ - kz * ngy) * GKH;
    const int gkh = (KH - ky0) < GKH ? (KH - ky0) : GKH;
    const int tpg = gkh * KW;
    const int tap0 = (kz * KH + ky0) * KW;
#pragma unroll
    for (int u = 0; u < WPF; ++u) {
      const int it = tid + u * NTHR;
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < tpg * BN * 4) {
        const int slot = it & 3;
        const int n = (it >> 2) % BN;
        const int tap = tap0 + (it >> 2) / BN;
        if (n0 + n < a.Cout)
          f = *reinterpret_cast<const float4*>(
              reinterpret_cast<const char*>(e.wh) +
              (((size_t)tap * a.Cout + n0 + n) * nchunk + ch_) * 64 + slot * 16);
      }
      wreg[u] = f;
    }
  };
  const int cpk = (nchunk + ksplit - 1) / ksplit;
  const int c_beg = kshare * cpk, c_end = (c_beg + cpk) < nchunk ? (c_beg + cpk) : nchunk;
  if (ksplit > 1) {
    a.y0 += (size_t)kshare * a.slab;
    a.ysplit = a.Cout;
    a.bias = nullptr;
    a.res = nullptr;
    a.part = nullptr;
  }
  if (wpipe && c_beg < c_end) wfetch(c_beg, 0);
  ADELL_STAMP(0);
  for (int ch = c_beg; ch < c_end; ++ch) {
    const int c0 = ch * CC;
    if (ch - c_beg < 4) ADELL_STAMP(1 + 4 * (ch - c_beg));
    float mx = 0.f;
    const bool skipA = (ADELL_DBG(e.dbg) & 1) && ch > 0;
    const bool presplit = SPEC != 0 && (ROWS == 2 || (ROWS == 1 && (c0 < a.C0 ? e.xs0 : e.xs1) != nullptr));
    if (skipA) {
    } else if (presplit) {
      if constexpr (SPEC) {
        const bool first = c0 < a.C0;
        const unsigned nch = (first ? a.C0 : a.C1) >> 4;       // chunks per voxel of this source
        const ADELL_GLOBAL char* src = adell_uniform_ptr(
            (first ? e.xs0 + ((size_t)vox0 * (a.C0 >> 4) + (c0 >> 4)) * 64
                   : e.xs1 + ((size_t)vox0 * (a.C1 >> 4) + ((c0 - a.C0) >> 4)) * 64));
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
          if (tid + NTHR * u < HV) {
            float4 f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gvk[u] >= 0) {
              const ADELL_GLOBAL char* p = src + (size_t)((unsigned)gvk[u] * nch) * 64u;
#pragma unroll
              for (int q = 0; q < 4; ++q) f[q] = adell_gload4(p + 16 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              keep[u][4 * q + 0] = f[q].x; keep[u][4 * q + 1] = f[q].y;
              keep[u][4 * q + 2] = f[q].z; keep[u][4 * q + 3] = f[q].w;
            }
          }
        }
      }
    } else if (SPEC) {
      if constexpr (SPEC != 0) {
        // branch-free: every load of the chunk is in flight before the first value is looked at
        // (voxels outside the tensor / past the halo read voxel 0 and are zeroed). With the
        // per-voxel early-out of load16 each voxel's four loads were waited for in turn.
        const bool first = c0 < a.C0;
        const ADELL_GLOBAL char* src = adell_uniform_ptr(first ? x0n + c0 : x1n + (c0 - a.C0));
        const unsigned cs = first ? a.C0 : a.C1;
        float4 f[KEEP][4];
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
          const ADELL_GLOBAL char* p = src + (unsigned)(gvk[u] >= 0 ? gvk[u] : 0) * cs * 4u;
#pragma unroll
          for (int q = 0; q < 4; ++q) f[u][q] = adell_gload4(p + 16 * q);
        }
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
          const bool ok = gvk[u] >= 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            keep[u][4 * q + 0] = ok ? f[u][q].x : 0.f;
            keep[u][4 * q + 1] = ok ? f[u][q].y : 0.f;
            keep[u][4 * q + 2] = ok ? f[u][q].z : 0.f;
            keep[u][4 * q + 3] = ok ? f[u][q].w : 0.f;
          }
#pragma unroll
          for (int j = 0; j < CC; ++j) mx = fmaxf(mx, fabsf(keep[u][j]));
        }
      }
    } else if (resident) {
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        if (tid + NTHR * u < HV) {
          load16(gvk[u], c0, keep[u]);
#pragma unroll
          for (int j = 0; j < CC; ++j) mx = fmaxf(mx, fabsf(keep[u][j]));
        }
      }
    } else {
      for (int hv = tid; hv < HV; hv += NTHR) {
        float v[CC];
        load16(halo_voxel(hv), c0, v);
#pragma unroll
        for (int j = 0; j < CC; ++j) mx = fmaxf(mx, fabsf(v[j]));
      }
    }
    int kA = 0;
    if (presplit) {
      __syncthreads();  // previous chunk's MFMAs are done: LDS may be overwritten
      kA = c0 < a.C0 ? e.xk0[nb * (a.C0 >> 4) + (c0 >> 4)]
                     : e.xk1[nb * (a.C1 >> 4) + ((c0 - a.C0) >> 4)];
    } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();  // previous chunk's MFMAs are done: LDS may be overwritten
    if (lane == 0) sMax[wave] = mx;
    __syncthreads();
    mx = sMax[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, sMax[w]);
    // by-product for the backward-weight kernel: tensor-wide absmax of the input
    if (e.amax_out != nullptr && tid == 0 && blockIdx.y == 0)
      atomicMax(e.amax_out, __float_as_uint(mx));
    {
      const int ebits = (__float_as_int(mx) >> 23) & 0xff;
      // max lands in [2^6, 2^14): multiples of 8 so the scale rarely changes per chunk
      if (ebits > 0 && ebits < 255) kA = 8 * ((13 - (ebits - 127)) >> 3);
      if (kA > 96) kA = 96;
      if (kA < -96) kA = -96;
    }
    }
    const float scaleA = __int_as_float((kA + 127) << 23);
    if (ch > c_beg && kA != kA_prev) {
      const float f = __int_as_float((kA - kA_prev + 127) << 23);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
    }
    kA_prev = kA;
    if (ch - c_beg < 4) ADELL_STAMP(2 + 4 * (ch - c_beg));
    auto store_split = [&](int hv, const float* v) {
      half8 h0, l0, h1, l1;
      adell_split8(v, scaleA, &h0, &l0);
      adell_split8(v + 8, scaleA, &h1, &l1);
      // SPEC: slot permutation by the halo voxel's x pair, (x >> 1) & 3. The 16-lane groups of a
      // ds_read_b128 hold 4 x-neighbours in each of 4 consecutive y (8x8 brick faces): with the
      // row's bank quarter (2 y + x) mod 4 every lane of a group lands in its own (quarter, slot)
      // cell for all 27 taps -- conflict-free fragment reads (the generic permutation costs 3x
      // the LDS cycles on this brick) -- and the eight consecutive voxels of a ds_write_b128 group
      // of the staging stores spread over the four slots as well (a permutation by y alone, round
      // 2, left them 4-way conflicted: 12-17 % of the kernel's LDS-active cycles;
      // searched exhaustively over the linear forms of (y, x)).
      const int sw = SPEC ? ((hv % 10) >> 1) & 3 : (hv >> 2) & 3;
      char* row = sA + (size_t)hv * 64;
      *reinterpret_cast<half8*>(row + ((0 ^ sw) << 4)) = h0;
      *reinterpret_cast<half8*>(row + ((1 ^ sw) << 4)) = h1;
      *reinterpret_cast<half8*>(row + ((2 ^ sw) << 4)) = l0;
      *reinterpret_cast<half8*>(row + ((3 ^ sw) << 4)) = l1;
    };
    if (skipA) {
    } else if (presplit) {
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        const int hv = tid + NTHR * u;
        if (hv < HV) {
          const int sw = ((hv % 10) >> 1) & 3;
          char* row = sA + (size_t)hv * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(row + ((q ^ sw) << 4)) =
                make_float4(keep[u][4 * q + 0], keep[u][4 * q + 1], keep[u][4 * q + 2],
                            keep[u][4 * q + 3]);
        }
      }
    } else if (resident) {
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        const int hv = tid + NTHR * u;
        if (hv < HV) store_split(hv, keep[u]);
      }
    } else {
      for (int hv = tid; hv < HV; hv += NTHR) {
        float v[CC];
        load16(halo_voxel(hv), c0, v);
        store_split(hv, v);
      }
    }
    if (ch - c_beg < 4) ADELL_STAMP(3 + 4 * (ch - c_beg));
#pragma unroll SPEC >= 2 ? NGRP : 1
    for (int grp = 0; grp < ngroups; ++grp) {
      // SPEC 1: group = kz plane; SPEC 2: group = taps [GT * grp, GT * grp + tpg) in (kz, ky, kx)
      // order (the loop is unrolled, so the tap coordinates are still compile-time)
      const int kz = SPEC >= 2 ? 0 : grp / ngy, ky0 = SPEC >= 2 ? 0 : (grp - kz * ngy) * GKH;
      const int gkh = (KH - ky0) < GKH ? (KH - ky0) : GKH;
      const int tpg = SPEC >= 2 ? ((27 - grp * GT) < GT ? (27 - grp * GT) : GT) : gkh * KW;
      const int tap0 = SPEC >= 2 ? grp * GT : (kz * KH + ky0) * KW;
      if (grp > 0 && !(ADELL_DBG(e.dbg) & 32)) __syncthreads();  // previous tap group consumed
      // ---- stage the weight slice of this tap group: [tpg][BN][4 slots] -----
      const bool skipB = (ADELL_DBG(e.dbg) & 2) && (ch > 0 || grp > 0);
      if (skipB) {
      } else if (SPEC) {
#pragma unroll
        for (int u = 0; u < WPF; ++u)
          if (u * TPU + wtap < tpg)
            *reinterpret_cast<float4*>(sB + wloff + u * (TPU * BN * 64)) = wreg[u];
      } else if (wpipe) {
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
          const int it = tid + u * NTHR;
          if (it < tpg * BN * 4) {
            const int slot = it & 3;
            const int n = (it >> 2) % BN;
            const int tl = (it >> 2) / BN;
            *reinterpret_cast<float4*>(sB + ((size_t)(tl * BN + n) * 4 + (slot ^ ((n >> 2) & 3))) * 16) =
                wreg[u];
          }
        }
      } else {
        for (int it = tid; it < tpg * BN * 4; it += NTHR) {
          const int slot = it & 3;
          const int n = (it >> 2) % BN;
          const int tl = (it >> 2) / BN;
          const int tap = tap0 + tl;
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n0 + n < a.Cout)
            f = *reinterpret_cast<const float4*>(
                reinterpret_cast<const char*>(e.wh) +
                (((size_t)tap * a.Cout + n0 + n) * nchunk + ch) * 64 + slot * 16);
          *reinterpret_cast<float4*>(sB + ((size_t)(tl * BN + n) * 4 + (slot ^ ((n >> 2) & 3))) * 16) = f;
        }
      }
      if (!(ADELL_DBG(e.dbg) & 32) || grp == 0) __syncthreads();   // (DBG 32: timing without the weight-group barriers)
      if (wpipe && !skipB) {
        // ONE fetch site: with two (next group / first group of the next chunk) the prefetch
        // registers meet in a phi, the copies behind it read them, and the wait for those copies
        // (vmcnt(0) right here) exposed the whole prefetch latency in front of every tap group
        const bool same_chunk = grp + 1 < ngroups;
        const int fch = same_chunk ? ch : ch + 1, fgrp = same_chunk ? grp + 1 : 0;
        if (fch < c_end) wfetch(fch, fgrp);
      }
      // ---- 3 f16 MFMAs per (tap, 32x32 tile) ---------------------------------------------
      const char* sAg = sA + (size_t)((kz * HY + ky0) * HX) * 64;
      auto load_frags = [&](int tl, half8* ah, half8* al, half8* bh, half8* bl) {
        // SPEC 2: absolute tap -> (kz, ky, kx); sAg is then the brick origin
        const int tabs = tap0 + tl;
        const int kzl = SPEC >= 2 ? tabs / 9 : 0;
        const int kyl = SPEC >= 2 ? (tabs - 9 * kzl) / 3 : tl / KW;
        const int kx = SPEC >= 2 ? tabs - 9 * kzl - 3 * kyl : tl - kyl * KW;
        const int aoff = (kzl * HY + kyl) * HX + kx;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int hv = arow[mt] + aoff;
          // SPEC: x of the halo voxel = (li & 7) + kx  (no carries: x + kx < 10)
          const int sw = SPEC ? (((li & 7) + kx) >> 1) & 3 : ((hv + (kz * HY + ky0) * HX) >> 2) & 3;
          const char* row = sAg + (size_t)hv * 64;
          ah[mt] = *reinterpret_cast<const half8*>(row + ((lh ^ sw) << 4));
          al[mt] = *reinterpret_cast<const half8*>(row + (((2 + lh) ^ sw) << 4));
        }
        const char* bt = sB + (size_t)tl * BN * 64;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = *reinterpret_cast<const half8*>(bt + boffh[nt]);
          bl[nt] = *reinterpret_cast<const half8*>(bt + boffl[nt]);
        }
      };
      auto do_mfma = [&](const half8* ah, const half8* al, const half8* bh, const half8* bl) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
          }
      };
      if (ADELL_DBG(e.dbg) & 8) continue;
      if constexpr (SPEC) {
        // 9 taps, statically indexed, two fragment sets: the LDS reads of tap t+1 are issued
        // before the MFMAs of tap t
        half8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
        load_frags(0, ah[0], al[0], bh[0], bl[0]);
#pragma unroll
        for (int tl = 0; tl < GT; ++tl) {
          if (tl < tpg) {
            if (tl + 1 < tpg)
              load_frags(tl + 1, ah[(tl + 1) & 1], al[(tl + 1) & 1], bh[(tl + 1) & 1], bl[(tl + 1) & 1]);
            do_mfma(ah[tl & 1], al[tl & 1], bh[tl & 1], bl[tl & 1]);
            // pin the interleave: one LDS read of the next tap behind each of the first MFMAs
            // (the 2x2 tile has no registers for the second fragment set: it spills when pinned)
            if (SPEC == 1 && MT * NT == 2 && tl + 1 < tpg) {
#pragma unroll
              for (int i = 0; i < 2 * MT + 2 * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              if constexpr (3 * MT * NT > 2 * MT + 2 * NT)
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * NT - (2 * MT + 2 * NT), 0);
            }
          }
        }
      } else if constexpr (MT * NT == 4) {
        // one fragment set: the MFMAs are asynchronous, the next tap's LDS reads follow them
        half8 ah0[MT], al0[MT], bh0[NT], bl0[NT];
        for (int tl = 0; tl < tpg; ++tl) {
          load_frags(tl, ah0, al0, bh0, bl0);
          do_mfma(ah0, al0, bh0, bl0);
        }
      } else {
        // the fragments of tap t+1 are read while the MFMAs of tap t run (two register sets)
        half8 ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
        load_frags(0, ah0, al0, bh0, bl0);
        for (int tl = 0; tl < tpg; tl += 2) {
          if (tl + 1 < tpg) load_frags(tl + 1, ah1, al1, bh1, bl1);
          do_mfma(ah0, al0, bh0, bl0);
          if (tl + 1 < tpg) {
            if (tl + 2 < tpg) load_frags(tl + 2, ah0, al0, bh0, bl0);
            do_mfma(ah1, al1, bh1, bl1);
          }
        }
      }
    }
  }

  ADELL_STAMP(20);
  // ---- epilogue (same contract as the fp32 kernel) ---------------------------
  // A row of the output is addressed as (64-bit block base, per lane and column tile) +
  // (32-bit offset of the row inside the brick); the brick spans < 2^31 elements (checked on
  // the host), so the per-row arithmetic is three 24-bit multiply-adds.
  const float ascale = __int_as_float((127 - kA_prev) << 23);
  const int fx = (shuffle & 1) + 1, fy = ((shuffle >> 1) & 1) + 1, fz = ((shuffle >> 2) & 1) + 1;
  // steps (in output rows) of one brick-local x / y / z step; destination grid when shuffling
  const int sX = fx, sY = fy * fx * a.Wo, sZ = fz * fy * a.Ho * fx * a.Wo;
  const size_t row0 =   // destination row of the brick origin
      ((size_t)((nb * fz * a.Do + fz * oz0) * (fy * a.Ho) + fy * oy0)) * (fx * a.Wo) + fx * ox0;
  const size_t rrow0 = ((size_t)((nb * a.Do + oz0) * a.Ho + oy0)) * a.Wo + ox0;  // residual row
  const bool res_like_y = (a.shuffle & 16) != 0;
  float s1[NT], s2[NT], bcol[NT], oscale[NT];
  float* colptr[NT];
  const float* resptr[NT];
  int rowmul[NT];
  bool nok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = s2[nt] = 0.f;
    const int n = n0 + (wn * NT + nt) * 32 + li;
    nok[nt] = n < a.Cout;
    bcol[nt] = 0.f;
    oscale[nt] = 0.f;
    colptr[nt] = a.y0;
    resptr[nt] = a.res;
    rowmul[nt] = 0;
    if (nok[nt]) {
      oscale[nt] = ascale * e.wscale[n];
      if (shuffle) {
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
      colptr[nt] += row0 * rowmul[nt];
      // residual: output-shaped rows, or (shuffle bit 4) laid out like the shuffled destination
      if (a.res)
        resptr[nt] = res_like_y ? a.res + (colptr[nt] - a.y0) : a.res + rrow0 * a.Cout + n;
    }
  }
  // EPI = 1: per 32-column sub-tile, the site behind its destination (a sub-tile never straddles
  // ysplit: the host takes the fused path only when ysplit is a multiple of 32)
  const float* adn_y[NT];
  const unsigned* adn_mk[NT];
  float adn_m[NT], adn_r[NT], adn_ks[NT], adn_ap[NT];
  int adn_act[NT];
  unsigned adn_e0[NT];
  if constexpr (EPI == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (wn * NT + nt) * 32 + li;
      const int dst = n < a.ysplit ? 0 : 1;
      const int cl = n - (dst ? a.ysplit : 0), cs = rowmul[nt];
      const ConvF16Extra::Adn& s = e.adn[dst];
      adn_y[nt] = nullptr;
      adn_mk[nt] = nullptr;
      adn_m[nt] = 0.f; adn_r[nt] = 1.f; adn_ks[nt] = 1.f; adn_ap[nt] = 0.f;
      adn_act[nt] = 0;
      adn_e0[nt] = 0;
      if (s.y != nullptr && nok[nt]) {
        adn_y[nt] = s.y + (colptr[nt] - (dst ? a.y1 : a.y0));
        adn_m[nt] = s.mean[(size_t)nb * cs + cl];
        adn_r[nt] = s.rstd[(size_t)nb * cs + cl];
        adn_ks[nt] = s.keep_scale;
        adn_ap[nt] = s.act_p;
        adn_act[nt] = s.act;
        // element index, inside the batch item, of this lane's column in the brick's origin voxel
        adn_e0[nt] = (unsigned)(row0 - (size_t)nb * a.Do * a.Ho * a.Wo) * (unsigned)cs + (unsigned)cl;
        if (s.mask != nullptr) adn_mk[nt] = s.mask + (size_t)nb * s.groups * 8;
      }
    }
  }
  bool stored = false;
  if constexpr (SPEC) {
    // interior brick, all columns valid: rows of an m-tile are 4 x-neighbours (r & 3) in 4
    // y-rows (r >> 2), so a row is the m-tile's base pointer + a compile-time multiple of two
    // per-lane strides; no bounds checks, the residual values of an m-tile are fetched at once
    const bool full = (ox0 + 8 <= a.Wo) & (oy0 + 8 <= a.Ho) & (oz0 + (SPEC == 3 ? 8 : 4) <= a.Do) &
                      (n0 + BN <= a.Cout) & !res_like_y;
    if (full) {
      stored = true;
      auto fast = [&](auto has_res) {
        // EPI = 1, one column tile per wave (the 32-column instances): the site inputs of m-tile
        // mt + 1 are fetched BEFORE the stores of m-tile mt (which may alias them as far as the
        // compiler knows), so each tile's memory round trip runs under the arithmetic of the
        // previous one. 16 more live registers (all the wave's tiles at once -- 64 -- lost).
        constexpr bool YPIPE = EPI == 1 && NT == 1;
        float ynext[16];
        auto fetch_y = [&](int mt_) {
          const int tile_ = wm * MT + mt_;
          const unsigned rb = (unsigned)(tile_ >> 1) * (unsigned)(a.Ho * a.Wo) +
                              (unsigned)((tile_ & 1) * 4 * a.Wo) + 4 * lh;
          const float* yp = (adn_y[0] != nullptr ? adn_y[0] : colptr[0]) + (size_t)rb * rowmul[0];
          const unsigned dY = a.Wo * rowmul[0], dX = rowmul[0];
#pragma unroll
          for (int r = 0; r < 16; ++r) ynext[r] = yp[(r >> 2) * dY + (r & 3) * dX];
        };
        if constexpr (YPIPE) fetch_y(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int tile = wm * MT + mt;
          const unsigned rbase =
              (unsigned)(tile >> 1) * (unsigned)(a.Ho * a.Wo) + (unsigned)((tile & 1) * 4 * a.Wo) +
              4 * lh;
          float ycur[16];
          if constexpr (YPIPE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ycur[r] = ynext[r];
            if (mt + 1 < MT) fetch_y(mt + 1);
          }
          float resv[NT][16];
          if constexpr (has_res.value) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float* rp = resptr[nt] + (size_t)rbase * a.Cout;
              const unsigned dY = a.Wo * a.Cout, dX = a.Cout;
#pragma unroll
              for (int r = 0; r < 16; ++r) resv[nt][r] = rp[(r >> 2) * dY + (r & 3) * dX];
            }
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float* p = colptr[nt] + (size_t)rbase * rowmul[nt];
            const unsigned dY = a.Wo * rowmul[nt], dX = rowmul[nt];
            if constexpr (EPI == 1) {
              if (adn_y[nt] != nullptr) {
                // ---- dt and its two sums instead of dout (see ConvF16Extra::Adn) --------------
                float yv[16];
                if constexpr (YPIPE) {
#pragma unroll
                  for (int r = 0; r < 16; ++r) yv[r] = ycur[r];
                } else {
                  const float* yp = adn_y[nt] + (size_t)rbase * rowmul[nt];
#pragma unroll
                  for (int r = 0; r < 16; ++r) yv[r] = yp[(r >> 2) * dY + (r & 3) * dX];
                }
                unsigned kw[16];
                const unsigned ebase = adn_e0[nt] + rbase * (unsigned)rowmul[nt];
                if (adn_mk[nt] != nullptr) {
                  // element el of the item -> bit (el >> 2) & 63 of 64-bit word (el >> 8) * 4 + (el & 3)
                  if (rowmul[nt] == 32) {
                    // 32 channels: the four x-neighbours of a brick row share one 32-bit word
                    // (el >> 7 = voxel >> 2, and x = 4 lh + (r & 3) with the brick 8-aligned)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      const unsigned el = ebase + j * dY;
                      const unsigned w = adn_mk[nt][((el >> 8) * 4u + (el & 3u)) * 2u + ((el >> 7) & 1u)];
                      kw[4 * j] = kw[4 * j + 1] = kw[4 * j + 2] = kw[4 * j + 3] = w;
                    }
                  } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                      const unsigned el = ebase + (r >> 2) * dY + (r & 3) * dX;
                      kw[r] = adn_mk[nt][((el >> 8) * 4u + (el & 3u)) * 2u + ((el >> 7) & 1u)];
                    }
                  }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  float v = acc[mt][nt][r] * oscale[nt] + bcol[nt];
                  if constexpr (has_res.value) v += resv[nt][r];
                  const float hn = (yv[r] - adn_m[nt]) * adn_r[nt];
                  bool keep = true;
                  if (adn_mk[nt] != nullptr) {
                    const unsigned el = ebase + (r >> 2) * dY + (r & 3) * dX;
                    keep = (kw[r] >> ((el >> 2) & 31u)) & 1u;
                  }
                  const float u = keep ? hn * adn_ks[nt] : 0.f;
                  float g = 1.f;
                  if (adn_act[nt] == ADELL_ACT_SILU) {
                    const float sg = adell_sigmoidf(u);
                    g = sg * (1.0f + u * (1.0f - sg));
                  } else if (adn_act[nt] == ADELL_ACT_RELU) {
                    g = u > 0.f ? 1.f : 0.f;
                  } else if (adn_act[nt] == ADELL_ACT_LEAKY_RELU) {
                    g = u > 0.f ? 1.f : adn_ap[nt];
                  }
                  const float du = v * g;
                  const float dt = keep ? du * adn_ks[nt] : 0.f;
                  p[(r >> 2) * dY + (r & 3) * dX] = dt;
                  s1[nt] += dt;
                  s2[nt] += dt * hn;
                }
                continue;
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[mt][nt][r] * oscale[nt] + bcol[nt];
              if constexpr (has_res.value) v += resv[nt][r];
              if (!(ADELL_DBG(e.dbg) & 16)) p[(r >> 2) * dY + (r & 3) * dX] = v;
              s1[nt] += v;
              s2[nt] += v * v;
            }
          }
        }
      };
      if (a.res)
        fast(std::true_type{});
      else
        fast(std::false_type{});
    }
  }
  if (!stored) {
  // the residual values of an m-tile are fetched together (16 loads in flight per column tile)
  // before its rows are stored; wide column tiles (NT > 2) read them row by row instead
  constexpr bool RES_BATCH = NT <= 2;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float resv[RES_BATCH ? NT : 1][16];
    if (RES_BATCH && a.res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = (wm * MT + mt) * 32 + row;
        const int xl = m & ((1 << lTX) - 1);
        const int yl = (m >> lTX) & ((1 << lTY) - 1);
        const int zl = m >> (lTX + lTY);
        const bool rok = (ox0 + xl < a.Wo) & (oy0 + yl < a.Ho) & (oz0 + zl < a.Do);
        const unsigned loc = (unsigned)zl * (unsigned)sZ + __umul24(yl, sY) + __umul24(xl, sX);
        const unsigned rloc = (unsigned)zl * (unsigned)(a.Ho * a.Wo) + __umul24(yl, a.Wo) + xl;
#pragma unroll
        for (int nt = 0; nt < (RES_BATCH ? NT : 1); ++nt)
          resv[nt][r] = (rok && nok[nt])
                            ? resptr[nt][res_like_y ? loc * (unsigned)rowmul[nt]
                                                    : rloc * (unsigned)a.Cout]
                            : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = (wm * MT + mt) * 32 + row;
      const int xl = m & ((1 << lTX) - 1);
      const int yl = (m >> lTX) & ((1 << lTY) - 1);
      const int zl = m >> (lTX + lTY);
      const bool rok = (ox0 + xl < a.Wo) & (oy0 + yl < a.Ho) & (oz0 + zl < a.Do);
      const unsigned loc = (unsigned)zl * (unsigned)sZ + __umul24(yl, sY) + __umul24(xl, sX);
      const unsigned rloc = (unsigned)zl * (unsigned)(a.Ho * a.Wo) + __umul24(yl, a.Wo) + xl;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (rok && nok[nt]) {
          float v = acc[mt][nt][r] * oscale[nt] + bcol[nt];
          if (a.res) {
            if constexpr (RES_BATCH)
              v += resv[nt][r];
            else
              v += resptr[nt][res_like_y ? loc * (unsigned)rowmul[nt] : rloc * (unsigned)a.Cout];
          }
          colptr[nt][loc * (unsigned)rowmul[nt]] = v;
          s1[nt] += v;
          s2[nt] += v * v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
  ADELL_STAMP(21);
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
        float* p = a.part + (((size_t)nb * nsp + tile_id) * a.Cout + n) * 2;
        p[0] = t1;
        p[1] = t2;
      }
    }
  }
  ADELL_STAMP(22);
}

#ifndef ADELL_NO_PACK_KERNELS   // (non-template kernels: defined once, in conv3d.hip's unit)
// ---------------------------------------------------------------------------
// Weight packing for the f16x3 kernel. One block per GEMM column n (an output
// channel): absmax over that column's taps x K -> power-of-two scale -> hi/lo split.
// mode 0: conv [Cout=A][Cin=B][taps] -> column n = cout, k = cin, tap order kept
// mode 1: same source -> column n = cin, k = cout, taps flipped (backward-data)
// ---------------------------------------------------------------------------
// (the column's taps x K source values are read ONCE, tap-fastest -- runs of `taps` contiguous floats,
// the canonical layout's innermost axis -- into LDS when they fit, and both passes (absmax, split)
// index the LDS copy; reading them k-fastest from global memory touched every 128-byte line ~27
// times: 323 MB fetched per step to repack 33 MB of weights)
// (27 taps x 512 channels: the 512-channel layers of the ResNet-backbone U-Net fell off the staged
// path at 8192 floats and their repack took 0.98 ms of every config-2b step)
constexpr int kPackLds = 14336;  // floats (56 KB): 27 taps x 512 channels, 125 x 64 and 343 x 16 fit
__device__ __forceinline__ void adell_pack_weight_f16_column(
    const float* __restrict__ w, _Float16* __restrict__ out, float* __restrict__ wscale,
    int mode, int A, int B, int taps, int n, float* smx) {
  __shared__ float scol[kPackLds];
  const int N = mode == 0 ? A : B;   // GEMM columns
  const int K = mode == 0 ? B : A;   // GEMM depth
  const int nchunk = (K + 15) / 16;
  auto src = [&](int tap, int k) -> long {
    return mode == 0 ? ((long)n * B + k) * taps + tap
                     : ((long)k * B + n) * taps + (taps - 1 - tap);
  };
  const bool staged = taps * K <= kPackLds;
  float mx = 0.f;
  if (staged) {
    // scol[k * taps + t] = w[...][t] (t = source tap index, before the flip of mode 1)
    for (int i = threadIdx.x; i < taps * K; i += 256) {
      const int k = i / taps, t = i - k * taps;
      const float v = w[mode == 0 ? ((long)n * B + k) * taps + t : ((long)k * B + n) * taps + t];
      scol[i] = v;
      mx = fmaxf(mx, fabsf(v));
    }
  } else {
    for (int i = threadIdx.x; i < taps * K; i += 256) mx = fmaxf(mx, fabsf(w[src(i / K, i % K)]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  int kw = 0;
  const int ebits = (__float_as_int(mx) >> 23) & 0xff;
  if (ebits > 0 && ebits < 255) kw = 13 - (ebits - 127);
  if (kw > 100) kw = 100;
  if (kw < -100) kw = -100;
  const float scale = __int_as_float((kw + 127) << 23);
  if (threadIdx.x == 0) wscale[n] = __int_as_float((127 - kw) << 23);
  const int total = taps * nchunk * 16;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int j = i & 15;
    const int ch = (i >> 4) % nchunk;
    const int tap = (i >> 4) / nchunk;
    const int k = ch * 16 + j;
    float v = 0.f;
    if (k < K) v = staged ? scol[k * taps + (mode == 0 ? tap : taps - 1 - tap)] : w[src(tap, k)];
    const float tsc = v * scale;
    const _Float16 h = (_Float16)tsc;
    _Float16* o = out + (((long)tap * N + n) * nchunk + ch) * 32;
    o[j] = h;
    o[16 + j] = (_Float16)(tsc - (float)h);
  }
}

__global__ __launch_bounds__(256) void adell_pack_weight_f16_kernel(
    const float* __restrict__ w, _Float16* __restrict__ out, float* __restrict__ wscale,
    int mode, int A, int B, int taps) {
  __shared__ float smx[4];
  adell_pack_weight_f16_column(w, out, wscale, mode, A, B, taps, blockIdx.x, smx);
}

// Every weight of a network in ONE launch (the per-weight launches are ~6 us each, 76 per
// training step). table[e] = {w, out, wscale (pointers), mode, A, B, taps, first block}: block b
// packs column b - first_block(e) of the entry e it falls into (binary search).
__global__ __launch_bounds__(256) void adell_pack_weight_f16_multi_kernel(
    const long* __restrict__ table, int entries) {
  __shared__ float smx[4];
  int lo = 0, hi = entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(size_t)mid * 8 + 7] <= (long)blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const long* t = table + (size_t)lo * 8;
  adell_pack_weight_f16_column(reinterpret_cast<const float*>(t[0]),
                               reinterpret_cast<_Float16*>(t[1]), reinterpret_cast<float*>(t[2]),
                               (int)t[3], (int)t[4], (int)t[5], (int)t[6],
                               (int)((long)blockIdx.x - t[7]), smx);
}
#endif  // ADELL_NO_PACK_KERNELS
