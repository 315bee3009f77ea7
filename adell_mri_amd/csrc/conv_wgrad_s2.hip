// Backward-weight of the U-Net downsampling convolution (3x3x3, stride 2, padding 1, 32 -> 32
// channels: unet.py:571-579 at the two high-resolution levels), f16x3 arithmetic -- the third
// kernel of the layer after csrc/conv_fwd_s2.hip and csrc/conv_dgrad_s2.hip.
//
//   dW[co][ci][t] = sum_{n, o} x[n, 2 o - 1 + t][ci] * dY[n, o][co]
//
// GEMM view of the other weight-gradient kernels (M = ci, N = co, K = voxels; fragments by the
// transposing LDS read over natural [voxel][32 channels] fp16 hi / lo planes). The decomposition
// is the forward kernel's: per axis the input index 2 o - 1 + t has parity 0 for t = 1 and parity 1
// for t in {0, 2}, so the eight parity sub-lattices of x pair with 1, 2, 4 or 8 taps each, and the
// part of ONE sub-lattice behind an 8 x 8 x 2 brick of dY is a 9 x 9 x 3 halo. A block (4 waves,
// two per CU so that one stages while the other multiplies) keeps the brick of dY (16 KB) in LDS,
// stages the eight sub-lattice halos of x in turn (31 KB, the next one in flight in registers) and
// accumulates all 27 taps in registers for its whole life: wave w owns 7 (6) of the 27 taps, dealt
// so that every sub-lattice keeps the four waves as evenly busy as its tap count allows. Every
// block writes ONE partial slab at the end, folded in fixed order by the slab fold of
// conv_wgrad_f16.hip. Operand scales: one power of
// two per tensor from the absmax words (by-products of the forward / backward-data kernels).
// Roofline: HBM (x 537 MB + dY 67 MB read once at 2 x 128^3).
#include <type_traits>
#include <utility>
#include "common.h"

typedef _Float16 ws2_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws2_half4 __attribute__((ext_vector_type(4)));
typedef __fp16 ws2_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

namespace {

constexpr int kBZ = 2;                                             // brick = 8 x 8 x kBZ voxels of dY
constexpr int kHX = 9, kHY = 9, kHZ = kBZ + 1, kHV = kHX * kHY * kHZ;   // sub-lattice halo of a brick
constexpr int kXPlane = kHV * 64;                                  // one fp16 plane of the halo
constexpr int kYRows = 64 * kBZ;
constexpr int kYPlane = kYRows * 64;                               // ... of the dY brick
constexpr int kLds = 2 * kXPlane + 2 * kYPlane;
constexpr int kThreads = 256;
constexpr int kXItems = kHV * 8, kXPer = (kXItems + kThreads - 1) / kThreads;   // float4 pieces per thread (8)
constexpr int kYPer = kYRows * 8 / kThreads;                       // 4

template <typename T>
__device__ __forceinline__ ADELL_GLOBAL T* uniform_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<ADELL_GLOBAL T*>(((uint64_t)hi << 32) | lo);
}

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& body, std::integer_sequence<int, I...>) {
  (body(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& body) {
  static_for_impl<N>(body, std::make_integer_sequence<int, N>{});
}

// two transposing reads: 16 voxels (rows p, p + 4 rows) x this lane's channel -> an MFMA fragment
__device__ __forceinline__ ws2_half8 tr_frag(const char* p) {
  typedef __attribute__((address_space(3))) ws2_fp16x4* lds_p;
  const ws2_fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p));
  const ws2_fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p + 4 * 64));
  ws2_half8 r;
  r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
  r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
  return r;
}

__device__ __forceinline__ void split_store(char* hi_plane, char* lo_plane, unsigned off, float4 f,
                                            float scale) {
  ws2_half4 h, l;
  const float t[4] = {f.x * scale, f.y * scale, f.z * scale, f.w * scale};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)t[j];
    l[j] = (_Float16)(t[j] - (float)h[j]);
  }
  *reinterpret_cast<ws2_half4*>(hi_plane + off) = h;
  *reinterpret_cast<ws2_half4*>(lo_plane + off) = l;
}

__device__ __forceinline__ int scale_exp(unsigned maxbits) {
  const int ebits = (int)((maxbits >> 23) & 0xff);
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 8 * ((13 - (ebits - 127)) >> 3);
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

// ---- the 27 taps by sub-lattice: class c = (pz, py, px); its j-th tap has, per axis, bit b of j's
// compressed index: parity 0 -> tap 1 at halo offset 1; parity 1 -> tap 0 at offset 0 (b = 0) or tap 2
// at offset 1 (b = 1) (halo origin = o0 - 1, see conv_fwd_s2.hip) -----------------------------
constexpr int cls_ntaps(int c) { return (1 << ((c >> 2) & 1)) * (1 << ((c >> 1) & 1)) * (1 << (c & 1)); }
// expand the j-th tap of class c into per-axis second-tap bits (sz, sy, sx)
constexpr int cls_bits(int c, int j) {
  int bits = 0, k = j;
  if (c & 1) { bits |= (k & 1); k >>= 1; }          // x
  if (c & 2) { bits |= (k & 1) << 1; k >>= 1; }     // y
  if (c & 4) { bits |= (k & 1) << 2; }              // z
  return bits;
}
constexpr int axis_tap(int p, int b) { return p ? 2 * b : 1; }
constexpr int axis_off(int p, int b) { return p ? b : 1; }
constexpr int cls_tap(int c, int j) {
  const int b = cls_bits(c, j);
  return (axis_tap((c >> 2) & 1, (b >> 2) & 1) * 3 + axis_tap((c >> 1) & 1, (b >> 1) & 1)) * 3 +
         axis_tap(c & 1, b & 1);
}
constexpr int cls_hoff(int c, int j) {   // halo row offset of the tap
  const int b = cls_bits(c, j);
  return (axis_off((c >> 2) & 1, (b >> 2) & 1) * kHY + axis_off((c >> 1) & 1, (b >> 1) & 1)) * kHX +
         axis_off(c & 1, b & 1);
}
// owner lane-set of the j-th tap of class c: round robin from a per-class start, chosen so that the
// totals come out 7 / 7 / 7 / 6
constexpr int cls_rot(int c) { return (c == 0 || c == 2) ? 2 : 0; }
constexpr int tap_owner(int c, int j) { return (j + cls_rot(c)) & 3; }
// accumulator slot of that tap inside its owner: taps of the same owner in earlier (class, j) order
constexpr int tap_slot(int c, int j) {
  int n = 0;
  for (int cc = 0; cc < 8; ++cc)
    for (int jj = 0; jj < cls_ntaps(cc); ++jj) {
      if (cc == c && jj == j) return n;
      if (tap_owner(cc, jj) == tap_owner(c, j)) ++n;
    }
  return n;
}
static_assert(tap_slot(7, 7) <= 6 && tap_slot(7, 4) <= 6 && tap_slot(7, 5) <= 6 && tap_slot(7, 6) <= 6,
              "seven accumulators per wave");

}  // namespace

struct WgradS2Args {
  const float* x;        // [N][D][H][W][32]
  const float* dy;       // [N][D/2][H/2][W/2][32]
  float* ws;             // [R][27][32][32]
  float* wsdb;           // [R][32] or null
  const unsigned* xmax;  // device absmax (float bits) of x and dY
  const unsigned* ymax;
  int N, D, H, W, Do, Ho, Wo;
  int ntx, nty, ntz, nbricks;
};

// DBG: timing experiments (-DADELL_DEBUG builds only; results are wrong when nonzero): 1 no MFMAs,
// 8 no split / LDS stores, 16 no loads after the first phase
template <int DBG>
__global__ __launch_bounds__(kThreads, 2) void adell_conv_wgrad_s2_kernel(WgradS2Args a) {
  extern __shared__ char smem[];
  char* sXh = smem;
  char* sXl = sXh + kXPlane;
  char* sYh = sXl + kXPlane;
  char* sYl = sYh + kYPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w4 = wave, lh = lane >> 5;
  const int nsp = a.ntx * a.nty * a.ntz;
  const int kX = scale_exp(a.xmax[0]), kY = scale_exp(a.ymax[0]);
  const float sX = __int_as_float((kX + 127) << 23), sY = __int_as_float((kY + 127) << 23);

  // transposed-read lane roles (conv_wgrad_zring.hip): lane 4 q + p of a 16-lane group addresses
  // voxel row q, channels 4 p .. 4 p + 3 of channel half cg; the lane halves take two brick rows
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const int colb = (16 * cg + 4 * tp) * 2;
  const int abase = (lh * kHX + tq) * 64 + colb;   // + plane / row pair / tap offsets (halo rows)
  const int bbase = (lh * 8 + tq) * 64 + colb;     // + plane / row pair (brick rows)

  // seven named accumulators, selected at compile time (an indexed array of them ends up in
  // scratch memory: the compiler sinks the identical MFMA chains of the tap branches into one block
  // with the slot as a phi, and 448 bytes of dynamically indexed private memory are not promoted)
  f32x16 acc0, acc1, acc2, acc3, acc4, acc5, acc6;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = acc4[r] = acc5[r] = acc6[r] = 0.f;
  auto accref = [&](auto S) -> f32x16& {
    constexpr int s = decltype(S)::value;
    if constexpr (s == 0) return acc0;
    else if constexpr (s == 1) return acc1;
    else if constexpr (s == 2) return acc2;
    else if constexpr (s == 3) return acc3;
    else if constexpr (s == 4) return acc4;
    else if constexpr (s == 5) return acc5;
    else return acc6;
  };

  const int c4 = tid & 7;
  auto brick_origin = [&](int t, int& nb, int& ox0, int& oy0, int& oz0) {
    const int tile = t % nsp;
    nb = t / nsp;
    int r = tile;
    const int tx = r % a.ntx;
    r /= a.ntx;
    const int ty = r % a.nty;
    const int tz = r / a.nty;
    ox0 = tx * 8;
    oy0 = ty * 8;
    oz0 = tz * kBZ;
  };
  float4 fx[kXPer], fy[kYPer];
  unsigned okx = 0, oky = 0;
  // loads of phase (brick, sub-lattice): the halo of x; with sub-lattice 0 also the brick of dY
  auto prefetch = [&](int phase) {
    const int t = blockIdx.x + (phase >> 3) * gridDim.x, cls = phase & 7;
    int nb, ox0, oy0, oz0;
    brick_origin(t, nb, ox0, oy0, oz0);
    const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
    const int bz = 2 * (oz0 - 1) + pz, by = 2 * (oy0 - 1) + py, bx = 2 * (ox0 - 1) + px;
    const ADELL_GLOBAL float* src = uniform_ptr(a.x + (size_t)nb * a.D * a.H * a.W * 32);
    int tt = tid;
    asm volatile("" : "+v"(tt));
    okx = 0;
#pragma unroll
    for (int u = 0; u < kXPer; ++u) {
      const int it = tt + kThreads * u;
      const int hv = it >> 3;
      const int hz = hv / (kHX * kHY), rem = hv - hz * (kHX * kHY);
      const int hy = rem / kHX, hx = rem - hy * kHX;
      const int iz = bz + 2 * hz, iy = by + 2 * hy, ix = bx + 2 * hx;
      const bool ok = (it < kXItems) & (iz >= 0) & (iz < a.D) & (iy >= 0) & (iy < a.H) & (ix >= 0) &
                      (ix < a.W);
      const unsigned rel = ok ? (unsigned)((iz * a.H + iy) * a.W + ix) * 32u + 4u * c4 : 0u;
      const f32x4 v = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(src + rel);
      fx[u] = make_float4(v.x, v.y, v.z, v.w);
      okx |= ok ? (1u << u) : 0u;
    }
    if (cls == 0) {
      const ADELL_GLOBAL float* ysrc = uniform_ptr(a.dy + (size_t)nb * a.Do * a.Ho * a.Wo * 32);
      oky = 0;
#pragma unroll
      for (int u = 0; u < kYPer; ++u) {
        const int it = tt + kThreads * u;
        const int v = it >> 3;                       // brick voxel: x + 8 (y + 8 z)
        const int oz = oz0 + (v >> 6), oy = oy0 + ((v >> 3) & 7), ox = ox0 + (v & 7);
        const bool ok = (oz < a.Do) & (oy < a.Ho) & (ox < a.Wo);
        const unsigned rel = ok ? (unsigned)((oz * a.Ho + oy) * a.Wo + ox) * 32u + 4u * c4 : 0u;
        const f32x4 q = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(ysrc + rel);
        fy[u] = make_float4(q.x, q.y, q.z, q.w);
        oky |= ok ? (1u << u) : 0u;
      }
    }
  };

  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int my_bricks = ((int)blockIdx.x < a.nbricks)
                            ? (a.nbricks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nphases = my_bricks * 8;
  // iteration ph: issue the loads of phase ph (one fetch site), run the MFMAs of phase ph - 1 out
  // of LDS, then move phase ph from registers to LDS
  for (int ph = 0; ph <= nphases; ++ph) {
    if (ph < nphases && !((DBG & 16) && ph > 0)) prefetch(ph);
    __builtin_amdgcn_sched_barrier(0);
    if (ph > 0 && !(DBG & 1)) {
      const int cls = (ph - 1) & 7;
      static_for<8>([&](auto CLS) {
        constexpr int c = decltype(CLS)::value;
        if (cls == c) {
          static_for<cls_ntaps(c)>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int slot = tap_slot(c, j), hoff = cls_hoff(c, j);
            if (w4 == tap_owner(c, j)) {
              // k-steps of 16 voxels: the planes of the brick, row pairs 0 .. 3
#pragma unroll
              for (int s = 0; s < 4 * kBZ; ++s) {
                const int z = s >> 2, yp = 2 * (s & 3);
                const int xo = abase + ((z * kHY + yp) * kHX + hoff) * 64;
                const int yo = bbase + ((z * 8 + yp) * 8) * 64;
                const ws2_half8 ah = tr_frag(sXh + xo), al = tr_frag(sXl + xo);
                const ws2_half8 bh = tr_frag(sYh + yo), bl = tr_frag(sYl + yo);
                f32x16& ac = accref(std::integral_constant<int, slot>{});
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ac, 0, 0, 0);
              }
            }
          });
        }
      });
    }
    if (ph < nphases) {
      __syncthreads();   // the fragments of phase ph - 1 are read
#pragma unroll
      for (int u = 0; u < kXPer; ++u) {
        const int it = tid + kThreads * u;
        if (it < kXItems && !(DBG & 8)) {
          const bool ok = (okx >> u) & 1u;
          const float4 f = make_float4(ok ? fx[u].x : 0.f, ok ? fx[u].y : 0.f, ok ? fx[u].z : 0.f,
                                       ok ? fx[u].w : 0.f);
          split_store(sXh, sXl, (unsigned)((it >> 3) * 64 + c4 * 8), f, sX);
        }
      }
      if ((ph & 7) == 0) {
#pragma unroll
        for (int u = 0; u < kYPer; ++u) {
          const int it = tid + kThreads * u;
          const bool ok = (oky >> u) & 1u;
          const float4 f = make_float4(ok ? fy[u].x : 0.f, ok ? fy[u].y : 0.f, ok ? fy[u].z : 0.f,
                                       ok ? fy[u].w : 0.f);
          dbacc.x += f.x; dbacc.y += f.y; dbacc.z += f.z; dbacc.w += f.w;
          split_store(sYh, sYl, (unsigned)((it >> 3) * 64 + c4 * 8), f, sY);
        }
      }
      __syncthreads();
    }
  }

  // ---- bias gradient of this block's bricks ------------------------------------------------
  const int region = blockIdx.x;
  if (a.wsdb) {
    __syncthreads();
    // threads with the same c4 hold the same four channels: fold the 64 of them in fixed order
    float4* red = reinterpret_cast<float4*>(smem);   // one float4 per thread (the halo image is free)
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 8) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = tid; k < kThreads; k += 8) {
        const float4 u = red[k];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float* o0 = a.wsdb + (size_t)region * 32 + 4 * tid;
      o0[0] = t.x; o0[1] = t.y; o0[2] = t.z; o0[3] = t.w;
    }
  }
  // ---- partial slab of (block, group): C row = ci, column = co (undo the operand scales) -------
  const float unscale = __int_as_float((127 - kX - kY) << 23);
  const int co = lane & 31;
  static_for<8>([&](auto CLS) {
    constexpr int c = decltype(CLS)::value;
    static_for<cls_ntaps(c)>([&](auto J) {
      constexpr int j = decltype(J)::value;
      constexpr int slot = tap_slot(c, j), tap = cls_tap(c, j);
      if (w4 == tap_owner(c, j)) {
        float* base = a.ws + (((size_t)region * 27 + tap) * 32 + 4 * lh) * 32 + co;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          base[(size_t)((r & 3) + 8 * (r >> 2)) * 32] = accref(std::integral_constant<int, slot>{})[r] * unscale;
      }
    });
  });
}

// ---- host side (called from conv_wgrad_f16.hip) ------------------------------------------------
struct WgradS2Plan {
  int ntx, nty, ntz, nbricks, blocks, R;
};

static int wgrad_s2_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// 1 + plan when the kernel takes the problem: 32 -> 32 channels, k = 3, stride 2, padding 1, even
// input dims, one source
extern "C" int adell_wgrad_s2_plan(int N, int D, int H, int W, int C0, int C1, int Cout, int KD,
                                   int KH, int KW, int SD, int SH, int SW, int PD, int PH, int PW,
                                   int Do, int Ho, int Wo, WgradS2Plan* p) {
  if (C1 != 0 || C0 != 32 || Cout != 32 || KD != 3 || KH != 3 || KW != 3 || SD != 2 || SH != 2 ||
      SW != 2 || PD != 1 || PH != 1 || PW != 1)
    return 0;
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || (D | H | W) & 1 || Do != D / 2 || Ho != H / 2 || Wo != W / 2)
    return 0;
  if ((size_t)D * H * W * 32 >= ((size_t)1 << 29)) return 0;   // 32-bit offsets inside an item
  if (g_adell_tune.wgrad_nozring) return 0;                     // the A/B switch of the z-ring kernel
  p->ntx = adell_cdiv(Wo, 8);
  p->nty = adell_cdiv(Ho, 8);
  p->ntz = adell_cdiv(Do, kBZ);
  const long nb = (long)N * p->ntx * p->nty * p->ntz;
  if (nb >= 0x0fffffffL) return 0;
  // every (block, group) writes a 108 KB slab: below a few bricks per block the slab fold costs more
  // than the kernel saves (2 x 64^3: 0.087 ms here against 0.077 on the generic kernel)
  if (nb < 8L * wgrad_s2_cus()) return 0;
  p->nbricks = (int)nb;
  const long want = 2L * wgrad_s2_cus();     // two resident blocks per CU
  p->blocks = (int)(nb < want ? nb : want);
  p->R = p->blocks;
  return 1;
}

extern "C" int adell_wgrad_s2_launch(const WgradS2Plan* p, int N, int D, int H, int W, const float* x,
                                     int Do, int Ho, int Wo, const float* dy, float* slabs,
                                     float* wsdb, const unsigned* xmax, const unsigned* ymax,
                                     hipStream_t st) {
  WgradS2Args a = {};
  a.x = x; a.dy = dy; a.ws = slabs; a.wsdb = wsdb; a.xmax = xmax; a.ymax = ymax;
  a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.ntx = p->ntx; a.nty = p->nty; a.ntz = p->ntz; a.nbricks = p->nbricks;
  auto launch = [&](auto kern) -> int {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p->blocks), dim3(kThreads), kLds, st, a);
    return ADELL_OK;
  };
  int rc = ADELL_OK;
#ifdef ADELL_DEBUG
  switch (g_adell_tune.zr_dbg) {
    case 1: rc = launch(adell_conv_wgrad_s2_kernel<1>); break;
    case 8: rc = launch(adell_conv_wgrad_s2_kernel<8>); break;
    case 16: rc = launch(adell_conv_wgrad_s2_kernel<16>); break;
    case 25: rc = launch(adell_conv_wgrad_s2_kernel<25>); break;
    default: rc = launch(adell_conv_wgrad_s2_kernel<0>); break;
  }
#else
  rc = launch(adell_conv_wgrad_s2_kernel<0>);
#endif
  if (rc != ADELL_OK) return rc;
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
