// Forward of the U-Net downsampling convolution (3x3x3, stride 2, padding 1, 32 -> 32 channels:
// unet.py:571-579 at the two high-resolution levels) in ONE persistent launch, f16x3 arithmetic --
// the companion of csrc/conv_dgrad_s2.hip.
//
//   y[o][co] = b[co] + sum_t x[2 o - 1 + t][.] w[co][.][t]
//
// Per axis the input index 2 o - 1 + t has parity 0 for t = 1 (sub-lattice index o) and parity 1
// for t = 0 (index o - 1) and t = 2 (index o). So the eight parity sub-lattices of x contribute
// through 1, 2, 4 or 8 taps each (27 in all), and the part of ONE sub-lattice behind an 8 x 8 x 4
// output brick is a 9 x 9 x 5 halo of 51 KB -- where the halo of the whole brick (17 x 17 x 9
// voxels, 333 KB) fits no LDS, which is why the implicit-GEMM kernel runs this layer on 64-voxel
// bricks at one wave per SIMD (0.56 ms at 2 x 128^3 for 0.1 ms of MFMA work and 0.13 ms of HBM).
// Here a persistent block (one per CU) of FOUR CONSUMER and FOUR PRODUCER waves (one of each per
// SIMD, so the MFMA pipe and the vector ALU / memory pipes of a SIMD are fed by different waves --
// the lock-step form of rounds 2-4, every wave fetching, splitting and multiplying in turn, cost
// the SUM of its phases: loads 67 + split 37 + y stores 38 + MFMAs 25 + skeleton 50 us of 231 us
// at 2 x 128^3, tools/fwd_s2_dbg.py):
//   * keeps the whole split weight (27 x 32 x 32 x (hi, lo) = 108 KB) in LDS for its lifetime,
//   * walks the eight sub-lattices of a brick in two half-phases each (channels 0-15, 16-31): the
//     producers split one 26 KB half image into LDS while the consumers run the taps of the OTHER
//     half image into 2 x 16 accumulator registers per wave (a wave owns one z plane of the brick:
//     two 32-voxel m-tiles sharing their B fragments); one s_barrier per half-phase, no other
//     synchronisation; the fp32 halos of the next TWO sub-lattices are in flight in the producers'
//     registers (a sub-lattice voxel is a full 128-byte line of x),
//   * takes the operand scale per (brick, sub-lattice) from the producers' absmax of the halo,
//     published one barrier ahead, and rescales the accumulators by the exact power of two when it
//     changes,
//   * emits bias, the per-channel (sum, sum of squares) partials of the following norm and the
//     absmax of x (for the weight-gradient kernel) like the implicit-GEMM epilogue -- from the
//     consumers, while the producers are already staging the next brick,
//   * bricks are dealt in contiguous ranges per XCD (blockIdx & 7) so that neighbouring halos meet in
//     one L2.
// Roofline: HBM (x read once: 537 MB at 2 x 128^3, y 67 MB).
#include <type_traits>
#include <utility>
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kHX = 9, kHY = 9, kHZ = 5, kHV = kHX * kHY * kHZ;
constexpr int kWBytes = 27 * 2 * 32 * 64;
constexpr int kImg = kHV * 64;                               // one 16-channel half image
constexpr int kCW = 4, kPW = 4;                              // consumer / producer waves
constexpr int kThreads = (kCW + kPW) * 64, kPThreads = kPW * 64;
constexpr int kRedFloats = kCW * 32 * 2;                     // statistics fold: [plane][channel][2]
constexpr int kLds = kWBytes + 2 * kImg + 64 + kRedFloats * 4;
static_assert(kLds <= 160 * 1024, "one block per CU: weights + two half images");
constexpr int kItems = kHV * 4;                              // 16-byte pieces of a half image
constexpr int kPer = (kItems + kPThreads - 1) / kPThreads;   // 7 per producer thread

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

struct FwdS2Args {
  const float* x;       // [N][D][H][W][32]
  const char* wpack;    // adell_pack_weight_f16x3 mode 0 of the full weight: [tap][co][chunk][64 B]
  const float* wscale;  // [32]
  const float* bias;    // [32] or null
  float* y;             // [N][D/2][H/2][W/2][32]
  float* part;          // [N][bricks per item][32][2] or null
  unsigned* amax_out;   // optional: absmax of x (float bits)
  int N, D, H, W, Do, Ho, Wo;
  int ntx, nty, ntz;
  int nbricks;
};

}  // namespace

// DBG: timing experiments (-DADELL_DEBUG builds only; results are wrong when nonzero): 1 no MFMAs,
// 2 no y stores / statistics, 8 no halo split / LDS stores, 16 no halo loads after the first phases
#define ADELL_S2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// power-of-two operand scale of a halo whose absmax is mx: multiples of 8 (the maximum lands in
// [2^6, 2^14)), so that it rarely changes inside a brick
__device__ __forceinline__ int fwd_s2_scale_exp(float mx) {
  int kA = 0;
  const int ebits = (__float_as_int(mx) >> 23) & 0xff;
  if (ebits > 0 && ebits < 255) kA = 8 * ((13 - (ebits - 127)) >> 3);
  if (kA > 96) kA = 96;
  if (kA < -96) kA = -96;
  return kA;
}

// wave-wide maximum on the DPP network (quad permutes, row mirrors, row broadcasts, one readlane):
// __shfl_xor is a ds_bpermute per step -- six dependent LDS latencies per phase
__device__ __forceinline__ float fwd_s2_wave_max(float v) {
  auto step = [&](auto CTRL, auto ROWS) {
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), decltype(CTRL)::value,
                                              decltype(ROWS)::value, 0xf, false);
    v = fmaxf(v, __int_as_float(o));
  };
  step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
  step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
  step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
  step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
  step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int DBG>
__global__ __launch_bounds__(kThreads, 1) void adell_fwd_s2_fused_kernel(FwdS2Args a) {
  extern __shared__ char smem[];
  char* sW = smem;
  char* sA = smem + kWBytes;                                  // [2 halves][kHV][64 B]
  float* sMax = reinterpret_cast<float*>(sA + 2 * kImg);      // [2 phases][kPW]
  float* sRed = sMax + 16;                                    // [kCW][32][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nsp = a.ntx * a.nty * a.ntz;

  // bricks of this block: first, first + stride, ... (count of them); blocks b, b + 8, ... run on one
  // XCD and share one contiguous eighth of the bricks
  int first, stride, count;
  if (gridDim.x >= 8) {
    const int xcd = blockIdx.x & 7, per = (a.nbricks + 7) >> 3;
    const int lo = xcd * per, hi = (lo + per < a.nbricks) ? lo + per : a.nbricks;
    stride = gridDim.x >> 3;
    first = lo + (blockIdx.x >> 3);
    count = first < hi ? (hi - 1 - first) / stride + 1 : 0;
  } else {
    stride = gridDim.x;
    first = blockIdx.x;
    count = first < a.nbricks ? (a.nbricks - 1 - first) / stride + 1 : 0;
  }
  if (count == 0) return;                                 // (whole block: no barrier is skipped)
  const int nphases = count * 8;

  auto brick_origin = [&](int t, int& nb, int& tile, int& ox0, int& oy0, int& oz0) {
    tile = t % nsp;
    nb = t / nsp;
    int r = tile;
    const int tx = r % a.ntx;
    r /= a.ntx;
    const int ty = r % a.nty;
    const int tz = r / a.nty;
    ox0 = tx * 8;
    oy0 = ty * 8;
    oz0 = tz * 4;
  };

  // ---- the split weight, once: global row (tap * 32 + n) * 2 + chunk -> [tap][chunk][n] ---------
  // (a plain strided loop: a register array of nine rows per pass ended up in scratch memory)
#pragma unroll 7
  for (int it = tid; it < 27 * 256; it += kThreads) {
    const float4 v = *reinterpret_cast<const float4*>(a.wpack + (size_t)it * 16);
    const int slot = it & 3, row = it >> 2;
    const int ch = row & 1, n = (row >> 1) & 31, tap = row >> 6;
    *reinterpret_cast<float4*>(sW + ((tap * 2 + ch) * 32 + n) * 64 + ((slot ^ ((n >> 2) & 3)) << 4)) = v;
  }

  if (wave >= kCW) {
    // =================================== producers ==============================================
    const int ptid = tid - kCW * 64, pw = wave - kCW;
    // Per-item constants of this thread, computed ONCE: position of item u in the halo (item = one
    // 16-byte piece of a half row: voxel j >> 2, channels 4 (j & 3) ... of the half), its element
    // offset from the halo origin in x, its LDS byte address (hi half; the lo half is that ^ 32).
    // A thread past the end of the half image (items 1620 .. 1791 of its last round) repeats items
    // 0 .. 171: the same bytes to the same LDS address as their owner -- no per-item branch.
    // (ldsoff: bits 0-15 the LDS address, 16-19 / 20-23 / 24-27 the halo position x / y / z)
    unsigned reloff[kPer], ldsoff[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      int j = ptid + kPThreads * u;
      j = j < kItems ? j : j - kItems;
      const int hv = j >> 2, q4 = j & 3;
      const int hz = hv / (kHX * kHY), rem = hv - hz * (kHX * kHY);
      const int hy = rem / kHX, hx = rem - hy * kHX;
      reloff[u] = (unsigned)((2 * hz * a.H + 2 * hy) * a.W + 2 * hx) * 32u + 4u * q4;
      const int sw = (hv >> 2) & 3;
      ldsoff[u] = (unsigned)(hv * 64 + (q4 & 1) * 8 + (((q4 >> 1) ^ sw) << 4)) |
                  (unsigned)((hx << 16) | (hy << 20) | (hz << 24));
    }
    // two register images (phases p and p + 1), each as two halves of kPer pieces
    float4 raw[2][2][kPer];
    unsigned okbuf[2] = {0u, 0u};
    int pf_nb = 0, pf_ox0 = 0, pf_oy0 = 0, pf_oz0 = 0;
    // loads of phase p into f (both halves of a 128-byte line by neighbouring instructions: issued
    // half a phase apart the second half missed the vector cache again); moves the brick origin
    // with sub-lattice 0 and sets okbits
    auto issue = [&](int p, float4 (&f)[2][kPer], unsigned& okbits) {
      const int cls = p & 7;
      if (cls == 0) {
        int tile;
        brick_origin(first + (p >> 3) * stride, pf_nb, tile, pf_ox0, pf_oy0, pf_oz0);
        pf_nb = __builtin_amdgcn_readfirstlane(pf_nb);
        pf_ox0 = __builtin_amdgcn_readfirstlane(pf_ox0);
        pf_oy0 = __builtin_amdgcn_readfirstlane(pf_oy0);
        pf_oz0 = __builtin_amdgcn_readfirstlane(pf_oz0);
      }
      const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
      const int bz = 2 * (pf_oz0 - 1) + pz, by = 2 * (pf_oy0 - 1) + py, bx = 2 * (pf_ox0 - 1) + px;
      const ADELL_GLOBAL float* src = uniform_ptr(a.x + (size_t)pf_nb * a.D * a.H * a.W * 32);
      // a halo that lies inside the volume (72 % of the bricks at 128^3): origin + constant offsets
      const bool inside = (bz >= 0) & (by >= 0) & (bx >= 0) & (bz + 2 * (kHZ - 1) < a.D) &
                          (by + 2 * (kHY - 1) < a.H) & (bx + 2 * (kHX - 1) < a.W);
      unsigned rel[kPer];
      unsigned ok = 0;
      if (inside) {
        const unsigned base = (unsigned)((bz * a.H + by) * a.W + bx) * 32u;
#pragma unroll
        for (int u = 0; u < kPer; ++u) rel[u] = base + reloff[u];
        ok = 0x80000000u;                 // (bit 31: nothing to mask)
      } else {
        const int base = ((bz * a.H + by) * a.W + bx) * 32;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const int hx = (ldsoff[u] >> 16) & 15, hy = (ldsoff[u] >> 20) & 15, hz = ldsoff[u] >> 24;
          const int iz = bz + 2 * hz, iy = by + 2 * hy, ix = bx + 2 * hx;
          const bool v = (iz >= 0) & (iz < a.D) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
          rel[u] = v ? (unsigned)(base + (int)reloff[u]) : 0u;
          ok |= v ? (1u << u) : 0u;
        }
      }
      okbits = ok;
      // ONE fetch site for both cases (two sites meeting in a phi made the register copies wait
      // for the loads)
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const f32x4 v0 = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(src + rel[u]);
        const f32x4 v1 = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(src + rel[u] + 16);
        f[0][u] = make_float4(v0.x, v0.y, v0.z, v0.w);
        f[1][u] = make_float4(v1.x, v1.y, v1.z, v1.w);
      }
    };
    // mask (in place) and absmax of a landed register image -> this wave's slot of sMax row `row`
    // (an image fetched from inside the volume needs no mask)
    auto publish_max = [&](float4 (&f)[2][kPer], unsigned okbits, int row) {
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (!(okbits >> 31)) {
#pragma unroll
          for (int u = 0; u < kPer; ++u) {
            const bool ok = (okbits >> u) & 1u;
            f[c][u] = make_float4(ok ? f[c][u].x : 0.f, ok ? f[c][u].y : 0.f, ok ? f[c][u].z : 0.f,
                                  ok ? f[c][u].w : 0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u)
          mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f[c][u].x)), fmaxf(fabsf(f[c][u].y), fabsf(f[c][u].z))),
                     fabsf(f[c][u].w));
      }
      mx = fwd_s2_wave_max(mx);
      if (lane == 0) sMax[row * kPW + pw] = mx;
    };
    auto convert = [&](const float4 (&f)[kPer], float scaleA, char* img) {
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        if (!(DBG & 8)) {
          const float v[4] = {f[u].x * scaleA, f[u].y * scaleA, f[u].z * scaleA, f[u].w * scaleA};
          half4 h, l;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            h[j] = (_Float16)v[j];
            l[j] = (_Float16)(v[j] - (float)h[j]);
          }
          const unsigned off = ldsoff[u] & 0xffffu;
          *reinterpret_cast<half4*>(img + off) = h;
          *reinterpret_cast<half4*>(img + (off ^ 32u)) = l;
        }
      }
    };

    float block_max = 0.f;
    // (every load of the loop is issued unconditionally -- past the last phase it fetches the last
    // phase again -- so that the counted waits the compiler derives hold on every path: a
    // conditional issue made every wait a vmcnt(0) and left no load in flight across the barriers)
    issue(0, raw[0], okbuf[0]);
    issue(1, raw[1], okbuf[1]);
    publish_max(raw[0], okbuf[0], 0);
    ADELL_S2_BARRIER();                                   // weights + the first absmax are in LDS
    // iteration p: halves of phase p from register image p & 1 to LDS (a barrier after each), the
    // loads of phase p + 2 into the registers just freed, the absmax of phase p + 1
    auto pstep = [&](int p, auto BUF) __attribute__((always_inline)) {
      constexpr int B = decltype(BUF)::value;
      const float* sm = sMax + kPW * (p & 1);
      const float mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
      block_max = fmaxf(block_max, mx);
      const float scaleA = __int_as_float((fwd_s2_scale_exp(mx) + 127) << 23);
      const int pn = p + 2 < nphases ? p + 2 : nphases - 1;
      convert(raw[B][0], scaleA, sA);
      ADELL_S2_BARRIER();
      convert(raw[B][1], scaleA, sA + kImg);
      if (!((DBG & 16) && p > 1)) issue(pn, raw[B], okbuf[B]);
      publish_max(raw[1 - B], okbuf[1 - B], (p + 1) & 1);
      ADELL_S2_BARRIER();
    };
    for (int p = 0; p < nphases; p += 2) {                // (nphases is a multiple of 8)
      pstep(p, std::integral_constant<int, 0>{});
      pstep(p + 1, std::integral_constant<int, 1>{});
    }
    ADELL_S2_BARRIER();                                   // the consumers' last statistics rows
    if (a.amax_out != nullptr && ptid == 0) atomicMax(a.amax_out, __float_as_uint(block_max));
    return;
  }

  // ===================================== consumers ================================================
  // wave w = output plane z = w of the brick: m-tile mt = rows y = 4 mt .. 4 mt + 3 (32 voxels)
  const int wz = wave;
  int arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) arow[mt] = (wz * kHY + (li >> 3) + 4 * mt) * kHX + (li & 7);
  const int bsw = (li >> 2) & 3;
  const int boffh = li * 64 + ((lh ^ bsw) << 4), boffl = li * 64 + (((2 + lh) ^ bsw) << 4);
  const float wsc = a.wscale[li];
  const float bcol = a.bias ? a.bias[li] : 0.f;
  f32x16 acc[2];
  int kprev = 0;

  // taps of sub-lattice cls on half image CH: per axis parity 0 -> tap 1 at offset 1; parity 1 ->
  // tap 0 at offset 0 and tap 2 at offset 1 (offsets in the halo whose origin is o0 - 1)
  auto taps = [&](int cls, auto CHT) __attribute__((always_inline)) {
    constexpr int ch = decltype(CHT)::value;
    const char* sAc = sA + ch * kImg;
    static_for<8>([&](auto CLS) {
      constexpr int c = decltype(CLS)::value;
      if (cls == c) {
        constexpr int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
        static_for<8>([&](auto T) {
          constexpr int tb = decltype(T)::value;   // bit a: the second tap of axis a (parity 1 only)
          constexpr int sz = tb >> 2, sy = (tb >> 1) & 1, sx = tb & 1;
          if constexpr ((sz <= pz) && (sy <= py) && (sx <= px) && !(DBG & 1)) {
            constexpr int tz = pz ? 2 * sz : 1, ty = py ? 2 * sy : 1, tx = px ? 2 * sx : 1;
            constexpr int dz = pz ? sz : 1, dy = py ? sy : 1, dx = px ? sx : 1;
            constexpr int tap = (tz * 3 + ty) * 3 + tx;
            const char* bt = sW + (tap * 2 + ch) * (32 * 64);
            const half8 bh = *reinterpret_cast<const half8*>(bt + boffh);
            const half8 bl = *reinterpret_cast<const half8*>(bt + boffl);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const int hv = arow[mt] + (dz * kHY + dy) * kHX + dx;
              const int sw = (hv >> 2) & 3;
              const char* row = sAc + hv * 64;
              const half8 ah = *reinterpret_cast<const half8*>(row + ((lh ^ sw) << 4));
              const half8 al = *reinterpret_cast<const half8*>(row + (((2 + lh) ^ sw) << 4));
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mt], 0, 0, 0);
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mt], 0, 0, 0);
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mt], 0, 0, 0);
            }
          }
        });
      }
    });
  };
  // statistics rows of brick i: the four planes' sums (sRed, written by the epilogue before the last
  // barrier) folded in plane order by the first 32 lanes of consumer wave 0
  auto fold_stats = [&](int i) {
    if (a.part && tid < 32 && !(DBG & 2)) {
      int nb, tile, ox0, oy0, oz0;
      brick_origin(first + i * stride, nb, tile, ox0, oy0, oz0);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < kCW; ++w) {
        t1 += sRed[(w * 32 + tid) * 2 + 0];
        t2 += sRed[(w * 32 + tid) * 2 + 1];
      }
      float* p = a.part + (((size_t)nb * nsp + tile) * 32 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
  };

  ADELL_S2_BARRIER();                                     // weights + the first absmax are in LDS
  for (int p = 0; p < nphases; ++p) {
    const int cls = p & 7;
    {
      const float* sm = sMax + kPW * (p & 1);
      const float mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
      const int kA = fwd_s2_scale_exp(mx);
      if (cls == 0) {   // first sub-lattice of a brick: fresh accumulators
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
      } else if (kA != kprev) {
        const float fix = __int_as_float((kA - kprev + 127) << 23);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][r] *= fix;
      }
      kprev = kA;
    }
    ADELL_S2_BARRIER();                                   // half image 0 of phase p is in LDS
    if (cls == 0 && p > 0) fold_stats((p >> 3) - 1);
    taps(cls, std::integral_constant<int, 0>{});
    ADELL_S2_BARRIER();                                   // half image 1
    taps(cls, std::integral_constant<int, 1>{});
    if (cls == 7) {
      // ---- epilogue of the brick: C row r of m-tile mt = output (x = (r & 3) + 4 lh,
      // y = (r >> 2) + 4 mt) of plane wz
      int nb, tile, ox0, oy0, oz0;
      brick_origin(first + (p >> 3) * stride, nb, tile, ox0, oy0, oz0);
      const float oscale = __int_as_float((127 - kprev) << 23) * wsc;
      const int z = oz0 + wz;
      float s1 = 0.f, s2 = 0.f;
      if (z < a.Do && !(DBG & 2)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int x = ox0 + (r & 3) + 4 * lh, yy = oy0 + (r >> 2) + 4 * mt;
            if (x < a.Wo && yy < a.Ho) {
              const float v = acc[mt][r] * oscale + bcol;
              a.y[((((size_t)nb * a.Do + z) * a.Ho + yy) * a.Wo + x) * 32 + li] = v;
              s1 += v;
              s2 += v * v;
            }
          }
      }
      if (a.part) {
        // fixed order: rows of a lane (m-tile 0 then 1), lane halves, then (fold_stats) the planes
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lh == 0) {
          sRed[(wz * 32 + li) * 2 + 0] = s1;
          sRed[(wz * 32 + li) * 2 + 1] = s2;
        }
      }
    }
  }
  ADELL_S2_BARRIER();                                     // the last brick's statistics rows
  if (count > 0) fold_stats(count - 1);
}

namespace {

int cu_count() {
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

bool fwd_s2_fused_ok(const adell_conv3d_desc* d) {
  return d && d->C1 == 0 && d->C0 == 32 && d->Cout == 32 && d->KD == 3 && d->KH == 3 && d->KW == 3 &&
         d->SD == 2 && d->SH == 2 && d->SW == 2 && d->PD == 1 && d->PH == 1 && d->PW == 1 &&
         d->D % 2 == 0 && d->H % 2 == 0 && d->W % 2 == 0 && d->N > 0 && d->D > 0 && d->H > 0 &&
         d->W > 0 && d->Do == d->D / 2 && d->Ho == d->H / 2 && d->Wo == d->W / 2 &&
         (size_t)d->D * d->H * d->W * 32 < ((size_t)1 << 29);   // 32-bit offsets inside an item
}

}  // namespace

// 1 when adell_conv3d_fwd_s2_fused takes this layer (the conditions of the backward-data twin).
extern "C" int adell_conv3d_fwd_s2_fused_applicable(const adell_conv3d_desc* d) {
  return fwd_s2_fused_ok(d) ? 1 : 0;
}

// statistics partial rows per batch item (stat_partials is [N][this][32][2]), or an error code
extern "C" int adell_conv3d_fwd_s2_fused_ntiles(const adell_conv3d_desc* d) {
  if (!fwd_s2_fused_ok(d)) return ADELL_E_UNSUPPORTED;
  return adell_cdiv(d->Wo, 8) * adell_cdiv(d->Ho, 8) * adell_cdiv(d->Do, 4);
}

// y (+ bias, + statistics partials, + absmax of x) of such a layer in one launch. w_split / wscale:
// adell_pack_weight_f16x3 mode 0 of the [32][32][3][3][3] weight (the pack adell_conv3d_fwd_f16x3 takes).
extern "C" int adell_conv3d_fwd_s2_fused(const adell_conv3d_desc* d, const float* x,
                                         const void* w_split, const float* wscale,
                                         const float* bias, float* y, float* stat_partials,
                                         int partial_rows, uint32_t* in_absmax, void* stream) {
  ADELL_REQUIRE(fwd_s2_fused_ok(d),
                "conv_fwd_s2_fused: needs 32 -> 32 channels, k = 3, stride 2, padding 1, even dims");
  ADELL_REQUIRE_ROWS(stat_partials, partial_rows,
                     (long)adell_cdiv(d->Wo, 8) * adell_cdiv(d->Ho, 8) * adell_cdiv(d->Do, 4),
                     "conv_fwd_s2_fused");
  ADELL_REQUIRE(x && w_split && wscale && y, "conv_fwd_s2_fused: null pointer");
  ADELL_REQUIRE(((uintptr_t)x & 15) == 0, "conv_fwd_s2_fused: x must be 16-byte aligned");
  FwdS2Args a;
  a.x = x;
  a.wpack = reinterpret_cast<const char*>(w_split);
  a.wscale = wscale;
  a.bias = bias;
  a.y = y;
  a.part = stat_partials;
  a.amax_out = in_absmax;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ntx = adell_cdiv(d->Wo, 8);
  a.nty = adell_cdiv(d->Ho, 8);
  a.ntz = adell_cdiv(d->Do, 4);
  const long nbricks = (long)d->N * a.ntx * a.nty * a.ntz;
  ADELL_REQUIRE(nbricks < 0x0fffffffL, "conv_fwd_s2_fused: too many bricks");
  a.nbricks = (int)nbricks;
  int grid = (int)(nbricks < cu_count() ? nbricks : cu_count());   // one block per CU
  if (grid >= 8) grid &= ~7;                                       // whole blocks per XCD
  auto launch = [&](auto kern) -> int {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), kLds, (hipStream_t)stream, a);
    return ADELL_OK;
  };
  int rc = ADELL_OK;
#ifdef ADELL_DEBUG
  switch (g_adell_tune.igemm_dbg) {
    case 1: rc = launch(adell_fwd_s2_fused_kernel<1>); break;
    case 2: rc = launch(adell_fwd_s2_fused_kernel<2>); break;
    case 3: rc = launch(adell_fwd_s2_fused_kernel<3>); break;
    case 8: rc = launch(adell_fwd_s2_fused_kernel<8>); break;
    case 16: rc = launch(adell_fwd_s2_fused_kernel<16>); break;
    case 27: rc = launch(adell_fwd_s2_fused_kernel<27>); break;
    default: rc = launch(adell_fwd_s2_fused_kernel<0>); break;
  }
#else
  rc = launch(adell_fwd_s2_fused_kernel<0>);
#endif
  if (rc != ADELL_OK) return rc;
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
