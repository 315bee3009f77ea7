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
// Here a persistent block (one per CU, 8 waves: a wave alone on its SIMD issues one vector
// instruction per 4 cycles, two waves one per 2 -- and the staging arithmetic, not memory, was what
// the four-wave form spent its time on)
//   * keeps the whole split weight (27 x 32 x 32 x (hi, lo) = 108 KB) in LDS for its lifetime,
//   * walks the eight sub-lattices of a brick: stage + split one 51 KB halo, run its taps into the
//     SAME 2 x 16 accumulator registers per wave, while the halos of the next TWO sub-lattices are
//     in flight in registers (a sub-lattice voxel is a full 128-byte line of x),
//   * takes the operand scale per (brick, sub-lattice) from the block-wide absmax of the staged
//     halo and rescales the accumulators by the exact power of two when it changes,
//   * emits bias, the per-channel (sum, sum of squares) partials of the following norm and the
//     absmax of x (for the weight-gradient kernel) like the implicit-GEMM epilogue.
// Roofline: HBM (x read once: 537 MB at 2 x 128^3, y 67 MB).
#include <type_traits>
#include <utility>
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kHX = 9, kHY = 9, kHZ = 5, kHV = kHX * kHY * kHZ;
constexpr int kWBytes = 27 * 2 * 32 * 64;
constexpr int kABytes = kHV * 128;
constexpr int kThreads = 512, kWaves = kThreads / 64;
constexpr int kRedFloats = 4 * 32 * 2;                       // statistics fold: [plane][channel][2]
constexpr int kLds = kWBytes + kABytes + 64 + kRedFloats * 4;
static_assert(kLds <= 160 * 1024, "one block per CU: weights + one halo image");
constexpr int kItems = kHV * 8;
constexpr int kPer = (kItems + kThreads - 1) / kThreads;

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
// 2 no y stores / statistics, 8 no halo split / LDS stores, 16 no halo loads after the first phase
template <int DBG>
__global__ __launch_bounds__(kThreads, 1) void adell_fwd_s2_fused_kernel(FwdS2Args a) {
  extern __shared__ char smem[];
  char* sW = smem;
  char* sA = smem + kWBytes;
  float* sMax = reinterpret_cast<float*>(sA + kABytes);
  float* sRed = sMax + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int q = tid & 7;
  const int nsp = a.ntx * a.nty * a.ntz;

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
  // halo of sub-lattice (pz, py, px) behind brick t: halo voxel h <-> x index 2 (o0 - 1 + h) + p per axis
  // two register images: the loads of a phase are issued two phases ahead of their use (a
  // one-tap sub-lattice runs 12 MFMAs per wave: far less than a memory round trip)
  float4 fbuf[2][kPer];
  unsigned okbuf[2] = {0u, 0u};
  // Per-item constants of this thread, computed ONCE (the staging arithmetic -- two divisions, six
  // comparisons, the swizzled LDS address per 16 bytes -- was what the kernel spent its time on, not
  // memory): position of item u in the halo, its element offset from the halo origin in x, its
  // LDS byte address (hi half; the lo half is that ^ 32).
  unsigned hpos[kPer], reloff[kPer], ldsoff[kPer];
  unsigned okstatic = 0;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int it = tid + kThreads * u;
    const bool in = it < kItems;
    const int hv = in ? it >> 3 : 0;
    const int hz = hv / (kHX * kHY), rem = hv - hz * (kHX * kHY);
    const int hy = rem / kHX, hx = rem - hy * kHX;
    hpos[u] = (unsigned)(hx | (hy << 8) | (hz << 16));
    reloff[u] = (unsigned)((2 * hz * a.H + 2 * hy) * a.W + 2 * hx) * 32u + 4u * q;
    const int sw = (hv >> 2) & 3, slot = (q & 3) >> 1;
    ldsoff[u] = (unsigned)((q >> 2) * (kHV * 64) + hv * 64 + (q & 1) * 8 + ((slot ^ sw) << 4));
    okstatic |= in ? (1u << u) : 0u;
  }
  // origin of the brick whose sub-lattices are being fetched (moves on with sub-lattice 0)
  int pf_nb = 0, pf_ox0 = 0, pf_oy0 = 0, pf_oz0 = 0;
  auto prefetch = [&](int phase, float4 (&f)[kPer], unsigned& okbits) {
    const int cls = phase & 7;
    if (cls == 0) {
      int tile;
      brick_origin(blockIdx.x + (phase >> 3) * gridDim.x, pf_nb, tile, pf_ox0, pf_oy0, pf_oz0);
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
    if (inside) {
      const unsigned base = (unsigned)((bz * a.H + by) * a.W + bx) * 32u;
#pragma unroll
      for (int u = 0; u < kPer; ++u) rel[u] = base + reloff[u];
      okbits = okstatic | 0x80000000u;      // (bit 31: nothing to mask)
    } else {
      const int base = ((bz * a.H + by) * a.W + bx) * 32;
      okbits = 0;
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int hx = hpos[u] & 255, hy = (hpos[u] >> 8) & 255, hz = hpos[u] >> 16;
        const int iz = bz + 2 * hz, iy = by + 2 * hy, ix = bx + 2 * hx;
        const bool ok = ((okstatic >> u) & 1u) & (iz >= 0) & (iz < a.D) & (iy >= 0) & (iy < a.H) &
                        (ix >= 0) & (ix < a.W);
        rel[u] = ok ? (unsigned)(base + (int)reloff[u]) : 0u;
        okbits |= ok ? (1u << u) : 0u;
      }
    }
    // ONE fetch site for both cases (two sites meeting in a phi made the register copies wait
    // for the loads)
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const f32x4 v = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(src + rel[u]);
      f[u] = make_float4(v.x, v.y, v.z, v.w);
    }
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

  // A rows of this lane at offset (0, 0, 0): wave w = output plane z = w & 3, rows y = 4 wm .. + 3
  // with wm = w >> 2 (one 32-voxel m-tile per wave)
  const int wz = wave & 3, wm = wave >> 2;
  const int arow = (wz * kHY + (li >> 3) + 4 * wm) * kHX + (li & 7);
  const int bsw = (li >> 2) & 3;
  const int boffh = li * 64 + ((lh ^ bsw) << 4), boffl = li * 64 + (((2 + lh) ^ bsw) << 4);
  const float wsc = a.wscale[li];
  const float bcol = a.bias ? a.bias[li] : 0.f;
  float block_max = 0.f;

  const int my_bricks = ((int)blockIdx.x < a.nbricks)
                            ? (a.nbricks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nphases = my_bricks * 8;
  f32x16 acc;
  int kprev = 0;
  // one phase = one sub-lattice of one brick.
  // iteration ph: issue the loads of phase ph (into register image ph & 1), run the MFMAs of phase
  // ph - 2 out of LDS (and close its brick after the eighth sub-lattice), move phase ph - 1 from
  // its register image to LDS. Unrolled by two so that each image has ONE fetch site.
  auto step = [&](int ph, auto BUF) __attribute__((always_inline)) {
    constexpr int B = decltype(BUF)::value;
    if (ph < nphases && !((DBG & 16) && ph > 1)) prefetch(ph, fbuf[B], okbuf[B]);
    __builtin_amdgcn_sched_barrier(0);
    if (ph > 1) {
      const int cls = (ph - 2) & 7;
      // ---- taps of sub-lattice cls: per axis parity 0 -> tap 1 at offset 1; parity 1 -> tap 0 at
      // offset 0 and tap 2 at offset 1 (offsets in the halo whose origin is o0 - 1)
      static_for<8>([&](auto CLS) {
        constexpr int c = decltype(CLS)::value;
        if (cls == c) {
          constexpr int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
          static_for<2>([&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            const char* sAc = sA + ch * (kHV * 64);
            static_for<8>([&](auto T) {
              constexpr int tb = decltype(T)::value;   // bit a: the second tap of axis a (parity 1 only)
              constexpr int sz = tb >> 2, sy = (tb >> 1) & 1, sx = tb & 1;
              if constexpr ((sz <= pz) && (sy <= py) && (sx <= px) && !(DBG & 1)) {
                constexpr int tz = pz ? 2 * sz : 1, ty = py ? 2 * sy : 1, tx = px ? 2 * sx : 1;
                constexpr int dz = pz ? sz : 1, dy = py ? sy : 1, dx = px ? sx : 1;
                constexpr int tap = (tz * 3 + ty) * 3 + tx;
                const int hv = arow + (dz * kHY + dy) * kHX + dx;
                const int sw = (hv >> 2) & 3;
                const char* row = sAc + hv * 64;
                const half8 ah = *reinterpret_cast<const half8*>(row + ((lh ^ sw) << 4));
                const half8 al = *reinterpret_cast<const half8*>(row + (((2 + lh) ^ sw) << 4));
                const char* bt = sW + (tap * 2 + ch) * (32 * 64);
                const half8 bh = *reinterpret_cast<const half8*>(bt + boffh);
                const half8 bl = *reinterpret_cast<const half8*>(bt + boffl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
              }
            });
          });
        }
      });
      if (cls == 7) {
        // ---- epilogue of the brick: C row r = output (x = (r & 3) + 4 lh, y = (r >> 2) + 4 wm)
        const int t = blockIdx.x + ((ph - 2) >> 3) * gridDim.x;
        int nb, tile, ox0, oy0, oz0;
        brick_origin(t, nb, tile, ox0, oy0, oz0);
        const float oscale = __int_as_float((127 - kprev) << 23) * wsc;
        const int z = oz0 + wz;
        float s1 = 0.f, s2 = 0.f;
        if (z < a.Do && !(DBG & 2)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int x = ox0 + (r & 3) + 4 * lh, yy = oy0 + (r >> 2) + 4 * wm;
            if (x < a.Wo && yy < a.Ho) {
              const float v = acc[r] * oscale + bcol;
              a.y[((((size_t)nb * a.Do + z) * a.Ho + yy) * a.Wo + x) * 32 + li] = v;
              s1 += v;
              s2 += v * v;
            }
          }
        }
        if (a.part) {
          // fold in a fixed order: lane halves, then the two row groups of a plane (waves w + 4
          // hand theirs to waves w through sRed), then the four planes
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (lh == 0 && wm == 1) {
            sRed[(wz * 32 + li) * 2 + 0] = s1;
            sRed[(wz * 32 + li) * 2 + 1] = s2;
          }
          __syncthreads();
          if (lh == 0 && wm == 0) {
            sRed[(wz * 32 + li) * 2 + 0] += s1;
            sRed[(wz * 32 + li) * 2 + 1] += s2;
          }
          __syncthreads();
          if (tid < 32) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              t1 += sRed[(w * 32 + tid) * 2 + 0];
              t2 += sRed[(w * 32 + tid) * 2 + 1];
            }
            float* p = a.part + (((size_t)nb * nsp + tile) * 32 + tid) * 2;
            p[0] = t1;
            p[1] = t2;
          }
        }
      }
    }
    if (ph > 0 && ph - 1 < nphases) {
      float4 (&f)[kPer] = fbuf[1 - B];
      const unsigned okbits = okbuf[1 - B];
      // ---- mask, absmax -> power-of-two scale of this sub-lattice halo ----------------------------
      // (an image fetched from inside the volume needs no mask: the items past the end of the
      // halo were read from its first voxel and change no maximum)
      float mx = 0.f;
      if (!(okbits >> 31)) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
          const bool ok = (okbits >> u) & 1u;
          f[u] = make_float4(ok ? f[u].x : 0.f, ok ? f[u].y : 0.f, ok ? f[u].z : 0.f, ok ? f[u].w : 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u)
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f[u].x)), fmaxf(fabsf(f[u].y), fabsf(f[u].z))), fabsf(f[u].w));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      // (two sMax rows used in turn: a row is rewritten two phases after it was read, with a
      // barrier in between, so one barrier serves "fragments of phase ph - 2 are read" and "the
      // four wave maxima are visible")
      float* sm = sMax + kWaves * (ph & 1);
      if (lane == 0) sm[wave] = mx;
      __syncthreads();
      mx = fmaxf(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])),
                 fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7])));
      block_max = fmaxf(block_max, mx);
      int kA = 0;
      {
        const int ebits = (__float_as_int(mx) >> 23) & 0xff;
        // multiples of 8 (max lands in [2^6, 2^14)): the scale rarely changes inside a brick
        if (ebits > 0 && ebits < 255) kA = 8 * ((13 - (ebits - 127)) >> 3);
        if (kA > 96) kA = 96;
        if (kA < -96) kA = -96;
      }
      if (((ph - 1) & 7) == 0) {   // first sub-lattice of a brick: fresh accumulators
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      } else if (kA != kprev) {
        const float fix = __int_as_float((kA - kprev + 127) << 23);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= fix;
      }
      kprev = kA;
      const float scaleA = __int_as_float((kA + 127) << 23);
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        if (((okstatic >> u) & 1u) && !(DBG & 8)) {
          const float v[4] = {f[u].x * scaleA, f[u].y * scaleA, f[u].z * scaleA, f[u].w * scaleA};
          half4 h, l;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            h[j] = (_Float16)v[j];
            l[j] = (_Float16)(v[j] - (float)h[j]);
          }
          *reinterpret_cast<half4*>(sA + ldsoff[u]) = h;
          *reinterpret_cast<half4*>(sA + (ldsoff[u] ^ 32u)) = l;
        }
      }
      __syncthreads();
    }
  };
  for (int ph = 0; ph <= nphases + 1; ph += 2) {
    step(ph, std::integral_constant<int, 0>{});
    step(ph + 1, std::integral_constant<int, 1>{});
  }
  if (a.amax_out != nullptr && tid == 0) atomicMax(a.amax_out, __float_as_uint(block_max));
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
  const int grid = (int)(nbricks < cu_count() ? nbricks : cu_count());   // one block per CU
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
