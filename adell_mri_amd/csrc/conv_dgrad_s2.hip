// Backward-data of the U-Net downsampling convolution (3x3x3, stride 2, padding 1, 32 -> 32
// channels: unet.py:571-579 at the two high-resolution levels) in ONE launch, f16x3 arithmetic.
//
//   dX[2i + p][ci] = sum over the taps t = p + 1 (mod 2) per axis of dY[i + (p + 1 - t) / 2][.] w[.][ci][t]
//
// Per axis: p = 0 takes tap 1 at dY offset 0; p = 1 takes tap 2 at offset 0 and tap 0 at offset 1.
// So the eight parity classes p of the dX voxels behind one dY brick read the same 9 x 9 x 5 halo of
// that brick at the eight offsets delta in {0,1}^3, 27 (class, tap) products in all. The parity-class
// formulation on the implicit-GEMM kernel (adell_conv3d_bwd_data_s2_f16x3) runs it as eight launches
// that each stage and split that halo again and restage their sub-kernel per block: 0.54 ms at
// 2 x 128^3 for 0.1 ms of MFMA work. Here a persistent block (one per CU, 4 waves)
//   * keeps the whole split weight (27 taps x 32 x 32 x (hi, lo) = 108 KB) in LDS for its lifetime,
//   * stages and splits the dY halo of a brick ONCE (51 KB, both 16-channel chunks),
//   * holds the 8 classes x 2 m-tiles of fp32 accumulators in registers (256 per lane), reads the
//     A fragments of an offset once for all the classes that use it (LDS: 86 fragment reads per 162
//     MFMAs per chunk -- the eight-launch form reads 6 per 6),
//   * prefetches the next brick's halo into registers while the MFMAs run,
//   * stores dX (+ the skip fork's parked gradient, functional.GradCarry) as full 128-byte lines.
// Roofline: HBM, dX written once (+ add0 read once) + dY read once: 17 B per dX element with add0.
#include <type_traits>
#include <utility>
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kHX = 9, kHY = 9, kHZ = 5, kHV = kHX * kHY * kHZ;   // halo of an 8 x 8 x 4 dY brick
constexpr int kWBytes = 27 * 2 * 32 * 64;                          // [tap][chunk][n][64 B]
constexpr int kABytes = kHV * 128;                                 // [chunk][halo voxel][64 B]
constexpr int kLds = kWBytes + kABytes + 64;
constexpr int kItems = kHV * 8;                                    // float4 loads per halo
constexpr int kPer = (kItems + 255) / 256;                         // ... per thread (13)

// a wave-uniform pointer pinned into SGPRs: `p[(unsigned)lane_offset]` is then one scalar-base +
// 32-bit-VGPR-offset access
// (typed ADELL_GLOBAL, see common.h: no flat_load / flat_store in the MFMA loop)
template <typename T>
__device__ __forceinline__ ADELL_GLOBAL T* uniform_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<ADELL_GLOBAL T*>(((uint64_t)hi << 32) | lo);
}

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N)
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& body, std::integer_sequence<int, I...>) {
  (body(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& body) {
  static_for_impl<N>(body, std::make_integer_sequence<int, N>{});
}

struct DgradS2Args {
  const float* dy;      // [N][Do][Ho][Wo][32]
  const char* wpack;    // adell_pack_weight_f16x3 mode 1 of the full weight: [26 - tap][ci][chunk][64 B]
  const float* wscale;  // [32]
  const float* add0;    // like dx, or null
  float* dx;            // [N][2 Do][2 Ho][2 Wo][32]
  unsigned* amax_out;   // optional: absmax of dY (float bits)
  int N, Do, Ho, Wo;
  int ntx, nty, ntz;
  int nbricks;
};

}  // namespace

// (at global scope so that profilers print its name)
// DBG: timing experiments, results are wrong when nonzero (instantiated by -DADELL_DEBUG builds
// only): 1 no MFMAs, 2 no dX stores, 4 no weight staging, 8 no halo split / LDS stores, 16 no halo
// loads after the first brick
template <int DBG>
__global__ __launch_bounds__(256, 1) void adell_dgrad_s2_fused_kernel(DgradS2Args a) {
  extern __shared__ char smem[];
  char* sW = smem;
  char* sA = smem + kWBytes;
  float* sMax = reinterpret_cast<float*>(sA + kABytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // ---- per-thread halo items: voxel hv = it >> 3, channels 4 q .. 4 q + 3 ------------------
  const int q = tid & 7;   // (256 is a multiple of 8: the same for every item of a thread)
  float4 f[kPer];
  auto brick_origin = [&](int t, int& nb, int& ox0, int& oy0, int& oz0) {
    const int tx = t % a.ntx;
    t /= a.ntx;
    const int ty = t % a.nty;
    t /= a.nty;
    const int tz = t % a.ntz;
    nb = t / a.ntz;
    ox0 = tx * 8;
    oy0 = ty * 8;
    oz0 = tz * 4;
  };
  auto prefetch = [&](int t) {
    int nb, ox0, oy0, oz0;
    brick_origin(t, nb, ox0, oy0, oz0);
    const size_t base = (((size_t)nb * a.Do + oz0) * a.Ho + oy0) * a.Wo + ox0;
    const ADELL_GLOBAL float* src = uniform_ptr(a.dy + base * 32);   // the halo spans < 2^32 bytes
    // The halo coordinates of an item do not depend on the brick; recomputed per brick on purpose
    // (hoisted out of the brick loop they are spilled, and every reload waits for vmcnt(0), i.e.
    // for the load before it). Loads are unconditional (the brick origin stands in for voxels
    // past the tensor) so that the 13 of them are in flight together.
    int tt = tid;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int it = tt + 256 * u;
      const int hv = it >> 3;
      const int hz = hv / (kHX * kHY), rem = hv - hz * (kHX * kHY);
      const int hy = rem / kHX, hx = rem - hy * kHX;
      const bool ok = (it < kItems) & (ox0 + hx < a.Wo) & (oy0 + hy < a.Ho) & (oz0 + hz < a.Do);
      const unsigned rel = ok ? (unsigned)((hz * a.Ho + hy) * a.Wo + hx) * 32u + 4u * q : 0u;
      const f32x4 v = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(src + rel);
      f[u].x = ok ? v.x : 0.f;
      f[u].y = ok ? v.y : 0.f;
      f[u].z = ok ? v.z : 0.f;
      f[u].w = ok ? v.w : 0.f;
    }
  };


  // ---- the split weight, once: global row (tap' * 32 + n) * 2 + chunk -> [tap'][chunk][n] ----
  // (27 x 256 slots of 16 bytes, nine loads in flight per thread)
  if (!(DBG & 4)) {
    // (a plain strided loop: a register array of nine rows per pass ended up in scratch memory)
#pragma unroll 9
    for (int it = tid; it < 27 * 256; it += 256) {
      const float4 v = *reinterpret_cast<const float4*>(a.wpack + (size_t)it * 16);
      const int slot = it & 3, row = it >> 2;
      const int ch = row & 1, n = (row >> 1) & 31, tap = row >> 6;
      *reinterpret_cast<float4*>(sW + ((tap * 2 + ch) * 32 + n) * 64 + ((slot ^ ((n >> 2) & 3)) << 4)) = v;
    }
  }

  int t = blockIdx.x;
  if (t < a.nbricks) prefetch(t);

  // A rows of this lane: m-tile mt of wave w is the z = w plane, y = 4 mt .. 4 mt + 3, x = 0 .. 7
  int arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) arow[mt] = (wave * kHY + (li >> 3) + 4 * mt) * kHX + (li & 7);
  const int bsw = (li >> 2) & 3;
  const int boffh = li * 64 + ((lh ^ bsw) << 4), boffl = li * 64 + (((2 + lh) ^ bsw) << 4);
  const float wsc = a.wscale[li];
  float block_max = 0.f;

  for (; t < a.nbricks; t += gridDim.x) {
    int nb, ox0, oy0, oz0;
    brick_origin(t, nb, ox0, oy0, oz0);
    // ---- absmax of the halo -> power-of-two scale ------------------------------------------
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f[u].x)), fmaxf(fabsf(f[u].y), fabsf(f[u].z))), fabsf(f[u].w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __syncthreads();   // the previous brick's fragment reads are done (first pass: weights staged)
    if (lane == 0) sMax[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sMax[0], sMax[1]), fmaxf(sMax[2], sMax[3]));
    block_max = fmaxf(block_max, mx);
    int kA = 0;
    {
      const int ebits = (__float_as_int(mx) >> 23) & 0xff;
      if (ebits > 0 && ebits < 255) kA = 13 - (ebits - 127);   // max lands in [2^13, 2^14)
      if (kA > 96) kA = 96;
      if (kA < -96) kA = -96;
    }
    const float scaleA = __int_as_float((kA + 127) << 23);
    const float oscale = __int_as_float((127 - kA) << 23) * wsc;
    // ---- split to (hi, lo) halves and store: row = 64 B per (chunk, voxel) --------------------
    // (the LDS row addresses do not depend on the brick either: recomputed per brick on purpose,
    // hoisted they are 13 more spilled registers)
    int ts = tid;
    asm volatile("" : "+v"(ts));
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int it = ts + 256 * u;
      if (it < kItems && !(DBG & 8)) {
        const int hv = it >> 3;
        const float v[4] = {f[u].x * scaleA, f[u].y * scaleA, f[u].z * scaleA, f[u].w * scaleA};
        half4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = (_Float16)v[j];
          l[j] = (_Float16)(v[j] - (float)h[j]);
        }
        const int qs = ts & 7;
        const int sw = (hv >> 2) & 3, slot = (qs & 3) >> 1;
        char* row = sA + (qs >> 2) * (kHV * 64) + hv * 64 + (qs & 1) * 8;
        *reinterpret_cast<half4*>(row + ((slot ^ sw) << 4)) = h;
        *reinterpret_cast<half4*>(row + (((2 + slot) ^ sw) << 4)) = l;
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < a.nbricks && !(DBG & 16)) prefetch(t + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);   // ... and in flight under the MFMAs, not sunk below them

    // ---- 27 (class, tap) products per chunk, A fragments shared by offset -------------------
    // Offsets run in the order 0 .. 7 of their bits (dz, dy, dx): class p only reads offsets that are
    // bitwise subsets of p, so it is complete once offset p is done and its rows are stored (and its
    // add0 rows fetched) under the MFMAs of the next offset -- the dX traffic of a brick is spread
    // over its MFMA phase instead of following it.
    // The phase is compiled three times -- whole brick with / without add0 (straight-line code: the
    // waits on the memory counters are then exact, and the stores of a class drain under the MFMAs of
    // the next), and the ragged-brick form with its per-row bounds checks.
    auto phase = [&](auto FULL_, auto ADD_) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(FULL_)::value, ADD = decltype(ADD_)::value;
    f32x16 acc[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][mt][r] = 0.f;
    // C row r of m-tile mt = dY voxel (x = (r & 3) + 4 lh, y = (r >> 2) + 4 mt, z = wave)
    const int W2 = 2 * a.Wo, H2 = 2 * a.Ho;
    const int bx = ox0 + 4 * lh, bz = oz0 + wave;
    const bool zok = bz < a.Do && !(DBG & 2);
    // element steps on the dX grid (32-bit: an item spans < 2^31 elements, host check)
    // (opaque to the optimiser: the per-row offsets below do not depend on the brick, and hoisted
    // out of the brick loop they are ~350 live values, i.e. spills)
    unsigned eY = (unsigned)W2 * 32u;
    asm volatile("" : "+s"(eY));
    const unsigned eZ = (unsigned)H2 * eY;
    // wave-uniform origin of this wave's z plane of the brick + this lane's (x half, channel)
    const size_t ubase = ((((size_t)nb * 2 * a.Do + 2 * bz) * H2 + 2 * oy0) * W2 + 2 * ox0) * 32;
    ADELL_GLOBAL float* dxu = uniform_ptr(a.dx + ubase);
    const ADELL_GLOBAL float* addu = (a.add0 && (ADD || !FULL)) ? uniform_ptr(a.add0 + ubase) : nullptr;
    unsigned lane_off = 8u * lh * 32u + li;
    asm volatile("" : "+v"(lane_off));
    // row r of (class c, m-tile mt): uniform offset, the x part is a compile-time constant
    auto row_off = [&](const int c, const int mt, const int r) __attribute__((always_inline)) -> unsigned {
      const int pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
      return (unsigned)pz * eZ + (unsigned)(8 * mt + py + 2 * (r >> 2)) * eY +
             (unsigned)(px + 2 * (r & 3)) * 32u;
    };
    // (opaque per brick: otherwise the 64 fragment addresses of the 8 offsets x 2 m-tiles x 2 chunks x
    // (hi, lo) are hoisted out of the brick loop and spilled)
    int arow_b[2] = {arow[0], arow[1]};
    asm volatile("" : "+v"(arow_b[0]), "+v"(arow_b[1]));
    float rv[2][16];
    auto fetch_add0 = [&](const int c) __attribute__((always_inline)) {
      if constexpr (FULL && ADD) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[mt][r] = addu[lane_off + row_off(c, mt, r)];
      }
    };
    auto store_class = [&](const int c) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if constexpr (FULL) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[c][mt][r] * oscale;
            if constexpr (ADD) v += rv[mt][r];
            if (!(DBG & 2)) dxu[lane_off + row_off(c, mt, r)] = v;
          }
        } else if (zok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (bx + (r & 3) < a.Wo && oy0 + 4 * mt + (r >> 2) < a.Ho) {
              const unsigned off = lane_off + row_off(c, mt, r);
              float v = acc[c][mt][r] * oscale;
              if (addu) v += addu[off];
              dxu[off] = v;
            }
          }
        }
      }
    };
    static_for<8>([&](auto D) {
      constexpr int d = decltype(D)::value;
      constexpr int dz = d >> 2, dy_ = (d >> 1) & 1, dx_ = d & 1;
      // (the fences keep the address arithmetic of a class next to its loads / stores: hoisted to
      // the top of this fully unrolled body it costs hundreds of spilled registers)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (d > 0) fetch_add0(D.value - 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DBG & 1)) {
        static_for<2>([&](auto CH) {
          constexpr int ch = decltype(CH)::value;
          const char* sAc = sA + ch * (kHV * 64);
          half8 ah[2], al[2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int hv = arow_b[mt] + (dz * kHY + dy_) * kHX + dx_;
            const int sw = (hv >> 2) & 3;
            const char* row = sAc + hv * 64;
            ah[mt] = *reinterpret_cast<const half8*>(row + ((lh ^ sw) << 4));
            al[mt] = *reinterpret_cast<const half8*>(row + (((2 + lh) ^ sw) << 4));
          }
          // class p reads offset d when d is a bitwise subset of p; per axis, offset 1 is tap 0,
          // offset 0 is tap 1 (p = 0) or tap 2 (p = 1)
          static_for<8>([&](auto P) {
            constexpr int cls = decltype(P)::value;
            if constexpr ((d & ~cls) == 0) {
              constexpr int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
              constexpr int tz = dz ? 0 : 1 + pz, ty = dy_ ? 0 : 1 + py, tx = dx_ ? 0 : 1 + px;
              constexpr int tapp = 26 - ((tz * 3 + ty) * 3 + tx);
              const char* bt = sW + (tapp * 2 + ch) * (32 * 64);
              const half8 bh = *reinterpret_cast<const half8*>(bt + boffh);
              const half8 bl = *reinterpret_cast<const half8*>(bt + boffl);
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                acc[cls][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[cls][mt], 0, 0, 0);
                acc[cls][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[cls][mt], 0, 0, 0);
                acc[cls][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[cls][mt], 0, 0, 0);
              }
            }
          });
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (d > 0) store_class(D.value - 1);
    });
    __builtin_amdgcn_sched_barrier(0);
    fetch_add0(7);
    store_class(7);
    };
    const bool whole = (ox0 + 8 <= a.Wo) & (oy0 + 8 <= a.Ho) & (oz0 + 4 <= a.Do);   // block-uniform
    if (whole) {
      if (a.add0)
        phase(std::true_type{}, std::true_type{});
      else
        phase(std::true_type{}, std::false_type{});
    } else {
      phase(std::false_type{}, std::false_type{});
    }
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

bool dgrad_s2_fused_ok(const adell_conv3d_desc* d) {
  return d && d->C1 == 0 && d->C0 == 32 && d->Cout == 32 && d->KD == 3 && d->KH == 3 && d->KW == 3 &&
         d->SD == 2 && d->SH == 2 && d->SW == 2 && d->PD == 1 && d->PH == 1 && d->PW == 1 &&
         d->D % 2 == 0 && d->H % 2 == 0 && d->W % 2 == 0 && d->N > 0 && d->D > 0 && d->H > 0 &&
         d->W > 0 && d->Do == d->D / 2 && d->Ho == d->H / 2 && d->Wo == d->W / 2 &&
         (size_t)d->D * d->H * d->W * 32 < ((size_t)1 << 29);   // 32-bit offsets inside an item
}

}  // namespace

// 1 when adell_conv3d_bwd_data_s2_fused takes this layer (32 -> 32 channels, k = 3, stride 2,
// padding 1, even input dims, one destination).
extern "C" int adell_conv3d_bwd_data_s2_fused_applicable(const adell_conv3d_desc* d) {
  return dgrad_s2_fused_ok(d) ? 1 : 0;
}

// dX of such a layer in one launch. w_split_bwd / wscale: adell_pack_weight_f16x3 mode 1 of the
// full [32][32][3][3][3] weight (the pack the zero-insertion backward-data call takes); add0: null
// or a dX-shaped tensor added in the epilogue; dy_absmax: optional by-product as in the other calls.
extern "C" int adell_conv3d_bwd_data_s2_fused(const adell_conv3d_desc* d, const float* dy,
                                              const void* w_split_bwd, const float* wscale,
                                              const float* add0, float* dx, uint32_t* dy_absmax,
                                              void* stream) {
  ADELL_REQUIRE(dgrad_s2_fused_ok(d),
                "conv_bwd_data_s2_fused: needs 32 -> 32 channels, k = 3, stride 2, padding 1, even dims");
  ADELL_REQUIRE(dy && w_split_bwd && wscale && dx, "conv_bwd_data_s2_fused: null pointer");
  DgradS2Args a;
  a.dy = dy;
  a.wpack = reinterpret_cast<const char*>(w_split_bwd);
  a.wscale = wscale;
  a.add0 = add0;
  a.dx = dx;
  a.amax_out = dy_absmax;
  a.N = d->N; a.Do = d->Do; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ntx = adell_cdiv(d->Wo, 8);
  a.nty = adell_cdiv(d->Ho, 8);
  a.ntz = adell_cdiv(d->Do, 4);
  const long nbricks = (long)d->N * a.ntx * a.nty * a.ntz;
  ADELL_REQUIRE(nbricks < 0x7fffffffL, "conv_bwd_data_s2_fused: too many bricks");
  a.nbricks = (int)nbricks;
  const int grid = (int)(nbricks < cu_count() ? nbricks : cu_count());   // one block per CU
  auto launch = [&](auto kern) -> int {
    // (the attribute is per kernel symbol; set on every call: a cheap host-side table write)
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, (hipStream_t)stream, a);
    return ADELL_OK;
  };
  int rc = ADELL_OK;
#ifdef ADELL_DEBUG
  switch (g_adell_tune.igemm_dbg) {
    case 1: rc = launch(adell_dgrad_s2_fused_kernel<1>); break;
    case 2: rc = launch(adell_dgrad_s2_fused_kernel<2>); break;
    case 3: rc = launch(adell_dgrad_s2_fused_kernel<3>); break;
    case 4: rc = launch(adell_dgrad_s2_fused_kernel<4>); break;
    case 8: rc = launch(adell_dgrad_s2_fused_kernel<8>); break;
    case 16: rc = launch(adell_dgrad_s2_fused_kernel<16>); break;
    case 27: rc = launch(adell_dgrad_s2_fused_kernel<27>); break;
    default: rc = launch(adell_dgrad_s2_fused_kernel<0>); break;
  }
#else
  rc = launch(adell_dgrad_s2_fused_kernel<0>);
#endif
  if (rc != ADELL_OK) return rc;
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
