// Depthwise 7 x 7 x 7 convolution (ConvNeXt's dwconv, res_blocks.py:540-557: groups == channels,
// stride 1, padding 3) on the MATRIX pipe: forward, and backward-data as the same kernel on the
// flipped taps.
//
// A depthwise stencil has no channel reduction to feed an MFMA with, but for a fixed (kz, ky) its
// seven taps along x are a banded Toeplitz matrix:
//
//   Y[y][x] += sum_k X[z + kz - 3][y + ky - 3][k - 3] * T[k][x],   T[k][x] = w[kz][ky][k - x] (0 <= k - x < 7)
//
// i.e. a [16 rows y] x [K = 32: 22 input columns, zero-padded] x [16 columns x] product per channel
// and (kz, ky): 49 accumulating v_mfma_f32_16x16x32_f16 steps per output plane, each as the three
// products of the f16x3 split (hi hi + lo hi + hi lo, fp32 accumulate: the arithmetic of the conv /
// GEMM kernels). 7 useful taps of 32 k-slots is 22 % of the f16 rate -- 180 TF-equivalent, against
// the ~37 TF of fp32 FMAs the vector-ALU kernels (csrc/ssl.hip) reach at this layer.
//
// Work split: block = (item, 4 consecutive channels), wave = channel. Each wave marches along z with
// G = 4 output planes in flight (their accumulators share every B fragment), a ring of G + 6 input
// planes of its channel in LDS: rows of 22 + 10 k-slots, [32 halfs hi | 32 halfs lo], zero halo
// written once. All 256 threads stage a plane together -- a thread's float4 is the four channels of
// one voxel -- with the next group's four planes in flight in registers under the MFMAs. Operand
// scales: a power of two per (item, channel) from the block's own absmax pre-pass over its 4-channel
// column (L2-resident for the staging that follows), one per channel for the 343 taps.
//
// B fragments: lane (x = lane & 15, k-group = lane >> 4) needs w[8 kg + i - x] for i = 0..7, zero
// outside 0..6: eight 2-byte LDS reads from a 9-entry zero-ended table per (kz, ky), at lane offsets
// that do not depend on the tap (clamped once) -- amortised over the G planes.
#include <type_traits>
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DM_G = 4;                        // output planes in flight
constexpr int DM_CG = 4;                       // channels per block (= waves)
constexpr int DM_ROWS = 22;                    // input rows of a plane: y - 3 .. y + 18
constexpr int DM_ROWB = 144;                   // bytes per row: 32 halfs hi | 32 halfs lo | 16 pad (banks)
constexpr int DM_PLANEB = DM_ROWS * DM_ROWB;
constexpr int DM_RING = DM_G + 6;
constexpr int DM_TABH = 49 * 18;               // halfs: [tap (kz, ky)][9 entries: 0 w0..w6 0][hi, lo] (a 32-bit word per entry)
constexpr int DM_CHB = DM_RING * DM_PLANEB + ((DM_TABH * 2 + 15) & ~15);
constexpr int DM_OUTROW = 68;                  // floats per (plane, y) row of the output staging: 16 x 4 channels + 4 pad
constexpr int DM_OUTB = DM_G * 16 * DM_OUTROW * 4;
constexpr int DM_LDS = DM_CG * DM_CHB + DM_OUTB + 64;    // + output staging + block reduction scratch
static_assert(DM_LDS <= 160 * 1024, "one block per CU");

struct DwMfmaArgs {
  const float* x;
  const float* w;   // [C][7][7][7]
  const float* b;   // [C] or null
  float* y;
  int N, C, D, H, W, flip;
  int total;        // work items = N * (C / 4): item n = work / (C / 4), channels 4 (work % (C / 4)) ..
};

__device__ __forceinline__ int dm_scale_exp(float amax) {
  const unsigned bits = __float_as_uint(amax);
  const int ebits = (bits >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);
  return k > 100 ? 100 : (k < -100 ? -100 : k);
}

}  // namespace

// REGCOL (D <= 16): the block's whole 4-channel column of the item -- 16 planes x one float4 per
// thread -- is loaded ONCE into registers: the absmax pass, the priming and the staging of every
// group read registers. A voxel's four channels are 16 bytes of a 384-byte row, so every lane of
// every load is its own 128-byte line request; measured with the phases switched off one at a time
// (64 x 96 x 16^3): absmax pass 103 us, the groups' loads 69 us, stores 60 us, staging 40 us, the
// MFMA loop 122 us -- the line requests, not the matrix pipe, were 2/3 of the kernel.
template <bool REGCOL>
__global__ __launch_bounds__(256, 1) void adell_dw_mfma_kernel(DwMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sOut = reinterpret_cast<float*>(smem + DM_CG * DM_CHB);   // [plane][y][x][channel], rows padded
  float* sRed = reinterpret_cast<float*>(smem + DM_CG * DM_CHB + DM_OUTB);   // [4 waves][4 channels]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vy = tid >> 4, vx = tid & 15;            // the voxel of a plane this thread stages
  const bool vok = vy < a.H && vx < a.W;
  const int groups = a.C / DM_CG;
  const size_t zstride = (size_t)a.H * a.W * a.C;
  // this thread's voxel of plane z of work item `work` (zeros outside the volume / past the last item)
  auto load_plane = [&](int work, int z) -> float4 {
    // (a select on the loaded value, not a branch around the load: the loads of a group stay in flight together)
    const bool ok = vok && z >= 0 && z < a.D && work < a.total;
    const int wn = work / groups, wc0 = (work - wn * groups) * DM_CG;
    const float4 f = *reinterpret_cast<const float4*>(
        ok ? a.x + ((size_t)wn * a.D * a.H * a.W + (size_t)vy * a.W + vx) * a.C + wc0 + (size_t)z * zstride : a.x);
    return make_float4(ok ? f.x : 0.f, ok ? f.y : 0.f, ok ? f.z : 0.f, ok ? f.w : 0.f);
  };

  // ---- zero the rings (the halo cells are never written again) ---------------------------------
  for (int i = tid; i < DM_CG * DM_CHB / 16; i += 256)
    reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // REGCOL: persistent blocks -- the column of the NEXT work item is loaded into a second register
  // set under this item's MFMAs (a column's 16 loads per thread are 4 096 line requests per block:
  // ~17 us that nothing overlapped when every block loaded its own column first)
  float4 colnext[REGCOL ? 16 : 1];
  if constexpr (REGCOL) {
#pragma unroll
    for (int z = 0; z < 16; ++z) colnext[z] = load_plane((int)blockIdx.x, z);
  }
  for (int work = blockIdx.x; work < a.total; work += gridDim.x) {
  const int n = work / groups, c0 = (work - n * groups) * DM_CG;
  const size_t item = (size_t)n * a.D * a.H * a.W;

  // ---- operand scale per channel: absmax of this block's 4-channel column of the item ----------
  float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 col[REGCOL ? 16 : 1];
  if constexpr (REGCOL) {
#pragma unroll
    for (int z = 0; z < 16; ++z) col[z] = colnext[z];
#pragma unroll
    for (int z = 0; z < 16; ++z) colnext[z] = load_plane(work + (int)gridDim.x, z);
#pragma unroll
    for (int z = 0; z < 16; ++z) {
      mx.x = fmaxf(mx.x, fabsf(col[z].x)); mx.y = fmaxf(mx.y, fabsf(col[z].y));
      mx.z = fmaxf(mx.z, fabsf(col[z].z)); mx.w = fmaxf(mx.w, fabsf(col[z].w));
    }
  } else {
    for (int z0 = 0; z0 < a.D; z0 += 8) {
      float4 f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = load_plane(work, z0 + u);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        mx.x = fmaxf(mx.x, fabsf(f[u].x)); mx.y = fmaxf(mx.y, fabsf(f[u].y));
        mx.z = fmaxf(mx.z, fabsf(f[u].z)); mx.w = fmaxf(mx.w, fabsf(f[u].w));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
    mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
  }
  __syncthreads();   // the zero fill is complete / the previous item's last group is consumed
  if (lane == 0) {
    sRed[wave * 4 + 0] = mx.x; sRed[wave * 4 + 1] = mx.y;
    sRed[wave * 4 + 2] = mx.z; sRed[wave * 4 + 3] = mx.w;
  }
  __syncthreads();
  int kx4[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
    kx4[ch] = dm_scale_exp(fmaxf(fmaxf(sRed[ch], sRed[4 + ch]), fmaxf(sRed[8 + ch], sRed[12 + ch])));
  float sx4[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) sx4[ch] = __int_as_float((kx4[ch] + 127) << 23);

  // ---- this wave's channel: tap table (split, scaled, zero-ended, flipped for backward-data) ---
  char* R = smem + wave * DM_CHB;
  _Float16* tab = reinterpret_cast<_Float16*>(R + DM_RING * DM_PLANEB);
  const int ch_w = c0 + wave;
  float wv[6];
  float wmax = 0.f;
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int i = lane + 64 * u;
    wv[u] = (i < 343) ? a.w[(size_t)ch_w * 343 + (a.flip ? 342 - i : i)] : 0.f;
    wmax = fmaxf(wmax, fabsf(wv[u]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
  const int kw = dm_scale_exp(wmax);
  const float sw = __int_as_float((kw + 127) << 23);
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int i = lane + 64 * u;
    if (i < 343) {
      const int t = i / 7, kxx = i - 7 * t;
      const float v = wv[u] * sw;
      const _Float16 h = (_Float16)v;
      tab[t * 18 + 2 * (1 + kxx)] = h;
      tab[t * 18 + 2 * (1 + kxx) + 1] = (_Float16)(v - (float)h);
    }
  }
  // (entries 0 and 8 of every tap stay zero from the fill)

  // staging of one plane into ring slot `slot` (all four channel regions)
  auto store_plane = [&](int slot, const float4& f) {
    if (!vok) return;
    const float v[4] = {f.x * sx4[0], f.y * sx4[1], f.z * sx4[2], f.w * sx4[3]};
    char* p = smem + slot * DM_PLANEB + (vy + 3) * DM_ROWB + (vx + 3) * 2;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const _Float16 h = (_Float16)v[ch];
      *reinterpret_cast<_Float16*>(p + ch * DM_CHB) = h;
      *reinterpret_cast<_Float16*>(p + ch * DM_CHB + 64) = (_Float16)(v[ch] - (float)h);
    }
  };

  // ---- prime the ring: input planes -3 .. 6 into slots 0 .. 9 (planes < 0: zeros, over whatever
  // the previous work item left in those slots) -----------------------------------------------------
#pragma unroll
  for (int u = 0; u < 3; ++u) store_plane(u, make_float4(0.f, 0.f, 0.f, 0.f));
  if constexpr (REGCOL) {
#pragma unroll
    for (int u = 0; u < 7; ++u) store_plane(3 + u, col[u]);
  } else {
    float4 f[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) f[u] = load_plane(work, u);
#pragma unroll
    for (int u = 0; u < 7; ++u) store_plane(3 + u, f[u]);
  }
  __syncthreads();

  // lane constants of the fragments
  const int mrow = lane & 15, kq = lane >> 4;
  int off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int d = 8 * kq + i - mrow;                    // tap index along x of B[k = 8 kq + i][x = mrow]
    d = d < -1 ? -1 : (d > 7 ? 7 : d);
    off[i] = d + 1;
  }
  const char* arow = R + mrow * DM_ROWB + kq * 16;   // A[m = y][k-group]: + slot, + ky rows
  // the two scales undone one after the other: each exponent is clamped to +-100, their sum
  // can pass the exponent range of a float (tiny dY times near-zero taps)
  const float oscale = __int_as_float((127 - kx4[wave]) << 23), oscale2 = __int_as_float((127 - kw) << 23);
  const float bias = a.b ? a.b[ch_w] : 0.f;

  const int ngroups = (a.D + DM_G - 1) / DM_G;
  int s0 = 0;                                       // ring slot of input plane 4 g - 3
  // one group of four output planes; `gq`: the group index, an integral_constant in the REGCOL form
  // (at most four groups: the register column is indexed statically)
  auto do_group = [&](auto gq) {
    const int g = gq;
    // the next group's four new planes (4 g + 7 .. 4 g + 10): from the register column, or loaded
    // under this group's MFMAs
    float4 pf[DM_G];
#pragma unroll
    for (int u = 0; u < DM_G; ++u) {
      if constexpr (REGCOL) {
        constexpr int z = 4 * decltype(gq)::value + 7;
        pf[u] = (z + u < 16) ? col[(z + u) & 15] : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        pf[u] = load_plane(work, 4 * g + 7 + u);
      }
    }

    f32x4 acc[DM_G];
#pragma unroll
    for (int j = 0; j < DM_G; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Loop order: ky outermost, its seven B fragment pairs (one per kz) in registers; then the ten
    // ring planes in turn -- ONE A fragment pair per plane serves every (output plane j, kz) with
    // j + kz = the plane's position, up to four MFMA triples. (With (kz, ky) outermost every output
    // plane re-read its A rows per tap: 8 KB of LDS reads per 12 MFMAs and wave, four waves on one
    // LDS -- the kernel ran LDS-bound, 10 % slower than the vector-ALU form.)
    int slotb[DM_RING];
#pragma unroll
    for (int t = 0; t < DM_RING; ++t) {
      int sl = s0 + t;
      sl = sl >= DM_RING ? sl - DM_RING : sl;
      slotb[t] = sl * DM_PLANEB;
    }
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      // B fragments of the seven kz at this ky: one 32-bit LDS read per entry brings (hi, lo); two
      // byte-permutes per entry pair sort them into the hi and the lo fragment
      half8 bh[7], bl[7];
      const uint32_t* tp = reinterpret_cast<const uint32_t*>(tab) + ky * 9;
#pragma unroll
      for (int kz = 0; kz < 7; ++kz) {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = tp[kz * 63 + off[i]];
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ph[q] = __builtin_amdgcn_perm(e[2 * q + 1], e[2 * q], 0x05040100u);   // hi halves of the pair
          pl[q] = __builtin_amdgcn_perm(e[2 * q + 1], e[2 * q], 0x07060302u);   // lo halves
        }
        __builtin_memcpy(&bh[kz], ph, 16);
        __builtin_memcpy(&bl[kz], pl, 16);
      }
      const char* prow = arow + ky * DM_ROWB;
#pragma unroll
      for (int t = 0; t < DM_RING; ++t) {
        const half8 ah = *reinterpret_cast<const half8*>(prow + slotb[t]);
        const half8 al = *reinterpret_cast<const half8*>(prow + slotb[t] + 64);
        // input plane 4 g - 3 + t feeds output plane 4 g + j through tap kz = t - j; the three
        // products of the split are issued across the accumulators (consecutive MFMAs independent)
#pragma unroll
        for (int j = 0; j < DM_G; ++j)
          if (t - j >= 0 && t - j <= 6)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[t - j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DM_G; ++j)
          if (t - j >= 0 && t - j <= 6)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[t - j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < DM_G; ++j)
          if (t - j >= 0 && t - j <= 6)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[t - j], acc[j], 0, 0, 0);
      }
    }
    // C / D layout: column x = lane & 15, rows y = 4 (lane >> 4) + r. The four channels of a voxel
    // meet in LDS and leave as ONE 16-byte store (a wave's own 4-byte stores would be 64 partial
    // lines per instruction: the channel stride is the row of the tensor)
#pragma unroll
    for (int j = 0; j < DM_G; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sOut[(j * 16 + 4 * kq + r) * DM_OUTROW + mrow * 4 + wave] = acc[j][r] * oscale * oscale2 + bias;
    __syncthreads();     // the staging is complete, and every wave has read the four oldest planes
    if (vok) {
#pragma unroll
      for (int j = 0; j < DM_G; ++j) {
        const int zo = 4 * g + j;
        if (zo < a.D)
          *reinterpret_cast<float4*>(a.y + (item + ((size_t)zo * a.H + vy) * a.W + vx) * a.C + c0) =
              *reinterpret_cast<const float4*>(sOut + (j * 16 + vy) * DM_OUTROW + vx * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < DM_G; ++u) {
      int s = s0 + u;    // slots of planes 4 g - 3 .. 4 g: overwritten by planes 4 g + 7 .. 4 g + 10
      s = s >= DM_RING ? s - DM_RING : s;
      store_plane(s, pf[u]);
    }
    s0 += DM_G;
    s0 = s0 >= DM_RING ? s0 - DM_RING : s0;
    __syncthreads();
  };
  if constexpr (REGCOL) {
    do_group(std::integral_constant<int, 0>{});
    if (ngroups > 1) do_group(std::integral_constant<int, 1>{});
    if (ngroups > 2) do_group(std::integral_constant<int, 2>{});
    if (ngroups > 3) do_group(std::integral_constant<int, 3>{});
  } else {
    for (int g = 0; g < ngroups; ++g) do_group(g);
  }
  }   // work items
}

// K = 7 cubic, rows of 9 .. 16 voxels in x and y, channels in fours, 16-byte aligned tensors
extern "C" int adell_dw_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                const float* x, const float* y) {
  return KD == 7 && KH == 7 && KW == 7 && W > 8 && W <= 16 && H > 8 && H <= 16 && D >= 1 &&
         C % DM_CG == 0 && N <= 65535 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 &&
         !g_adell_tune.dw_nomfma;
}

extern "C" int adell_dw_mfma_launch(const float* x, const float* w, const float* b, float* y, int N,
                                    int C, int D, int H, int W, int flip, void* stream) {
  ADELL_REQUIRE(adell_dw_mfma_ok(N, C, D, H, W, 7, 7, 7, x, y), "dw_mfma: shape not covered");
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dw_mfma_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dw_mfma_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const long total = (long)N * (C / DM_CG);
  ADELL_REQUIRE(total <= 0x7fffffffL - 4096, "dw_mfma: too many work items");
  DwMfmaArgs a = {x, w, b, y, N, C, D, H, W, flip, (int)total};
  if (D <= 16) {
    // one block per CU (LDS): persistent blocks, each prefetching its next column
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      ADELL_CHECK_HIP(hipGetDevice(&dev));
      ADELL_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
      cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = total < cus ? (int)total : cus;
    hipLaunchKernelGGL(adell_dw_mfma_kernel<true>, dim3(grid), dim3(256), DM_LDS, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(adell_dw_mfma_kernel<false>, dim3((unsigned)total), dim3(256), DM_LDS,
                       (hipStream_t)stream, a);
  }
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
