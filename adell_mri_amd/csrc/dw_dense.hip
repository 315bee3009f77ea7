// Depthwise 7^3 convolution on volumes of at most 4 x 4 x 4 voxels (ConvNeXt's third stage: 4^3 at
// 384 channels, res_blocks.py:540-557) as a DENSE per-channel matrix product on the f16x3 MFMA.
//
// With 'same' padding 3 every output voxel of such a volume sees every input voxel: per channel
//   Y[item][o] = b + sum_i X[item][i] * M[i][o],   M[i][o] = w[iz - oz + 3][iy - oy + 3][ix - ox + 3]
// a [16 items] x [64 inputs] x [64 outputs] product -- 4 096 multiply-adds per item where the stencil
// form walks 343 taps per output (21 952, five sixths of them on the zero padding). Forward, and
// backward-data as the same kernel on the flipped taps.
//
// Block = (16 items, 16 or 4 channels), wave = channel. The wave builds its 8 B fragment pairs (2 k-steps x
// 4 output tiles, hi and lo) once from the channel's 343 taps and keeps them in registers; the 16 x 64
// inputs of its channel are staged split in LDS (a thread's float4 is the four channels of one
// voxel, all of a block's loads in flight together); 24 MFMAs per wave; the four channels of an
// output voxel meet in LDS and leave as one 16-byte store. Scales: a power of two per (block,
// channel) from the block's own inputs, one per channel for the taps.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DD_IT = 16;                      // items per block (one M tile)
constexpr int DD_AROW = 272;                   // bytes per (channel, item): 64 halfs hi | 64 halfs lo | 16 pad
constexpr int DD_AB = DD_IT * DD_AROW;         // per channel
constexpr int DD_TAPW = 344;                   // packed (hi | lo << 16) tap words per channel (+ 1 zero)
// CG channels (= waves) per block: 4, or 16 (a voxel's 16 channels are a 64-byte run shared by four
// lanes of a load, where four channels are 16 bytes of a line per lane -- measured slower at the
// shapes in hand, see the launch)
template <int CG> constexpr int dd_outrow() { return 64 * CG + 4; }   // floats per item of the output staging
template <int CG> constexpr int dd_lds() {
  // the output staging [item][o][channel] reuses the input region (16 x dd_outrow floats <= CG x DD_AB for CG = 16)
  return (CG * DD_AB > DD_IT * dd_outrow<CG>() * 4 ? CG * DD_AB : DD_IT * dd_outrow<CG>() * 4) +
         CG * DD_TAPW * 4 + (CG * (CG / 4) * 4 + CG) * 4 + 64;
}

struct DwDenseArgs {
  const float* x;
  const float* w;   // [C][7][7][7]
  const float* b;
  float* y;
  int N, C, D, H, W, flip;
};

__device__ __forceinline__ int dd_scale_exp(float amax) {
  const unsigned bits = __float_as_uint(amax);
  const int ebits = (bits >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);
  return k > 100 ? 100 : (k < -100 ? -100 : k);
}

}  // namespace

template <int CG>
__global__ __launch_bounds__(64 * CG) void adell_dw_dense_kernel(DwDenseArgs a) {
  constexpr int Q = CG / 4, NT = 64 * CG;
  constexpr int AREG = CG * DD_AB > DD_IT * dd_outrow<CG>() * 4 ? CG * DD_AB : DD_IT * dd_outrow<CG>() * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* sTap = reinterpret_cast<uint32_t*>(smem + AREG);
  float* sRed = reinterpret_cast<float*>(smem + AREG + CG * DD_TAPW * 4);   // [wave][quad][4], then [wave] exponents
  float* sOut = reinterpret_cast<float*>(smem);                            // after the MFMAs: over the inputs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * CG, n0 = blockIdx.y * DD_IT;
  const int V = a.D * a.H * a.W;                 // <= 64

  // ---- loads: four float4 per thread over [item][voxel][channel quad], quads fastest ------------
  float4 f[4];
  const int quad = tid % Q;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + u * NT, v = (idx / Q) & 63, it = idx / (64 * Q);
    const bool ok = v < V && n0 + it < a.N;
    const float4 g = *reinterpret_cast<const float4*>(
        ok ? a.x + ((size_t)(n0 + it) * V + v) * a.C + c0 + 4 * quad : a.x);
    f[u] = make_float4(ok ? g.x : 0.f, ok ? g.y : 0.f, ok ? g.z : 0.f, ok ? g.w : 0.f);
  }
  // ---- this wave's channel: taps split, scaled, packed (flipped for backward-data) --------------
  const int ch_w = c0 + wave;
  {
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
    const int kw = dd_scale_exp(wmax);
    const float sw = __int_as_float((kw + 127) << 23);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = lane + 64 * u;
      if (i < DD_TAPW) {
        const float v = wv[u] * sw;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        sTap[wave * DD_TAPW + i] = (uint32_t)__builtin_bit_cast(unsigned short, h) |
                                   ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
      }
    }
    if (lane == 0) sRed[CG * Q * 4 + wave] = __int_as_float(kw);   // (bit pattern of the exponent)
  }
  // ---- operand scale per channel: absmax of the block's inputs (lanes of one quad together) -----
  float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    mx.x = fmaxf(mx.x, fabsf(f[u].x)); mx.y = fmaxf(mx.y, fabsf(f[u].y));
    mx.z = fmaxf(mx.z, fabsf(f[u].z)); mx.w = fmaxf(mx.w, fabsf(f[u].w));
  }
#pragma unroll
  for (int o = 32; o >= Q; o >>= 1) {
    mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
    mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
  }
  if (lane < Q) {
    float* r = sRed + (wave * Q + lane) * 4;
    r[0] = mx.x; r[1] = mx.y; r[2] = mx.z; r[3] = mx.w;
  }
  __syncthreads();
  auto chan_exp = [&](int ch) {     // channel ch = 4 quad + j of the block
    float m = 0.f;
    for (int w = 0; w < CG; ++w) m = fmaxf(m, sRed[(w * Q + (ch >> 2)) * 4 + (ch & 3)]);
    return dd_scale_exp(m);
  };
  int kx4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) kx4[j] = chan_exp(4 * quad + j);
  const int kx_w = chan_exp(wave);
  // ---- stage the inputs: [channel][item][hi 64 | lo 64] ------------------------------------------
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + u * NT, v = (idx / Q) & 63, it = idx / (64 * Q);
    const float vals[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
    char* p = smem + (4 * quad) * DD_AB + it * DD_AROW + v * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = vals[j] * __int_as_float((kx4[j] + 127) << 23);
      const _Float16 h = (_Float16)s;
      *reinterpret_cast<_Float16*>(p + j * DD_AB) = h;
      *reinterpret_cast<_Float16*>(p + j * DD_AB + 128) = (_Float16)(s - (float)h);
    }
  }
  __syncthreads();

  // ---- B fragments of this channel: M[i][o] for i = 32 ks + 8 kq + j, o = 16 nt + (lane & 15) -----
  const int col = lane & 15, kq = lane >> 4;
  const uint32_t* tap = sTap + wave * DD_TAPW;
  const int HW = a.H * a.W;
  half8 bh[2][4], bl[2][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int o = nt * 16 + col;
    const int oz = o / HW, oy = (o - oz * HW) / a.W, ox = o - oz * HW - oy * a.W;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = ks * 32 + 8 * kq + j;
        const int iz = i / HW, iy = (i - iz * HW) / a.W, ix = i - iz * HW - iy * a.W;
        // (taps are in range by construction: |difference| <= 3 for axes of at most 4 voxels)
        const int t = ((iz - oz + 3) * 7 + (iy - oy + 3)) * 7 + (ix - ox + 3);
        e[j] = (i < V && o < V) ? tap[t] : 0u;
      }
      uint32_t ph[4], pl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ph[q] = __builtin_amdgcn_perm(e[2 * q + 1], e[2 * q], 0x05040100u);
        pl[q] = __builtin_amdgcn_perm(e[2 * q + 1], e[2 * q], 0x07060302u);
      }
      __builtin_memcpy(&bh[ks][nt], ph, 16);
      __builtin_memcpy(&bl[ks][nt], pl, 16);
    }
  }
  // ---- A fragments: item = lane & 15, inputs 32 ks + 8 kq .. + 7 ---------------------------------
  const char* ap = smem + wave * DD_AB + col * DD_AROW + kq * 16;
  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const half8 ah = *reinterpret_cast<const half8*>(ap + ks * 64);
    const half8 al = *reinterpret_cast<const half8*>(ap + 128 + ks * 64);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks][nt], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks][nt], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks][nt], acc[nt], 0, 0, 0);
  }
  // ---- outputs: D[item = 4 kq + r][o = 16 nt + col] -> [item][o][channel] in LDS -> 16-byte stores --
  const int kw = __float_as_int(sRed[CG * Q * 4 + wave]);
  // the two scales undone one after the other: each exponent is clamped to +-100, their sum
  // can pass the exponent range of a float (tiny dY times near-zero taps)
  const float oscale = __int_as_float((127 - kx_w) << 23), oscale2 = __int_as_float((127 - kw) << 23);
  const float bias = a.b ? a.b[ch_w] : 0.f;
  __syncthreads();     // every wave has read its A fragments: the staging may overwrite the inputs
  constexpr int OUTROW = dd_outrow<CG>();
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sOut[(4 * kq + r) * OUTROW + (nt * 16 + col) * CG + wave] = acc[nt][r] * oscale * oscale2 + bias;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + u * NT, v = (idx / Q) & 63, it = idx / (64 * Q);
    if (v < V && n0 + it < a.N)
      *reinterpret_cast<float4*>(a.y + ((size_t)(n0 + it) * V + v) * a.C + c0 + 4 * quad) =
          *reinterpret_cast<const float4*>(sOut + it * OUTROW + v * CG + 4 * quad);
  }
}

// 7^3 taps on volumes of at most 4 x 4 x 4 voxels (more than 16 of them), channels in fours
extern "C" int adell_dw_dense_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                 const float* x, const float* y) {
  return KD == 7 && KH == 7 && KW == 7 && D <= 4 && H <= 4 && W <= 4 && D * H * W > 16 &&
         C % 4 == 0 && (N + DD_IT - 1) / DD_IT <= 65535 &&
         ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && !g_adell_tune.dw_nomfma;
}

template <int CG>
static int adell_dw_dense_go(const DwDenseArgs& a, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dw_dense_kernel<CG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(adell_dw_dense_kernel<CG>, dim3(a.C / CG, (a.N + DD_IT - 1) / DD_IT), dim3(64 * CG),
                     dd_lds<CG>(), st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

extern "C" int adell_dw_dense_launch(const float* x, const float* w, const float* b, float* y, int N,
                                     int C, int D, int H, int W, int flip, void* stream) {
  ADELL_REQUIRE(adell_dw_dense_ok(N, C, D, H, W, 7, 7, 7, x, y), "dw_dense: shape not covered");
  DwDenseArgs a = {x, w, b, y, N, C, D, H, W, flip};
  // (16 channels per block quarter the line requests but leave 96 blocks of 1 024 threads for 256 CUs at
  // ConvNeXt's 64 crops x 384 channels: 30.4 us against 24.4 -- A/B switch only)
  return adell_dw_dense_go<4>(a, (hipStream_t)stream);
}
