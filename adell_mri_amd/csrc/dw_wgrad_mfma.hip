// Weight gradient of the depthwise 7 x 7 x 7 convolution on the MATRIX pipe (companion of
// csrc/dw_mfma.hip; ConvNeXt's dwconv, res_blocks.py:540-557).
//
//   dW[c][kz][ky][kx] = sum over (n, z, y, x) of dY[n][z][y][x][c] * X[n][z + kz - 3][y + ky - 3][x + kx - 3][c]
//
// For a fixed (kz, ky) the sum over the rows (n, z, y) of  P[xo][k] = dY[row][xo] * X[row'][k - 3]
// is a matrix product with the ROWS as reduction dimension -- M = 16 output columns xo, N = 32 input
// columns k (22 used), K = 32 rows per v_mfma_f32_16x16x32_f16 (16 y of two consecutive planes) --
// and dW[kz][ky][kx] is the sum of the diagonal k - xo = kx of P. Same product count as the forward
// (7 useful diagonals of 32 columns), f16x3 split, fp32 accumulate.
//
// Block = (4 channels, a chunk of items), wave = channel, ONE block per CU: the 49 x 2 accumulator
// tiles of a channel (392 registers) stay in the wave's register file for the whole block. Per pair
// of dY planes the A fragments (dY transposed: [xo][y]) are read once and meet all 49 taps; the B
// fragments of the seven ky at one (kz, column) are windows of ONE run of 14 packed (hi | lo) words
// along y of the transposed X plane ([k][y], ring of eight planes) -- 28 LDS reads per kz and wave.
// Operand scales: one power of two per tensor (absmax words made by adell_absmax_f32 before the
// launch), so that the accumulators never need rescaling between items.
// The blocks write per-chunk partial sums [chunk][C][344] (column 343: the bias gradient, summed
// from the staged dY); adell_dw_wgrad_reduce_kernel (csrc/ssl.hip) folds them in chunk order.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WM_CG = 4;                      // channels per block (= waves)
constexpr int WM_XROW = 25;                   // words per (plane, column k) run along y: 22 rows + pad (banks)
constexpr int WM_XPLANE = 32 * WM_XROW * 4;   // bytes: [k = 32][y rows] packed (hi | lo << 16)
constexpr int WM_XRING = 8;                   // X planes z0 - 3 .. z0 + 4 of a dY plane pair
constexpr int WM_DROW = 48;                   // bytes per (plane, xo) run along y: 16 halfs + pad (banks)
constexpr int WM_DPLANE = 16 * WM_DROW;       // one of hi / lo
constexpr int WM_DYB = 2 * 2 * WM_DPLANE;     // [plane of the pair][hi | lo]
constexpr int WM_CHB = WM_XRING * WM_XPLANE + WM_DYB;
constexpr int WM_LDS = WM_CG * WM_CHB + 64;
static_assert(WM_LDS <= 160 * 1024, "one block per CU");

struct DwWgMfmaArgs {
  const float* x;
  const float* dy;
  float* part;              // [chunks][C][344]
  const uint32_t* amax;     // [2]: absmax bits of x, dy
  int N, C, D, H, W, items_per_chunk;
};

__device__ __forceinline__ int wm_scale_exp(uint32_t bits) {
  const int ebits = (bits >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);
  // +-63: the two tensors' exponents are undone as ONE factor 2^-(kx + kd), whose exponent field
  // must stay inside a float's (a tensor whose absmax is below 2^-50 keeps fewer bits: it is zero
  // for every purpose of a gradient)
  return k > 63 ? 63 : (k < -63 ? -63 : k);
}

}  // namespace

__global__ __launch_bounds__(256, 1) void adell_dw_wgrad_mfma_kernel(DwWgMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sRed = reinterpret_cast<float*>(smem + WM_CG * WM_CHB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * WM_CG, chunk = blockIdx.y;
  const int vy = tid >> 4, vx = tid & 15;
  const bool vok = vy < a.H && vx < a.W;
  const int n_beg = chunk * a.items_per_chunk;
  int n_end = n_beg + a.items_per_chunk;
  n_end = n_end < a.N ? n_end : a.N;
  const size_t zstride = (size_t)a.H * a.W * a.C;
  const size_t voff = ((size_t)vy * a.W + vx) * a.C + c0;
  auto load_plane = [&](const float* t, int n, int z) -> float4 {
    const bool ok = vok && z >= 0 && z < a.D && n < n_end;
    const float4 f = *reinterpret_cast<const float4*>(ok ? t + ((size_t)n * a.D + z) * zstride + voff : t);
    return make_float4(ok ? f.x : 0.f, ok ? f.y : 0.f, ok ? f.z : 0.f, ok ? f.w : 0.f);
  };
  const int kxs = wm_scale_exp(a.amax[0]), kds = wm_scale_exp(a.amax[1]);
  const float sx = __int_as_float((kxs + 127) << 23), sd = __int_as_float((kds + 127) << 23);

  for (int i = tid; i < WM_CG * WM_CHB / 16; i += 256)
    reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  // staging (all four channel regions): X plane -> [k = x + 3][y + 3] packed words; dY plane ->
  // [xo][y] halfs, hi and lo apart
  auto store_x = [&](int slot, const float4& f) {
    if (!vok) return;
    const float v[4] = {f.x * sx, f.y * sx, f.z * sx, f.w * sx};
    char* p = smem + slot * WM_XPLANE + ((vx + 3) * WM_XROW + vy + 3) * 4;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const _Float16 h = (_Float16)v[ch];
      const _Float16 l = (_Float16)(v[ch] - (float)h);
      *reinterpret_cast<uint32_t*>(p + ch * WM_CHB) =
          (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
    }
  };
  auto store_dy = [&](int pl, const float4& f) {
    if (!vok) return;
    const float v[4] = {f.x * sd, f.y * sd, f.z * sd, f.w * sd};
    char* p = smem + WM_XRING * WM_XPLANE + pl * 2 * WM_DPLANE + vx * WM_DROW + vy * 2;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const _Float16 h = (_Float16)v[ch];
      *reinterpret_cast<_Float16*>(p + ch * WM_CHB) = h;
      *reinterpret_cast<_Float16*>(p + ch * WM_CHB + WM_DPLANE) = (_Float16)(v[ch] - (float)h);
    }
  };

  // lane constants of the fragments
  const int col = lane & 15, kq = lane >> 4, zsel = kq >> 1, yb = kq & 1;
  const char* R = smem + wave * WM_CHB;
  const char* abase = R + WM_XRING * WM_XPLANE + zsel * 2 * WM_DPLANE + col * WM_DROW + yb * 16;
  const uint32_t* bbase = reinterpret_cast<const uint32_t*>(R) + col * WM_XROW + 8 * yb;

  f32x4 P[7][7][2];
#pragma unroll
  for (int kz = 0; kz < 7; ++kz)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int t = 0; t < 2; ++t) P[kz][ky][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

  const int npairs = (a.D + 1) / 2;
  for (int n = n_beg; n < n_end; ++n) {
    // item prologue: X planes -3 .. 2 (slots 0 .. 5); planes 3, 4 come with the first pair
    {
      float4 f[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) f[u] = load_plane(a.x, n, u);
      __syncthreads();    // the previous item's last pair is consumed
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        store_x(u, make_float4(0.f, 0.f, 0.f, 0.f));
        store_x(3 + u, f[u]);
      }
    }
    float4 pdy[2], px[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pdy[u] = load_plane(a.dy, n, u);
      px[u] = load_plane(a.x, n, 3 + u);
    }
    for (int pr = 0; pr < npairs; ++pr) {
      const int z0 = 2 * pr;
      // stage this pair (loaded an iteration ago): dY planes z0, z0 + 1; X planes z0 + 3, z0 + 4
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        store_dy(u, pdy[u]);
        store_x((z0 + 6 + u) & 7, px[u]);
        dbs.x += pdy[u].x; dbs.y += pdy[u].y; dbs.z += pdy[u].z; dbs.w += pdy[u].w;
      }
      __syncthreads();
      // the next pair's loads under this pair's MFMAs
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pdy[u] = load_plane(a.dy, n, z0 + 2 + u);
        px[u] = load_plane(a.x, n, z0 + 5 + u);
      }
      const half8 ah = *reinterpret_cast<const half8*>(abase);
      const half8 al = *reinterpret_cast<const half8*>(abase + WM_DPLANE);
#pragma unroll
      for (int kz = 0; kz < 7; ++kz) {
        // X plane z0 + zsel + kz - 3 sits in slot (z0 + zsel + kz) & 7
        const uint32_t* bp = bbase + ((z0 + zsel + kz) & 7) * (WM_XPLANE / 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint32_t e[14];
#pragma unroll
          for (int i = 0; i < 14; ++i) e[i] = bp[t * 16 * WM_XROW + i];
          // two ky at a time: their MFMA triples alternate between two accumulators
#pragma unroll
          for (int ky = 0; ky < 7; ky += 2) {
            half8 bh[2], bl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              if (ky + u > 6) continue;
              uint32_t ph[4], pl[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                ph[q] = __builtin_amdgcn_perm(e[ky + u + 2 * q + 1], e[ky + u + 2 * q], 0x05040100u);
                pl[q] = __builtin_amdgcn_perm(e[ky + u + 2 * q + 1], e[ky + u + 2 * q], 0x07060302u);
              }
              __builtin_memcpy(&bh[u], ph, 16);
              __builtin_memcpy(&bl[u], pl, 16);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (ky + u <= 6)
                P[kz][ky + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], P[kz][ky + u][t], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (ky + u <= 6)
                P[kz][ky + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], P[kz][ky + u][t], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (ky + u <= 6)
                P[kz][ky + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], P[kz][ky + u][t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- diagonals: tile t holds P[xo = 4 kq + r][k = col + 16 t]; dW[kx] = sum_xo P[xo][xo + kx].
  // Per kz the wave dumps its 14 tiles into its own (now idle) ring region and lanes 0 .. 48 each
  // walk one diagonal -- the register tiles are only ever indexed statically.
  const float oscale = __int_as_float((127 - (kxs + kds)) << 23);
  float* out = a.part + ((size_t)chunk * a.C + c0 + wave) * 344;
  __syncthreads();   // no staging store of another wave is still on its way into this region
  float* T = reinterpret_cast<float*>(smem + wave * WM_CHB);       // [ky][t][lane][r]
  const int oky = lane / 7, okx = lane - 7 * oky;
#pragma unroll
  for (int kz = 0; kz < 7; ++kz) {
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<f32x4*>(T + ((ky * 2 + t) * 64 + lane) * 4) = P[kz][ky][t];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own stores have landed
    __builtin_amdgcn_wave_barrier();
    if (lane < 49) {
      float s = 0.f;
#pragma unroll
      for (int xo = 0; xo < 16; ++xo) {
        const int k = xo + okx;                                     // < 22
        s += T[((oky * 2 + (k >> 4)) * 64 + (k & 15) + 16 * (xo >> 2)) * 4 + (xo & 3)];
      }
      out[(kz * 7 + oky) * 7 + okx] = s * oscale;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  // ---- bias gradient: sum of the staged dY per channel ------------------------------------------
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dbs.x += __shfl_xor(dbs.x, o, 64); dbs.y += __shfl_xor(dbs.y, o, 64);
    dbs.z += __shfl_xor(dbs.z, o, 64); dbs.w += __shfl_xor(dbs.w, o, 64);
  }
  __syncthreads();
  if (lane == 0) {
    sRed[wave * 4 + 0] = dbs.x; sRed[wave * 4 + 1] = dbs.y;
    sRed[wave * 4 + 2] = dbs.z; sRed[wave * 4 + 3] = dbs.w;
  }
  __syncthreads();
  if (tid < 4)
    a.part[((size_t)chunk * a.C + c0 + tid) * 344 + 343] =
        (sRed[tid] + sRed[4 + tid]) + (sRed[8 + tid] + sRed[12 + tid]);
}

extern "C" int adell_absmax_f32(const float* x, long n, uint32_t* out, void* stream);

static int adell_dw_wgrad_mfma_chunks(int N, int C, int* items_per_chunk) {
  const int groups = C / WM_CG;
  int chunks = 256 / groups;
  chunks = chunks < 1 ? 1 : (chunks > N ? N : chunks);
  const int ipc = (N + chunks - 1) / chunks;
  *items_per_chunk = ipc;
  return (N + ipc - 1) / ipc;
}

extern "C" int adell_dw_wgrad_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                                      const float* x, const float* dy) {
  return KD == 7 && KH == 7 && KW == 7 && W > 8 && W <= 16 && H > 8 && H <= 16 && D >= 1 &&
         C % WM_CG == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0 &&
         !g_adell_tune.dw_nomfma && !g_adell_tune.dw_wgrad_nomfma;
}

extern "C" long adell_dw_wgrad_mfma_workspace_floats(int N, int C) {
  int ipc;
  const int chunks = adell_dw_wgrad_mfma_chunks(N, C, &ipc);
  return (long)chunks * C * 344 + 4;
}

// dw [C][343], db [C] or null; workspace: adell_dw_wgrad_mfma_workspace_floats floats
extern "C" int adell_dw_wgrad_mfma_launch(const float* x, const float* dy, float* workspace, int N,
                                          int C, int D, int H, int W, int* chunks_out, void* stream) {
  ADELL_REQUIRE(workspace, "dw_wgrad_mfma: workspace required");
  hipStream_t st = (hipStream_t)stream;
  int ipc;
  const int chunks = adell_dw_wgrad_mfma_chunks(N, C, &ipc);
  uint32_t* words = reinterpret_cast<uint32_t*>(workspace + (long)chunks * C * 344);
  ADELL_CHECK_HIP(hipMemsetAsync(words, 0, 8, st));
  const long total = (long)N * C * D * H * W;
  int rc = adell_absmax_f32(x, total, words, stream);
  if (rc != ADELL_OK) return rc;
  rc = adell_absmax_f32(dy, total, words + 1, stream);
  if (rc != ADELL_OK) return rc;
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_dw_wgrad_mfma_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  DwWgMfmaArgs a = {x, dy, workspace, words, N, C, D, H, W, ipc};
  hipLaunchKernelGGL(adell_dw_wgrad_mfma_kernel, dim3(C / WM_CG, chunks), dim3(256), WM_LDS, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  *chunks_out = chunks;
  return ADELL_OK;
}
