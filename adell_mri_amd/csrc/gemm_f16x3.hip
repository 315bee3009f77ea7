// Row-major fp32 GEMM on the f16 MFMA by error-compensated splitting ("f16x3", see
// conv_igemm_f16.h) for the Linear layers of the token / feature paths (ConvNeXt point-wise MLPs
// res_blocks.py:559-566, ViT / SWIN projections linear_blocks.py, projection heads
// res_net.py:278-324): same contract as adell_gemm_f32 (csrc/gemm.hip),
//
//     C[M][N] = sum_k A(m, k) * B(k, n)  (+ bias[n]) (+ residual[m][n])
//
// with each operand k-contiguous (KC: A[m*lda + k], B[n*ldb + k]) or outer-contiguous (A[k*lda + m],
// B[k*ldb + n]) -- forward X W^T (KC, KC), backward-data dY W (KC, outer), backward-weight dY^T X
// (outer, outer) -- at 3 f16 MFMAs per product instead of the fp32 MFMA's 5.3x lower rate.
//
// Range: power-of-two operand scales, undone in the epilogue: per block and 64-k stage from the
// absmax of the staged tiles (default: no extra pass over the operands; the accumulators are
// rescaled by the exact ratio when the exponents change), or one per operand TENSOR from absmax
// words the caller provides (adell_absmax_f32).
//
// A block of 4 waves owns a 128 x 128 tile (wave = 64 x 64 = 2 x 2 MFMA tiles); a stage is 64 k
// values: both operand tiles go global -> registers -> split to (hi, lo) halves -> LDS
// (one 64 KB image, the next stage's loads in flight in registers under the MFMAs; two resident
// blocks per CU).
// LDS image of an operand tile: [16-k chunk][row][64 B = hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15],
// rows permuted inside blocks of 16 and slots XOR-swizzled so that the ds_read_b128 fragment
// reads are conflict-free AND the 8-byte writes of a transposed (outer-contiguous) operand spread
// over the banks. A KC tile is loaded as rows of 256 contiguous bytes (16 lanes per row); an
// outer-contiguous tile as k-rows of 512 contiguous bytes, each thread holding 4 k x 4 rows and
// writing the four 4-k half-slots of its rows -- the transpose costs no extra pass.
// Skinny problems split K (deterministic: slabs + fixed-order fold with bias / residual).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

struct GemmHArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;
  float* slab;
  const unsigned* amaxA;
  const unsigned* amaxB;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int splits, stages_per_split;
  // wide epilogue / the fold: an activation around the output (ld = ldc for both tensors)
  float* act_out;        // also store act(C) here (Linear -> activation: both tensors in one pass)
  const float* dact_in;  // multiply the output by act'(dact_in) (the backward of that pair)
  int act;
  float act_p;
};

constexpr int BM = 128, BN = 128, BK = 64, NCH = BK / 16;
constexpr int kTileBytes = NCH * 128 * 64;   // one operand tile: [4 chunks][128 rows][64 B] = 32 KB

// byte offset of (row, 16-byte slot) inside a chunk image
__device__ __forceinline__ int lds_off(int r, int slot) {
  const int r2 = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
  const int s2 = slot ^ (r & 3) ^ ((r >> 4) & 3);
  return r2 * 64 + s2 * 16;
}

__device__ __forceinline__ int scale_exp(unsigned amax_bits) {
  const int ebits = (amax_bits >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);
  if (k > 100) k = 100;
  if (k < -100) k = -100;
  return k;
}

__device__ __forceinline__ void split4(const float4& f, float scale, half4* h, half4* l) {
  const float v[4] = {f.x * scale, f.y * scale, f.z * scale, f.w * scale};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    (*h)[j] = (_Float16)v[j];
    (*l)[j] = (_Float16)(v[j] - (float)(*h)[j]);
  }
}

}  // namespace

// KC operand tile: X[(r0 + row) * ld + k0 + k], 128 rows x 64 k. Thread t: k quad q = t & 15 of
// rows (t >> 4) + 16 u (a row = 256 contiguous bytes over 16 lanes). Outer operand tile:
// X[(k0 + k) * ld + r0 + row]. Thread t: rows 4 (t & 31) .. + 3, k = 32 u + 4 (t >> 5) .. + 3 (a k-row
// = 512 contiguous bytes over 32 lanes).
template <bool KC>
__device__ __forceinline__ unsigned gemm_h_fetch(float4 (&f)[8], const float* X, long ld, int rows,
                                                 int kext, int r0, int k0, int tid) {
  unsigned okbits = 0;   // bit u: f[u] is inside the matrix (applied by gemm_h_put: a select here
                         // would wait for the loads right after they are issued)
  // branch-free: positions past the matrix read element (0, 0) and are zeroed, so that the eight
  // loads are in flight together (a conditional load per element made them wait for each other)
  if constexpr (KC) {
    const int k = k0 + 4 * (tid & 15);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + (tid >> 4) + 16 * u;
      const bool ok = (r < rows) & (k < kext);
      f[u] = *reinterpret_cast<const float4*>(X + (ok ? (long)r * ld + k : 0L));
      okbits |= ok ? (1u << u) : 0u;
    }
  } else {
    const int r = r0 + 4 * (tid & 31);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 32 * u + 4 * (tid >> 5) + j;
        const bool ok = (r < rows) & (k < kext);
        f[4 * u + j] = *reinterpret_cast<const float4*>(X + (ok ? (long)k * ld + r : 0L));
        okbits |= ok ? (1u << (4 * u + j)) : 0u;
      }
  }
  return okbits;
}

// zero the positions past the matrix; returns the absmax of what is left
__device__ __forceinline__ float gemm_h_mask(float4 (&f)[8], unsigned okbits) {
  float mx = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = (okbits >> u) & 1u;
    f[u] = make_float4(ok ? f[u].x : 0.f, ok ? f[u].y : 0.f, ok ? f[u].z : 0.f, ok ? f[u].w : 0.f);
    mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f[u].x)), fmaxf(fabsf(f[u].y), fabsf(f[u].z))), fabsf(f[u].w));
  }
  return mx;
}

template <bool KC>
__device__ __forceinline__ void gemm_h_put(char* tile, const float4 (&f)[8], float scale, int tid) {
  if constexpr (KC) {
    const int q = tid & 15, chunk = q >> 2, slot = (q & 3) >> 1, half = q & 1;
    char* base = tile + chunk * (128 * 64) + half * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = (tid >> 4) + 16 * u;
      half4 h, l;
      split4(f[u], scale, &h, &l);
      *reinterpret_cast<half4*>(base + lds_off(r, slot)) = h;
      *reinterpret_cast<half4*>(base + lds_off(r, 2 + slot)) = l;
    }
  } else {
    const int kg = tid >> 5, slot = (kg & 3) >> 1, half = kg & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      char* base = tile + (2 * u + (kg >> 2)) * (128 * 64) + half * 8;
      const float4* g = f + 4 * u;
      const float v[4][4] = {{g[0].x, g[0].y, g[0].z, g[0].w}, {g[1].x, g[1].y, g[1].z, g[1].w},
                             {g[2].x, g[2].y, g[2].z, g[2].w}, {g[3].x, g[3].y, g[3].z, g[3].w}};
#pragma unroll
      for (int c = 0; c < 4; ++c) {   // row 4 (t & 31) + c: its k = 32 u + 4 kg .. + 3
        const float4 col = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);
        half4 h, l;
        split4(col, scale, &h, &l);
        const int r = 4 * (tid & 31) + c;
        *reinterpret_cast<half4*>(base + lds_off(r, slot)) = h;
        *reinterpret_cast<half4*>(base + lds_off(r, 2 + slot)) = l;
      }
    }
  }
}

template <bool AKC, bool BKC, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void adell_gemm_f16x3_kernel(GemmHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [A | B] 64 KB + 8 floats
  float* sMax = reinterpret_cast<float*>(smem + 2 * kTileBytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, split = blockIdx.z;
  const int nstages = (a.K + BK - 1) / BK;
  const int s_beg = split * a.stages_per_split;
  const int s_end = (s_beg + a.stages_per_split) < nstages ? (s_beg + a.stages_per_split) : nstages;
  // operand scales: per tensor from the caller's absmax words, or (no words) per block and stage
  // from the absmax of the staged tiles, the accumulators rescaled by the exact power of two
  // whenever the pair of exponents changes (as the conv kernels do per chunk)
  const bool dyn = a.amaxA == nullptr;
  int kA = dyn ? 0 : scale_exp(*a.amaxA), kB = dyn ? 0 : scale_exp(*a.amaxB);
  int kprev = kA + kB;

#ifdef ADELL_GEMM_MFMA16
  // experiment: the same wave tile on v_mfma_f32_16x16x32_f16 (K = 32 = two 16-k chunks)
  f32x4 acc4[2][2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc4[i][j][ih][jh][r] = 0.f;
#else
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#endif

  // One LDS image, the next stage in registers: with two resident blocks per CU a stage's loads
  // are in flight for two MFMA phases (2 x 48 MFMAs per wave), enough to cover the memory latency
  // that a 32-k double-buffered stage (24 MFMAs) exposed.
  float4 fa[8], fb[8];
  unsigned oka = 0, okb = 0;
  auto fetch = [&](int s) {
    oka = gemm_h_fetch<AKC>(fa, a.A, a.lda, a.M, a.K, m0, s * BK, tid);
    okb = gemm_h_fetch<BKC>(fb, a.B, a.ldb, a.N, a.K, n0, s * BK, tid);
  };
  const char* tA = smem;
  const char* tB = smem + kTileBytes;
  // iteration s: issue the loads of stage s, run the MFMAs of stage s - 1 out of LDS, then move
  // stage s from registers to LDS. ONE fetch site: with a prologue fetch + a loop fetch the
  // registers meet in a phi whose copies wait for the loads right after they are issued.
  for (int s = s_beg; s <= s_end; ++s) {
    if (s < s_end) fetch(s);
    if (s > s_beg) {
#ifdef ADELL_GEMM_MFMA16
    const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int cp = 0; cp < NCH / 2; ++cp) {
      const int choff = (2 * cp + (kq >> 1)) * (128 * 64);
      half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
          const int r = wm * 64 + i * 32 + ih * 16 + l16;
          ah[i][ih] = *reinterpret_cast<const half8*>(tA + choff + lds_off(r, kq & 1));
          al[i][ih] = *reinterpret_cast<const half8*>(tA + choff + lds_off(r, 2 + (kq & 1)));
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          const int r = wn * 64 + j * 32 + jh * 16 + l16;
          bh[j][jh] = *reinterpret_cast<const half8*>(tB + choff + lds_off(r, kq & 1));
          bl[j][jh] = *reinterpret_cast<const half8*>(tB + choff + lds_off(r, 2 + (kq & 1)));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ih = 0; ih < 2; ++ih)
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
              f32x4& c = acc4[i][j][ih][jh];
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i][ih], bh[j][jh], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i][ih], bl[j][jh], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i][ih], bh[j][jh], c, 0, 0, 0);
            }
    }
    }
#else
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + li;
        ah[i] = *reinterpret_cast<const half8*>(tA + ch * (128 * 64) + lds_off(r, lh));
        al[i] = *reinterpret_cast<const half8*>(tA + ch * (128 * 64) + lds_off(r, 2 + lh));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + li;
        bh[j] = *reinterpret_cast<const half8*>(tB + ch * (128 * 64) + lds_off(r, lh));
        bl[j] = *reinterpret_cast<const half8*>(tB + ch * (128 * 64) + lds_off(r, 2 + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    }
#endif
    if (s < s_end) {
      float ma = gemm_h_mask(fa, oka), mb = gemm_h_mask(fb, okb);
      if (dyn) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          ma = fmaxf(ma, __shfl_xor(ma, o, 64));
          mb = fmaxf(mb, __shfl_xor(mb, o, 64));
        }
      }
      __syncthreads();   // the fragments of stage s - 1 are read (and sMax of the previous stage)
      if (dyn) {
        if (lane == 0) {
          sMax[wave] = ma;
          sMax[4 + wave] = mb;
        }
        __syncthreads();
        ma = fmaxf(fmaxf(sMax[0], sMax[1]), fmaxf(sMax[2], sMax[3]));
        mb = fmaxf(fmaxf(sMax[4], sMax[5]), fmaxf(sMax[6], sMax[7]));
        // multiples of 8 (max lands in [2^6, 2^14)) so that the pair rarely changes
        kA = 8 * ((scale_exp(__float_as_uint(ma))) >> 3);
        kB = 8 * ((scale_exp(__float_as_uint(mb))) >> 3);
        if (s > s_beg && kA + kB != kprev) {
          const float fix = __int_as_float((kA + kB - kprev + 127) << 23);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
#ifdef ADELL_GEMM_MFMA16
                acc4[i][j][r >> 3][(r >> 2) & 1][r & 3] *= fix;
#else
                acc[i][j][r] *= fix;
#endif
              }
        }
        kprev = kA + kB;
      }
      gemm_h_put<AKC>(smem, fa, __int_as_float((kA + 127) << 23), tid);
      gemm_h_put<BKC>(smem + kTileBytes, fb, __int_as_float((kB + 127) << 23), tid);
      __syncthreads();
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const float oscale = __int_as_float((127 - kprev) << 23);
  const bool direct = a.splits == 1;
#ifdef ADELL_GEMM_MFMA16
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          const int col = n0 + wn * 64 + j * 32 + jh * 16 + (lane & 15);
          if (col >= a.N) continue;
          const float bv = (direct && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 32 + ih * 16 + 4 * (lane >> 4) + r;
            if (row >= a.M) continue;
            float v = acc4[i][j][ih][jh][r] * oscale;
            if (direct) {
              v += bv;
              if (a.residual) v += a.residual[(long)row * a.ldr + col];
              a.C[(long)row * a.ldc + col] = v;
            } else {
              a.slab[((long)split * a.M + row) * a.N + col] = v;
            }
          }
        }
#else
  if constexpr (WIDE) {
    // Wide epilogue (N and the leading dimensions multiples of 4): the tile goes through the LDS
    // image (free after the last stage: 128 x 128 floats = 64 KB) and ONE rolled loop applies bias /
    // residual / activation to 16-byte row pieces -- 16 stores of 16 B per thread instead of 64 of
    // 4 B (VICReg ConvNeXt step -0.5 ms). The activation lives only here: inlined into the 64
    // unrolled stores of the scalar epilogue it made the kernel 190 KB of code (9 activations x 64
    // sites) and the step 3 ms SLOWER than the element-wise passes it replaced -- instruction-cache
    // misses.
    __syncthreads();
    float* sC = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * 64 + j * 32 + li] =
              acc[i][j][r] * oscale;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < BM * BN / 4 / 256; ++it) {
      const int idx = it * 256 + tid, rl = idx >> 5, cl = (idx & 31) * 4;
      const int row = m0 + rl, col = n0 + cl;
      if (row >= a.M || col >= a.N) continue;
      float4 v4 = *reinterpret_cast<const float4*>(sC + rl * BN + cl);
      if (!direct) {   // split K: this block's slab (bias / residual / activation ride the fold)
        *reinterpret_cast<float4*>(a.slab + ((long)split * a.M + row) * a.N + col) = v4;
        continue;
      }
      float v[4] = {v4.x, v4.y, v4.z, v4.w};
      const long o = (long)row * a.ldc + col;
      if (a.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + col);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (a.residual) {
        const float4 r4 = *reinterpret_cast<const float4*>(a.residual + (long)row * a.ldr + col);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
      if (a.dact_in) {
        const float4 d4 = *reinterpret_cast<const float4*>(a.dact_in + o);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll 1
        for (int q = 0; q < 4; ++q) v[q] *= adell_act_grad(a.act, d[q], a.act_p);
      }
      *reinterpret_cast<float4*>(a.C + o) = make_float4(v[0], v[1], v[2], v[3]);
      if (a.act_out) {
        float g[4];
#pragma unroll 1
        for (int q = 0; q < 4; ++q) g[q] = adell_act_fwd(a.act, v[q], a.act_p);
        *reinterpret_cast<float4*>(a.act_out + o) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= a.N) continue;
      const float bv = (direct && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= a.M) continue;
        float v = acc[i][j][r] * oscale;
        if (direct) {
          v += bv;
          if (a.residual) v += a.residual[(long)row * a.ldr + col];
          a.C[(long)row * a.ldc + col] = v;
        } else {
          a.slab[((long)split * a.M + row) * a.N + col] = v;
        }
      }
    }
#endif
}

// fixed-order fold of the split-K slabs (+ bias, residual). A block takes OPB consecutive outputs
// (x V floats each) and spreads the slabs over G thread groups of OPB threads, four independent
// loads in flight per thread: group q sums slabs q, q + G, ... in that
// order, the G group sums are added in group order -- one fixed tree per (splits, shape). G follows
// the slab count (1 / 4 / 16): one thread per output walking all the slabs took 120 us for 171
// slabs of 96 x 384 (25 MB behind 144 blocks' worth of load latency), 16 groups on a two-slab
// fold cost 10 us instead of 5.
template <int V, int G>
__global__ __launch_bounds__(G == 16 ? 1024 : 256) void adell_gemm_f16x3_fold_kernel(GemmHArgs a) {
  constexpr int NT = G == 16 ? 1024 : 256, OPB = NT / G;
  __shared__ float sh[G][OPB * V];
  const long total = (long)a.M * a.N;
  const int e = threadIdx.x % OPB, q = threadIdx.x / OPB;
  const long i = ((long)blockIdx.x * OPB + e) * V;
  float acc[4][V];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[u][v] = 0.f;
  auto add = [&](int u, int sp) {
    const float* src = a.slab + (long)sp * total + i;
    if constexpr (V == 4) {
      const float4 f = *reinterpret_cast<const float4*>(src);
      acc[u][0] += f.x; acc[u][1] += f.y; acc[u][2] += f.z; acc[u][3] += f.w;
    } else {
      acc[u][0] += *src;
    }
  };
  if (i < total) {
    int sp = q;
    for (; sp + 3 * G < a.splits; sp += 4 * G) {
#pragma unroll
      for (int u = 0; u < 4; ++u) add(u, sp + G * u);
    }
    for (; sp < a.splits; sp += G) add(0, sp);
  }
  float t[V];
#pragma unroll
  for (int v = 0; v < V; ++v) t[v] = (acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v]);
  if constexpr (G > 1) {
#pragma unroll
    for (int v = 0; v < V; ++v) sh[q][e * V + v] = t[v];
    __syncthreads();
    if (q != 0) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      t[v] = 0.f;
#pragma unroll
      for (int k = 0; k < G; ++k) t[v] += sh[k][e * V + v];
    }
  }
  if (i >= total) return;
  const int row = (int)(i / a.N), col = (int)(i - (long)row * a.N);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float s = t[v];
    if (a.bias) s += a.bias[col + v];
    if (a.residual) s += a.residual[(long)row * a.ldr + col + v];
    if (a.dact_in) s *= adell_act_grad(a.act, a.dact_in[(long)row * a.ldc + col + v], a.act_p);
    if (a.act_out) a.act_out[(long)row * a.ldc + col + v] = adell_act_fwd(a.act, s, a.act_p);
    a.C[(long)row * a.ldc + col + v] = s;
  }
}

template <int V>
static void adell_gemm_f16x3_fold(const GemmHArgs& a, hipStream_t st) {
  const long units = ((long)a.M * a.N + V - 1) / V;
  if (a.splits >= 32)
    hipLaunchKernelGGL((adell_gemm_f16x3_fold_kernel<V, 16>), dim3((unsigned)((units + 63) / 64)),
                       dim3(1024), 0, st, a);
  else if (a.splits >= 8)
    hipLaunchKernelGGL((adell_gemm_f16x3_fold_kernel<V, 4>), dim3((unsigned)((units + 63) / 64)),
                       dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((adell_gemm_f16x3_fold_kernel<V, 1>), dim3((unsigned)((units + 255) / 256)),
                       dim3(256), 0, st, a);
}

// absmax (float bits, atomicMax into a zero-initialised word) of n floats
__global__ __launch_bounds__(256) void adell_absmax_f32_kernel(const float* __restrict__ x, long n,
                                                               unsigned* __restrict__ out) {
  float mx = 0.f;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    const float4 f = x4[i];
    mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f.x)), fmaxf(fabsf(f.y), fabsf(f.z))), fabsf(f.w));
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
    mx = fmaxf(mx, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  // one atomic per block, and only when it can raise the word (thousands of waves hitting one
  // address serialised: 100 us for a 50 MB tensor)
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const unsigned bits = __float_as_uint(mx);
    if (bits > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, bits);
  }
}

// the streaming kernel for many-row Linear layers (gemm_rows.hip)
bool adell_gemm_rows_ok(int M, int N, int K, const float* A, long lda, int a_kc, const float* B, long ldb,
                        int act, bool epi, bool dact);
long adell_gemm_rows_workspace_floats(int M, int N, int K);
int adell_gemm_rows_run(int M, int N, int K, const float* A, long lda, const float* B, long ldb, int b_kc,
                        float* C, long ldc, const float* bias, const float* residual, long ldr,
                        float* workspace, hipStream_t st, int act, float act_p, float* act_out,
                        const float* dact_in);

namespace {

struct GemmHPlan {
  int splits, stages_per_split;
};

GemmHPlan gemm_h_plan(int M, int N, int K) {
  GemmHPlan p;
  const long tiles = (long)adell_cdiv(M, BM) * adell_cdiv(N, BN);
  const int stages = adell_cdiv(K, BK);
  // two resident blocks per CU, and never one block more than that: 3 tiles x 171 shares = 513 blocks
  // left the last block alone on the chip for a second round (dW of 262 144 x 96 / 384: 194 us)
  long s = 512 / (tiles < 512 ? tiles : 512);
  if (s > stages / 4) s = stages / 4;                          // at least four stages per split
  const long slab_cap = (16L << 20) / ((long)M * N);           // slabs capped at 64 MB
  if (s > slab_cap) s = slab_cap;
  if (s > 65535) s = 65535;
  if (s < 1) s = 1;
  p.stages_per_split = adell_cdiv(stages, (int)s);
  p.splits = adell_cdiv(stages, p.stages_per_split);
  return p;
}

bool gemm_h_ok(int M, int N, int K, const float* A, long lda, int a_kc, const float* B, long ldb,
               int b_kc) {
  if (M <= 0 || N <= 0 || K <= 0) return false;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return false;
  if ((lda | ldb) & 3) return false;
  // whole 16-byte groups along each operand's contiguous axis
  if (K & 3) return (false);
  if (!a_kc && (M & 3)) return false;
  if (!b_kc && (N & 3)) return false;
  return true;
}

}  // namespace

// absmax of n floats as float bits into *out (zero-initialised by the caller): the operand scale
// words of adell_gemm_f16x3.
extern "C" int adell_absmax_f32(const float* x, long n, uint32_t* out, void* stream) {
  ADELL_REQUIRE(x && out && n > 0, "absmax_f32: bad arguments");
  ADELL_REQUIRE(((uintptr_t)x & 15) == 0, "absmax_f32: x must be 16-byte aligned");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adell_absmax_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     x, n, out);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// 1 when adell_gemm_f16x3 takes the problem (16-byte aligned operands, leading dimensions and K
// multiples of 4, outer-contiguous operands with a multiple-of-4 outer extent).
extern "C" int adell_gemm_f16x3_applicable(int M, int N, int K, const float* A, long lda, int a_kc,
                                           const float* B, long ldb, int b_kc) {
  return gemm_h_ok(M, N, K, A, lda, a_kc, B, ldb, b_kc) ? 1 : 0;
}

extern "C" long adell_gemm_f16x3_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmHPlan p = gemm_h_plan(M, N, K);
  long need = p.splits > 1 ? (long)p.splits * M * N : 0;
  // the streaming kernel packs its weight operand into the workspace (shape-eligible problems only;
  // the pointer conditions are checked at the call)
  if (adell_gemm_rows_ok(M, N, K, nullptr, 0, 1, nullptr, 0, ADELL_ACT_GELU, false, false)) {
    const long rows = adell_gemm_rows_workspace_floats(M, N, K);
    need = need > rows ? need : rows;
  }
  return need;
}

// Same contract as adell_gemm_f32 + a_absmax / b_absmax: both NULL (operand scales chosen per block
// and 64-k stage inside the kernel), or device words holding the float bits of the absmax of the
// A / B tensors (adell_absmax_f32, or any upper bound within a factor of 2^10): one scale per tensor.
static int gemm_h_run(int M, int N, int K, const float* A, long lda, int a_kc,
                      const float* B, long ldb, int b_kc, float* C, long ldc,
                      const float* bias, const float* residual, long ldr,
                      const uint32_t* a_absmax, const uint32_t* b_absmax,
                      float* workspace, void* stream, int act, float act_p, float* act_out,
                      const float* dact_in) {
  ADELL_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f16x3: bad dims");
  ADELL_REQUIRE(A && B && C, "gemm_f16x3: null pointer");
  ADELL_REQUIRE((a_absmax == nullptr) == (b_absmax == nullptr),
                "gemm_f16x3: give both absmax words or neither");
  ADELL_REQUIRE(lda >= (a_kc ? K : M) && ldb >= (b_kc ? K : N) && ldc >= N, "gemm_f16x3: bad strides");
  ADELL_REQUIRE(!residual || ldr >= N, "gemm_f16x3: bad residual stride");
  if (!gemm_h_ok(M, N, K, A, lda, a_kc, B, ldb, b_kc)) {
    adell_set_error("gemm_f16x3: operands need 16-byte alignment, leading dimensions / K multiples of 4");
    return ADELL_E_UNSUPPORTED;
  }
  if (workspace && (((uintptr_t)workspace) & 15) == 0 && (((uintptr_t)C) & 3) == 0 &&
      adell_gemm_rows_ok(M, N, K, A, lda, a_kc, B, ldb, act, act_out || dact_in, dact_in != nullptr))
    return adell_gemm_rows_run(M, N, K, A, lda, B, ldb, b_kc, C, ldc, bias, residual, ldr, workspace,
                               (hipStream_t)stream, act, act_p, act_out, dact_in);
  const GemmHPlan p = gemm_h_plan(M, N, K);
  ADELL_REQUIRE(p.splits == 1 || workspace, "gemm_f16x3: workspace required for this shape");
  GemmHArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.residual = residual; a.slab = workspace;
  a.amaxA = a_absmax; a.amaxB = b_absmax;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr;
  a.splits = p.splits;
  a.stages_per_split = p.stages_per_split;
  a.act = act; a.act_p = act_p; a.act_out = act_out; a.dact_in = dact_in;
  const bool epi = act_out || dact_in;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(adell_cdiv(M, BM), adell_cdiv(N, BN), p.splits);
  ADELL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm_f16x3: grid too large");
  constexpr size_t kLds = 2 * kTileBytes + 8 * sizeof(float);
  auto launch = [&](auto kern) -> int {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    hipLaunchKernelGGL(kern, grid, dim3(256), kLds, st, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  };
  int rc;
  const bool wide = N % 4 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0) &&
                    ((((uintptr_t)C) | ((uintptr_t)bias) | ((uintptr_t)residual) |
                      ((uintptr_t)workspace)) & 15) == 0;
  ADELL_REQUIRE(wide || !epi, "gemm_f16x3_act: operands do not qualify for the wide epilogue");
  if (wide) {
    if (a_kc && b_kc)
      rc = launch(adell_gemm_f16x3_kernel<true, true, true>);
    else if (a_kc && !b_kc)
      rc = launch(adell_gemm_f16x3_kernel<true, false, true>);
    else if (!a_kc && !b_kc)
      rc = launch(adell_gemm_f16x3_kernel<false, false, true>);
    else
      rc = launch(adell_gemm_f16x3_kernel<false, true, true>);
  } else if (a_kc && b_kc)
    rc = launch(adell_gemm_f16x3_kernel<true, true>);
  else if (a_kc && !b_kc)
    rc = launch(adell_gemm_f16x3_kernel<true, false>);
  else if (!a_kc && !b_kc)
    rc = launch(adell_gemm_f16x3_kernel<false, false>);
  else
    rc = launch(adell_gemm_f16x3_kernel<false, true>);
  if (rc != ADELL_OK) return rc;
  if (p.splits > 1) {
    if (N % 4 == 0 && (((uintptr_t)a.slab) & 15) == 0)
      adell_gemm_f16x3_fold<4>(a, st);
    else
      adell_gemm_f16x3_fold<1>(a, st);
    ADELL_CHECK_HIP(hipGetLastError());
  }
  return ADELL_OK;
}

extern "C" int adell_gemm_f16x3(int M, int N, int K, const float* A, long lda, int a_kc,
                                const float* B, long ldb, int b_kc, float* C, long ldc,
                                const float* bias, const float* residual, long ldr,
                                const uint32_t* a_absmax, const uint32_t* b_absmax,
                                float* workspace, void* stream) {
  return gemm_h_run(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, bias, residual, ldr, a_absmax,
                    b_absmax, workspace, stream, 0, 0.f, nullptr, nullptr);
}

// The same GEMM with an activation in its epilogue -- Linear -> activation pairs without an
// element-wise pass (ConvNeXt's pwconv1 -> GELU -> pwconv2, res_blocks.py:559-566):
//   act_out != NULL: C = A B^T + bias (+ residual) and act_out = act(C), both [M][ldc];
//   dact_in != NULL: C = (A B^T ...) * act'(dact_in)  (dact_in = the saved pre-activation):
//                    the gradient of the pair's input side in the GEMM that produces it.
// A must be K-contiguous (forward and dX of a Linear layer).
extern "C" int adell_gemm_f16x3_act(int M, int N, int K, const float* A, long lda, int a_kc,
                                    const float* B, long ldb, int b_kc, float* C, long ldc,
                                    const float* bias, const float* residual, long ldr,
                                    const uint32_t* a_absmax, const uint32_t* b_absmax,
                                    float* workspace, int act, float act_p, float* act_out,
                                    const float* dact_in, void* stream) {
  ADELL_REQUIRE(act_out || dact_in, "gemm_f16x3_act: give act_out or dact_in");
  ADELL_REQUIRE(act >= 0 && act <= ADELL_ACT_ELU, "gemm_f16x3_act: unknown activation %d", act);
  ADELL_REQUIRE(a_kc, "gemm_f16x3_act: A must be K-contiguous");
  ADELL_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && (!residual || ldr % 4 == 0),
                "gemm_f16x3_act: N and the leading dimensions must be multiples of 4");
  ADELL_REQUIRE(((((uintptr_t)C) | ((uintptr_t)act_out) | ((uintptr_t)dact_in) | ((uintptr_t)bias) |
                  ((uintptr_t)residual)) & 15) == 0,
                "gemm_f16x3_act: C, act_out, dact_in, bias and residual must be 16-byte aligned");
  return gemm_h_run(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, bias, residual, ldr, a_absmax,
                    b_absmax, workspace, stream, act, act_p, act_out, dact_in);
}
