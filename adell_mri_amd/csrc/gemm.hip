// Row-major fp32 GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32) for the Linear layers of the
// token / feature paths (ViT encoder of UNETR, ConvNeXt point-wise layers, projection
// heads; torch.nn.Linear in adell_mri/modules/layers/linear_blocks.py, res_blocks.py:559-566,
// res_net.py:278-324):
//
//     C[M][N] = sum_k A(m, k) * B(k, n)  (+ bias[n]) (+ residual[m][n])
//
// Each operand is either "k-contiguous" (KC: A[m*lda + k], B[n*ldb + k]) or "outer-
// contiguous" (MC: A[k*lda + m], B[k*ldb + n]); that covers forward (X W^T: KC, KC),
// backward-data (dY W: KC, MC) and backward-weight (dY^T X: MC, MC) without materialising
// a transpose. A block of 4 waves owns a BM x BN tile; per step of 16 k values the two
// operand tiles go global -> registers -> LDS (double-buffered, one barrier per step,
// the next step's global loads in flight during the MFMAs). Within a group of 8 k values
// lane half h = lane >> 5 takes k = 4h + s for MFMA s = 0..3 (the same permutation for
// both operands), so a KC operand is one ds_read_b128 per 4 MFMAs. LDS strides are chosen
// so that every read and write is bank-conflict-free.
// Skinny problems are split along k (deterministic: each split writes its own slab, a
// second kernel sums the slabs in fixed order and applies bias / residual).
#include "common.h"

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;
  float* slab;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int splits, ksteps_per_split;
  int a_vec, b_vec;  // operand may be read with 16-byte loads
};

constexpr int GEMM_BK = 16;
constexpr int GEMM_KC_STRIDE = GEMM_BK + 4;  // [row][k] layout: 20 words

template <int BR>
struct GemmOpTile {
  static constexpr int MC_STRIDE = BR + 8;  // [k][row] layout: 4*stride = 32 (mod 64)
  static constexpr int FLOATS_KC = BR * GEMM_KC_STRIDE;
  static constexpr int FLOATS_MC = GEMM_BK * MC_STRIDE;
  static constexpr int FLOATS = FLOATS_KC > FLOATS_MC ? FLOATS_KC : FLOATS_MC;
  static constexpr int F4_PER_THREAD = (BR * 4 + 255) / 256;  // float4 loads per thread per step
};

__device__ __forceinline__ f32x4 adell_gemm_load4(const float* __restrict__ p, long off, int valid,
                                                  int vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p + off);
  } else {
    if (valid > 0) v.x = p[off];
    if (valid > 1) v.y = p[off + 1];
    if (valid > 2) v.z = p[off + 2];
    if (valid > 3) v.w = p[off + 3];
  }
  return v;
}

// fetch this thread's share of a BR x 16 operand tile (rows r0.., k from k0, k < kend)
template <int BR, bool KC>
__device__ __forceinline__ void adell_gemm_fetch(const float* __restrict__ P, long ld, int rows,
                                                 int r0, int k0, int kend, int vec, int tid,
                                                 f32x4 (&reg)[GemmOpTile<BR>::F4_PER_THREAD]) {
#pragma unroll
  for (int i = 0; i < GemmOpTile<BR>::F4_PER_THREAD; ++i) {
    const int idx = tid + i * 256;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (idx < BR * 4) {
      if (KC) {
        const int row = r0 + (idx >> 2), k = k0 + (idx & 3) * 4;
        if (row < rows) v = adell_gemm_load4(P, (long)row * ld + k, kend - k, vec);
      } else {
        const int k = k0 + idx / (BR / 4), row = r0 + (idx % (BR / 4)) * 4;
        if (k < kend) v = adell_gemm_load4(P, (long)k * ld + row, rows - row, vec);
      }
    }
    reg[i] = v;
  }
}

template <int BR, bool KC>
__device__ __forceinline__ void adell_gemm_stash(float* lds, int tid,
                                                 const f32x4 (&reg)[GemmOpTile<BR>::F4_PER_THREAD]) {
#pragma unroll
  for (int i = 0; i < GemmOpTile<BR>::F4_PER_THREAD; ++i) {
    const int idx = tid + i * 256;
    if (idx < BR * 4) {
      float* dst = KC ? lds + (idx >> 2) * GEMM_KC_STRIDE + (idx & 3) * 4
                      : lds + (idx / (BR / 4)) * GemmOpTile<BR>::MC_STRIDE + (idx % (BR / 4)) * 4;
      *reinterpret_cast<f32x4*>(dst) = reg[i];
    }
  }
}

// the 4 k values (4h + s, s = 0..3) of k-group g for tile row `r`
template <int BR, bool KC>
__device__ __forceinline__ f32x4 adell_gemm_frag(const float* lds, int r, int g, int h) {
  if (KC) return *reinterpret_cast<const f32x4*>(lds + r * GEMM_KC_STRIDE + g * 8 + 4 * h);
  const float* p = lds + (g * 8 + 4 * h) * GemmOpTile<BR>::MC_STRIDE + r;
  f32x4 v;
  v.x = p[0];
  v.y = p[GemmOpTile<BR>::MC_STRIDE];
  v.z = p[2 * GemmOpTile<BR>::MC_STRIDE];
  v.w = p[3 * GemmOpTile<BR>::MC_STRIDE];
  return v;
}

template <int WM, int WN, int TM, int TN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void adell_gemm_f32_kernel(GemmArgs a) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using TA = GemmOpTile<BM>;
  using TB = GemmOpTile<BN>;
  __shared__ __attribute__((aligned(16))) float lds[2 * (TA::FLOATS + TB::FLOATS)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int split = blockIdx.z;
  const int k_begin = split * a.ksteps_per_split * GEMM_BK;
  int k_end = k_begin + a.ksteps_per_split * GEMM_BK;
  if (k_end > a.K) k_end = a.K;
  const int nsteps = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[TA::F4_PER_THREAD], rb[TB::F4_PER_THREAD];
  adell_gemm_fetch<BM, AKC>(a.A, a.lda, a.M, m0, k_begin, k_end, a.a_vec, tid, ra);
  adell_gemm_fetch<BN, BKC>(a.B, a.ldb, a.N, n0, k_begin, k_end, a.b_vec, tid, rb);
  for (int step = 0; step < nsteps; ++step) {
    float* la = lds + (step & 1) * (TA::FLOATS + TB::FLOATS);
    float* lb = la + TA::FLOATS;
    adell_gemm_stash<BM, AKC>(la, tid, ra);
    adell_gemm_stash<BN, BKC>(lb, tid, rb);
    __syncthreads();
    if (step + 1 < nsteps) {
      const int k0 = k_begin + (step + 1) * GEMM_BK;
      adell_gemm_fetch<BM, AKC>(a.A, a.lda, a.M, m0, k0, k_end, a.a_vec, tid, ra);
      adell_gemm_fetch<BN, BKC>(a.B, a.ldb, a.N, n0, k0, k_end, a.b_vec, tid, rb);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = adell_gemm_frag<BM, AKC>(la, (wm * TM + i) * 32 + l31, g, h);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = adell_gemm_frag<BN, BKC>(lb, (wn * TN + j) * 32 + l31, g, h);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * h
  const bool direct = a.splits == 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      if (col >= a.N) continue;
      const float bv = (direct && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= a.M) continue;
        float v = acc[i][j][r];
        if (direct) {
          v += bv;
          if (a.residual) v += a.residual[(long)row * a.ldr + col];
          a.C[(long)row * a.ldc + col] = v;
        } else {
          a.slab[((long)split * a.M + row) * a.N + col] = v;
        }
      }
    }
}

// 64 outputs x 16 lanes per block; lane l sums splits l, l+16, ... then a fixed-order fold
__global__ __launch_bounds__(1024) void adell_gemm_reduce_kernel(GemmArgs a) {
  __shared__ float sh[16][64];
  const long total = (long)a.M * a.N;
  const int cl = threadIdx.x & 63, vl = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + cl;
  float s = 0.f;
  if (i < total)
    for (int sp = vl; sp < a.splits; sp += 16) s += a.slab[(long)sp * total + i];
  sh[vl][cl] = s;
  __syncthreads();
  if (vl != 0 || i >= total) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[k][cl];
  const int row = (int)(i / a.N), col = (int)(i - (long)row * a.N);
  if (a.bias) s += a.bias[col];
  if (a.residual) s += a.residual[(long)row * a.ldr + col];
  a.C[(long)row * a.ldc + col] = s;
}

// few splits: one thread per output (float4 when the row length allows), fixed order
__global__ __launch_bounds__(256) void adell_gemm_reduce_flat_kernel(GemmArgs a) {
  const long total = (long)a.M * a.N;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    float s = 0.f;
    for (int sp = 0; sp < a.splits; ++sp) s += a.slab[(long)sp * total + i];
    const int row = (int)(i / a.N), col = (int)(i - (long)row * a.N);
    if (a.bias) s += a.bias[col];
    if (a.residual) s += a.residual[(long)row * a.ldr + col];
    a.C[(long)row * a.ldc + col] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Linear layers with a handful of features over millions of rows (SWIN-UNet's per-voxel 2 -> 8 -> 2
// and 8 -> 32 layers at 256 x 256 x 128: 8.4 M rows). They are streaming operations -- 335 MB for
// 2 -> 8 at 8.4 M rows -- that ran 0.7-1.0 ms each on 128-wide MFMA tiles (0.3-0.5 TB/s).
//   rows kernel:  C[m][n] = sum_k A[m][k] B(k, n)   K, N <= 32, A row-major: a thread per row,
//                 the weights in LDS (broadcast reads), 16-byte loads / stores when K / N allow;
//   tall kernel:  C[m][n] = sum_k A[k][m] B[k][n]   M, N <= 32, K rows long (the weight gradient):
//                 lane = (m, row group), N accumulators per lane, block partials folded in block
//                 order by adell_gemm_reduce_kernel (deterministic).
constexpr int GEMM_SMALL = 32;

constexpr int GEMM_ROWS_K = 64, GEMM_ROWS_N = 64;   // rows kernel: K <= 64, N <= 64, N K <= 512
constexpr int GEMM_ROWS_W = 512 + 4 * GEMM_ROWS_N;    // weight image: N rows of K + 4 floats

template <bool BKC>
__global__ __launch_bounds__(256) void adell_gemm_rows_small_kernel(GemmArgs a) {
  // a thread per OUTPUT element (m, n): the N lanes of a row read the same K inputs (one
  // transaction, broadcast) and store N contiguous floats; weights [n][k] (rows padded by four
  // floats: a stride of 32 floats put every n on one bank) and bias in LDS
  __shared__ __attribute__((aligned(16))) float sw[GEMM_ROWS_W + GEMM_ROWS_N];
  const int K = a.K, N = a.N, KP = K + 4;
  for (int i = threadIdx.x; i < N * K; i += 256) {
    const int n = i / K, k = i - n * K;
    sw[n * KP + k] = BKC ? a.B[(long)n * a.ldb + k] : a.B[(long)k * a.ldb + n];
  }
  float* sb = sw + GEMM_ROWS_W;
  for (int i = threadIdx.x; i < N; i += 256) sb[i] = a.bias ? a.bias[i] : 0.f;
  __syncthreads();
  const bool vk = a.a_vec && (K & 3) == 0;
  // N in fours: a thread per (row, four outputs) -- a quarter of the row loads and 16-byte stores
  // (8 388 608 x 8 x 2 spent its time on 4-byte stores and two loads per output)
  if ((N & 3) == 0 && (N & (N - 1)) == 0 && (a.ldc & 3) == 0 && (((uintptr_t)a.C) & 15) == 0 &&
      (!a.residual || ((a.ldr & 3) == 0 && (((uintptr_t)a.residual) & 15) == 0))) {
    const int qshift = __ffs(N) - 3;                  // log2(N / 4)
    const long total4 = (long)a.M << qshift;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total4; e += (long)gridDim.x * 256L) {
      const long m = e >> qshift;
      const int n = (int)(e & ((N >> 2) - 1)) << 2;
      const float* ar = a.A + m * a.lda;
      const float* w0 = sw + n * KP;
      f32x4 s = {sb[n], sb[n + 1], sb[n + 2], sb[n + 3]};
      if (vk) {
        for (int k = 0; k < K; k += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ar + k);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w0 + j * KP + k);
            s[j] = fmaf(v.x, w.x, s[j]);
            s[j] = fmaf(v.y, w.y, s[j]);
            s[j] = fmaf(v.z, w.z, s[j]);
            s[j] = fmaf(v.w, w.w, s[j]);
          }
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const float v = ar[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] = fmaf(v, w0[j * KP + k], s[j]);
        }
      }
      if (a.residual) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(a.residual + m * a.ldr + n);
        s[0] += r.x; s[1] += r.y; s[2] += r.z; s[3] += r.w;
      }
      *reinterpret_cast<f32x4*>(a.C + m * a.ldc + n) = s;
    }
    return;
  }
  const long total = (long)a.M * N;
  // (N a power of two -- every caller's is --: shift and mask instead of a 64-bit division per
  // output, which cost more than the whole dot product of a 2- or 8-feature row)
  const bool pow2 = (N & (N - 1)) == 0;
  const int nshift = __ffs(N) - 1;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
    const long m = pow2 ? (e >> nshift) : e / N;
    const int n = pow2 ? (int)(e & (N - 1)) : (int)(e - m * N);
    const float* ar = a.A + m * a.lda;
    const float* wr = sw + n * KP;
    float s = sb[n];
    if (vk) {
      for (int k = 0; k < K; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ar + k);
        const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
        s = fmaf(v.x, w.x, s);
        s = fmaf(v.y, w.y, s);
        s = fmaf(v.z, w.z, s);
        s = fmaf(v.w, w.w, s);
      }
    } else {
      for (int k = 0; k < K; ++k) s = fmaf(ar[k], wr[k], s);
    }
    if (a.residual) s += a.residual[m * a.ldr + n];
    a.C[m * a.ldc + n] = s;
  }
}

__global__ __launch_bounds__(256) void adell_gemm_tall_small_kernel(GemmArgs a) {
  __shared__ float red[256 * GEMM_SMALL];
  const int M = a.M, N = a.N;
  const int groups = 256 / M;                 // row groups of a block (lanes past groups * M idle)
  const int m = threadIdx.x % M, grp = threadIdx.x / M;
  const bool active = grp < groups;
  float acc[GEMM_SMALL];
#pragma unroll
  for (int n = 0; n < GEMM_SMALL; ++n) acc[n] = 0.f;
  // rows of this block: a contiguous range (k0 .. k1), dealt to the row groups in turn
  const long per = ((long)a.K + gridDim.x - 1) / gridDim.x;
  const long k0 = (long)blockIdx.x * per, k1 = (k0 + per) < a.K ? (k0 + per) : a.K;
  const bool vn = a.b_vec && (N & 3) == 0;
  if (active)
#pragma unroll 4
    for (long k = k0 + grp; k < k1; k += groups) {
      const float av = a.A[k * a.lda + m];
      const float* br = a.B + k * a.ldb;
      if (vn) {
#pragma unroll
        for (int q = 0; q < GEMM_SMALL / 4; ++q)
          if (4 * q < N) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(br + 4 * q);
            acc[4 * q] = fmaf(av, v.x, acc[4 * q]);
            acc[4 * q + 1] = fmaf(av, v.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(av, v.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(av, v.w, acc[4 * q + 3]);
          }
      } else {
#pragma unroll
        for (int n = 0; n < GEMM_SMALL; ++n)
          if (n < N) acc[n] = fmaf(av, br[n], acc[n]);
      }
    }
#pragma unroll
  for (int n = 0; n < GEMM_SMALL; ++n)
    if (n < N) red[threadIdx.x * GEMM_SMALL + n] = active ? acc[n] : 0.f;
  __syncthreads();
  // fold the row groups in order; one partial [M][N] per block
  for (int i = threadIdx.x; i < M * N; i += 256) {
    const int mm = i / N, n = i - mm * N;
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += red[(g * M + mm) * GEMM_SMALL + n];
    a.slab[((long)blockIdx.x * M + mm) * N + n] = s;
  }
}

// The same product for dense operands (lda == M, ldb == N, 16-byte aligned) and power-of-two M, N
// up to 64 with M N <= 512: a block walks a contiguous range of rows in chunks; a chunk of both
// operands is copied to LDS as one flat stream of 16-byte pieces (the kernel above reads 4-byte
// pieces, a few per lane in flight: 1.3 TB/s on 8 x 2 over 8.4 M rows, and 64 x 8 fell to the
// 128-wide MFMA tiles at 0.9 TB/s); thread = (m, NB outputs n, row group g) sums the rows
// r = g (mod groups) of the chunk from LDS; groups folded in order, one partial [M][N] per block.
constexpr int GEMM_TALL_LDS_FLOATS = 8192;      // 32 KB of operands per chunk

__global__ __launch_bounds__(256) void adell_gemm_tall_lds_kernel(GemmArgs a, int R, int NB) {
  __shared__ __attribute__((aligned(16))) float sAB[GEMM_TALL_LDS_FLOATS];
  __shared__ float red[256 * 8];
  const int M = a.M, N = a.N;
  float* sA = sAB;                 // [R][M]
  float* sB = sAB + R * M;         // [R][N]
  const int nq = N / NB;           // output groups per m
  const int P = M * nq;            // threads per row slice
  const int groups = 256 / P;
  const int p = threadIdx.x % P, g = threadIdx.x / P;
  const int m = p / nq, n0 = (p - m * nq) * NB;
  const bool active = g < groups;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const long nchunks = ((long)a.K + R - 1) / R;
  const long per = (nchunks + gridDim.x - 1) / gridDim.x;
  const long c0 = (long)blockIdx.x * per, c1 = (c0 + per) < nchunks ? (c0 + per) : nchunks;
  for (long c = c0; c < c1; ++c) {
    const long k0 = c * R;
    const int rows = (int)(((long)a.K - k0) < R ? ((long)a.K - k0) : R);
    const f32x4* gA = reinterpret_cast<const f32x4*>(a.A + k0 * M);
    const f32x4* gB = reinterpret_cast<const f32x4*>(a.B + k0 * N);
    const int nA = rows * M / 4, nB = rows * N / 4;       // (rows % 4 == 0: the chunk length is a
    __syncthreads();                                      //  multiple of 64 and K of 4)
    for (int i = threadIdx.x; i < nA; i += 256) reinterpret_cast<f32x4*>(sA)[i] = gA[i];
    for (int i = threadIdx.x; i < nB; i += 256) reinterpret_cast<f32x4*>(sB)[i] = gB[i];
    __syncthreads();
    if (active) {
      if (NB == 8) {                                      // (n0 and N in eights: 16-byte reads)
        for (int r = g; r < rows; r += groups) {
          const float av = sA[r * M + m];
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sB + r * N + n0);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB + r * N + n0 + 4);
          acc[0] = fmaf(av, b0.x, acc[0]); acc[1] = fmaf(av, b0.y, acc[1]);
          acc[2] = fmaf(av, b0.z, acc[2]); acc[3] = fmaf(av, b0.w, acc[3]);
          acc[4] = fmaf(av, b1.x, acc[4]); acc[5] = fmaf(av, b1.y, acc[5]);
          acc[6] = fmaf(av, b1.z, acc[6]); acc[7] = fmaf(av, b1.w, acc[7]);
        }
      } else {
        for (int r = g; r < rows; r += groups) {
          const float av = sA[r * M + m];
          const float* br = sB + r * N + n0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < NB) acc[j] = fmaf(av, br[j], acc[j]);
        }
      }
    }
  }
  // fold the row groups in order
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = active ? acc[j] : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < M * N; i += 256) {
    const int mm = i / N, n = i - mm * N;
    const int pp = mm * nq + n / NB, j = n % NB;
    float sum = 0.f;
    for (int gg = 0; gg < groups; ++gg) sum += red[(gg * P + pp) * 8 + j];
    a.slab[((long)blockIdx.x * M + mm) * N + n] = sum;
  }
}

// chunk rows / outputs per thread / blocks of the LDS-staged form, or 0 when it does not apply
static int adell_gemm_tall_lds_plan(int M, int N, int K, long lda, long ldb, const float* A,
                                    const float* B, int* R, int* NB) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (!pow2(M) || !pow2(N) || M > 64 || N > 64 || M * N > 512 || M * N < 8 || K < 16384 || (K & 3)) return 0;
  if (A != nullptr && (lda != M || ldb != N || (((uintptr_t)A | (uintptr_t)B) & 15))) return 0;
  int nb = N < 8 ? N : 8;
  while (M * (N / nb) > 256) nb *= 2;           // (M N <= 512, nb <= 8: never needed beyond 8)
  if (nb > 8) return 0;
  int r = GEMM_TALL_LDS_FLOATS / (M + N);
  r &= ~63;
  if (r < 64) return 0;
  *R = r;
  *NB = nb;
  long b = ((long)K + r - 1) / r;               // chunks
  b = (b + 3) / 4;                              // at least four chunks per block
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

static int adell_gemm_tall_blocks(int M, int N, int K) {
  if (M > GEMM_SMALL || N > GEMM_SMALL || K < 16384) return 0;
  long b = K / 512;
  if (b > 1024) b = 1024;
  if (b < 2) b = 2;
  return (int)b;
}

struct GemmPlan {
  int skinny;  // 32 x 128 tile instead of 128 x 128
  int BM, BN, splits, ksteps_per_split;
};

static GemmPlan adell_gemm_plan(int M, int N, int K) {
  GemmPlan p;
  p.skinny = M <= 32;
  p.BM = p.skinny ? 32 : 128;
  p.BN = 128;
  const long tiles = (long)adell_cdiv(M, p.BM) * adell_cdiv(N, p.BN);
  const int ksteps = adell_cdiv(K, GEMM_BK);
  long s = adell_cdiv(512, tiles);
  // very long reductions over a small output (per-voxel weight gradients: k = voxels): more,
  // shorter splits -- at most 256 k-steps each, slabs capped at 64 MB
  const long by_len = adell_cdiv(ksteps, 256);
  if (s < by_len) s = by_len;
  if (s > ksteps / 4) s = ksteps / 4;
  const long slab_cap = (16L << 20) / ((long)M * N);
  if (s > slab_cap) s = slab_cap;
  if (s > 65535) s = 65535;
  if (s < 1) s = 1;
  p.ksteps_per_split = adell_cdiv(ksteps, (int)s);
  p.splits = adell_cdiv(ksteps, p.ksteps_per_split);
  return p;
}

extern "C" long adell_gemm_f32_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmPlan p = adell_gemm_plan(M, N, K);
  long need = p.splits > 1 ? (long)p.splits * M * N : 0;
  // (the layout decides at launch whether the tall small-output kernel runs: room for either)
  long tall = (long)adell_gemm_tall_blocks(M, N, K) * M * N;
  int r_ = 0, nb_ = 0;
  const long tall2 = (long)adell_gemm_tall_lds_plan(M, N, K, 0, 0, nullptr, nullptr, &r_, &nb_) * M * N;
  if (tall2 > tall) tall = tall2;
  return need > tall ? need : tall;
}

template <int WM, int WN, int TM, int TN>
static int adell_gemm_launch(const GemmArgs& a, int a_kc, int b_kc, dim3 grid, hipStream_t st) {
  if (a_kc && b_kc)
    hipLaunchKernelGGL((adell_gemm_f32_kernel<WM, WN, TM, TN, true, true>), grid, dim3(256), 0, st, a);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((adell_gemm_f32_kernel<WM, WN, TM, TN, true, false>), grid, dim3(256), 0, st, a);
  else if (!a_kc && !b_kc)
    hipLaunchKernelGGL((adell_gemm_f32_kernel<WM, WN, TM, TN, false, false>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((adell_gemm_f32_kernel<WM, WN, TM, TN, false, true>), grid, dim3(256), 0, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

// C[M][N] (row stride ldc) = A x B (+ bias[N]) (+ residual, row stride ldr).
// a_kc != 0: A element (m, k) at A[m*lda + k], else at A[k*lda + m];
// b_kc != 0: B element (k, n) at B[n*ldb + k], else at B[k*ldb + n].
// workspace: adell_gemm_f32_workspace_floats(M, N, K) floats (NULL when that is 0).
extern "C" int adell_gemm_f32(int M, int N, int K, const float* A, long lda, int a_kc,
                              const float* B, long ldb, int b_kc, float* C, long ldc,
                              const float* bias, const float* residual, long ldr,
                              float* workspace, void* stream) {
  ADELL_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad dims");
  ADELL_REQUIRE(A && B && C, "gemm: null pointer");
  ADELL_REQUIRE(lda >= (a_kc ? K : M) && ldb >= (b_kc ? K : N) && ldc >= N, "gemm: bad strides");
  ADELL_REQUIRE(!residual || ldr >= N, "gemm: bad residual stride");
  const GemmPlan p = adell_gemm_plan(M, N, K);
  int tallR = 0, tallNB = 0;
  const int tall_lds = (!a_kc && !b_kc) ? adell_gemm_tall_lds_plan(M, N, K, lda, ldb, A, B, &tallR, &tallNB) : 0;
  const int tall = tall_lds ? tall_lds : ((!a_kc && !b_kc) ? adell_gemm_tall_blocks(M, N, K) : 0);
  // (measured on SWIN-UNet's shapes: 10x / 5x for 8 -> 2 / 2 -> 8 features at 8.4 M rows; from
  // N K = 256 on the MFMA tiles are as fast or faster, 32 x 8 at 2 M rows 180 vs 250 us)
  // (round 5: without the 64-bit division per output the thread-per-output kernel wins up to
  // N K = 512 -- 2 097 152 x 8 x 32: 249 us on the tiles)
  const bool rows_small = a_kc && K <= GEMM_ROWS_K && N <= GEMM_ROWS_N && (long)N * K <= 512 && M >= 65536;
  ADELL_REQUIRE(rows_small || (tall ? workspace != nullptr : (p.splits == 1 || workspace)),
                "gemm: workspace required for this shape");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.residual = residual; a.slab = workspace;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr;
  a.splits = p.splits;
  a.ksteps_per_split = p.ksteps_per_split;
  a.a_vec = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
  a.b_vec = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (rows_small) {
    long blocks = ((long)M * N + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (b_kc)
      hipLaunchKernelGGL(adell_gemm_rows_small_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(adell_gemm_rows_small_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  if (tall) {
    a.splits = tall;
    if (tall_lds)
      hipLaunchKernelGGL(adell_gemm_tall_lds_kernel, dim3((unsigned)tall), dim3(256), 0, st, a, tallR, tallNB);
    else
      hipLaunchKernelGGL(adell_gemm_tall_small_kernel, dim3((unsigned)tall), dim3(256), 0, st, a);
    const long blocks = ((long)M * N + 63) / 64;
    hipLaunchKernelGGL(adell_gemm_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  dim3 grid(adell_cdiv(M, p.BM), adell_cdiv(N, p.BN), p.splits);
  ADELL_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
  int rc = p.skinny ? adell_gemm_launch<1, 4, 1, 1>(a, a_kc, b_kc, grid, st)
                    : adell_gemm_launch<2, 2, 2, 2>(a, a_kc, b_kc, grid, st);
  if (rc != ADELL_OK) return rc;
  if (p.splits > 1) {
    if (p.splits <= 8) {
      long blocks = ((long)M * N + 255) / 256;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(adell_gemm_reduce_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
      const long blocks = ((long)M * N + 63) / 64;
      hipLaunchKernelGGL(adell_gemm_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, a);
    }
    ADELL_CHECK_HIP(hipGetLastError());
  }
  return ADELL_OK;
}
