// Streaming form of the f16x3 GEMM (gemm_f16x3.hip) for Linear layers over MANY rows with a SMALL
// weight: C[M][N] = A[M][K] W (+ bias) (+ residual) (x act' / + act output), A K-contiguous, M in the
// tens of thousands to millions, K x N a few 10^4 .. 10^6 -- ConvNeXt's point-wise MLP
// (res_blocks.py:588-604: 262 144 x 96 -> 384 -> 96 at BASELINE config 4), the token-wise Linear
// layers of ViT / SWIN (linear_blocks.py:85-103, vit.py:531-536) and their backward-data products.
// These problems are HBM-bound (arithmetic intensity ~40 FLOP/B against a ridge of ~100 for the
// split arithmetic); the tile kernel of gemm_f16x3.hip ran them at 2 - 2.9 TB/s because every
// 128 x 128 tile is its own block with a two-stage k-loop (nothing in flight across tiles) and
// because each stage passes through registers, a split pass and three barriers.
//
// Here
//   * W is split ONCE per call into the LDS image of the B operand (hi | lo fp16 rows, one
//     power-of-two scale per output column) by a small pack kernel; it is streamed from L2 by LDS-DMA;
//   * a persistent block (one per CU: 4 compute waves + 4 loader waves, one of each per SIMD) walks
//     a contiguous range of (128-row tile, 128-column slice) pairs;
//   * the loader waves fill a ring of four 32-k stages by global_load_lds_dwordx4 -- the fp32 A tile
//     as it lies in memory (no registers, no split pass, no ds_write) and the packed W slice -- three
//     stages (48 KB of A) in flight per CU ACROSS tile boundaries, retired by counted vmcnt waits
//     that only the loader waves execute (the compute waves' stores live on their own counters);
//   * a compute wave owns 32 rows x all columns of the slice: it reads its fp32 A rows from LDS
//     (16-byte pieces XOR-permuted through the DMA's source addresses: conflict-free
//     ds_read_b128), takes the power-of-two scale of the 32 x 32 block from its absmax, splits it to
//     (hi, lo) in registers -- each A element once -- and runs a_lo b_hi + a_hi b_lo + a_hi b_hi on
//     v_mfma_f32_32x32x16_f16 against up to four 32-column B fragments; accumulators are rescaled
//     by the exact ratio when the block exponent changes;
//   * one barrier per stage; the epilogue (scale undo, bias, residual, activation pair) stores from
//     registers: 128-byte row segments per half wave.
// Numerics: those of gemm_f16x3.hip (22 mantissa bits per product, fp32 accumulation).
#include "common.h"
#include <type_traits>

typedef _Float16 rb_half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char rb_lds_char;
typedef __attribute__((address_space(1))) const char rb_glb_char;

namespace {

constexpr int RB_CW = 8;                        // compute waves (two per SIMD), + 4 loader waves
constexpr int RB_M = 32 * RB_CW;                // rows per tile
constexpr int RB_N = 128;                       // columns per slice (up to 4 MFMA tiles per wave)
constexpr int RB_K = 32;                        // k per stage (two 16-k chunks)
constexpr int RB_D = 3;                         // ring depth
constexpr int RB_A_BYTES = RB_M * RB_K * 4;     // 32 KB: [256 rows][8 pieces of 16 B], pieces permuted
constexpr int RB_B_BYTES = 2 * RB_N * 64;       // 16 KB: [2 chunks][128 rows][64 B = hi | lo]
constexpr int RB_STAGE = RB_A_BYTES + RB_B_BYTES;
constexpr int RB_LDS = RB_D * RB_STAGE;         // 144 KB
constexpr int RB_NA = RB_A_BYTES / 1024 / 4;    // A / B DMA instructions per loader wave and stage
constexpr int RB_NB = RB_B_BYTES / 1024 / 4;
constexpr int RB_NDMA = RB_NA + RB_NB;
static_assert(RB_LDS <= 160 * 1024 && (RB_D - 1) * RB_NDMA < 64, "ring fits LDS, counted waits fit vmcnt");

struct RowsArgs {
  const float* A;
  const char* Bimg;       // [slice][k stage][2 chunks][128 rows][64 B], rows permuted as rb_lds_off
  const float* bscale;    // [slices * 128]: 2^-k of the column's pack scale
  float* C;
  const float* bias;
  const float* residual;
  float* act_out;
  const float* dact_in;
  int M, N, K;
  long lda, ldc, ldr;
  int nslices, kstages, tiles, tiles_per_block;
  float act_p;
};

// byte offset of (row, 16-byte slot) inside a 128-row chunk image (the permutation of gemm_f16x3.hip)
__device__ __host__ __forceinline__ int rb_lds_off(int r, int slot) {
  const int r2 = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
  const int s2 = slot ^ (r & 3) ^ ((r >> 4) & 3);
  return r2 * 64 + s2 * 16;
}

__device__ __forceinline__ int rb_scale_exp(unsigned amax_bits) {
  const int ebits = (amax_bits >> 23) & 0xff;
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 13 - (ebits - 127);
  if (k > 100) k = 100;
  if (k < -100) k = -100;
  return k;
}

// max over the 64 lanes as a wave-uniform value: four DPP steps inside each row of 16 lanes, two
// row broadcasts, one readlane -- no LDS round trips (__shfl_xor is a ds_bpermute per step: six
// dependent LDS latencies per stage)
__device__ __forceinline__ float rb_wave_max(float v) {
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

__device__ __forceinline__ void rb_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

// ---- W -> packed B image --------------------------------------------------------------------------
// A block owns R rows (output columns n) of the weight over the whole K and reads them ONCE into LDS
// (one global round trip: a two-pass form -- absmax, then split -- took 11 us per call, 0.27 ms of a
// config-4 step): row absmax -> power-of-two scale -> (hi, lo) pieces of every 16-k chunk.
// W(k, n) = b_kc ? B[n * ldb + k] : B[k * ldb + n]. R (a power of two <= 32) divides 128; grid = N
// padded to 128, / R.
__global__ __launch_bounds__(256) void adell_gemm_rows_pack_kernel(const float* __restrict__ B, long ldb,
                                                                   int b_kc, int N, int K, int R,
                                                                   char* __restrict__ img,
                                                                   float* __restrict__ bscale) {
  extern __shared__ __attribute__((aligned(16))) float ptile[];   // [R][K + 4], then R scale words
  const int tid = threadIdx.x, ldt = K + 4;
  const int nfirst = blockIdx.x * R;                    // first padded column of this block
  unsigned* sexp = reinterpret_cast<unsigned*>(ptile + (size_t)R * ldt);
  const int kstages = K / RB_K;
  // the tile, coalesced along the operand's contiguous axis
  if (b_kc) {
    const int k4 = K >> 2;                              // K is a multiple of 32
    for (int i = tid; i < R * k4; i += 256) {
      const int rr = i / k4, q = i - rr * k4, n = nfirst + rr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) v = *reinterpret_cast<const float4*>(B + (long)n * ldb + 4 * q);
      *reinterpret_cast<float4*>(ptile + rr * ldt + 4 * q) = v;
    }
  } else {
    const int lr = __ffs(R) - 1;
    for (int i = tid; i < R * K; i += 256) {
      const int rr = i & (R - 1), k = i >> lr, n = nfirst + rr;
      ptile[rr * ldt + k] = n < N ? B[(long)k * ldb + n] : 0.f;
    }
  }
  __syncthreads();
  // row absmax: 256 / R threads per row, folded through LDS-free shuffles where they share a wave
  {
    const int tpr = 256 / R, rr = tid / tpr, sub = tid - rr * tpr;
    float mx = 0.f;
    for (int k = sub; k < K; k += tpr) mx = fmaxf(mx, fabsf(ptile[rr * ldt + k]));
    for (int o = 1; o < tpr && o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (sub == 0) sexp[rr] = 0u;
    __syncthreads();
    if ((sub & 63) == 0) atomicMax(&sexp[rr], __float_as_uint(mx));    // <= 4 waves per row
  }
  __syncthreads();
  if (tid < R) {
    const int kb = rb_scale_exp(sexp[tid]);
    bscale[nfirst + tid] = __int_as_float((127 - kb) << 23);
    sexp[tid] = (unsigned)(kb + 127);
  }
  __syncthreads();
  // item = (row rr, 16-k chunk): 64 bytes of the image
  const int nchunks = K / 16;
  for (int it = tid; it < R * nchunks; it += 256) {
    const int rr = it / nchunks, ch = it - rr * nchunks;
    const float scale = __int_as_float(sexp[rr] << 23);
    const float* src = ptile + rr * ldt + ch * 16;
    rb_half8 h[2], l[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 f = *reinterpret_cast<const float4*>(src + 4 * q);
      const float v[4] = {f.x * scale, f.y * scale, f.z * scale, f.w * scale};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int jx = 4 * q + e;
        const _Float16 hh = (_Float16)v[e];
        h[jx >> 3][jx & 7] = hh;
        l[jx >> 3][jx & 7] = (_Float16)(v[e] - (float)hh);
      }
    }
    const int np = nfirst + rr, slice = np / RB_N, r = np - slice * RB_N;
    const int ks = ch >> 1, c = ch & 1;
    char* base = img + (((size_t)slice * kstages + ks) * 2 + c) * (RB_N * 64);
    *reinterpret_cast<rb_half8*>(base + rb_lds_off(r, 0)) = h[0];
    *reinterpret_cast<rb_half8*>(base + rb_lds_off(r, 1)) = h[1];
    *reinterpret_cast<rb_half8*>(base + rb_lds_off(r, 2)) = l[0];
    *reinterpret_cast<rb_half8*>(base + rb_lds_off(r, 3)) = l[1];
  }
}

// ---- the streaming kernel -------------------------------------------------------------------------
// EPI 0: C = acc + bias (+ residual);  1: also act_out = act(C);  2: C = (acc + bias ...) * act'(dact_in).
// ACT: the activation of EPI 1 / 2 (compile time: one inlined activation at the 64 store sites).
template <int EPI, int ACT>
__global__ __launch_bounds__(64 * (RB_CW + 4), (RB_CW + 4) / 4) void adell_gemm_rows_f16x3_kernel(RowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t_beg = blockIdx.x * a.tiles_per_block;
  const int t_end = (t_beg + a.tiles_per_block) < a.tiles ? (t_beg + a.tiles_per_block) : a.tiles;
  if (t_beg >= t_end) return;
  const int KS = a.kstages;
  const int total = (t_end - t_beg) * KS;      // stages of this block

  if (wave >= RB_CW) {
    // ================================ loader waves ===============================================
    const int lw = wave - RB_CW;
    // A: instruction ia = RB_NA lw + u covers tile rows 8 ia .. + 7; lane: row 8 ia + (lane >> 3), LDS
    // position lane & 7 holds the row's piece (lane & 7) ^ ((row >> 1) & 7)
    int arow[RB_NA], apiece[RB_NA];
#pragma unroll
    for (int u = 0; u < RB_NA; ++u) {
      const int r = 8 * (RB_NA * lw + u) + (lane >> 3);
      arow[u] = r;
      apiece[u] = (lane & 7) ^ ((r >> 1) & 7);
    }
    // issue cursor: (tile it, k stage iks, ring slot islot); per-lane source pointers advance by one
    // stage per issue and are rebuilt once per tile (no division, no 64-bit multiply per stage)
    int it = t_beg, iks = 0, islot = 0;
    const char* aptr[RB_NA];
    const char* bptr;
    auto set_tile = [&](int t) {
      const int mt = t / a.nslices, sl = t - mt * a.nslices;
#pragma unroll
      for (int u = 0; u < RB_NA; ++u) {
        int row = mt * RB_M + arow[u];
        row = row < a.M ? row : a.M - 1;          // rows past the matrix: a valid row, never stored
        aptr[u] = reinterpret_cast<const char*>(a.A) + (size_t)row * a.lda * 4 + apiece[u] * 16;
      }
      bptr = a.Bimg + (size_t)sl * KS * RB_B_BYTES + lane * 16 + lw * (RB_NB * 1024);
    };
    set_tile(it);
    auto issue = [&]() {
      const unsigned sbase = (unsigned)(islot * RB_STAGE);
#pragma unroll
      for (int u = 0; u < RB_NA; ++u) {
        const unsigned off = __builtin_amdgcn_readfirstlane(sbase + (unsigned)((RB_NA * lw + u) * 1024));
        __builtin_amdgcn_global_load_lds((rb_glb_char*)aptr[u], (rb_lds_char*)smem + off, 16, 0, 0);
        aptr[u] += RB_K * 4;
      }
#pragma unroll
      for (int u = 0; u < RB_NB; ++u) {
        const unsigned off = __builtin_amdgcn_readfirstlane(
            sbase + (unsigned)(RB_A_BYTES + (RB_NB * lw + u) * 1024));
        __builtin_amdgcn_global_load_lds((rb_glb_char*)(bptr + u * 1024), (rb_lds_char*)smem + off, 16, 0, 0);
      }
      bptr += RB_B_BYTES;
      islot = islot + 1 == RB_D ? 0 : islot + 1;
      if (++iks == KS) {
        iks = 0;
        if (++it < t_end) set_tile(it);
      }
    };
    int issued = 0;
    for (; issued < RB_D - 1 && issued < total; ++issued) issue();
    for (int g = 0; g < total; ++g) {
      // stage g has landed when at most the younger issued stages are outstanding
      const int younger = issued - 1 - g;
      if (younger >= 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RB_NDMA) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      rb_barrier();                    // b_g: stage g readable, the slot of stage g - 1 free
      if (issued < total) {
        issue();
        ++issued;
      }
    }
    return;
  }

  // ================================== compute waves ================================================
  __builtin_amdgcn_s_setprio(2);
  const int li = lane & 31, lh = lane >> 5;
  const int arow = 32 * wave + li;                       // this lane's A row inside the tile
  const int asw = (arow >> 1) & 7;
  const int aoff = arow * 128;
  int boff[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    boff[j][0] = rb_lds_off(32 * j + li, lh);
    boff[j][1] = rb_lds_off(32 * j + li, 2 + lh);
  }
  f32x16 acc[4];
  int kprev = 0;
  int slot = 0;
#pragma unroll 1
  for (int t = t_beg; t < t_end; ++t) {
    const int mt = t / a.nslices, sl = t - mt * a.nslices;
    const int n0 = sl * RB_N, m0 = mt * RB_M;
    int nt = (a.N - n0 + 31) >> 5;
    nt = nt > 4 ? 4 : nt;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // this lane's column constants of the slice, in flight under the k loop
    float cs[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + 32 * j + li;
      cs[j] = a.bscale[col];
      bv[j] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
    }
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
      rb_barrier();                                    // b_g
      const char* sb = smem + slot * RB_STAGE;
      slot = slot + 1 == RB_D ? 0 : slot + 1;
      float4 fa[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int q0 = 4 * c + 2 * lh;
        fa[c][0] = *reinterpret_cast<const float4*>(sb + aoff + ((q0 ^ asw) << 4));
        fa[c][1] = *reinterpret_cast<const float4*>(sb + aoff + (((q0 + 1) ^ asw) << 4));
      }
      // block exponent of this wave's 32 x 32 piece of A (multiples of 8: it rarely changes)
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          mx = fmaxf(fmaxf(fmaxf(mx, fabsf(fa[c][h].x)), fmaxf(fabsf(fa[c][h].y), fabsf(fa[c][h].z))),
                     fabsf(fa[c][h].w));
      mx = rb_wave_max(mx);
      const int kA = 8 * (rb_scale_exp(__float_as_uint(mx)) >> 3);
      if (ks > 0 && kA != kprev) {
        const float fix = __int_as_float((kA - kprev + 127) << 23);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] *= fix;
      }
      kprev = kA;
      const float sA = __int_as_float((kA + 127) << 23);
      rb_half8 ah[2], al[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float v[8] = {fa[c][0].x * sA, fa[c][0].y * sA, fa[c][0].z * sA, fa[c][0].w * sA,
                            fa[c][1].x * sA, fa[c][1].y * sA, fa[c][1].z * sA, fa[c][1].w * sA};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const _Float16 hh = (_Float16)v[q];
          ah[c][q] = hh;
          al[c][q] = (_Float16)(v[q] - (float)hh);
        }
      }
      const char* sbB = sb + RB_A_BYTES;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nt) {
            const rb_half8 bh = *reinterpret_cast<const rb_half8*>(sbB + c * (RB_N * 64) + boff[j][0]);
            const rb_half8 bl = *reinterpret_cast<const rb_half8*>(sbB + c * (RB_N * 64) + boff[j][1]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c], bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bh, acc[j], 0, 0, 0);
          }
    }
    // ---- epilogue of the tile: C row (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32, column li of tile j.
    // Wave-uniform tile base + 32-bit element offsets (one scalar base, no 64-bit address arithmetic per
    // store); the loads of an epilogue operand are issued 16 at a time, ahead of their use.
    const float osc = __int_as_float((127 - kprev) << 23);
    const int rw = m0 + 32 * wave;                        // first row of this wave
    const bool full = rw + 32 <= a.M;
    float* cb = a.C + (size_t)rw * a.ldc + n0;
    const float* rb = a.residual ? a.residual + (size_t)rw * a.ldr + n0 : nullptr;
    const float* db = EPI == 2 ? a.dact_in + (size_t)rw * a.ldc + n0 : nullptr;
    float* ob = EPI == 1 ? a.act_out + (size_t)rw * a.ldc + n0 : nullptr;
    const unsigned ldc = (unsigned)a.ldc, ldr = (unsigned)a.ldr;
    // Address of (register r, tile j) = [uniform row pointer of r] + [ONE per-lane element offset] + 32 j:
    // the row pointers are scalar arithmetic, the lane offset is shared by all 64 stores (per-store
    // lane offsets are loop invariants the compiler hoists and spills: 64 of them)
    const unsigned lo_c = 4u * lh * ldc + li, lo_r = 4u * lh * ldr + li;
    const int rows_left = a.M - rw - 4 * lh;               // rows this lane's half may still touch
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + 32 * j + li;
      if (j < nt && col < a.N) {
        const float csj = osc * cs[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * csj + bv[j];
        // an epilogue operand: its 16 loads first, all in flight together (interleaved with the
        // stores they would wait for each other: the compiler cannot tell the tensors apart)
        if (rb != nullptr) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ru = (r & 3) + 8 * (r >> 2);
            t[r] = (full || ru < rows_left) ? (rb + (size_t)ru * ldr + 32 * j)[lo_r] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] += t[r];
        }
        if constexpr (EPI == 2) {
          float t[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ru = (r & 3) + 8 * (r >> 2);
            t[r] = (full || ru < rows_left) ? (db + (size_t)ru * ldc + 32 * j)[lo_c] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] *= adell_act_grad(ACT, t[r], a.act_p);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ru = (r & 3) + 8 * (r >> 2);            // wave-uniform part of the row
          if (full || ru < rows_left) {
            (cb + (size_t)ru * ldc + 32 * j)[lo_c] = acc[j][r];
            if constexpr (EPI == 1) (ob + (size_t)ru * ldc + 32 * j)[lo_c] = adell_act_fwd(ACT, acc[j][r], a.act_p);
          }
        }
      }
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------
namespace {

int rb_cu_count() {
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

}  // namespace

// Whether the streaming kernel takes this problem (gemm_f16x3.hip asks before it plans its tiles):
// A K-contiguous, K a multiple of 32, many rows, a weight small enough to pack per call, and an
// epilogue it has an instance of (no activation, or GELU for the activation pair).
bool adell_gemm_rows_ok(int M, int N, int K, const float* A, long lda, int a_kc, const float* B, long ldb,
                        int act, bool epi, bool dact) {
  if (g_adell_tune.gemm_norows) return false;
  if (!a_kc || (K % RB_K) != 0 || K > 8192 || N < 32 || M < 2048) return false;
  if ((long)K * N > (4L << 20)) return false;                 // pack per call: <= 16 MB of weight
  if ((((uintptr_t)A) & 15) || (lda & 3) || (ldb & 3) || (((uintptr_t)B) & 15)) return false;
  if (epi && act != ADELL_ACT_GELU) return false;
  // (the act' epilogue -- C * act'(saved) -- reads a second [M][N] tensor with 4-byte loads from the
  // accumulator layout: measured slower than the tile kernel's LDS-transposed epilogue, 591 vs 473 us
  // at 262 144 x 96 -> 384; the instance stays for completeness, the dispatch does not pick it)
  if (dact) return false;
  // enough tiles for the persistent grid: at least one per CU
  const long tiles = (long)adell_cdiv(M, RB_M) * adell_cdiv(N, RB_N);
  return tiles >= (long)rb_cu_count();
}

// floats of workspace: the packed image + the column scales
long adell_gemm_rows_workspace_floats(int M, int N, int K) {
  (void)M;
  const long slices = adell_cdiv(N, RB_N);
  return slices * RB_N * (long)K + slices * RB_N;
}

int adell_gemm_rows_run(int M, int N, int K, const float* A, long lda, const float* B, long ldb, int b_kc,
                        float* C, long ldc, const float* bias, const float* residual, long ldr,
                        float* workspace, hipStream_t st, int act, float act_p, float* act_out,
                        const float* dact_in) {
  ADELL_REQUIRE(workspace && (((uintptr_t)workspace) & 15) == 0, "gemm_rows: workspace required (16-byte aligned)");
  RowsArgs a;
  const int slices = adell_cdiv(N, RB_N);
  char* img = reinterpret_cast<char*>(workspace);
  float* bscale = workspace + (size_t)slices * RB_N * K;
  // rows per pack block: the tile [R][K + 4] fp32 stays within 64 KB of LDS (K <= 8192: R >= 1);
  // outer-contiguous weights want many rows per block (their rows are the contiguous axis)
  int R = b_kc ? 8 : 32;
  while (R > 1 && (size_t)R * (K + 4) * 4 > 64 * 1024) R >>= 1;
  const size_t plds = (size_t)R * (K + 4) * 4 + 32 * sizeof(unsigned);
  ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_gemm_rows_pack_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  hipLaunchKernelGGL(adell_gemm_rows_pack_kernel, dim3((unsigned)(slices * RB_N / R)), dim3(256), plds, st,
                     B, ldb, b_kc, N, K, R, img, bscale);
  a.A = A; a.Bimg = img; a.bscale = bscale; a.C = C; a.bias = bias; a.residual = residual;
  a.act_out = act_out; a.dact_in = dact_in;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr;
  a.nslices = slices; a.kstages = K / RB_K;
  const long tiles = (long)adell_cdiv(M, RB_M) * slices;
  ADELL_REQUIRE(tiles < 0x7fffffffL, "gemm_rows: too many tiles");
  a.tiles = (int)tiles;
  int blocks = rb_cu_count();
  if (blocks > tiles) blocks = (int)tiles;
  a.tiles_per_block = (int)((tiles + blocks - 1) / blocks);
  blocks = (int)((tiles + a.tiles_per_block - 1) / a.tiles_per_block);
  a.act_p = act_p;
  auto launch = [&](auto kern) -> int {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, RB_LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * (RB_CW + 4)), RB_LDS, st, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  };
  (void)act;
  if (act_out) return launch(adell_gemm_rows_f16x3_kernel<1, ADELL_ACT_GELU>);
  if (dact_in) return launch(adell_gemm_rows_f16x3_kernel<2, ADELL_ACT_GELU>);
  return launch(adell_gemm_rows_f16x3_kernel<0, ADELL_ACT_IDENTITY>);
}
