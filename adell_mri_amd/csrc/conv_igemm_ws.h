// Persistent, wave-specialised instance of the f16x3 implicit-GEMM convolution for the layers
// that carry the FLOPs of a U-Net: 3x3x3 taps, stride 1, 16-channel-aligned sources
// (same arithmetic, LDS images and epilogue contract as conv_igemm_f16.h, SPEC = 1 / 3).
//
// Why: in the one-brick-per-block kernel 30-45 % of a block's life is halo / weight staging,
// the block-wide absmax and the epilogue, and the two blocks sharing a CU overlap those phases
// with each other's MFMA loops only by chance (MFMA pipe busy 47-55 %). Here ONE 512-thread
// block per CU walks a contiguous range of bricks and splits its waves by role:
//
//   waves 0-3  (compute)  fragment reads + MFMAs of stage s, epilogue of a finished brick
//   waves 4-7  (loaders)  meanwhile: weight slice of stage s+1 -> sB[(s+1)&1]; halo of the NEXT
//                         16-channel chunk (or of the next brick's first chunk): global loads +
//                         absmax in the first stage of the current chunk, fp16 hi/lo split +
//                         LDS store in the second
//
// A stage = one tap group (9 taps of a kz plane for the 8x8x4 brick, <= 7 taps for 8x8x8) of one
// chunk; one workgroup barrier per stage is the only synchronisation. Both LDS images are double
// buffered: 2 x 37.5 KB halo + 2 x 36 KB weights (64-channel tile) or 2 x 62.5 KB + 2 x 14 KB
// (32-channel tile, 8x8x8 brick). Each SIMD hosts one compute and one loader wave, so the loader's
// VALU / memory work issues in the shadow of the other wave's MFMAs.
//
// Blocks are dealt to the XCDs in contiguous brick ranges (blockIdx & 7 labels the blocks that
// share an L2), so neighbouring halos and the weights meet in one L2.
#pragma once
#include "conv_igemm_f16.h"

// one LDS-visible rendezvous of all 8 waves; the asm memory clobber keeps the compiler from moving
// LDS traffic across it, lgkmcnt(0) retires this wave's own LDS operations first
__device__ __forceinline__ void adell_ws_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef ADELL_WS_CENSUS
// cycle census of the two roles (debug builds only; read back with adell_debug_ws_prof):
// [0] compute: stage bodies, [1] compute: barrier waits, [2] compute: epilogues,
// [3] loader: stage work, [4] loader: barrier waits, [5] blocks counted
__device__ unsigned long long g_ws_prof[16];
#define WS_T0() const unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define WS_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define WS_STAMP(var)
#endif

struct ConvWsItem {
  int nb, ncol, tile, ox0, oy0, oz0;
};

// ROWS = 1 (round 4): every source holds SPLIT ROWS (ConvF16Extra::xs0 / xs1: the 64-byte hi | lo rows
// of this very LDS image, written by the producer, adell_norm_act_fwd_split). The loaders then
// have no arithmetic at all: the halo of the next chunk arrives by LDS-DMA like the weights -- a
// wave-instruction fills 16 rows, the slot permutation and the zero padding ride on each lane's
// SOURCE address (rows outside the tensor read e.zeros) -- no halo registers, no absmax, no split, no
// ds_write. (Round 2 measured the register-staged halo path of the loaders as what costs this design
// its lead: compute waves alone 453-470 TF, with that path 345 TF.)
template <int MT, int NT, int BZ, int ROWS = 0>
__global__ __launch_bounds__(512, 2)
void adell_conv_igemm_ws_kernel(ConvArgs a, ConvF16Extra e, int n_items, int nct) {
  constexpr int BN = NT * 32, CC = 16;
  constexpr int HX = 10, HY = 10, HZ = BZ + 2, HV = HX * HY * HZ;
  constexpr int GT = BZ == 8 ? 7 : 9;                 // taps per weight group
  constexpr int NGRP = (27 + GT - 1) / GT;            // stages per chunk (3 or 4)
  constexpr int LTZ = BZ == 8 ? 3 : 2;
  constexpr int NLD = 256;                            // loader threads
  constexpr int KEEP = ROWS ? 1 : (HV + NLD - 1) / NLD;   // halo voxels per loader thread (fp32 path)
  // (ROWS: the halo image is padded to whole 1 KB DMA pieces; the pad rows read the zero page)
  constexpr int HPIECES = (HV + 15) / 16;             // 16-row pieces of a halo image
  constexpr int HP = (HPIECES + 3) / 4;               // ... per loader wave
  constexpr size_t A_BYTES = ROWS ? (size_t)HPIECES * 1024 : (size_t)HV * 64,
                   B_BYTES = (size_t)GT * BN * 64;
  static_assert(NGRP >= 3, "the halo of the next chunk is staged over three stages");

  extern __shared__ float smem[];
  char* sA = reinterpret_cast<char*>(smem);           // [2][HV][64 B]
  char* sB = sA + 2 * A_BYTES;                        // [2][GT][BN][64 B]
  float* sMaxL = reinterpret_cast<float*>(sB + 2 * B_BYTES);   // [4] loader-wave absmax
  int* sK = reinterpret_cast<int*>(sMaxL + 4);                 // [2] scale exponent per halo buffer
  float* sRed = reinterpret_cast<float*>(sK + 2);              // [4][BN][2] statistics of a brick

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // ---- this block's contiguous range of work items (brick x column tile x batch item) -----------
  const int nsp = a.ntx * a.nty * a.ntz;
  int first, count;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int Wx = (n_items + 7) >> 3, Wb = (Wx + nbx - 1) / nbx;
    first = xcd * Wx + j * Wb;
    int last = first + Wb;
    const int xend = (xcd + 1) * Wx < n_items ? (xcd + 1) * Wx : n_items;
    if (last > xend) last = xend;
    count = last > first ? last - first : 0;
  }
  if (count == 0) return;   // whole block: before the first barrier
  auto item_at = [&](int idx) -> ConvWsItem {
    ConvWsItem it;
    int t = idx % nsp;
    const int rest = idx / nsp;
    it.tile = t;
    it.ncol = rest % nct;
    it.nb = rest / nct;
    const int tx = t % a.ntx;
    t /= a.ntx;
    it.ox0 = tx << 3;
    it.oy0 = (t % a.nty) << 3;
    it.oz0 = (t / a.nty) << LTZ;
    return it;
  };
  const int nchunk = a.Cin / CC;

  if (wave >= 4) {
    // =============================== loader waves ==============================================
    const int t = tid - 256;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(1))) char glb_char;
    if constexpr (ROWS != 0) {
      // ============================ loaders, split-row sources: DMA only ==========================
      constexpr int PIECES = (GT * BN / 16 + 3) / 4;      // weight pieces per loader wave
      const int lw = wave - 4;
      // per piece of this wave: byte offset of this lane's 16 bytes inside the batch item's chunk-0
      // rows (voxel * C * 4 + slot * 16; the chunk adds 64 per chunk), or -1: zero page
      int voff[HP];
      auto set_item = [&](const ConvWsItem& it) {
        const int lx0 = it.ox0 - a.PW, ly0 = it.oy0 - a.PH, lz0 = it.oz0 - a.PD;
#pragma unroll
        for (int j = 0; j < HP; ++j) {
          const int hv = (lw + 4 * j) * 16 + (lane >> 2);
          int off = -1;
          if (hv < HV) {
            const int hz = hv / (HX * HY), rem = hv - hz * (HX * HY);
            const int hy = rem / HX, hx = rem - hy * HX;
            const int rx = lx0 + hx, ry = ly0 + hy, rz = lz0 + hz;
            if ((rx >= 0) & (ry >= 0) & (rz >= 0) & (rx < a.W) & (ry < a.H) & (rz < a.D)) {
              // physical slot lane & 3 of the row holds logical piece slot ^ sw (conv_igemm_f16.h)
              const int sw = (hx >> 1) & 3;
              off = ((rz * a.H + ry) * a.W + rx);          // voxel index: scaled by the source below
              off = off * 4 + ((lane & 3) ^ sw);           // (voxel, logical piece)
            }
          }
          voff[j] = off;
        }
      };
      // halo pieces [j0, j1) of this wave for chunk ch of item `it` -> halo buffer `buf`
      auto dma_halo = [&](const ConvWsItem& it, int ch, int buf, int j0, int j1) {
        const int c0 = ch * CC;
        const bool firstsrc = c0 < a.C0;
        const size_t vox0 = (size_t)it.nb * a.D * a.H * a.W;
        const unsigned cs = firstsrc ? a.C0 : a.C1;       // channels of the source: cs * 4 bytes per voxel
        const char* src = firstsrc ? e.xs0 + vox0 * a.C0 * 4 + (size_t)c0 * 4
                                   : e.xs1 + vox0 * a.C1 * 4 + (size_t)(c0 - a.C0) * 4;
        const ADELL_GLOBAL char* base = adell_uniform_ptr(src);
        const ADELL_GLOBAL char* zero = adell_uniform_ptr(e.zeros);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
          if (j < j0 || j >= j1) continue;
          const int piece = lw + 4 * j;
          if (piece < HPIECES) {    // wave-uniform
            const int o = voff[j];
            const ADELL_GLOBAL char* p =
                o >= 0 ? base + ((unsigned)(o >> 2) * cs * 4u + (unsigned)(o & 3) * 16u)
                       : zero + (lane & 3) * 16;
            const unsigned off = __builtin_amdgcn_readfirstlane(
                (unsigned)(buf * A_BYTES + piece * 1024));
            __builtin_amdgcn_global_load_lds((glb_char*)p, (lds_char*)smem + off, 16, 0, 0);
          }
        }
      };
      auto dma_weights = [&](int ch, int grp, int ncol, int buf) {
        const int n0 = ncol * BN;
        const int tpg = (27 - grp * GT) < GT ? (27 - grp * GT) : GT;
        const int rows = tpg * BN;
        const char* wbase = reinterpret_cast<const char*>(e.wh) +
                            ((size_t)(grp * GT) * a.Cout * nchunk + ch) * 64;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
          const int piece = k * 4 + lw;
          if (piece * 16 < rows) {   // wave-uniform
            const int row = piece * 16 + (lane >> 2);
            const int tl = row / BN, n = row % BN;
            int col = n0 + n;
            if (col >= a.Cout) col = a.Cout - 1;
            const int lslot = (lane & 3) ^ ((n >> 2) & 3);
            const char* src = wbase + ((size_t)tl * a.Cout + col) * nchunk * 64 + lslot * 16;
            const unsigned off = __builtin_amdgcn_readfirstlane(
                (unsigned)(2 * A_BYTES + buf * B_BYTES + piece * 1024));
            __builtin_amdgcn_global_load_lds((glb_char*)src, (lds_char*)smem + off, 16, 0, 0);
          }
        }
      };
      auto chunk_exp = [&](const ConvWsItem& it, int ch) -> int {
        const int c0 = ch * CC;
        return c0 < a.C0 ? e.xk0[it.nb * (a.C0 >> 4) + (c0 >> 4)]
                         : e.xk1[it.nb * (a.C1 >> 4) + ((c0 - a.C0) >> 4)];
      };
      auto flush_stats = [&](const ConvWsItem& it) {
        if (a.part == nullptr || t >= BN) return;
        const int n = it.ncol * BN + t;
        if (n >= a.Cout) return;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          t1 += sRed[(w * BN + t) * 2 + 0];
          t2 += sRed[(w * BN + t) * 2 + 1];
        }
        float* p = a.part + (((size_t)it.nb * nsp + it.tile) * a.Cout + n) * 2;
        p[0] = t1;
        p[1] = t2;
      };
      ConvWsItem cur = item_at(first);
      set_item(cur);
      dma_halo(cur, 0, 0, 0, HP);
      dma_weights(0, 0, cur.ncol, 0);
      if (t == 0) sK[0] = chunk_exp(cur, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      adell_ws_barrier();                       // P1
      adell_ws_barrier();                       // P2
      int stage = 0, cidx = 0;
      for (int i = 0; i < count; ++i) {
        const bool has_next_item = i + 1 < count;
        ConvWsItem nxt = cur;
        if (has_next_item) nxt = item_at(first + i + 1);
        for (int ch = 0; ch < nchunk; ++ch, ++cidx) {
          const bool last_chunk = ch + 1 == nchunk;
          const bool stage_next = !last_chunk || has_next_item;
#pragma unroll
          for (int g = 0; g < NGRP; ++g, ++stage) {
            if (g == 0 && ch == 0 && i > 0) flush_stats(item_at(first + i - 1));
            // weights of stage s + 1 (buffer released by the barrier that ended stage s - 1)
            if (g + 1 < NGRP)
              dma_weights(ch, g + 1, cur.ncol, (stage + 1) & 1);
            else if (!last_chunk)
              dma_weights(ch + 1, 0, cur.ncol, (stage + 1) & 1);
            else if (has_next_item)
              dma_weights(0, 0, nxt.ncol, (stage + 1) & 1);
            // a share of the next chunk's halo (its buffer is idle for the whole of this chunk)
            if (stage_next) {
              if (g == 0 && last_chunk) set_item(nxt);
              const int j0 = (HP * g) / NGRP, j1 = (HP * (g + 1)) / NGRP;
              dma_halo(last_chunk ? nxt : cur, last_chunk ? 0 : ch + 1, (cidx + 1) & 1, j0, j1);
              if (g == 0 && t == 0)
                sK[(cidx + 1) & 1] = chunk_exp(last_chunk ? nxt : cur, last_chunk ? 0 : ch + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            adell_ws_barrier();
          }
        }
        cur = nxt;
      }
      flush_stats(item_at(first + count - 1));
      return;
    }
    float keep[KEEP][CC];
    int gvk[KEEP];

    auto set_item = [&](const ConvWsItem& it) {   // halo voxel -> input voxel of item `it`
      const int lx0 = it.ox0 - a.PW, ly0 = it.oy0 - a.PH, lz0 = it.oz0 - a.PD;
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        const int hv = t + NLD * u;
        int g = -1;
        if (hv < HV) {
          const int hz = hv / (HX * HY), rem = hv - hz * (HX * HY);
          const int hy = rem / HX, hx = rem - hy * HX;
          const int rx = lx0 + hx, ry = ly0 + hy, rz = lz0 + hz;
          if ((rx >= 0) & (ry >= 0) & (rz >= 0) & (rx < a.W) & (ry < a.H) & (rz < a.D))
            g = (rz * a.H + ry) * a.W + rx;
        }
        gvk[u] = g;
      }
    };
    // phase A (first stage of a chunk): the 16 channels [16 ch, 16 ch + 16) of every halo voxel of
    // the NEXT chunk -> registers (loads only: nothing waits for them in this stage)
    auto issue_halo = [&](const ConvWsItem& it, int ch) {
      const int c0 = ch * CC;
      const size_t vox0 = (size_t)it.nb * a.D * a.H * a.W;
      const bool firstsrc = c0 < a.C0;
      const ADELL_GLOBAL char* src = adell_uniform_ptr(firstsrc ? a.x0 + vox0 * a.C0 + c0
                                                   : a.x1 + vox0 * a.C1 + (c0 - a.C0));
      const unsigned cs = firstsrc ? a.C0 : a.C1;
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        // exactly 4 loads per voxel whatever the lane's voxel is (the counted vmcnt below relies
        // on it): voxels outside the tensor read voxel 0 and are zeroed
        const bool ok = gvk[u] >= 0;
        const ADELL_GLOBAL char* p = src + (unsigned)(ok ? gvk[u] : 0) * cs * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 f = adell_gload4(p + 16 * q);
          keep[u][4 * q + 0] = ok ? f.x : 0.f;
          keep[u][4 * q + 1] = ok ? f.y : 0.f;
          keep[u][4 * q + 2] = ok ? f.z : 0.f;
          keep[u][4 * q + 3] = ok ? f.w : 0.f;
        }
      }
    };
    // phase A' (second stage): absmax of the loaded values, one float per loader wave
    auto halo_absmax = [&]() {
      float mx = 0.f;
#pragma unroll
      for (int u = 0; u < KEEP; ++u)
#pragma unroll
        for (int j = 0; j < CC; ++j) mx = fmaxf(mx, fabsf(keep[u][j]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if (lane == 0) sMaxL[wave - 4] = mx;
    };
    // phase B: scale from the four loader waves' absmax, split, store into halo buffer `buf`
    auto store_halo = [&](int buf, bool first_col) {
      const float mx = fmaxf(fmaxf(sMaxL[0], sMaxL[1]), fmaxf(sMaxL[2], sMaxL[3]));
      if (e.amax_out != nullptr && t == 0 && first_col) atomicMax(e.amax_out, __float_as_uint(mx));
      int kA = 0;
      const int ebits = (__float_as_int(mx) >> 23) & 0xff;
      // max lands in [2^6, 2^14): multiples of 8 so the scale rarely changes per chunk
      if (ebits > 0 && ebits < 255) kA = 8 * ((13 - (ebits - 127)) >> 3);
      if (kA > 96) kA = 96;
      if (kA < -96) kA = -96;
      if (t == 0) sK[buf] = kA;
      const float scaleA = __int_as_float((kA + 127) << 23);
      char* base = sA + (size_t)buf * A_BYTES;
#pragma unroll
      for (int u = 0; u < KEEP; ++u) {
        const int hv = t + NLD * u;
        if (hv < HV) {
          half8 h0, l0, h1, l1;
          adell_split8(keep[u], scaleA, &h0, &l0);
          adell_split8(keep[u] + 8, scaleA, &h1, &l1);
          const int sw = ((hv % 10) >> 1) & 3;   // slot permutation by the voxel's x pair (conv_igemm_f16.h)
          char* row = base + (size_t)hv * 64;
          *reinterpret_cast<half8*>(row + ((0 ^ sw) << 4)) = h0;
          *reinterpret_cast<half8*>(row + ((1 ^ sw) << 4)) = h1;
          *reinterpret_cast<half8*>(row + ((2 ^ sw) << 4)) = l0;
          *reinterpret_cast<half8*>(row + ((3 ^ sw) << 4)) = l1;
        }
      }
    };
    // weight slice [tpg][BN][4 slots] of (chunk ch, tap group grp, column tile ncol) -> sB[buf] by
    // LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write): the packed weights already hold
    // the hi | lo halves the LDS image wants. One wave-instruction fills 1 KB = 16 rows of 64 B in
    // lane order, so the per-row slot permutation is applied to each lane's SOURCE address.
    // Columns past Cout (ragged last tile) read a clamped column: their outputs are never stored.
    constexpr int PIECES = (GT * BN / 16 + 3) / 4;      // 1 KB pieces per loader wave
    auto dma_weights = [&](int ch, int grp, int ncol, int buf) {
      const int n0 = ncol * BN;
      const int tpg = (27 - grp * GT) < GT ? (27 - grp * GT) : GT;
      const int rows = tpg * BN;
      const char* wbase = reinterpret_cast<const char*>(e.wh) +
                          ((size_t)(grp * GT) * a.Cout * nchunk + ch) * 64;
#pragma unroll
      for (int k = 0; k < PIECES; ++k) {
        const int piece = k * 4 + (wave - 4);
        if (piece * 16 < rows) {   // wave-uniform
          const int row = piece * 16 + (lane >> 2);
          const int tl = row / BN, n = row % BN;
          int col = n0 + n;
          if (col >= a.Cout) col = a.Cout - 1;
          const int lslot = (lane & 3) ^ ((n >> 2) & 3);
          const char* src = wbase + ((size_t)tl * a.Cout + col) * nchunk * 64 + lslot * 16;
          const unsigned off = __builtin_amdgcn_readfirstlane(
              (unsigned)(2 * A_BYTES + buf * B_BYTES + piece * 1024));
          __builtin_amdgcn_global_load_lds((glb_char*)src, (lds_char*)smem + off, 16, 0, 0);
        }
      }
    };
    // statistics of a finished brick: fold the four compute waves' sums (written to sRed before the
    // barrier that ended the brick's last stage) into the per-brick partial
    auto flush_stats = [&](const ConvWsItem& it) {
      if (a.part == nullptr || t >= BN) return;
      const int n = it.ncol * BN + t;
      if (n >= a.Cout) return;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += sRed[(w * BN + t) * 2 + 0];
        t2 += sRed[(w * BN + t) * 2 + 1];
      }
      float* p = a.part + (((size_t)it.nb * nsp + it.tile) * a.Cout + n) * 2;
      p[0] = t1;
      p[1] = t2;
    };

    ConvWsItem cur = item_at(first);
    // prologue: chunk 0 of the first brick and the weights of stage 0
    set_item(cur);
    issue_halo(cur, 0);
    halo_absmax();
    dma_weights(0, 0, cur.ncol, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    adell_ws_barrier();                       // P1
    store_halo(0, cur.ncol == 0);
    adell_ws_barrier();                       // P2
    int stage = 0, cidx = 0;
#ifdef ADELL_WS_CENSUS
    unsigned long long lw = 0, lb = 0, lg[4] = {0, 0, 0, 0};
    const unsigned long long lstart = __builtin_amdgcn_s_memtime();
#endif
    for (int i = 0; i < count; ++i) {
      const bool has_next_item = i + 1 < count;
      ConvWsItem nxt = cur;
      if (has_next_item) nxt = item_at(first + i + 1);
      for (int ch = 0; ch < nchunk; ++ch, ++cidx) {
        const bool last_chunk = ch + 1 == nchunk;
        const bool stage_next = !last_chunk || has_next_item;   // another chunk follows
#pragma unroll
        for (int g = 0; g < NGRP; ++g, ++stage) {
          const bool skipW = ADELL_DBG(e.dbg) & 2, skipA = ADELL_DBG(e.dbg) & 1;
          WS_STAMP(ls0);
          // weights of stage s+1 -> sB[(s+1)&1] (released by the barrier that ended stage s-1)
          auto next_weights = [&]() {
            if (skipW) return;
            if (g + 1 < NGRP)
              dma_weights(ch, g + 1, cur.ncol, (stage + 1) & 1);
            else if (!last_chunk)
              dma_weights(ch + 1, 0, cur.ncol, (stage + 1) & 1);
            else if (has_next_item)
              dma_weights(0, 0, nxt.ncol, (stage + 1) & 1);
          };
          const bool halo = stage_next && !skipA;
          if (g == 0) {
            // DMAs first, halo loads after them: the counted wait below retires the DMAs and leaves
            // the halo loads (consumed in the next stage) in flight
            if (ch == 0 && i > 0) flush_stats(item_at(first + i - 1));
            next_weights();
            __builtin_amdgcn_sched_barrier(0);
            if (halo) {
              if (last_chunk) set_item(nxt);
              issue_halo(last_chunk ? nxt : cur, last_chunk ? 0 : ch + 1);
              __builtin_amdgcn_sched_barrier(0);
              if (!(ADELL_DBG(e.dbg) & 32)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP * 4) : "memory");
            } else {
              if (!(ADELL_DBG(e.dbg) & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
          } else if (g == 1) {
            next_weights();
            __builtin_amdgcn_sched_barrier(0);
            if (halo) halo_absmax();          // first use of the halo registers
            if (!(ADELL_DBG(e.dbg) & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            next_weights();
            if (g == 2 && halo) store_halo((cidx + 1) & 1, (last_chunk ? nxt.ncol : cur.ncol) == 0);
            if (!(ADELL_DBG(e.dbg) & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          WS_STAMP(ls1);
          adell_ws_barrier();
#ifdef ADELL_WS_CENSUS
          lb += __builtin_amdgcn_s_memtime() - ls1;
          lg[g] += ls1 - ls0;
#endif
        }
      }
      cur = nxt;
    }
    flush_stats(item_at(first + count - 1));
#ifdef ADELL_WS_CENSUS
    lw = __builtin_amdgcn_s_memtime() - lstart - lb;
    if (tid == 256) {
      for (int q = 0; q < 4; ++q) atomicAdd(&g_ws_prof[8 + q], lg[q]);
      atomicAdd(&g_ws_prof[3], lw);
      atomicAdd(&g_ws_prof[4], lb);
    }
#endif
    return;
  }

  // ================================= compute waves ===============================================
  // each SIMD hosts one compute and one loader wave: the compute wave wins every issue
  // arbitration (the loaders have slack: alone they need ~60 % of the compute waves' time)
  __builtin_amdgcn_s_setprio(3);
  const int wm = wave;   // WM = 4, WN = 1
  int arow[MT];          // halo voxel index of this lane's A row at tap (0,0,0)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + li;
    arow[mt] = (((m >> 6) * HY) + ((m >> 3) & 7)) * HX + (m & 7);
  }
  int boffh[NT], boffl[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 32 + li;
    const int sw = (n >> 2) & 3;
    boffh[nt] = n * 64 + ((lh ^ sw) << 4);
    boffl[nt] = n * 64 + (((2 + lh) ^ sw) << 4);
  }
  f32x16 acc[MT][NT];
  adell_ws_barrier();   // P1
  adell_ws_barrier();   // P2
  int stage = 0, cidx = 0;
#ifdef ADELL_WS_CENSUS
  unsigned long long cw = 0, cb = 0, ce = 0;
  const unsigned long long cstart = __builtin_amdgcn_s_memtime();
#endif
  for (int i = 0; i < count; ++i) {
    const ConvWsItem it = item_at(first + i);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int nj = 0; nj < NT; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;
    int kA_prev = 0;
    for (int ch = 0; ch < nchunk; ++ch, ++cidx) {
      const char* sAc = sA + (size_t)(cidx & 1) * A_BYTES;
      const int kA = sK[cidx & 1];
      if (ch > 0 && kA != kA_prev) {
        const float f = __int_as_float((kA - kA_prev + 127) << 23);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int nj = 0; nj < NT; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nj][r] *= f;
      }
      kA_prev = kA;
#pragma unroll
      for (int g = 0; g < NGRP; ++g, ++stage) {
        const int tpg = (27 - g * GT) < GT ? (27 - g * GT) : GT;
        const char* sBc = sB + (size_t)(stage & 1) * B_BYTES;
        auto load_frags = [&](int tl, half8* ah, half8* al, half8* bh, half8* bl) {
          const int tabs = g * GT + tl;                      // compile-time (g, tl unrolled)
          const int kz = tabs / 9, ky = (tabs - 9 * kz) / 3, kx = tabs - 9 * kz - 3 * ky;
          const int aoff = (kz * HY + ky) * HX + kx;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            // x of the halo voxel = (li & 7) + kx: slot permutation by the x pair
            const int sw = (((li & 7) + kx) >> 1) & 3;
            const char* row = sAc + (size_t)(arow[mt] + aoff) * 64;
            ah[mt] = *reinterpret_cast<const half8*>(row + ((lh ^ sw) << 4));
            al[mt] = *reinterpret_cast<const half8*>(row + (((2 + lh) ^ sw) << 4));
          }
          const char* bt = sBc + (size_t)tl * BN * 64;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = *reinterpret_cast<const half8*>(bt + boffh[nt]);
            bl[nt] = *reinterpret_cast<const half8*>(bt + boffl[nt]);
          }
        };
        auto do_mfma = [&](const half8* ah, const half8* al, const half8* bh, const half8* bl) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        };
        half8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
        if (!(ADELL_DBG(e.dbg) & 8)) {
        load_frags(0, ah[0], al[0], bh[0], bl[0]);
#pragma unroll
        for (int tl = 0; tl < GT; ++tl) {
          if (tl < tpg) {
            // all LDS reads of tap t+1 are issued BEFORE the MFMAs of tap t (left alone, hipcc reads
            // each fragment right before its MFMA and waits lgkmcnt(0): with one compute wave per
            // SIMD nothing else hides that latency)
            if (tl + 1 < tpg)
              load_frags(tl + 1, ah[(tl + 1) & 1], al[(tl + 1) & 1], bh[(tl + 1) & 1], bl[(tl + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            do_mfma(ah[tl & 1], al[tl & 1], bh[tl & 1], bl[tl & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        }
        if (g + 1 < NGRP || ch + 1 < nchunk) {
          WS_STAMP(cs1);
          adell_ws_barrier();
#ifdef ADELL_WS_CENSUS
          cb += __builtin_amdgcn_s_memtime() - cs1;
#endif
        }
      }
    }
    // ---- epilogue of the brick (before the barrier that closes its last stage) -------------------
    WS_STAMP(ce0);
    if (!(ADELL_DBG(e.dbg) & 16)) {
      const int n0 = it.ncol * BN;
      const float ascale = __int_as_float((127 - kA_prev) << 23);
      const size_t row0 = ((size_t)(it.nb * a.Do + it.oz0) * a.Ho + it.oy0) * a.Wo + it.ox0;
      float s1[NT], s2[NT], bcol[NT], oscale[NT];
      float* colptr[NT];
      const float* resptr[NT];
      int rowmul[NT];
      bool nok[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = s2[nt] = 0.f;
        const int n = n0 + nt * 32 + li;
        nok[nt] = n < a.Cout;
        bcol[nt] = 0.f;
        oscale[nt] = 0.f;
        colptr[nt] = a.y0;
        resptr[nt] = a.res;
        rowmul[nt] = 0;
        if (nok[nt]) {
          oscale[nt] = ascale * e.wscale[n];
          if (n < a.ysplit) {
            colptr[nt] = a.y0 + n;
            rowmul[nt] = a.ysplit;
          } else {
            colptr[nt] = a.y1 + (n - a.ysplit);
            rowmul[nt] = a.Cout - a.ysplit;
          }
          if (a.bias) bcol[nt] = a.bias[n];
          colptr[nt] += row0 * rowmul[nt];
          if (a.res) resptr[nt] = a.res + row0 * a.Cout + n;
        }
      }
      const bool full = (it.ox0 + 8 <= a.Wo) & (it.oy0 + 8 <= a.Ho) & (it.oz0 + BZ <= a.Do) &
                        (n0 + BN <= a.Cout);
      if (full) {
        // interior brick, all columns valid: rows of an m-tile are 4 x-neighbours (r & 3) in 4
        // y-rows (r >> 2): the m-tile's base + compile-time multiples of two per-lane strides
        auto fast = [&](auto has_res) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int tile = wm * MT + mt;
            const unsigned rbase = (unsigned)(tile >> 1) * (unsigned)(a.Ho * a.Wo) +
                                   (unsigned)((tile & 1) * 4 * a.Wo) + 4 * lh;
            float resv[NT][16];
            if constexpr (has_res.value) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const float* rp = resptr[nt] + (size_t)rbase * a.Cout;
                const unsigned dY = a.Wo * a.Cout, dX = a.Cout;
#pragma unroll
                for (int r = 0; r < 16; ++r) resv[nt][r] = rp[(r >> 2) * dY + (r & 3) * dX];
              }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float* p = colptr[nt] + (size_t)rbase * rowmul[nt];
              const unsigned dY = a.Wo * rowmul[nt], dX = rowmul[nt];
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] * oscale[nt] + bcol[nt];
                if constexpr (has_res.value) v += resv[nt][r];
                p[(r >> 2) * dY + (r & 3) * dX] = v;
                s1[nt] += v;
                s2[nt] += v * v;
              }
            }
          }
        };
        if (a.res)
          fast(std::true_type{});
        else
          fast(std::false_type{});
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int m = (wm * MT + mt) * 32 + row;
            const int xl = m & 7, yl = (m >> 3) & 7, zl = m >> 6;
            const bool rok = (it.ox0 + xl < a.Wo) & (it.oy0 + yl < a.Ho) & (it.oz0 + zl < a.Do);
            const unsigned rloc = (unsigned)zl * (unsigned)(a.Ho * a.Wo) + __umul24(yl, a.Wo) + xl;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (rok && nok[nt]) {
                float v = acc[mt][nt][r] * oscale[nt] + bcol[nt];
                if (a.res) v += resptr[nt][rloc * (unsigned)a.Cout];
                colptr[nt][rloc * (unsigned)rowmul[nt]] = v;
                s1[nt] += v;
                s2[nt] += v * v;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (a.part) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float t1 = s1[nt] + __shfl_xor(s1[nt], 32, 64);
          const float t2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
          if (lh == 0) {
            const int col = nt * 32 + li;
            sRed[(wm * BN + col) * 2 + 0] = t1;
            sRed[(wm * BN + col) * 2 + 1] = t2;
          }
        }
      }
    }
    WS_STAMP(ce1);
    adell_ws_barrier();   // closes the brick's last stage (the loaders count it as a stage barrier)
#ifdef ADELL_WS_CENSUS
    ce += ce1 - ce0;
    cb += __builtin_amdgcn_s_memtime() - ce1;
#endif
  }
#ifdef ADELL_WS_CENSUS
  cw = __builtin_amdgcn_s_memtime() - cstart - cb - ce;   // everything that is not barrier / epilogue
  if (tid == 0) {
    atomicAdd(&g_ws_prof[0], cw);
    atomicAdd(&g_ws_prof[1], cb);
    atomicAdd(&g_ws_prof[2], ce);
    atomicAdd(&g_ws_prof[5], 1ull);
  }
#endif
}
