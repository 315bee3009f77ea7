// Backward-weight on the f16 MFMA with error-compensated operand splitting
// ("f16x3", see conv_igemm_f16.h):
//
//   dW[tap][ci][co] = sum_{n,v} X[n, S*v + tap - P][ci] * dY[n, v][co]
//
// GEMM view: M = ci, N = co, K = output voxels, v_mfma_f32_32x32x16_f16. The K index
// of both operands is the voxel, i.e. the STRIDED dimension of NDHWC data; the
// fragments are therefore fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16): LDS keeps the natural [voxel][32 channels] layout (64-byte
// rows, fp16 hi plane + lo plane, filled by straight coalesced copies + conversion),
// a tap shift is just a different row, and every 16-lane group pulls a 4-voxel x
// 16-channel block column-major into its lanes.
// Brick = 8 x TY x 1 output voxels, one kz plane of taps per block (blockIdx.z); a
// wave owns one 32x32 channel sub-tile and every (4/nsub)-th tap; the dY fragment is
// read once per k-step and shared by its taps. Ranges: X and dY carry one power-of-two
// scale per tensor (absmax kernels below), undone when the slab is written. Split-K
// over brick regions + fixed-order slab reduction as in conv_wgrad.hip.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// two transposed reads (voxels x'..x'+3 and x'+4..x'+7) -> one 8-half MFMA fragment
__device__ __forceinline__ half8 adell_tr_frag(const char* base, int off, int step) {
  typedef __attribute__((address_space(3))) fp16x4* lds_p;
  const fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(base + off));
  const fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(base + off + step));
  half8 r;
  r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
  r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
  return r;
}

struct WgradF16Args {
  const float* x0;
  const float* x1;
  const float* dy;
  float* ws;       // [R][ntap][Cin][Cout]
  float* wsdb;     // [R][Cout] or null
  const unsigned* xmax;  // device absmax (float bits) of X and dY
  const unsigned* ymax;
  int N, D, H, W;
  int C0, C1, Cin, Cout;
  int KD, KH, KW, SD, SH, SW, PD, PH, PW;
  int Do, Ho, Wo;
  int lTY;            // log2 of brick rows (TX = 8, TZ = 1)
  int ntx, nty;       // bricks per row / column (per z slice)
  int HX, HY;         // halo brick of X: (8-1)*SW + KW by (TY-1)*SH + GKH
  int GKH, NGY;       // ky rows per tap group, groups per kz plane
  int TCI, TCO, nci, nco;
  int R;
  int vecx, vecy;
  int ksplit;         // 1: a wave owns every tap of its sub-tile and a share of the k-steps
};

__device__ __forceinline__ int adell_scale_exp(unsigned maxbits) {
  const int ebits = (int)((maxbits >> 23) & 0xff);
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 8 * ((13 - (ebits - 127)) >> 3);
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

__device__ __forceinline__ void adell_split4_store(char* hi_plane, char* lo_plane, size_t off,
                                                   float a, float b, float c, float d,
                                                   float scale) {
  half4 h, l;
  const float t[4] = {a * scale, b * scale, c * scale, d * scale};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)t[j];
    l[j] = (_Float16)(t[j] - (float)h[j]);
  }
  *reinterpret_cast<half4*>(hi_plane + off) = h;
  *reinterpret_cast<half4*>(lo_plane + off) = l;
}

// PFX / PFY: 16-byte loads per thread of the register-prefetch pipeline (0: stage in place)
template <int MAXJ, int PFX, int PFY, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void adell_conv_wgrad_f16_kernel(WgradF16Args a) {
  extern __shared__ float smem[];
  const int TY = 1 << a.lTY;
  const int TV = 8 * TY;
  const int HV = a.HX * a.HY;
  const int sci = a.TCI >> 5, sco = a.TCO >> 5;
  const size_t xplane = (size_t)sci * HV * 64;
  const size_t yplane = (size_t)sco * TV * 64;
  char* sXh = reinterpret_cast<char*>(smem);
  char* sXl = sXh + xplane;
  char* sYh = sXl + xplane;
  char* sYl = sYh + yplane;
  // (the wave index is uniform: pinned to a scalar register so that the roles derived from it --
  // sub-tile, tap group, k-group -- cost no vector registers)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5;
  const int region = blockIdx.x;
  const int cit = blockIdx.y % a.nci, cot = blockIdx.y / a.nci;
  const int ci0 = cit * a.TCI, co0 = cot * a.TCO;
  const int kz = blockIdx.z / a.NGY;
  const int ky0 = (blockIdx.z % a.NGY) * a.GKH;
  const int tapsg = ((a.KH - ky0) < a.GKH ? (a.KH - ky0) : a.GKH) * a.KW;
  const int nsub = sci * sco;
  const int sub = wave % nsub;
  const int cis = sub % sci, cos = sub / sci;
  // Work split inside a block. Default: the 4 / nsub waves of a sub-tile share the taps. With
  // ksplit (fewer than 4 sub-tiles, e.g. 32 x 32 channels) they share the k-steps instead:
  // every wave then amortises its dY fragment over all taps of the group and no wave idles on
  // a ragged tap count; the waves' accumulators are folded through LDS after the region loop.
  // (a.ksplit = number of k-groups: 1, 2 or 4; the remaining factor of the 4 / nsub waves of a
  // sub-tile splits the taps.)
  const int kgroups = (PFX > 0 && a.ksplit > 1) ? a.ksplit : 1;   // only the pipelined variant
  const int wi = wave / nsub;                  // wave index inside its sub-tile
  const int tgroups = (4 / nsub) / kgroups;
  const int kw = wi % kgroups;
  const int tstride = tgroups, tfirst = wi / kgroups;

  const int kX = adell_scale_exp(a.xmax[0]), kY = adell_scale_exp(a.ymax[0]);
  const float sX = __int_as_float((kX + 127) << 23), sY = __int_as_float((kY + 127) << 23);

  // transposed-read lane roles: 16-lane group g = (channel group cg, voxel half lh);
  // lane 4q+p of a group addresses row q (voxel x' = q), columns 4p..4p+3
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const int colb = (16 * cg + 4 * tp) * 2;
  // byte offset of this lane's X block at k-step 0, tap (0,0): brick row lh, x' = tq
  const int abase = (cis * HV + lh * a.SH * a.HX + tq * a.SW) * 64 + colb;
  const int astep = 4 * a.SW * 64;          // x' + 4
  const int akstep = 2 * a.SH * a.HX * 64;  // two brick rows
  const int bbase = (cos * TV + lh * 8 + tq) * 64 + colb;
  int aoffj[MAXJ];
  bool jok[MAXJ];
#pragma unroll
  for (int q = 0; q < MAXJ; ++q) {
    int tl = tfirst + tstride * q;
    jok[q] = tl < tapsg;
    if (!jok[q]) tl = 0;
    const int kx = tl % a.KW, ky = tl / a.KW;
    aoffj[q] = abase + (ky * a.HX + kx) * 64;
  }

  f32x16 acc[MAXJ];
#pragma unroll
  for (int q = 0; q < MAXJ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const long bricks_per_item = (long)a.ntx * a.nty * a.Do;
  const long nbricks = bricks_per_item * a.N;
  const int c4x = a.TCI >> 2, c4y = a.TCO >> 2;
  const bool do_db = a.wsdb != nullptr && cit == 0 && blockIdx.z == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  if constexpr (PFX > 0) {
  // Software pipeline over this region's bricks: the global loads of brick i+1 are issued
  // into registers right after brick i has been written to LDS, so they are in flight while
  // the MFMAs of brick i run (PFX + PFY 16-byte loads per thread; the plan guarantees the fit).
  float4 xr[PFX > 0 ? PFX : 1], yr[PFY > 0 ? PFY : 1];
  auto fetch = [&](long brick) {
    long t = brick;
    const int tx = (int)(t % a.ntx); t /= a.ntx;
    const int ty = (int)(t % a.nty); t /= a.nty;
    const int oz = (int)(t % a.Do);
    const int nb = (int)(t / a.Do);
    const int ox0 = tx * 8, oy0 = ty << a.lTY;
    const int iz = oz * a.SD - a.PD + kz;
    const bool zok = iz >= 0 && iz < a.D;
    const int ix0 = ox0 * a.SW - a.PW, iy0 = oy0 * a.SH - a.PH + ky0;
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      int it = tid + u * 256;
      if (PFX >= 7) asm volatile("" : "+v"(it));  // opaque: keeps the index math out of the
                                                    // loop-invariant set (register pressure)
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < HV * c4x) {
        const int c4 = it % c4x, hv = it / c4x;
        const int hy = hv / a.HX, hx = hv - hy * a.HX;
        const int ix = ix0 + hx, iy = iy0 + hy;
        const int c = ci0 + 4 * c4;
        if (zok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          const size_t gv = ((size_t)(nb * a.D + iz) * a.H + iy) * a.W + ix;
          if (a.vecx) {
            if (c < a.C0)
              f = *reinterpret_cast<const float4*>(a.x0 + gv * a.C0 + c);
            else if (c < a.Cin)
              f = *reinterpret_cast<const float4*>(a.x1 + gv * a.C1 + (c - a.C0));
          } else {
            float uu[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int cc = c + j;
              uu[j] = cc < a.C0 ? a.x0[gv * a.C0 + cc]
                                : (cc < a.Cin ? a.x1[gv * a.C1 + (cc - a.C0)] : 0.f);
            }
            f = make_float4(uu[0], uu[1], uu[2], uu[3]);
          }
        }
      }
      xr[u] = f;
    }
#pragma unroll
    for (int u = 0; u < PFY; ++u) {
      int it = tid + u * 256;
      if (PFX >= 7) asm volatile("" : "+v"(it));  // opaque: keeps the index math out of the
                                                    // loop-invariant set (register pressure)
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < TV * c4y) {
        const int c4 = it % c4y, v = it / c4y;
        const int ox = ox0 + (v & 7), oy = oy0 + (v >> 3);
        const int c = co0 + 4 * c4;
        if (oy < a.Ho && ox < a.Wo) {
          const size_t gv = ((size_t)(nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
          const float* ptr = a.dy + gv * a.Cout + c;
          if (a.vecy) {
            if (c < a.Cout) f = *reinterpret_cast<const float4*>(ptr);
          } else {
            if (c + 0 < a.Cout) f.x = ptr[0];
            if (c + 1 < a.Cout) f.y = ptr[1];
            if (c + 2 < a.Cout) f.z = ptr[2];
            if (c + 3 < a.Cout) f.w = ptr[3];
          }
        }
      }
      yr[u] = f;
    }
  };
  if (region < nbricks) fetch(region);
  for (long brick = region; brick < nbricks; brick += a.R) {
    __syncthreads();
    // ---- registers -> LDS: [32-channel group][voxel][32 halfs], hi and lo planes ----
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      int it = tid + u * 256;
      if (PFX >= 7) asm volatile("" : "+v"(it));  // opaque: keeps the index math out of the
                                                    // loop-invariant set (register pressure)
      if (it < HV * c4x) {
        const int c4 = it % c4x, hv = it / c4x;
        const size_t off = ((size_t)((4 * c4) >> 5) * HV + hv) * 64 + ((4 * c4) & 31) * 2;
        adell_split4_store(sXh, sXl, off, xr[u].x, xr[u].y, xr[u].z, xr[u].w, sX);
      }
    }
#pragma unroll
    for (int u = 0; u < PFY; ++u) {
      int it = tid + u * 256;
      if (PFX >= 7) asm volatile("" : "+v"(it));  // opaque: keeps the index math out of the
                                                    // loop-invariant set (register pressure)
      if (it < TV * c4y) {
        const int c4 = it % c4y, v = it / c4y;
        dbacc.x += yr[u].x; dbacc.y += yr[u].y; dbacc.z += yr[u].z; dbacc.w += yr[u].w;
        const size_t off = ((size_t)((4 * c4) >> 5) * TV + v) * 64 + ((4 * c4) & 31) * 2;
        adell_split4_store(sYh, sYl, off, yr[u].x, yr[u].y, yr[u].z, yr[u].w, sY);
      }
    }
    __syncthreads();
    if (brick + a.R < nbricks) fetch(brick + a.R);
    // ---- k-steps of 16 voxels (two brick rows), fragments by transposed reads ----
    for (int s = kw; s < (TY >> 1); s += kgroups) {
      const half8 bh = adell_tr_frag(sYh, bbase + s * 16 * 64, 4 * 64);
      const half8 bl = adell_tr_frag(sYl, bbase + s * 16 * 64, 4 * 64);
#pragma unroll
      for (int q = 0; q < MAXJ; ++q) {
        const half8 ah = adell_tr_frag(sXh, aoffj[q] + s * akstep, astep);
        const half8 al = adell_tr_frag(sXl, aoffj[q] + s * akstep, astep);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q], 0, 0, 0);
      }
    }
  }

  } else {
  for (long brick = region; brick < nbricks; brick += a.R) {
    long t = brick;
    const int tx = (int)(t % a.ntx); t /= a.ntx;
    const int ty = (int)(t % a.nty); t /= a.nty;
    const int oz = (int)(t % a.Do);
    const int nb = (int)(t / a.Do);
    const int ox0 = tx * 8, oy0 = ty << a.lTY;
    const int iz = oz * a.SD - a.PD + kz;
    const bool zok = iz >= 0 && iz < a.D;
    const int ix0 = ox0 * a.SW - a.PW, iy0 = oy0 * a.SH - a.PH + ky0;
    __syncthreads();
    // ---- stage the X halo brick: [32-channel group][halo voxel][32 halfs] ------
    for (int it = tid; it < HV * c4x; it += 256) {
      const int c4 = it % c4x, hv = it / c4x;
      const int hy = hv / a.HX, hx = hv - hy * a.HX;
      const int ix = ix0 + hx, iy = iy0 + hy;
      const int c = ci0 + 4 * c4;
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (zok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        const size_t gv = ((size_t)(nb * a.D + iz) * a.H + iy) * a.W + ix;
        if (a.vecx) {
          if (c < a.C0)
            f = *reinterpret_cast<const float4*>(a.x0 + gv * a.C0 + c);
          else if (c < a.Cin)
            f = *reinterpret_cast<const float4*>(a.x1 + gv * a.C1 + (c - a.C0));
        } else {
          float u[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cc = c + j;
            u[j] = cc < a.C0 ? a.x0[gv * a.C0 + cc]
                             : (cc < a.Cin ? a.x1[gv * a.C1 + (cc - a.C0)] : 0.f);
          }
          f = make_float4(u[0], u[1], u[2], u[3]);
        }
      }
      const size_t off = ((size_t)((4 * c4) >> 5) * HV + hv) * 64 + ((4 * c4) & 31) * 2;
      adell_split4_store(sXh, sXl, off, f.x, f.y, f.z, f.w, sX);
    }
    // ---- stage the dY brick: [32-channel group][brick voxel][32 halfs] ---------
    for (int it = tid; it < TV * c4y; it += 256) {
      const int c4 = it % c4y, v = it / c4y;
      const int ox = ox0 + (v & 7), oy = oy0 + (v >> 3);
      const int c = co0 + 4 * c4;
      float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < a.Ho && ox < a.Wo) {
        const size_t gv = ((size_t)(nb * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
        const float* ptr = a.dy + gv * a.Cout + c;
        if (a.vecy) {
          if (c < a.Cout) f = *reinterpret_cast<const float4*>(ptr);
        } else {
          if (c + 0 < a.Cout) f.x = ptr[0];
          if (c + 1 < a.Cout) f.y = ptr[1];
          if (c + 2 < a.Cout) f.z = ptr[2];
          if (c + 3 < a.Cout) f.w = ptr[3];
        }
      }
      dbacc.x += f.x; dbacc.y += f.y; dbacc.z += f.z; dbacc.w += f.w;
      const size_t off = ((size_t)((4 * c4) >> 5) * TV + v) * 64 + ((4 * c4) & 31) * 2;
      adell_split4_store(sYh, sYl, off, f.x, f.y, f.z, f.w, sY);
    }
    __syncthreads();
    // ---- k-steps of 16 voxels (two brick rows), fragments by transposed reads ----
    for (int s = kw; s < (TY >> 1); s += kgroups) {
      const half8 bh = adell_tr_frag(sYh, bbase + s * 16 * 64, 4 * 64);
      const half8 bl = adell_tr_frag(sYl, bbase + s * 16 * 64, 4 * 64);
#pragma unroll
      for (int q = 0; q < MAXJ; ++q) {
        const half8 ah = adell_tr_frag(sXh, aoffj[q] + s * akstep, astep);
        const half8 al = adell_tr_frag(sXl, aoffj[q] + s * akstep, astep);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q], 0, 0, 0);
      }
    }
  }

  }

  if (do_db) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < c4y) {
      float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = tid; k < 256; k += c4y) {
        const float4 u = red[k];
        tsum.x += u.x; tsum.y += u.y; tsum.z += u.z; tsum.w += u.w;
      }
      const int c = co0 + 4 * tid;
      float* o = a.wsdb + (size_t)region * a.Cout + c;
      if (c + 0 < a.Cout) o[0] = tsum.x;
      if (c + 1 < a.Cout) o[1] = tsum.y;
      if (c + 2 < a.Cout) o[2] = tsum.z;
      if (c + 3 < a.Cout) o[3] = tsum.w;
    }
  }
  if (PFX > 0 && a.ksplit > 1) {
    // fold the k-groups of each (sub-tile, tap group) in fixed order: red[group][job][r][lane]
    __syncthreads();
    float* red = smem;
    for (int g = 0; g < kgroups; ++g) {
      if (kw == g) {
#pragma unroll
        for (int q = 0; q < MAXJ; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* slot = red + ((size_t)((sub * tgroups + tfirst) * MAXJ + q) * 16 + r) * 64 + lane;
            *slot = (g == 0 ? 0.f : *slot) + acc[q][r];
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
      if (kw != 0) jok[q] = false;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[q][r] = red[((size_t)((sub * tgroups + tfirst) * MAXJ + q) * 16 + r) * 64 + lane];
    }
  }
  // ---- partial slab (undo the operand scales) --------------------------------
  const float unscale = __int_as_float((127 - kX - kY) << 23);
  const int ntap = a.KD * a.KH * a.KW;
  // (the lane / wave roles are derived again from an opaque copy of the thread index: kept live
  // across the region loop of the nine-accumulator instance they were spilled to scratch memory)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int li_e = tid_e & 31, lh_e = (tid_e >> 5) & 1, sub_e = (tid_e >> 6) % nsub;
  const int cis_e = sub_e % sci, cos_e = sub_e / sci;
  const int co = co0 + cos_e * 32 + li_e;
#pragma unroll
  for (int q = 0; q < MAXJ; ++q) {
    const int tl = tfirst + tstride * q;
    const int tap = (kz * a.KH + ky0) * a.KW + tl;
    const int cib = ci0 + cis_e * 32 + 4 * lh_e;
    float* base = a.ws + (((size_t)region * ntap + tap) * a.Cin + cib) * a.Cout + co;
    if (jok[q] && co < a.Cout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        if (cib + row < a.Cin) base[(size_t)row * a.Cout] = acc[q][r] * unscale;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void adell_absmax2_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  float mx = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    mx = fmaxf(mx, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(mx));
}

// implemented in conv_wgrad.hip
extern "C" int adell_wgrad_reduce_launch(const float* ws, float* out, int R, int ntap, int Cin,
                                         int Cout, const float* wsdb, float* db, void* stream);

// z-ring kernel for 3^3 stride-1 convolutions (conv_wgrad_zring.hip)
struct WgradZrPlan {
  int ntx, nty, nseg, seglen, nci, nco, R, t16;
};
extern "C" int adell_wgrad_zring_plan(int N, int D, int H, int W, int C0, int C1, int Cout, int KD,
                                      int KH, int KW, int SD, int SH, int SW, int Do, int Ho, int Wo,
                                      int* plan);   // plan: the eight ints of a WgradZrPlan
extern "C" size_t adell_wgrad_zring_ws_floats(const WgradZrPlan* p, int Cin, int Cout);
// the 32 -> 32 stride-2 downsampling layer (conv_wgrad_s2.hip)
struct WgradS2Plan {
  int ntx, nty, ntz, nbricks, blocks, R;
};
extern "C" int adell_wgrad_s2_plan(int N, int D, int H, int W, int C0, int C1, int Cout, int KD,
                                   int KH, int KW, int SD, int SH, int SW, int PD, int PH, int PW,
                                   int Do, int Ho, int Wo, WgradS2Plan* p);
extern "C" int adell_wgrad_s2_launch(const WgradS2Plan* p, int N, int D, int H, int W, const float* x,
                                     int Do, int Ho, int Wo, const float* dy, float* slabs,
                                     float* wsdb, const unsigned* xmax, const unsigned* ymax,
                                     hipStream_t st);
extern "C" int adell_wgrad_zring_launch(const WgradZrPlan* p, int N, int D, int H, int W, int C0,
                                        int C1, const float* x0, const float* x1, int Cout, int Do,
                                        int Ho, int Wo, const float* dy, int PD, int PH, int PW,
                                        float* slabs, float* wsdb, const unsigned* xmax,
                                        const unsigned* ymax, hipStream_t st,
                                        const int* xk0 = nullptr, const int* xk1 = nullptr);

struct WgradF16Plan {
  int lTY, HX, HY, TCI, TCO, nci, nco, maxj, R, ntx, nty, GKH, NGY, ksplit;
  size_t lds;
};

static int adell_wgrad_f16_plan(int N, int Cin, int Cout, int KD, int KH, int KW, int SH, int SW,
                                int Do, int Ho, int Wo, WgradF16Plan* p) {
  p->TCI = Cin > 32 ? 64 : 32;
  p->TCO = Cout > 32 ? 64 : 32;
  p->nci = adell_cdiv(Cin, p->TCI);
  p->nco = adell_cdiv(Cout, p->TCO);
  const int nsub = (p->TCI / 32) * (p->TCO / 32);
  // a block owns the taps of GKH consecutive ky rows of one kz plane (<= 9 per wave)
  // one 32 x 32 channel tile: two k-groups x two tap groups (TY = 8 gives four k-steps)
  p->ksplit = (nsub == 1 && Ho > 4 && KW <= 3 && KH <= 3 && SH == 1 && SW == 1) ? 2 : 1;
  const int tgroups = (4 / nsub) / p->ksplit;
  p->GKH = KH;
  while (p->GKH > 1 && adell_cdiv(p->GKH * KW, tgroups) > 9) --p->GKH;
  p->maxj = adell_cdiv(p->GKH * KW, tgroups);
  p->NGY = adell_cdiv(KH, p->GKH);
  if (p->maxj > 9) {
    adell_set_error("wgrad f16x3: %d jobs per wave unsupported", p->maxj);
    return ADELL_E_UNSUPPORTED;
  }
  p->HX = 7 * SW + KW;
  p->lTY = Ho > 4 ? 3 : (Ho > 2 ? 2 : 1);  // TY = 8, 4 or 2
  for (;;) {
    const int TY = 1 << p->lTY;
    p->HY = (TY - 1) * SH + p->GKH;
    p->lds = 2 * ((size_t)(p->TCI / 32) * p->HX * p->HY * 64 + (size_t)(p->TCO / 32) * 8 * TY * 64);
    if (p->lds < 4096) p->lds = 4096;
    if (p->ksplit > 1) {  // room for the accumulator fold (MAXJ template sizes: 2, 3, 5, 7, 9)
      const int mj = p->maxj <= 2 ? 2 : (p->maxj <= 3 ? 3 : (p->maxj <= 5 ? 5 : (p->maxj <= 7 ? 7 : 9)));
      const size_t red = (size_t)nsub * tgroups * mj * 16 * 64 * sizeof(float);
      if (p->lds < red) p->lds = red;
    }
    if (p->lds <= 80 * 1024 || p->lTY == 1) break;
    --p->lTY;
  }
  if (p->lds > 160 * 1024) {
    adell_set_error("wgrad f16x3: cannot fit LDS (%zu B)", p->lds);
    return ADELL_E_UNSUPPORTED;
  }
  p->ntx = adell_cdiv(Wo, 8);
  p->nty = adell_cdiv(Ho, 1 << p->lTY);
  const long nbricks = (long)N * p->ntx * p->nty * Do;
  const long chan_blocks = (long)p->nci * p->nco * KD * p->NGY;
  int per_cu = (int)((160 * 1024) / p->lds);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  long R = (256L * per_cu) / chan_blocks;
  if (R > nbricks) R = nbricks;
  if (R < 1) R = 1;
  p->R = (int)R;
  return ADELL_OK;
}

static size_t adell_wgrad_f16_ws(const WgradF16Plan& p, int ntap, int Cin, int Cout) {
  // slabs + db slabs + two absmax words
  return ((size_t)p.R * ntap * Cin * Cout + (size_t)p.R * Cout + 4) * sizeof(float);
}

template <int MAXJ, int PFX = 0, int PFY = 0, int MINB = 2>
static int adell_launch_wgrad_f16(const WgradF16Args& a, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  auto kern = adell_conv_wgrad_f16_kernel<MAXJ, PFX, PFY, MINB>;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}

static int adell_wgrad_f16_core(int N, int D, int H, int W, int C0, int C1, const float* x0,
                                const float* x1, int Cout, int Do, int Ho, int Wo,
                                const float* dy, int KD, int KH, int KW, int SD, int SH, int SW,
                                int PD, int PH, int PW, float* out, float* db,
                                const uint32_t* xmax_in, const uint32_t* ymax_in, void* ws,
                                size_t ws_bytes, hipStream_t st, const int* xk0 = nullptr,
                                const int* xk1 = nullptr) {
  const int Cin = C0 + C1;
  WgradF16Plan p = {};
  WgradZrPlan zp;
  const bool zring = adell_wgrad_zring_plan(N, D, H, W, C0, C1, Cout, KD, KH, KW, SD, SH, SW, Do, Ho,
                                            Wo, reinterpret_cast<int*>(&zp)) != 0;
  // split-row sources: the z-ring kernel only, and no 32-channel tile across the two sources
  ADELL_REQUIRE((!xk0 && !xk1) || (zring && (C1 == 0 || C0 % 32 == 0)),
                "wgrad f16x3: this problem does not take split-row sources "
                "(adell_conv3d_bwd_weight_f16x3_rows_ok)");
  WgradS2Plan sp;
  const bool s2 = !zring && adell_wgrad_s2_plan(N, D, H, W, C0, C1, Cout, KD, KH, KW, SD, SH, SW, PD,
                                                PH, PW, Do, Ho, Wo, &sp) != 0;
  int rc = ADELL_OK;
  if (zring)
    p.R = zp.R;
  else if (s2)
    p.R = sp.R;
  else
    rc = adell_wgrad_f16_plan(N, Cin, Cout, KD, KH, KW, SH, SW, Do, Ho, Wo, &p);
  if (rc != ADELL_OK) return rc;
  const int ntap = KD * KH * KW;
  const size_t need = adell_wgrad_f16_ws(p, ntap, Cin, Cout);
  ADELL_REQUIRE(ws != nullptr && ws_bytes >= need, "wgrad f16x3: workspace too small (%zu < %zu)",
                ws_bytes, need);
  float* slabs = (float*)ws;
  float* wsdb = slabs + (size_t)p.R * ntap * Cin * Cout;
  unsigned* amax = reinterpret_cast<unsigned*>(wsdb + (size_t)p.R * Cout);
  // operand scales: taken from the caller when the forward / backward-data kernels
  // already produced them as a by-product, otherwise one reduction pass each
  const long nx0 = (long)N * D * H * W * C0, nx1 = (long)N * D * H * W * C1;
  const long ny = (long)N * Do * Ho * Wo * Cout;
  auto blocks_for = [](long n) { long b = (n / 4 + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); };
  if (!xmax_in || !ymax_in) ADELL_CHECK_HIP(hipMemsetAsync(amax, 0, 4 * sizeof(unsigned), st));
  if (!xmax_in) {   // (a split-row source carries its own exponent: its bytes are not fp32 values)
    if (!xk0)
      hipLaunchKernelGGL(adell_absmax2_kernel, dim3(blocks_for(nx0)), dim3(256), 0, st, x0, nx0, amax);
    if (C1 > 0 && !xk1)
      hipLaunchKernelGGL(adell_absmax2_kernel, dim3(blocks_for(nx1)), dim3(256), 0, st, x1, nx1, amax);
  }
  if (!ymax_in)
    hipLaunchKernelGGL(adell_absmax2_kernel, dim3(blocks_for(ny)), dim3(256), 0, st, dy, ny, amax + 1);
  if (zring) {
    rc = adell_wgrad_zring_launch(&zp, N, D, H, W, C0, C1, x0, x1, Cout, Do, Ho, Wo, dy, PD, PH, PW,
                                  slabs, db ? wsdb : nullptr, xmax_in ? xmax_in : amax,
                                  ymax_in ? ymax_in : amax + 1, st, xk0, xk1);
    if (rc != ADELL_OK) return rc;
    return adell_wgrad_reduce_launch(slabs, out, zp.R, ntap, Cin, Cout, db ? wsdb : nullptr, db, st);
  }
  if (s2) {
    rc = adell_wgrad_s2_launch(&sp, N, D, H, W, x0, Do, Ho, Wo, dy, slabs, db ? wsdb : nullptr,
                               xmax_in ? xmax_in : amax, ymax_in ? ymax_in : amax + 1, st);
    if (rc != ADELL_OK) return rc;
    return adell_wgrad_reduce_launch(slabs, out, sp.R, ntap, Cin, Cout, db ? wsdb : nullptr, db, st);
  }
  WgradF16Args a = {};
  a.x0 = x0; a.x1 = x1; a.dy = dy; a.ws = slabs; a.wsdb = db ? wsdb : nullptr;
  a.xmax = xmax_in ? xmax_in : amax;
  a.ymax = ymax_in ? ymax_in : amax + 1;
  a.N = N; a.D = D; a.H = H; a.W = W;
  a.C0 = C0; a.C1 = C1; a.Cin = Cin; a.Cout = Cout;
  a.KD = KD; a.KH = KH; a.KW = KW; a.SD = SD; a.SH = SH; a.SW = SW;
  a.PD = PD; a.PH = PH; a.PW = PW;
  a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.lTY = p.lTY; a.ntx = p.ntx; a.nty = p.nty; a.HX = p.HX; a.HY = p.HY;
  a.GKH = p.GKH; a.NGY = p.NGY;
  a.TCI = p.TCI; a.TCO = p.TCO; a.nci = p.nci; a.nco = p.nco; a.R = p.R;
  a.ksplit = p.ksplit;
  a.vecx = (C0 % 4 == 0) && (C1 % 4 == 0) && (((uintptr_t)x0 & 15) == 0) &&
           (((uintptr_t)x1 & 15) == 0);
  a.vecy = (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
  dim3 grid((unsigned)p.R, (unsigned)(p.nci * p.nco), (unsigned)(KD * p.NGY));
  if (p.maxj <= 2)
    rc = adell_launch_wgrad_f16<2>(a, grid, p.lds, st);
  else if (p.maxj <= 3)
    rc = adell_launch_wgrad_f16<3>(a, grid, p.lds, st);
  else if (p.maxj <= 5 && p.ksplit > 1 && (long)p.HX * p.HY * (p.TCI / 4) <= 4 * 256 &&
           8L * (1 << p.lTY) * (p.TCO / 4) <= 2 * 256)
    rc = adell_launch_wgrad_f16<5, 4, 2>(a, grid, p.lds, st);  // 32 x 32 channels: pipelined
  else if (p.maxj <= 5 && (long)p.HX * p.HY * (p.TCI / 4) <= 7 * 256 &&
           8L * (1 << p.lTY) * (p.TCO / 4) <= 2 * 256)
    rc = adell_launch_wgrad_f16<5, 7, 2>(a, grid, p.lds, st);  // 64 x 32 channels: pipelined
  else if (p.maxj <= 5 && (long)p.HX * p.HY * (p.TCI / 4) <= 4 * 256 &&
           8L * (1 << p.lTY) * (p.TCO / 4) <= 4 * 256)
    rc = adell_launch_wgrad_f16<5, 4, 4>(a, grid, p.lds, st);  // 32 x 64 channels: pipelined
  else if (p.maxj <= 5)
    rc = adell_launch_wgrad_f16<5>(a, grid, p.lds, st);
  else if (p.maxj <= 7)
    rc = adell_launch_wgrad_f16<7>(a, grid, p.lds, st);
  else  // 64 x 64 channels: the prefetch does not fit 256 registers, and at one block per CU
        // (512 registers, accumulators in AGPRs) it measured 152 TF against 194 TF in place
    rc = adell_launch_wgrad_f16<9>(a, grid, p.lds, st);
  if (rc != ADELL_OK) return rc;
  return adell_wgrad_reduce_launch(slabs, out, p.R, ntap, Cin, Cout, a.wsdb, db, st);
}

// small-channel paths (conv_small.hip)
extern "C" long adell_wgrad_small_workspace(const adell_conv3d_desc* d);
extern "C" int adell_wgrad_small(const adell_conv3d_desc* d, const float* x0, const float* x1,
                                 const float* dy, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream);

extern "C" long adell_conv3d_bwd_weight_f16x3_workspace(const adell_conv3d_desc* d) {
  if (!d) return ADELL_E_BADARG;
  if (adell_wgrad_small_workspace(d) > 0) return adell_wgrad_small_workspace(d);
  WgradF16Plan p = {};
  WgradZrPlan zp;
  const int Cin = d->C0 + d->C1;
  if (adell_wgrad_zring_plan(d->N, d->D, d->H, d->W, d->C0, d->C1, d->Cout, d->KD, d->KH, d->KW,
                             d->SD, d->SH, d->SW, d->Do, d->Ho, d->Wo, reinterpret_cast<int*>(&zp))) {
    // the larger of the two plans, so that a later call may take either kernel
    WgradF16Plan q = {};
    if (adell_wgrad_f16_plan(d->N, Cin, d->Cout, d->KD, d->KH, d->KW, d->SH, d->SW, d->Do, d->Ho,
                             d->Wo, &q) != ADELL_OK)
      q.R = 0;
    p.R = zp.R > q.R ? zp.R : q.R;
  } else if (adell_wgrad_f16_plan(d->N, Cin, d->Cout, d->KD, d->KH, d->KW, d->SH, d->SW, d->Do,
                                  d->Ho, d->Wo, &p) != ADELL_OK)
    return ADELL_E_UNSUPPORTED;
  WgradS2Plan sp;
  if (adell_wgrad_s2_plan(d->N, d->D, d->H, d->W, d->C0, d->C1, d->Cout, d->KD, d->KH, d->KW, d->SD,
                          d->SH, d->SW, d->PD, d->PH, d->PW, d->Do, d->Ho, d->Wo, &sp) &&
      sp.R > p.R)
    p.R = sp.R;   // the larger plan, so that a later call may take either kernel
  return (long)adell_wgrad_f16_ws(p, d->KD * d->KH * d->KW, Cin, d->Cout);
}

extern "C" int adell_conv3d_bwd_weight_f16x3(const adell_conv3d_desc* d, const float* x0,
                                             const float* x1, const float* dy, float* dw,
                                             float* db, const uint32_t* x_absmax,
                                             const uint32_t* dy_absmax, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && x0 && dy && dw, "conv_bwd_weight_f16x3: null pointer");
  ADELL_REQUIRE(d->C1 == 0 || x1, "conv_bwd_weight_f16x3: C1 > 0 needs x1");
  if (adell_wgrad_small_workspace(d) > 0)  // Cin <= 4: vector-ALU kernel, exact fp32
    return adell_wgrad_small(d, x0, x1, dy, dw, db, workspace, workspace_bytes, stream);
  return adell_wgrad_f16_core(d->N, d->D, d->H, d->W, d->C0, d->C1, x0, x1, d->Cout, d->Do,
                              d->Ho, d->Wo, dy, d->KD, d->KH, d->KW, d->SD, d->SH, d->SW, d->PD,
                              d->PH, d->PW, dw, db, x_absmax, dy_absmax, workspace, workspace_bytes,
                              (hipStream_t)stream);
}

// The same with split-row sources of X (adell_norm_act_fwd_split; include/adell_hip.h): xk0 / xk1
// non-null = that source holds rows (ONE exponent per tensor: xk[0]). _rows_ok: 1 when this problem
// runs on the z-ring kernel with every 32-channel tile inside one source.
extern "C" int adell_conv3d_bwd_weight_f16x3_rows_ok(const adell_conv3d_desc* d) {
  if (!d || adell_wgrad_small_workspace(d) > 0) return 0;
  WgradZrPlan zp;
  if (!adell_wgrad_zring_plan(d->N, d->D, d->H, d->W, d->C0, d->C1, d->Cout, d->KD, d->KH, d->KW,
                              d->SD, d->SH, d->SW, d->Do, d->Ho, d->Wo, reinterpret_cast<int*>(&zp)))
    return 0;
  return (d->C1 == 0 || d->C0 % 32 == 0) ? 1 : 0;
}

extern "C" int adell_conv3d_bwd_weight_f16x3_rows(const adell_conv3d_desc* d, const void* x0,
                                                  const int* xk0, const void* x1, const int* xk1,
                                                  const float* dy, float* dw, float* db,
                                                  const uint32_t* x_absmax,
                                                  const uint32_t* dy_absmax, void* workspace,
                                                  size_t workspace_bytes, void* stream) {
  ADELL_REQUIRE(d && x0 && dy && dw, "conv_bwd_weight_f16x3_rows: null pointer");
  ADELL_REQUIRE(d->C1 == 0 || x1, "conv_bwd_weight_f16x3_rows: C1 > 0 needs x1");
  ADELL_REQUIRE(xk0 || xk1, "conv_bwd_weight_f16x3_rows: no split-row source");
  ADELL_REQUIRE(adell_conv3d_bwd_weight_f16x3_rows_ok(d),
                "conv_bwd_weight_f16x3_rows: this problem does not take split-row sources");
  return adell_wgrad_f16_core(d->N, d->D, d->H, d->W, d->C0, d->C1, (const float*)x0,
                              (const float*)x1, d->Cout, d->Do, d->Ho, d->Wo, dy, d->KD, d->KH,
                              d->KW, d->SD, d->SH, d->SW, d->PD, d->PH, d->PW, dw, db, x_absmax,
                              dy_absmax, workspace, workspace_bytes, (hipStream_t)stream, xk0, xk1);
}
