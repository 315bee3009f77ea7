// Backward-weight of 3x3x3 stride-1 convolutions on the f16 MFMA (f16x3 operand split, see
// conv_igemm_f16.h / conv_wgrad_f16.hip), marching along z with a ring of input planes.
//
//   dW[kz][ky][kx][ci][co] = sum_{n,z,y,x} X[n, z-PD+kz, y-PH+ky, x-PW+kx][ci] * dY[n, z, y, x][co]
//
// GEMM view as in conv_wgrad_f16.hip (M = ci, N = co, K = voxels, transposing LDS reads over the
// natural [voxel][32 channels] fp16 hi / lo planes). What is different is the work decomposition:
// a block owns ONE 32 x 32 channel tile and ALL 27 taps, and walks a column of 8 x 8 x 1 output
// bricks along z. LDS keeps a ring of 10 x 10 input planes (three live, a fourth being written), so each step converts and
// stores ONE new input plane (100 rows) and ONE dY plane (64 rows) and then runs 27 taps x 4
// k-steps x 3 MFMAs on them -- three times the MFMA work per staged byte of the per-kz-plane
// kernel, whose staging (fp32 -> fp16 hi/lo conversion on the vector ALU) bounds it.
// Wave w owns taps w, w+4, w+8, ... (7 accumulators of 16 registers), every k-step.
// Work units are (column, z segment) pairs dealt round-robin to the blocks; every block writes
// one partial slab, folded in fixed order by adell_wgrad_reduce_kernel (deterministic).
#include "common.h"

typedef _Float16 zr_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 zr_half4 __attribute__((ext_vector_type(4)));
typedef __fp16 zr_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ zr_half8 adell_zr_frag(const char* p) {
  typedef __attribute__((address_space(3))) zr_fp16x4* lds_p;
  const zr_fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p));
  const zr_fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p + 4 * 64));
  zr_half8 r;
  r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
  r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
  return r;
}

struct WgradZrArgs {
  const float* x0;
  const float* x1;
  const float* dy;
  float* ws;             // [R][27][Cin][Cout]
  float* wsdb;           // [R][Cout] or null
  const unsigned* xmax;  // device absmax (float bits) of X and dY
  const unsigned* ymax;
  // split-row sources (adell_norm_act_fwd_split): xk0 / xk1 non-null = that source of X holds the
  // 64-byte hi | lo rows of 16-channel chunks scaled by 2^xk[0] (ONE exponent per tensor: the
  // accumulators of a block sum over batch items). A row occupies the bytes of its fp32 chunk, so
  // the loads are the fp32 path's; staging stores the 16-byte pieces as they are.
  const int* xk0;
  const int* xk1;
  int N, D, H, W;
  int C0, C1, Cin, Cout;
  int PD, PH, PW;
  int Do, Ho, Wo;
  int ntx, nty;          // 8 x 8 bricks per output plane
  int nseg, seglen;      // z segments per column and their length
  int nci, nco;          // 32-channel tiles
  int R;                 // blocks per channel tile (= slabs)
  int dbg;               // timing experiments (ADELL_ZR_DBG): 1 no global loads after priming,
                         // 2 no conversion / LDS stores after priming, 4 no MFMAs
};

__device__ __forceinline__ int adell_zr_scale_exp(unsigned maxbits) {
  const int ebits = (int)((maxbits >> 23) & 0xff);
  int k = 0;
  if (ebits > 0 && ebits < 255) k = 8 * ((13 - (ebits - 127)) >> 3);
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

// 4 floats -> 4 fp16 hi + 4 fp16 lo, 8 bytes each, at byte offset `off` of the two planes
__device__ __forceinline__ void adell_zr_split_store(char* hi_plane, char* lo_plane, unsigned off,
                                                     float4 f, float scale) {
  zr_half4 h, l;
  const float t[4] = {f.x * scale, f.y * scale, f.z * scale, f.w * scale};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)t[j];
    l[j] = (_Float16)(t[j] - (float)h[j]);
  }
  *reinterpret_cast<zr_half4*>(hi_plane + off) = h;
  *reinterpret_cast<zr_half4*>(lo_plane + off) = l;
}

// (ADELL_GLOBAL, common.h: the plane prefetches are in flight under the MFMAs of a z step; as flat
// loads they made every LDS fragment wait of that step a wait for them)
__device__ __forceinline__ const ADELL_GLOBAL char* adell_zr_uniform(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const ADELL_GLOBAL char*>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ float4 adell_zr_gload4(const ADELL_GLOBAL char* p) {
  const f32x4 v = *reinterpret_cast<const ADELL_GLOBAL f32x4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int ZR_HX = 10, ZR_HV = 100;      // halo plane of an 8 x 8 brick
constexpr int ZR_PLANE = ZR_HV * 64;        // bytes of one fp16 plane of 32 channels
constexpr int ZR_NX = 4, ZR_NY = 2;         // 16-byte loads per thread: 100 x 8 and 64 x 8 slots
constexpr int ZR_MAXJ = 7;                  // taps per wave
// Ring of FOUR plane slots and two dY buffers: step z reads slots z..z+2 (mod 4) and dY buffer
// z & 1 while step z+1 is already being written into slot z+3 and buffer (z+1) & 1 -- one barrier
// per step (stores done -> reads), none after the reads.
constexpr int ZR_SLOTS = 4;

__global__ __launch_bounds__(256, 2) void adell_conv_wgrad_zring_kernel(WgradZrArgs a) {
  extern __shared__ float smem[];
  char* sXh = reinterpret_cast<char*>(smem);          // [4 ring slots][100][32 halfs]
  char* sXl = sXh + ZR_SLOTS * ZR_PLANE;
  char* sYh = sXl + ZR_SLOTS * ZR_PLANE;              // [2 steps][64][32 halfs]
  char* sYl = sYh + 2 * 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // consecutive blocks go round-robin to the 8 XCDs (one L2 each): an XCD takes a contiguous eighth
  // of the regions, i.e. neighbouring columns, whose 10 x 10 input halos then meet in one L2 (the
  // blocks of one region over the channel tiles share an XCD already: R is a multiple of 8)
  const int region = (a.R & 7) == 0 ? (int)((blockIdx.x & 7) * (a.R >> 3) + (blockIdx.x >> 3))
                                    : (int)blockIdx.x;
  const int cit = blockIdx.y % a.nci, cot = blockIdx.y / a.nci;
  const int ci0 = cit * 32, co0 = cot * 32;

  // (the tile's source decides the format: the host keeps tiles that straddle two sources off
  // this path when either holds rows)
  const int* xkt = ci0 < a.C0 ? a.xk0 : a.xk1;
  const bool xrows = xkt != nullptr;
  const int kX = xrows ? xkt[0] : adell_zr_scale_exp(a.xmax[0]);
  const int kY = adell_zr_scale_exp(a.ymax[0]);
  const float sX = __int_as_float((kX + 127) << 23), sY = __int_as_float((kY + 127) << 23);

  // Whole tile inside one source (every 32-channel-aligned layer): one wave-uniform base per
  // plane. Otherwise (channel counts that are multiples of 16 only: a tile that straddles the two
  // sources of a virtual concat, or the ragged last tile) every thread addresses its own 4-channel
  // piece -- pieces never straddle a source (C0 % 16 == 0) -- and pieces past Cin are zero.
  const bool first = ci0 < a.C0;
  const bool uni = first ? (ci0 + 32 <= a.C0) : (ci0 + 32 <= a.Cin);
  const float* xsrc = first ? a.x0 + ci0 : a.x1 + (ci0 - a.C0);
  const unsigned xcs = first ? a.C0 : a.C1;

  // transposed-read lane roles (conv_wgrad_f16.hip): lane 4q+p of a 16-lane group addresses
  // voxel row q, channels 4p..4p+3 of channel half cg; the two lane halves take two brick rows
  const int cg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const int colb = (16 * cg + 4 * tp) * 2;
  const int abase = (lh * ZR_HX + tq) * 64 + colb;   // + tap offset + k-step * 2 rows
  const int bbase = (lh * 8 + tq) * 64 + colb;       // + k-step * 16 rows
  int tapoff[ZR_MAXJ], tapkz[ZR_MAXJ];
  bool jok[ZR_MAXJ];
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q) {
    int t = wave + 4 * q;
    jok[q] = t < 27;
    if (!jok[q]) t = 0;
    const int kz = t / 9, ky = (t - 9 * kz) / 3, kx = t - 9 * kz - 3 * ky;
    tapkz[q] = kz;
    tapoff[q] = abase + (ky * ZR_HX + kx) * 64;
  }
  f32x16 acc[ZR_MAXJ];
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // staging slots of this thread: 16-byte piece c4 = tid & 7 of rows tid / 8 + 32 u
  const int c4 = tid & 7, row0 = tid >> 3;
  const int cpiece = ci0 + 4 * c4;                     // first input channel of this piece
  const bool pvalid = cpiece < a.Cin, pfirst = cpiece < a.C0;
  const float* xsrc_t = pfirst ? a.x0 + cpiece : a.x1 + (pvalid ? cpiece - a.C0 : 0);
  const unsigned xcs_t = pfirst ? a.C0 : a.C1;
  const bool yvalid = co0 + 4 * c4 < a.Cout;
  const bool do_db = a.wsdb != nullptr && cit == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  const int ncols = a.N * a.ntx * a.nty;
  const long nunits = (long)ncols * a.nseg;
  for (long unit = region; unit < nunits; unit += a.R) {
    const int col = (int)(unit / a.nseg), seg = (int)(unit - (long)col * a.nseg);
    const int tx = col % a.ntx, ty = (col / a.ntx) % a.nty, nb = col / (a.ntx * a.nty);
    const int ox0 = tx * 8, oy0 = ty * 8;
    const int z0 = seg * a.seglen;
    const int z1 = (z0 + a.seglen < a.Do) ? z0 + a.seglen : a.Do;
    if (z0 >= z1) continue;
    // per-column geometry of this thread's slots: offsets inside an input / output plane
    unsigned xoff[ZR_NX], yoff[ZR_NY];
    bool xok[ZR_NX], yok[ZR_NY];
#pragma unroll
    for (int u = 0; u < ZR_NX; ++u) {
      const int hv = row0 + 32 * u;
      const int hy = hv / ZR_HX, hx = hv - hy * ZR_HX;
      const int ix = ox0 - a.PW + hx, iy = oy0 - a.PH + hy;
      xok[u] = (hv < ZR_HV) & (ix >= 0) & (ix < a.W) & (iy >= 0) & (iy < a.H) & (uni | pvalid);
      xoff[u] = !xok[u] ? 0u
                : uni   ? ((unsigned)(iy * a.W + ix) * xcs + 4 * c4) * 4u
                        : ((unsigned)(iy * a.W + ix) * xcs_t) * 4u;
    }
#pragma unroll
    for (int u = 0; u < ZR_NY; ++u) {
      const int v = row0 + 32 * u;
      const int ox = ox0 + (v & 7), oy = oy0 + (v >> 3);
      yok[u] = (ox < a.Wo) & (oy < a.Ho) & yvalid;
      yoff[u] = yok[u] ? ((unsigned)(oy * a.Wo + ox) * (unsigned)a.Cout + co0 + 4 * c4) * 4u : 0u;
    }
    float4 xr[ZR_NX], yr[ZR_NY];
    auto fetch_x = [&](int p) {   // input plane p of item nb (zeros outside the tensor)
      const bool pok = p >= 0 && p < a.D;
      const size_t plane = (size_t)(nb * a.D + (pok ? p : 0)) * a.H * a.W;
      if (uni) {
        const ADELL_GLOBAL char* base = adell_zr_uniform(xsrc + plane * xcs);
#pragma unroll
        for (int u = 0; u < ZR_NX; ++u) {
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pok && xok[u]) f = adell_zr_gload4(base + xoff[u]);
          xr[u] = f;
        }
      } else {
        const char* base = reinterpret_cast<const char*>(xsrc_t + plane * xcs_t);
#pragma unroll
        for (int u = 0; u < ZR_NX; ++u) {
          float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pok && xok[u]) f = *reinterpret_cast<const float4*>(base + xoff[u]);
          xr[u] = f;
        }
      }
    };
    auto fetch_y = [&](int z) {
      const ADELL_GLOBAL char* base =
          adell_zr_uniform(a.dy + ((size_t)(nb * a.Do + z) * a.Ho * a.Wo) * a.Cout);
#pragma unroll
      for (int u = 0; u < ZR_NY; ++u) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yok[u]) f = adell_zr_gload4(base + yoff[u]);
        yr[u] = f;
      }
    };
    auto store_x = [&](int slot) {
      if (xrows) {
        // piece c4 of the tile's 128 bytes: chunk c4 >> 2, pieces 0, 1 = hi halves (channels 0-7,
        // 8-15 of the chunk), 2, 3 = lo halves
        char* plane = (c4 & 2) ? sXl : sXh;
        const unsigned po = (unsigned)((c4 >> 2) * 32 + (c4 & 1) * 16);
#pragma unroll
        for (int u = 0; u < ZR_NX; ++u) {
          const int hv = row0 + 32 * u;
          if (hv < ZR_HV)
            *reinterpret_cast<float4*>(plane + slot * ZR_PLANE + hv * 64 + po) = xr[u];
        }
        return;
      }
#pragma unroll
      for (int u = 0; u < ZR_NX; ++u) {
        const int hv = row0 + 32 * u;
        if (hv < ZR_HV)
          adell_zr_split_store(sXh, sXl, (unsigned)(slot * ZR_PLANE + hv * 64 + c4 * 8), xr[u], sX);
      }
    };
    auto store_y = [&](int buf) {
#pragma unroll
      for (int u = 0; u < ZR_NY; ++u) {
        const int v = row0 + 32 * u;
        dbacc.x += yr[u].x; dbacc.y += yr[u].y; dbacc.z += yr[u].z; dbacc.w += yr[u].w;
        adell_zr_split_store(sYh, sYl, (unsigned)(buf * 64 * 64 + v * 64 + c4 * 8), yr[u], sY);
      }
    };
    // prime the ring: planes of taps kz = 0, 1 of the first step. Plane p = z - PD + kz sits in
    // slot (z + kz) & 3.
    __syncthreads();   // the previous unit's MFMAs are done with LDS
    fetch_x(z0 - a.PD);
    store_x(z0 & 3);
    fetch_x(z0 - a.PD + 1);
    store_x((z0 + 1) & 3);
    fetch_x(z0 - a.PD + 2);
    fetch_y(z0);
    for (int z = z0; z < z1; ++z) {
      // registers -> LDS: the new input plane (tap kz = 2 of this step) and this step's dY
      if (!(ADELL_DBG(a.dbg) & 2) || z == z0) {
        store_x((z + 2) & 3);
        store_y(z & 1);
      }
      if (z + 1 < z1 && !(ADELL_DBG(a.dbg) & 1)) {  // next step's loads fly during this step's MFMAs
        fetch_x(z + 1 - a.PD + 2);
        fetch_y(z + 1);
      }
      __syncthreads();
      int slotoff[3];   // byte offset of the ring slot holding tap plane kz
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) slotoff[kz] = ((z + kz) & 3) * ZR_PLANE;
      const int ybuf = (z & 1) * 64 * 64;
      // 4 k-steps of 16 voxels (two brick rows) x 7 taps, flattened and software-pipelined: the
      // fragments of job i+1 are read while the 3 MFMAs of job i run (two register sets)
      int tslot[ZR_MAXJ];
#pragma unroll
      for (int q = 0; q < ZR_MAXJ; ++q)
        tslot[q] = tapoff[q] + (tapkz[q] == 0 ? slotoff[0] : (tapkz[q] == 1 ? slotoff[1] : slotoff[2]));
      zr_half8 ah[2], al[2], bh[2], bl[2];
      if (ADELL_DBG(a.dbg) & 4) continue;
      bh[0] = adell_zr_frag(sYh + ybuf + bbase);
      bl[0] = adell_zr_frag(sYl + ybuf + bbase);
      ah[0] = adell_zr_frag(sXh + tslot[0]);
      al[0] = adell_zr_frag(sXl + tslot[0]);
#pragma unroll
      for (int i = 0; i < 4 * ZR_MAXJ; ++i) {
        const int s = i / ZR_MAXJ, q = i - s * ZR_MAXJ;
        const int cur = i & 1, nxt = cur ^ 1;
        const bool more = i + 1 < 4 * ZR_MAXJ;
        const bool newb = more && q + 1 == ZR_MAXJ;
        if (more) {
          const int s2 = (i + 1) / ZR_MAXJ, q2 = (i + 1) - s2 * ZR_MAXJ;
          if (newb) {
            bh[s2 & 1] = adell_zr_frag(sYh + ybuf + bbase + s2 * 16 * 64);
            bl[s2 & 1] = adell_zr_frag(sYl + ybuf + bbase + s2 * 16 * 64);
          }
          ah[nxt] = adell_zr_frag(sXh + tslot[q2] + s2 * 2 * ZR_HX * 64);
          al[nxt] = adell_zr_frag(sXl + tslot[q2] + s2 * 2 * ZR_HX * 64);
        }
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[s & 1], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[s & 1], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[s & 1], acc[q], 0, 0, 0);
        if (more) {  // pin: the next job's LDS reads go between this job's MFMAs
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (newb) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (newb) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    }
  }

  if (do_db) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 8) {
      float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = tid; k < 256; k += 8) {
        const float4 u = red[k];
        tsum.x += u.x; tsum.y += u.y; tsum.z += u.z; tsum.w += u.w;
      }
      if (co0 + 4 * tid < a.Cout) {
        float* o = a.wsdb + (size_t)region * a.Cout + co0 + 4 * tid;
        o[0] = tsum.x; o[1] = tsum.y; o[2] = tsum.z; o[3] = tsum.w;
      }
    }
  }
  // ---- partial slab (undo the operand scales) --------------------------------
  const float unscale = __int_as_float((127 - kX - kY) << 23);
  const int co = co0 + li;
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q) {
    const int tap = wave + 4 * q;
    const int cib = ci0 + 4 * lh;
    float* base = a.ws + (((size_t)region * 27 + tap) * a.Cin + cib) * a.Cout + co;
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

// ---------------------------------------------------------------------------------------------
// 16 x 16 channel tiles on v_mfma_f32_16x16x32_f16 (layers with 16 input or 16 output channels: the
// full-resolution levels of UNETR). On the 32 x 32 tile above such a layer fills a quarter (16 -> 16)
// or half (32 -> 16) of every MFMA: 54 / 75 TF algorithmic at 4 x 96^3. Same march, same ring, same
// slabs; what changes:
//  * a k-step is 32 voxels = FOUR brick rows (lane group g = lane >> 4 is the k-block = brick row
//    4 s + g; lane & 15 the channel), two k-steps per 8 x 8 plane, 27 taps over the four waves;
//  * plane images hold 16 channels (32-byte rows) with a row pitch of 12 voxels for X and dY alike:
//    the two 16-lane groups of a transposed read's 32-lane half are then 384 B = 128 B (mod 256)
//    apart -- disjoint bank halves for every tap offset;
//  * 43 KB of LDS and < 128 registers: three blocks per CU. Per plane a block moves 10.4 KB and runs
//    162 MFMAs of 16 cycles (650 cycles per SIMD): the launch is HBM-bound, not MFMA-bound.
constexpr int Z16_HXP = 12;
constexpr int Z16_XPLANE = 10 * Z16_HXP * 32;   // bytes of one fp16 plane (hi or lo), 16 channels
constexpr int Z16_YPLANE = 8 * Z16_HXP * 32;
constexpr int Z16_PF = 4;                       // steps of register prefetch

__device__ __forceinline__ zr_half8 adell_z16_frag(const char* p) {
  typedef __attribute__((address_space(3))) zr_fp16x4* lds_p;
  const zr_fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p));
  const zr_fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(p + 4 * 32));
  zr_half8 r;
  r[0] = (_Float16)lo4[0]; r[1] = (_Float16)lo4[1]; r[2] = (_Float16)lo4[2]; r[3] = (_Float16)lo4[3];
  r[4] = (_Float16)hi4[0]; r[5] = (_Float16)hi4[1]; r[6] = (_Float16)hi4[2]; r[7] = (_Float16)hi4[3];
  return r;
}

__device__ __forceinline__ void adell_z16_split_store(char* hi_plane, char* lo_plane, unsigned off,
                                                      float4 f, float scale) {
  adell_zr_split_store(hi_plane, lo_plane, off, f, scale);
}

__global__ __launch_bounds__(256, 3) void adell_conv_wgrad_zring16_kernel(WgradZrArgs a) {
  extern __shared__ float smem[];
  char* sXh = reinterpret_cast<char*>(smem);          // [4 ring slots][10 x 12 rows][16 halfs]
  char* sXl = sXh + ZR_SLOTS * Z16_XPLANE;
  char* sYh = sXl + ZR_SLOTS * Z16_XPLANE;            // [2 steps][8 x 12 rows][16 halfs]
  char* sYl = sYh + 2 * Z16_YPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int region = (a.R & 7) == 0 ? (int)((blockIdx.x & 7) * (a.R >> 3) + (blockIdx.x >> 3))
                                    : (int)blockIdx.x;   // (XCD-contiguous regions, see above)
  const int cit = blockIdx.y % a.nci, cot = blockIdx.y / a.nci;
  const int ci0 = cit * 16, co0 = cot * 16;

  const bool first = ci0 < a.C0;                      // a tile never straddles the sources (C0 % 16 == 0)
  const int* xkt = first ? a.xk0 : a.xk1;
  const bool xrows = xkt != nullptr;
  const int kX = xrows ? xkt[0] : adell_zr_scale_exp(a.xmax[0]);
  const int kY = adell_zr_scale_exp(a.ymax[0]);
  const float sX = __int_as_float((kX + 127) << 23), sY = __int_as_float((kY + 127) << 23);
  const float* xsrc = first ? a.x0 + ci0 : a.x1 + (ci0 - a.C0);
  const unsigned xcs = first ? a.C0 : a.C1;

  // transposed-read lane roles: lane 4q + p of 16-lane group g addresses voxel q of brick row g,
  // channels 4p .. 4p + 3; after the transpose lane j of the group holds channel j, 4 voxels
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int fbase = (g * Z16_HXP + tq) * 32 + tp * 8;   // + tap offset + k-step * 4 rows
  int tapoff[ZR_MAXJ], tapkz[ZR_MAXJ];
  bool jok[ZR_MAXJ];
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q) {
    int t = wave + 4 * q;
    jok[q] = t < 27;
    if (!jok[q]) t = 0;
    const int kz = t / 9, ky = (t - 9 * kz) / 3, kx = t - 9 * kz - 3 * ky;
    tapkz[q] = kz;
    tapoff[q] = fbase + (ky * Z16_HXP + kx) * 32;
  }
  f32x4 acc[ZR_MAXJ];
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;

  // staging slots of this thread: 16-byte piece c4 = tid & 3 of halo rows tid / 4 (+ 64) and of dY
  // row tid / 4
  const int c4 = tid & 3, row0 = tid >> 2;
  const bool do_db = a.wsdb != nullptr && cit == 0;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  const int ncols = a.N * a.ntx * a.nty;
  const long nunits = (long)ncols * a.nseg;
  for (long unit = region; unit < nunits; unit += a.R) {
    const int col = (int)(unit / a.nseg), seg = (int)(unit - (long)col * a.nseg);
    const int tx = col % a.ntx, ty = (col / a.ntx) % a.nty, nb = col / (a.ntx * a.nty);
    const int ox0 = tx * 8, oy0 = ty * 8;
    const int z0 = seg * a.seglen;
    const int z1 = (z0 + a.seglen < a.Do) ? z0 + a.seglen : a.Do;
    if (z0 >= z1) continue;
    unsigned xoff[2], xlds[2], yoff, ylds;
    bool xok[2], yok;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int hv = row0 + 64 * u;
      const int hy = hv / ZR_HX, hx = hv - hy * ZR_HX;
      const int ix = ox0 - a.PW + hx, iy = oy0 - a.PH + hy;
      xok[u] = (hv < ZR_HV) & (ix >= 0) & (ix < a.W) & (iy >= 0) & (iy < a.H);
      xoff[u] = xok[u] ? ((unsigned)(iy * a.W + ix) * xcs + 4 * c4) * 4u : 0u;
      xlds[u] = (unsigned)(hy * Z16_HXP + hx) * 32u;
    }
    {
      const int ox = ox0 + (row0 & 7), oy = oy0 + (row0 >> 3);
      yok = (ox < a.Wo) & (oy < a.Ho);
      yoff = yok ? ((unsigned)(oy * a.Wo + ox) * (unsigned)a.Cout + co0 + 4 * c4) * 4u : 0u;
      ylds = (unsigned)((row0 >> 3) * Z16_HXP + (row0 & 7)) * 32u;
    }
    // Register prefetch Z16_PF steps deep: a step's MFMAs take ~0.3 us of a SIMD, a load from HBM
    // 1.5-2 us -- with one step of lookahead (the 32 x 32 form's, whose steps are 4x longer) every
    // step waited for memory (2.7 us per step measured). Loads are unconditional (planes outside
    // the tensor / past the segment read a valid plane and are zeroed or dropped at store time), so
    // the compiler's vmcnt bookkeeping stays static.
    float4 xr[Z16_PF][2], yr[Z16_PF];
    auto xbase = [&](int p) {     // input plane p of item nb, clamped into the tensor
      const int pc = p < 0 ? 0 : (p < a.D ? p : a.D - 1);
      return adell_zr_uniform(xsrc + (size_t)(nb * a.D + pc) * a.H * a.W * xcs);
    };
    auto ybase = [&](int z) {
      const int zc = z < a.Do ? z : a.Do - 1;
      return adell_zr_uniform(a.dy + ((size_t)(nb * a.Do + zc) * a.Ho * a.Wo) * a.Cout);
    };
    auto put_x = [&](int slot, int p, const float4* v) {   // plane p (zeros outside the tensor)
      const bool pok = p >= 0 && p < a.D;
      if (xrows) {
        // piece c4 of the chunk's 64 bytes: 0, 1 = hi halves (channels 0-7, 8-15), 2, 3 = lo halves
        char* plane = ((c4 & 2) ? sXl : sXh) + slot * Z16_XPLANE + (c4 & 1) * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (row0 + 64 * u < ZR_HV)
            *reinterpret_cast<float4*>(plane + xlds[u]) =
                (pok && xok[u]) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        return;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (row0 + 64 * u < ZR_HV)
          adell_zr_split_store(sXh, sXl, (unsigned)(slot * Z16_XPLANE) + xlds[u] + c4 * 8,
                               (pok && xok[u]) ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f), sX);
    };
    auto put_y = [&](int buf, float4 v) {
      if (!yok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      dbacc.x += v.x; dbacc.y += v.y; dbacc.z += v.z; dbacc.w += v.w;
      adell_zr_split_store(sYh, sYl, (unsigned)(buf * Z16_YPLANE) + ylds + c4 * 8, v, sY);
    };
    __syncthreads();   // the previous unit's MFMAs are done with LDS
    {
      // prime the ring: planes of taps kz = 0, 1 of the first step (plane z - PD + kz sits in slot
      // (z + kz) & 3), then the first Z16_PF steps' new planes into the prefetch registers
      const ADELL_GLOBAL char* b0 = xbase(z0 - a.PD);
      const ADELL_GLOBAL char* b1 = xbase(z0 - a.PD + 1);
      float4 t0[2], t1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        t0[u] = adell_zr_gload4(b0 + xoff[u]);
        t1[u] = adell_zr_gload4(b1 + xoff[u]);
      }
#pragma unroll
      for (int j = 0; j < Z16_PF; ++j) {
        const ADELL_GLOBAL char* bx = xbase(z0 + j - a.PD + 2);
        const ADELL_GLOBAL char* by = ybase(z0 + j);
#pragma unroll
        for (int u = 0; u < 2; ++u) xr[j][u] = adell_zr_gload4(bx + xoff[u]);
        yr[j] = adell_zr_gload4(by + yoff);
      }
      put_x(z0 & 3, z0 - a.PD, t0);
      put_x((z0 + 1) & 3, z0 - a.PD + 1, t1);
    }
    for (int zb = z0; zb < z1; zb += Z16_PF) {
#pragma unroll
     for (int j = 0; j < Z16_PF; ++j) {
      const int z = zb + j;
      if (z >= z1) break;
      put_x((z + 2) & 3, z - a.PD + 2, xr[j]);
      put_y(z & 1, yr[j]);
      {  // the loads of step z + Z16_PF fly during the MFMAs of the steps in between
        const ADELL_GLOBAL char* bx = xbase(z + Z16_PF - a.PD + 2);
        const ADELL_GLOBAL char* by = ybase(z + Z16_PF);
#pragma unroll
        for (int u = 0; u < 2; ++u) xr[j][u] = adell_zr_gload4(bx + xoff[u]);
        yr[j] = adell_zr_gload4(by + yoff);
      }
      __syncthreads();
      int slotoff[3];
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) slotoff[kz] = ((z + kz) & 3) * Z16_XPLANE;
      const int ybuf = (z & 1) * Z16_YPLANE;
      int tslot[ZR_MAXJ];
#pragma unroll
      for (int q = 0; q < ZR_MAXJ; ++q)
        tslot[q] = tapoff[q] + (tapkz[q] == 0 ? slotoff[0] : (tapkz[q] == 1 ? slotoff[1] : slotoff[2]));
      // 2 k-steps of 32 voxels (four brick rows) x 7 taps, flattened and software-pipelined
      zr_half8 ah[2], al[2], bh[2], bl[2];
      bh[0] = adell_z16_frag(sYh + ybuf + fbase);
      bl[0] = adell_z16_frag(sYl + ybuf + fbase);
      ah[0] = adell_z16_frag(sXh + tslot[0]);
      al[0] = adell_z16_frag(sXl + tslot[0]);
#pragma unroll
      for (int i = 0; i < 2 * ZR_MAXJ; ++i) {
        const int s = i / ZR_MAXJ, q = i - s * ZR_MAXJ;
        const int cur = i & 1, nxt = cur ^ 1;
        const bool more = i + 1 < 2 * ZR_MAXJ;
        const bool newb = more && q + 1 == ZR_MAXJ;
        if (more) {
          const int s2 = (i + 1) / ZR_MAXJ, q2 = (i + 1) - s2 * ZR_MAXJ;
          if (newb) {
            bh[s2 & 1] = adell_z16_frag(sYh + ybuf + fbase + s2 * 4 * Z16_HXP * 32);
            bl[s2 & 1] = adell_z16_frag(sYl + ybuf + fbase + s2 * 4 * Z16_HXP * 32);
          }
          ah[nxt] = adell_z16_frag(sXh + tslot[q2] + s2 * 4 * Z16_HXP * 32);
          al[nxt] = adell_z16_frag(sXl + tslot[q2] + s2 * 4 * Z16_HXP * 32);
        }
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur], bh[s & 1], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bl[s & 1], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bh[s & 1], acc[q], 0, 0, 0);
      }
     }
    }
  }

  if (do_db) {
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = dbacc;
    __syncthreads();
    if (tid < 4) {
      float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = tid; k < 256; k += 4) {
        const float4 u = red[k];
        tsum.x += u.x; tsum.y += u.y; tsum.z += u.z; tsum.w += u.w;
      }
      float* o = a.wsdb + (size_t)region * a.Cout + co0 + 4 * tid;
      o[0] = tsum.x; o[1] = tsum.y; o[2] = tsum.z; o[3] = tsum.w;
    }
  }
  // ---- partial slab: C[row = ci 4 g + r][col = co lane & 15] -------------------
  const float unscale = __int_as_float((127 - kX - kY) << 23);
#pragma unroll
  for (int q = 0; q < ZR_MAXJ; ++q) {
    const int tap = wave + 4 * q;
    if (jok[q]) {
      float* base = a.ws + (((size_t)region * 27 + tap) * a.Cin + ci0 + 4 * g) * a.Cout + co0 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) base[(size_t)r * a.Cout] = acc[q][r] * unscale;
    }
  }
}

struct WgradZrPlan {
  int ntx, nty, nseg, seglen, nci, nco, R, t16;
};

// 1 when the z-ring kernel serves this problem (3^3 taps, stride 1, channel counts in whole 16s)
extern "C" int adell_wgrad_zring_plan(int N, int D, int H, int W, int C0, int C1, int Cout, int KD,
                                      int KH, int KW, int SD, int SH, int SW, int Do, int Ho, int Wo,
                                      int* plan) {
  static_assert(sizeof(WgradZrPlan) == 8 * sizeof(int), "plan[8] of include/adell_hip.h");
  WgradZrPlan* p = reinterpret_cast<WgradZrPlan*>(plan);
  const int Cin = C0 + C1;
  if (KD != 3 || KH != 3 || KW != 3 || SD != 1 || SH != 1 || SW != 1) return 0;
  // whole 16-channel pieces: tiles are 32 x 32 channels, a ragged last tile or one that straddles
  // the two sources of a virtual concat is masked per 4-channel piece (half the MFMA columns idle,
  // still far ahead of the generic kernel on the 16-channel decoder levels of UNETR)
  if (Cin % 16 || Cout % 16 || C0 % 16) return 0;
  if (Wo < 8 || Ho < 8 || Do < 4) return 0;   // small planes: the per-plane kernel wastes less
  const size_t cmax = (size_t)(C0 > C1 ? C0 : C1);
  if ((size_t)H * W * cmax >= ((size_t)1 << 30) || (size_t)Ho * Wo * Cout >= ((size_t)1 << 30))
    return 0;                                  // 32-bit byte offsets inside a plane
  if (g_adell_tune.wgrad_nozring) return 0;
  p->ntx = adell_cdiv(Wo, 8);
  p->nty = adell_cdiv(Ho, 8);
  // 16 x 16 tiles (adell_conv_wgrad_zring16_kernel) when a 32-wide tile would be half empty on
  // either side
  p->t16 = (Cin == 16 || Cout == 16) && !g_adell_tune.wgrad_no16 ? 1 : 0;
  p->nci = adell_cdiv(Cin, p->t16 ? 16 : 32);
  p->nco = adell_cdiv(Cout, p->t16 ? 16 : 32);
  const long ncols = (long)N * p->ntx * p->nty;
  const long chan_blocks = (long)p->nci * p->nco;
  // two (three: the 16-channel form) blocks per CU over all channel tiles
  long target = (p->t16 ? 768 : 512) / chan_blocks;
  if (target < 8) target = 8;
  // z segments: at least 4 steps each (2 priming planes), enough units to fill the target,
  // and -- round 4 -- a unit count the blocks share out evenly: units are dealt round-robin, so the
  // launch lasts ceil(units / blocks) units of (seglen + priming) steps. 4 x 96^3 has 576 columns:
  // as 576 units on 512 blocks an eighth of the blocks ran two whole columns while the rest idled
  // (makespan 2 x 98 steps; 8 segments: 9 x 14). Columns that fill the target exactly (2 x 128^3:
  // 512) keep one segment.
  const int minseg = 4;
  const long maxseg = Do / minseg > 0 ? Do / minseg : 1;
  long nseg = 1;
  {
    long best = -1;
    for (long cand = 1; cand <= maxseg; ++cand) {
      const long sl = adell_cdiv(Do, (int)cand), ns = adell_cdiv(Do, (int)sl);
      if (ns != cand) continue;                       // same partition as a smaller candidate
      const long units = ncols * ns;
      const long blocks = units < target ? units : target;
      const long cost = adell_cdiv(units, blocks) * (sl + 4);
      // (fewer, longer segments on a tie: fewer priming planes and less halo re-read)
      if (best < 0 || cost < best) { best = cost; nseg = cand; }
    }
  }
  p->seglen = (int)adell_cdiv(Do, (int)nseg);
  p->nseg = adell_cdiv(Do, p->seglen);
  const long nunits = ncols * p->nseg;
  p->R = (int)(nunits < target ? nunits : target);
  return 1;
}

extern "C" size_t adell_wgrad_zring_ws_floats(const WgradZrPlan* p, int Cin, int Cout) {
  return (size_t)p->R * 27 * Cin * Cout + (size_t)p->R * Cout + 4;
}

extern "C" int adell_wgrad_zring_launch(const WgradZrPlan* p, int N, int D, int H, int W, int C0,
                                        int C1, const float* x0, const float* x1, int Cout, int Do,
                                        int Ho, int Wo, const float* dy, int PD, int PH, int PW,
                                        float* slabs, float* wsdb, const unsigned* xmax,
                                        const unsigned* ymax, hipStream_t st,
                                        const int* xk0 = nullptr, const int* xk1 = nullptr) {
  WgradZrArgs a = {};
  a.x0 = x0; a.x1 = x1; a.dy = dy; a.ws = slabs; a.wsdb = wsdb; a.xmax = xmax; a.ymax = ymax;
  a.xk0 = xk0; a.xk1 = xk1;
  a.N = N; a.D = D; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.Cout = Cout;
  a.PD = PD; a.PH = PH; a.PW = PW; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.ntx = p->ntx; a.nty = p->nty; a.nseg = p->nseg; a.seglen = p->seglen;
  a.nci = p->nci; a.nco = p->nco; a.R = p->R;
  a.dbg = g_adell_tune.zr_dbg;
  if (p->t16) {
    const size_t lds16 = 2 * (ZR_SLOTS * (size_t)Z16_XPLANE + 2 * (size_t)Z16_YPLANE);
    hipLaunchKernelGGL(adell_conv_wgrad_zring16_kernel, dim3((unsigned)p->R, (unsigned)(p->nci * p->nco)),
                       dim3(256), lds16, st, a);
    ADELL_CHECK_HIP(hipGetLastError());
    return ADELL_OK;
  }
  const size_t lds = 2 * (ZR_SLOTS * (size_t)ZR_PLANE + 2 * 64 * 64);
  static bool attr_done = false;
  if (!attr_done) {
    ADELL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(adell_conv_wgrad_zring_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(adell_conv_wgrad_zring_kernel, dim3((unsigned)p->R, (unsigned)(p->nci * p->nco)),
                     dim3(256), lds, st, a);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
