// 3x3x3 stride-1 convolution with 16 input and 16 output channels (the full-resolution levels of
// UNETR: forward and backward-data of its 16 -> 16 layers at 4 x 96^3) on v_mfma_f32_16x16x32_f16,
// marching along z with a ring of halo planes (the work decomposition of conv_wgrad_zring.hip).
//
// On the implicit-GEMM instances (conv_igemm_f16.h) such a layer fills half of every 32-column MFMA
// and, with ONE 16-channel chunk per brick, pays a brick's whole staging and epilogue for a third of
// the usual MFMA work: 0.40 ms forward / 0.63 ms backward-data per launch at 4 x 96^3 (120 / 77 TF),
// where the tensors move in ~0.12 ms.
//
// Here a block owns an 8 x 8 column of output voxels and a z segment. Per step it stages ONE new
// 10 x 10 input plane (6.4 KB; halo re-read 1.56x in xy, none along z) and produces one 8 x 8 output
// plane. The split-row format (conv_igemm_f16.h: per voxel [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15]) IS
// the A operand of the 16x16x32 MFMA -- lane group g = lane >> 4 reads slot g of voxel lane & 15:
//
//     [A_hi | A_lo] x [B_hi ; B_hi] = A_hi B_hi + A_lo B_hi      (K = 32 = the row's 32 halfs)
//     [A_hi | A_lo] x [B_lo ;  0  ] = A_hi B_lo
//
// two MFMAs per tap and 16 voxels, one ds_read_b128 per A fragment (slots XOR-ed with hx & 3:
// conflict-free for all 9 in-plane shifts, searched exhaustively). The four waves split the 27 taps
// (their B fragments stay in registers for the whole launch) and fold their partial planes through
// LDS in wave order (double-buffered: the fold of a plane runs after the next step's barrier, so a
// step has ONE barrier). fp32 sources: one power-of-two scale per staged plane (block-wide absmax,
// published one step ahead, so no extra barrier); the three kz groups of a step accumulate
// separately and are combined with their planes' scales. Split-row sources carry their exponent.
// Loads run CZ_PF planes ahead in registers.
#define ADELL_NO_PACK_KERNELS
#include "conv_igemm_f16.h"

constexpr int CZ_HX = 10, CZ_HV = 100, CZ_PLANE = CZ_HV * 64, CZ_PF = 4, CZ_MAXJ = 7;

struct ConvZr16Args {
  ConvArgs a;
  ConvF16Extra e;
  int seglen, nseg;
};

__device__ __forceinline__ int adell_cz_exp(float mx) {
  const int ebits = (__float_as_int(mx) >> 23) & 0xff;
  int k = 0;
  // max lands in [2^6, 2^14): multiples of 8 so the scale rarely changes between planes
  if (ebits > 0 && ebits < 255) k = 8 * ((13 - (ebits - 127)) >> 3);
  if (k > 96) k = 96;
  if (k < -96) k = -96;
  return k;
}

template <int ROWS>
__global__ __launch_bounds__(256, 2) void adell_conv_zring16_kernel(ConvZr16Args args) {
  const ConvArgs& a = args.a;
  const ConvF16Extra& e = args.e;
  extern __shared__ float smem[];
  char* sX = reinterpret_cast<char*>(smem);                       // [4 ring slots][100 rows][64 B]
  float* sRed = reinterpret_cast<float*>(sX + 4 * CZ_PLANE);       // [2 steps][4 waves][64 voxels][16 co]
  float* sMax = sRed + 2 * 4 * 64 * 16;                            // [4 slots][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;

  // consecutive blocks go round-robin to the 8 XCDs: give every XCD one contiguous range of units
  const int per_item = a.ntx * a.nty * args.nseg;
  int t = (blockIdx.x & 7) * ((per_item + 7) >> 3) + (blockIdx.x >> 3);
  if (t >= per_item) return;
  const int nb = blockIdx.y;
  const int tile_id = t;
  const int seg = t % args.nseg;
  t /= args.nseg;
  const int tx = t % a.ntx, ty = t / a.ntx;
  const int ox0 = tx * 8, oy0 = ty * 8;
  const int z0 = seg * args.seglen;
  const int z1 = (z0 + args.seglen < a.Do) ? z0 + args.seglen : a.Do;
  if (z0 >= z1) {
    // (cannot happen: nseg = ceil(Do / seglen); the statistics row of an empty unit would be unset)
    return;
  }

  // ---- weights of this wave's taps: B fragments in registers ------------------------------------
  half8 b1[CZ_MAXJ], b2[CZ_MAXJ];
  int aoff[CZ_MAXJ], tkz[CZ_MAXJ];
#pragma unroll
  for (int q = 0; q < CZ_MAXJ; ++q) {
    int tap = wave + 4 * q;
    const bool ok = tap < 27;
    if (!ok) tap = 0;
    const int kz = tap / 9, ky = (tap - 9 * kz) / 3, kx = tap - 9 * kz - 3 * ky;
    tkz[q] = ok ? kz : 3;
    const int hx = (r & 7) + kx;
    aoff[q] = ((ky + (r >> 3)) * CZ_HX + hx) * 64 + (((g ^ hx) & 3) << 4);
    half8 z8;
#pragma unroll
    for (int j = 0; j < 8; ++j) z8[j] = (_Float16)0.f;
    b1[q] = z8;
    b2[q] = z8;
    if (ok) {
      const char* p = reinterpret_cast<const char*>(e.wh) + ((size_t)tap * 16 + r) * 64;
      b1[q] = *reinterpret_cast<const half8*>(p + 16 * (g & 1));           // [B_hi ; B_hi]
      if (g < 2) b2[q] = *reinterpret_cast<const half8*>(p + 32 + 16 * g);  // [B_lo ; 0]
    }
  }

  // ---- staging roles ------------------------------------------------------------------------------
  // fp32 source: thread = (halo voxel tid >> 1, channel half tid & 1), two 16-byte loads, writes the
  // hi and the lo piece of its 8 channels. Rows source: 16-byte pieces tid and tid + 256 of the 400.
  const size_t vox0 = (size_t)nb * a.D * a.H * a.W;
  unsigned goff[2], loff[2][2];
  bool pok_xy[2];
  if (ROWS) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u, hv = idx >> 2, q = idx & 3;
      const int hy = hv / CZ_HX, hx = hv - hy * CZ_HX;
      const int ix = ox0 - a.PW + hx, iy = oy0 - a.PH + hy;
      pok_xy[u] = (hv < CZ_HV) & (ix >= 0) & (ix < a.W) & (iy >= 0) & (iy < a.H);
      goff[u] = pok_xy[u] ? (unsigned)(iy * a.W + ix) * 64u + 16u * q : 0u;
      loff[u][0] = (unsigned)hv * 64u + (((q ^ hx) & 3) << 4);
      loff[u][1] = 0;
    }
  } else {
    const int hv = tid >> 1, j = tid & 1;
    const int hy = hv / CZ_HX, hx = hv - hy * CZ_HX;
    const int ix = ox0 - a.PW + hx, iy = oy0 - a.PH + hy;
    pok_xy[0] = pok_xy[1] = (hv < CZ_HV) & (ix >= 0) & (ix < a.W) & (iy >= 0) & (iy < a.H);
    goff[0] = pok_xy[0] ? ((unsigned)(iy * a.W + ix) * 16u + 8u * j) * 4u : 0u;
    goff[1] = goff[0] + 16u;
    loff[0][0] = (unsigned)hv * 64u + (((j ^ hx) & 3) << 4);          // hi piece of channels 8 j ..
    loff[0][1] = (unsigned)hv * 64u + ((((2 + j) ^ hx) & 3) << 4);    // lo piece
    loff[1][0] = loff[1][1] = 0;
  }
  const bool stager = ROWS ? true : (tid < 2 * CZ_HV);
  const int kRows = ROWS ? e.xk0[nb] : 0;     // (one 16-channel chunk per voxel: exponent [item])
  const char* src_item = ROWS ? e.xs0 + vox0 * 64 : reinterpret_cast<const char*>(a.x0 + vox0 * 16);
  const int p0 = z0 - a.PD;                   // input plane of pipeline step 0
  auto plane_base = [&](int p) {              // clamped into the tensor (masked when staged)
    const int pc = p < 0 ? 0 : (p < a.D ? p : a.D - 1);
    return adell_uniform_ptr(src_item + (size_t)pc * a.H * a.W * 64);
  };
  float4 xr[CZ_PF][2];
  auto fetch = [&](float4* v, int p) {
    const ADELL_GLOBAL char* base = plane_base(p);
    v[0] = adell_gload4(base + goff[0]);
    v[1] = adell_gload4(base + goff[1]);
  };
  auto masked = [&](const float4* v, int p, int u) {
    const bool ok = (p >= 0) & (p < a.D) & pok_xy[u] & stager;
    return ok ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float blockmax = 0.f;
  auto publish_max = [&](const float4* v, int p, int slot) {   // fp32 path: absmax of plane p
    if (ROWS) return;
    const float4 f0 = masked(v, p, 0), f1 = masked(v, p, 1);
    float mx = fmaxf(fmaxf(fmaxf(fabsf(f0.x), fabsf(f0.y)), fmaxf(fabsf(f0.z), fabsf(f0.w))),
                     fmaxf(fmaxf(fabsf(f1.x), fabsf(f1.y)), fmaxf(fabsf(f1.z), fabsf(f1.w))));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sMax[slot * 4 + wave] = mx;
  };
  auto plane_exp = [&](int slot) -> int {
    if (ROWS) return kRows;
    const float mx = fmaxf(fmaxf(sMax[slot * 4], sMax[slot * 4 + 1]),
                           fmaxf(sMax[slot * 4 + 2], sMax[slot * 4 + 3]));
    blockmax = fmaxf(blockmax, mx);
    return adell_cz_exp(mx);
  };
  auto put = [&](const float4* v, int p, int slot, int kA) {
    char* dst = sX + slot * CZ_PLANE;
    if (ROWS) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + 256 * u < 4 * CZ_HV) *reinterpret_cast<float4*>(dst + loff[u][0]) = masked(v, p, u);
      return;
    }
    if (!stager) return;
    const float4 f0 = masked(v, p, 0), f1 = masked(v, p, 1);
    const float vals[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    half8 hi, lo;
    adell_split8(vals, __int_as_float((kA + 127) << 23), &hi, &lo);
    *reinterpret_cast<half8*>(dst + loff[0][0]) = hi;
    *reinterpret_cast<half8*>(dst + loff[0][1]) = lo;
  };

  // ---- epilogue roles: thread = (voxel tid >> 2 of the plane, channels 4 (tid & 3) ..) ------------
  const int ev = tid >> 2, ec = 4 * (tid & 3);
  const int eox = ox0 + (ev & 7), eoy = oy0 + (ev >> 3);
  const bool eok = (eox < a.Wo) & (eoy < a.Ho);
  const size_t erow0 = ((size_t)nb * a.Do * a.Ho + eoy) * a.Wo + eox;     // + z * Ho * Wo
  const float4 wsc = *reinterpret_cast<const float4*>(e.wscale + ec);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + ec);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;

  // fold in wave order, weight scale, bias, residual, store, statistics. Runs one step LATE, after
  // the next step's barrier (the partial planes are double-buffered): one barrier per step.
  auto epilogue = [&](int z, float4 res4, const float* red) {
    const float4 u0 = *reinterpret_cast<const float4*>(red + (0 * 64 + ev) * 16 + ec);
    const float4 u1 = *reinterpret_cast<const float4*>(red + (1 * 64 + ev) * 16 + ec);
    const float4 u2 = *reinterpret_cast<const float4*>(red + (2 * 64 + ev) * 16 + ec);
    const float4 u3 = *reinterpret_cast<const float4*>(red + (3 * 64 + ev) * 16 + ec);
    float4 v;
    v.x = (((u0.x + u1.x) + u2.x) + u3.x) * wsc.x + bias4.x + res4.x;
    v.y = (((u0.y + u1.y) + u2.y) + u3.y) * wsc.y + bias4.y + res4.y;
    v.z = (((u0.z + u1.z) + u2.z) + u3.z) * wsc.z + bias4.z + res4.z;
    v.w = (((u0.w + u1.w) + u2.w) + u3.w) * wsc.w + bias4.w + res4.w;
    if (eok) {
      *reinterpret_cast<float4*>(a.y0 + (erow0 + (size_t)z * a.Ho * a.Wo) * 16 + ec) = v;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
    }
  };
  // ---- pipeline: step i stages plane p0 + i into slot i & 3 and (i >= 2) produces plane z0 + i - 2 --
  const int nsteps = (z1 - z0) + 2;
#pragma unroll
  for (int j = 0; j < CZ_PF; ++j) fetch(xr[j], p0 + j);
  publish_max(xr[0], p0, 0);
  __syncthreads();
  int kA0 = 0, kA1 = 0, kA2 = 0;    // exponents of the planes of taps kz = 0, 1, 2 of the current step
  float4 res_prev = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ib = 0; ib < nsteps; ib += CZ_PF) {
#pragma unroll
    for (int j = 0; j < CZ_PF; ++j) {
      const int i = ib + j;
      if (i >= nsteps) break;
      // slot of plane p0 + i is j (CZ_PF == ring size == 4)
      kA0 = kA1;
      kA1 = kA2;
      kA2 = plane_exp(j);
      put(xr[j], p0 + i, j, kA2);
      publish_max(xr[(j + 1) & 3], p0 + i + 1, (j + 1) & 3);
      fetch(xr[j], p0 + i + CZ_PF);       // flies during the next CZ_PF - 1 steps
      const int z = z0 + i - 2;
      float4 res4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i >= 2 && a.res != nullptr && eok)
        res4 = *reinterpret_cast<const float4*>(a.res + (erow0 + (size_t)z * a.Ho * a.Wo) * 16 + ec);
      __syncthreads();
      if (i >= 3) epilogue(z - 1, res_prev, sRed + ((j + 1) & 1) * (4 * 64 * 16));   // plane of step i - 1
      res_prev = res4;
      if (i < 2) continue;
      // ---- MFMAs: this wave's taps x 4 m-tiles (16 voxels = two rows of the plane) --------------
      f32x4 acc[3][4];
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) acc[kz][mt][rr] = 0.f;
      // slots of taps kz = 0, 1, 2: planes i - 2, i - 1, i
      const char* sl[3] = {sX + ((j + 2) & 3) * CZ_PLANE, sX + ((j + 3) & 3) * CZ_PLANE,
                           sX + j * CZ_PLANE};
      half8 fa[2][4];
      auto load_a = [&](int q, half8* f) {
        const char* base = (tkz[q] == 0 ? sl[0] : (tkz[q] == 1 ? sl[1] : sl[2])) + aoff[q];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          f[mt] = *reinterpret_cast<const half8*>(base + mt * (2 * CZ_HX * 64));
      };
      load_a(0, fa[0]);
#pragma unroll
      for (int q = 0; q < CZ_MAXJ; ++q) {
        if (q + 1 < CZ_MAXJ) load_a(q + 1, fa[(q + 1) & 1]);
        // (q == 6 of wave 3 is the 28th tap: zero weights)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          // taps of a wave in q order have kz = 0,0,(0|1),1,(1|2),2,2: select the group per wave
          // with wave-uniform branches on tkz (compile-time for q = 0, 1, 5, 6)
          if (q <= 1 || (q == 2 && tkz[2] == 0)) {
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b1[q], acc[0][mt], 0, 0, 0);
            acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b2[q], acc[0][mt], 0, 0, 0);
          } else if (q <= 3 || (q == 4 && tkz[4] == 1)) {
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b1[q], acc[1][mt], 0, 0, 0);
            acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b2[q], acc[1][mt], 0, 0, 0);
          } else {
            acc[2][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b1[q], acc[2][mt], 0, 0, 0);
            acc[2][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[q & 1][mt], b2[q], acc[2][mt], 0, 0, 0);
          }
        }
      }
      // ---- combine the kz groups with their planes' scales, park this wave's partial plane -------
      const float f0 = __int_as_float((127 - kA0) << 23), f1 = __int_as_float((127 - kA1) << 23),
                  f2 = __int_as_float((127 - kA2) << 23);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          sRed[(j & 1) * (4 * 64 * 16) + (wave * 64 + 16 * mt + 4 * g + rr) * 16 + r] =
              (acc[0][mt][rr] * f0 + acc[1][mt][rr] * f1) + acc[2][mt][rr] * f2;
      // (its epilogue runs after the next step's barrier; that buffer is written again two steps on)
    }
  }
  __syncthreads();
  epilogue(z1 - 1, res_prev, sRed + ((nsteps - 1) & 1) * (4 * 64 * 16));   // the last plane
  if (!ROWS && e.amax_out != nullptr && tid == 0) atomicMax(e.amax_out, __float_as_uint(blockmax));
  if (a.part) {
    __syncthreads();
    float* red = sRed;   // [256 threads][8]
    red[tid * 8 + 0] = s1.x; red[tid * 8 + 1] = s1.y; red[tid * 8 + 2] = s1.z; red[tid * 8 + 3] = s1.w;
    red[tid * 8 + 4] = s2.x; red[tid * 8 + 5] = s2.y; red[tid * 8 + 6] = s2.z; red[tid * 8 + 7] = s2.w;
    __syncthreads();
    if (tid < 16) {
      const int c4 = tid >> 2, cj = tid & 3;
      float t1 = 0.f, t2 = 0.f;
      for (int v = 0; v < 64; ++v) {
        t1 += red[(4 * v + c4) * 8 + cj];
        t2 += red[(4 * v + c4) * 8 + 4 + cj];
      }
      float* p = a.part + (((size_t)nb * per_item + tile_id) * 16 + tid) * 2;
      p[0] = t1;
      p[1] = t2;
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------
// 1 when this launch takes the kernel: 16 -> 16 channels, 3^3 taps, stride 1, one fp32 or split-row
// source, one destination, planes of at least 8 x 8.
extern "C" int adell_conv_zring16_ok(const ConvArgs* a) {
  if (g_adell_tune.igemm_no16) return 0;
  if (a->Cout != 16 || a->Cin != 16 || a->C0 != 16 || a->C1 != 0) return 0;
  if (a->KD != 3 || a->KH != 3 || a->KW != 3 || a->SD != 1 || a->SH != 1 || a->SW != 1) return 0;
  if (a->UPS != 1 || a->UPSY != 1 || a->UPSZ != 1 || a->shuffle != 0) return 0;
  if (a->ysplit != a->Cout) return 0;
  if (a->Wo < 8 || a->Ho < 8 || a->Do < 4) return 0;
  if ((size_t)a->D * a->H * a->W * 16 >= ((size_t)1 << 28)) return 0;   // 32-bit byte offsets in an item
  return 1;   // (pointer alignment is checked at launch: a plan must not depend on it)
}

// z segments of the launch: units (columns x segments x items) dealt to ~2 blocks per CU so that the
// blocks share them out evenly (the rule of adell_wgrad_zring_plan)
extern "C" void adell_conv_zring16_segments(int N, int Do, int Ho, int Wo, int* seglen, int* nseg) {
  const long ncols = (long)N * adell_cdiv(Wo, 8) * adell_cdiv(Ho, 8);
  const long target = 512;
  const long maxseg = Do / 4 > 0 ? Do / 4 : 1;
  long best = -1, pick = 1;
  for (long cand = 1; cand <= maxseg; ++cand) {
    const long sl = adell_cdiv(Do, (int)cand), ns = adell_cdiv(Do, (int)sl);
    if (ns != cand) continue;
    const long units = ncols * ns;
    const long blocks = units < target ? units : target;
    const long cost = adell_cdiv((int)units, (int)blocks) * (sl + 6);   // 6: steps a unit costs besides its planes (priming, epilogue)
    if (best < 0 || cost < best) { best = cost; pick = cand; }
  }
  *seglen = adell_cdiv(Do, (int)pick);
  *nseg = adell_cdiv(Do, *seglen);
}

extern "C" int adell_conv_zring16_launch(const ConvArgs* a, const ConvF16Extra* e, int N, int seglen,
                                         int nseg, hipStream_t st) {
  ADELL_REQUIRE((((uintptr_t)a->x0 | (uintptr_t)e->xs0 | (uintptr_t)a->y0 | (uintptr_t)a->res) & 15) == 0,
                "conv f16x3 (16-column z-ring): tensors must be 16-byte aligned");
  ConvZr16Args args;
  args.a = *a;
  args.e = *e;
  args.seglen = seglen;
  args.nseg = nseg;
  const size_t lds = 4 * (size_t)CZ_PLANE + (size_t)2 * 4 * 64 * 16 * 4 + 16 * 4;
  const int per_item = a->ntx * a->nty * nseg;
  dim3 grid((unsigned)(8 * ((per_item + 7) / 8)), (unsigned)N);
  if (e->xs0 != nullptr)
    hipLaunchKernelGGL(adell_conv_zring16_kernel<1>, grid, dim3(256), lds, st, args);
  else
    hipLaunchKernelGGL(adell_conv_zring16_kernel<0>, grid, dim3(256), lds, st, args);
  ADELL_CHECK_HIP(hipGetLastError());
  return ADELL_OK;
}
